for wl in c5shard8 c2 c3 k30; do tools/profile_workload.sh r2base $wl; done
