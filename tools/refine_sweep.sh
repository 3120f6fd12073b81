#!/bin/bash
# rows above the refine limit and the price of refining them, per bench workload: tools/refine_sweep.sh <limit> ...
for L in "$@"; do
  for w in c4 c5shard8 c3 c2 k30; do
    echo "== limit $L"; MALS_REFINE_LIMIT=$L tools/bq2.sh "--workload $w" | cut -c1-200
  done
done
