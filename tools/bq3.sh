#!/bin/bash
# A/B of an environment switch on the same box: tools/bq3.sh "<ENV=1>" "<bench args>" ...
E=$1; shift
for a in "$@"; do
  echo "-- default"; tools/bq2.sh "$a" | cut -c1-330
  echo "-- $E"; env $E tools/bq2.sh "$a" | cut -c1-330
done
