#!/bin/bash
for args in "" "--overlap-aux 1" "--overlap-aux 1 --aux-cus 0" "--overlap-aux 1 --main-chunks 4" "--overlap-aux 1 --aux-cus 0 --main-chunks 4" "--main-chunks 4"; do
  echo "== $args"; timeout 600 python tools/shard_sim.py $args 2>&1 | grep -E "^shards|shard0 stats:" | cut -c1-420
done
