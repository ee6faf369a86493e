#!/bin/bash
# second pass of tools/cull_crossover.sh: the point between 1 M and 2 M, and the shards as a group runs them (frame graphs)
set -u
for CULL in 0 1; do W=syn128x1500k MODES=three timeout -k 10 300 python3 tools/shard_cost.py $CULL 1 two 2>&1 | grep -E "workload|world"; done
for W in syn128x1m syn128x1500k syn128x2m syn128x3m; do
  for CULL in 0 1; do
    W=$W MODES=graph timeout -k 10 300 python3 tools/shard_cost.py $CULL 8 two 2>&1 | grep -E "workload|world"
  done
done
