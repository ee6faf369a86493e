#!/bin/bash
# Where group culling starts to pay: streamed frame time, three frames in flight, through the C++ loop
# (tools/shard_cost.py), LS_OPT_BLOCK_CULL off (0) and on (1), the full raster (world 1) and an eighth of a turn (world 8, the
# lightest and the heaviest sector), mesh sizes between SYN-1M and SYN-10M, and the same span with fewer rings.
# usage (GPU box, repo root): bash tools/cull_crossover.sh > gpurun_out/<dir>/cull_crossover.txt
set -u
for W in syn128x1m syn128x2m syn128x3m syn128x5m syn128x10m; do
  for CULL in 0 1; do
    W=$W MODES=three RANKS= timeout -k 10 300 python3 tools/shard_cost.py $CULL 1,8 two 2>&1 | grep -E "workload|world"
  done
done
for V in 32 64; do
  for W in syn128x1m syn128x2m syn128x5m; do
    for CULL in 0 1; do
      echo "channels $V"
      LS_BENCH_SYN_CHANNELS=$V W=$W MODES=three timeout -k 10 300 python3 tools/shard_cost.py $CULL 1 two 2>&1 | grep -E "workload|world"
    done
  done
done
