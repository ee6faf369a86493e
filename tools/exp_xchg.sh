#!/bin/bash
# how the fused finish + pack publishes its count: a store (shipped) against an atomic exchange (-DLS_EXP_PUBLISH_XCHG)
# tools/exp_xchg.sh <base variant> <xchg variant> <out dir>
set -e
cd "$(dirname "$0")/.."
OUT=$3; mkdir -p $OUT
for r in 1 2; do
for v in $1 $2; do
  E=$PWD/build/exp/$v/liblidarshooter_hip.so
  export LS_LIB_PATH=$E LD_PRELOAD=$E
  echo "$v full raster, one in flight, fused (2 048 workgroups): $(LS_FUSE_FINISH_PACK_SINGLE=1 LS_FUSE_FINISH_PACK_BLOCKS=4096 PROBE_WINDOWS=4 timeout -k 10 300 python tools/variance_probe.py 2 0 2>&1 | grep medians | cut -c1-40)"
  echo "$v full raster, three in flight, fused: $(LS_FUSE_FINISH_PACK_BLOCKS=4096 PROBE_WINDOWS=4 timeout -k 10 300 python tools/variance_probe.py 2 2 2>&1 | grep medians | cut -c1-40)"
  echo "$v 10M rank 5, graphs, fused in graph: $(W=syn128x10m RANKS=5 MODES=graph LS_FUSE_IN_GRAPH=1 timeout -k 10 300 python tools/shard_cost.py 2 8 all 2>&1 | grep 'world 8' | sed 's/.*three as frame graphs/graphs/' | cut -c1-60)"
  echo "$v 10M rank 5, graphs, two launches:   $(W=syn128x10m RANKS=5 MODES=graph timeout -k 10 300 python tools/shard_cost.py 2 8 all 2>&1 | grep 'world 8' | sed 's/.*three as frame graphs/graphs/' | cut -c1-60)"
done
done 2>&1 | tee $OUT/xchg.txt
