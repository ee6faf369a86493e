#!/bin/bash
# k_cull rounds per workgroup at SYN-10M, full raster, three frames in flight (LS_CULL_ROUNDS: an experimental build with that knob)
cd "$(dirname "$0")/.."
E=$PWD/build/exp/$1/liblidarshooter_hip.so
export LS_LIB_PATH=$E LD_PRELOAD=$E
for r in 1 2; do for rounds in 0 3 4 6 8; do
  echo "rounds=$rounds: $(LS_CULL_ROUNDS=$rounds W=syn128x10m MODES=one,three timeout -k 10 300 python tools/shard_cost.py 2 1 all 2>&1 | grep 'world 1' | sed 's/.*one in flight \([0-9.]*\) us.*three in flight \([0-9.]*\) us.*/one \1 three \2/')"
done; done
