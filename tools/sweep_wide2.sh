#!/bin/bash
# second sweep (experimental build): whole-wave refills only when every lane is idle (64), leaf phases at 8 / 16 lanes, both walks, both scenes
set -u
export LS_LIB_PATH=$(pwd)/build/exp/base/liblidarshooter_hip.so
for WL in syn128x1m cfg5; do
  for WIDE in 1 0; do
    for R in 56 64; do for LW in 8 16; do
      echo -n "$WL wide=$WIDE refill=$R leaf_wait=$LW: "
      env W=$WL LS_BENCH_BVH_WIDE=$WIDE LS_TRACE_REFILL_MIN=$R LS_TRACE_LEAF_WAIT=$LW timeout -k 10 200 python3 tools/bvh_frame_cost.py 300 1 2>&1 | grep -E "us per frame|Error|error" | tail -1
    done; done
  done
done
