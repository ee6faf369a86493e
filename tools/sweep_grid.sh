#!/bin/bash
# experimental build: the persistent trace grid at sizes between two and three blocks per CU (E8.3: 2.07 heavy batches per wave).
# NEEDS a knob that was not kept: in trace_grid_blocks (ls_kernels.hip), `const int absolute = lsi::tune_int("LS_TRACE_GRID_BLOCKS", 0);
# if (absolute > 0) return (uint32_t)absolute;` -- the sweep was flat (profiles/r06_bvh_grid_sweep.txt)
set -u
export LS_LIB_PATH=$(pwd)/build/exp/base/liblidarshooter_hip.so
for WL in syn128x1m cfg5; do
  for B in 512 544 576 608 640 704 768; do
    echo -n "$WL grid=$B blocks: "
    env W=$WL LS_TRACE_GRID_BLOCKS=$B timeout -k 10 200 python3 tools/bvh_frame_cost.py 300 1 2>&1 | grep -E "us per frame|rror" | tail -1
  done
done
