#!/bin/bash
# experimental builds side by side (tools/exp_build.sh <name> "<-D flags>" -> build/exp/<name>; selected through LS_LIB_PATH):
# VARIANTS="base other base other" bash tools/sweep_variants.sh -- the BVH engine's frame by the host clock + the cloud's hash, SYN-1M and configs[4].
# (Round 6 compared the five-comparator sort, nearest-only ordering and a leaf-record prefetch this way: EXPERIMENTS.md E8.3.)
set -u
for WL in syn128x1m cfg5; do
  for V in ${VARIANTS:-base nearest base nearest}; do
    echo -n "$WL $V: "
    env LS_LIB_PATH=$(pwd)/build/exp/$V/liblidarshooter_hip.so W=$WL timeout -k 10 200 python3 tools/bvh_frame_cost.py 300 1 2>&1 | grep -E "points|us per frame|rror" | tr '\n' ' '; echo
  done
done
