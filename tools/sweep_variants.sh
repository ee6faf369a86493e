#!/bin/bash
# experimental builds of the four-wide walk side by side (build/exp/<name>; LS_LIB_PATH)
set -u
for WL in syn128x1m cfg5; do
  for V in ${VARIANTS:-base nearest base nearest}; do
    echo -n "$WL $V: "
    env LS_LIB_PATH=$(pwd)/build/exp/$V/liblidarshooter_hip.so W=$WL timeout -k 10 200 python3 tools/bvh_frame_cost.py 300 1 2>&1 | grep -E "points|us per frame|rror" | tr '\n' ' '; echo
  done
done
