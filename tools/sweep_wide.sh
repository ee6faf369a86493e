#!/bin/bash
# NEEDS the experimental build (tools/exp_build.sh base ""): LS_TRACE_* are only read there.  The four-wide walk's knobs.
set -u
export LS_LIB_PATH=$(pwd)/build/exp/base/liblidarshooter_hip.so
run() { echo -n "$1: "; timeout -k 10 120 python3 tools/bvh_frame_cost.py 300 1 2>&1 | grep "us per frame"; }
for W in 1 0; do for B in 1 2 3 4 5; do LS_BENCH_BVH_WIDE=$W LS_TRACE_BLOCKS_PER_CU=$B run "wide=$W blocks/CU=$B"; done; done
for R in 24 40 48 64; do LS_BENCH_BVH_WIDE=1 LS_TRACE_REFILL_MIN=$R run "wide=1 refill_min=$R"; done
for LW in 0 8 24 32; do LS_BENCH_BVH_WIDE=1 LS_TRACE_LEAF_WAIT=$LW run "wide=1 leaf_wait=$LW"; done
for B in 3 4; do for R in 40 64; do LS_BENCH_BVH_WIDE=1 LS_TRACE_BLOCKS_PER_CU=$B LS_TRACE_REFILL_MIN=$R run "wide=1 blocks=$B refill=$R"; done; done
