#!/bin/bash
# the unculled k_project over pre-gathered 48-byte corner records (one memory round trip instead of index -> vertex), SYN-1M:
# tools/exp_corners.sh <base variant> <variant built with -DLS_EXP_CORNERS_DIRECT>   (PROBE_CULL=1 makes the commit build the records)
cd "$(dirname "$0")/.."
B=$PWD/build/exp/$1/liblidarshooter_hip.so; C=$PWD/build/exp/$2/liblidarshooter_hip.so
LS_LIB_PATH=$C LD_PRELOAD=$C PROBE_CULL=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do for m in 2 0; do
  echo "base, culling off,          mode $m: $(LS_LIB_PATH=$B LD_PRELOAD=$B PROBE_WINDOWS=5 timeout -k 10 300 python tools/variance_probe.py 2 $m 2>&1 | grep medians | cut -c1-40)"
  echo "corner records, no cull,    mode $m: $(LS_LIB_PATH=$C LD_PRELOAD=$C PROBE_CULL=1 PROBE_WINDOWS=5 timeout -k 10 300 python tools/variance_probe.py 2 $m 2>&1 | grep medians | cut -c1-40)"
  echo "base, culling forced on,    mode $m: $(LS_LIB_PATH=$B LD_PRELOAD=$B PROBE_CULL=1 PROBE_WINDOWS=5 timeout -k 10 300 python tools/variance_probe.py 2 $m 2>&1 | grep medians | cut -c1-40)"
done; done
