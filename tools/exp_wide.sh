#!/bin/bash
# finish + pack as one launch of wide workgroups for a frame that has the device to itself (LS_FUSE_WIDE, experimental builds):
# tools/exp_wide.sh <variant under build/exp> <out dir>.  Parity with the path forced, then one frame in flight alternating.
set -e
cd "$(dirname "$0")/.."
E=$PWD/build/exp/$1/liblidarshooter_hip.so
OUT=$2; mkdir -p $OUT
export LS_LIB_PATH=$E LD_PRELOAD=$E
LS_FUSE_WIDE=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull.py tests/test_gpu_dropin.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for r in 1 2 3; do
  for wv in 0 1; do
    echo "wide=$wv one in flight: $(LS_FUSE_WIDE=$wv PROBE_WINDOWS=5 timeout -k 10 300 python tools/variance_probe.py 2 0 2>&1 | grep medians | cut -c1-40)"
  done
done 2>&1 | tee $OUT/wide.txt
