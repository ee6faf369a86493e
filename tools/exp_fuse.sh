#!/bin/bash
# finish + pack as one launch at the full raster (LS_FUSE_FINISH_PACK_BLOCKS / _SINGLE, experimental builds only) against the two
# launches: tools/exp_fuse.sh <variant under build/exp> <out dir>.  Parity subset first (with the fused path forced), then
# tools/variance_probe.py alternating the two forms, three frames in flight and one.
set -e
cd "$(dirname "$0")/.."
E=$PWD/build/exp/$1/liblidarshooter_hip.so
OUT=$2; mkdir -p $OUT
export LS_LIB_PATH=$E LD_PRELOAD=$E
LS_FUSE_FINISH_PACK_BLOCKS=4096 LS_FUSE_FINISH_PACK_SINGLE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cull.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
for c in "0 0" "4096 1" "0 0" "4096 1"; do
  set -- $c
  for m in 2 0; do
    echo "== blocks=$1 single=$2 mode=$m"
    LS_FUSE_FINISH_PACK_BLOCKS=$1 LS_FUSE_FINISH_PACK_SINGLE=$2 timeout -k 10 300 python tools/variance_probe.py 2 $m 2>&1 | grep -E "us per frame per|medians" || true
  done
done > $OUT/fuse.txt 2>&1
cat $OUT/fuse.txt
