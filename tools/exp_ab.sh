#!/bin/bash
# same-box A/B of experimental builds (tools/exp_build.sh): tools/exp_ab.sh <out dir> <variant> [<variant> ...]
# Per variant: the parity subset once, then tools/variance_probe.py with three frames in flight and with one, the variants
# alternating, twice round.  The builds are preloaded (the host mirror links the shipped library by name).
set -e
cd "$(dirname "$0")/.."
OUT=$1; shift; mkdir -p $OUT
for v in "$@"; do
  E=$PWD/build/exp/$v/liblidarshooter_hip.so
  LS_LIB_PATH=$E LD_PRELOAD=$E timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "xt32 or full_size or two_step or edge" > $OUT/pytest_$v.log 2>&1 || { tail -20 $OUT/pytest_$v.log; exit 1; }
  echo "$v: $(tail -1 $OUT/pytest_$v.log)"
done
for round in 1 2; do
  for v in "$@"; do
    E=$PWD/build/exp/$v/liblidarshooter_hip.so
    for m in 2 0; do
      echo "== $v mode=$m: $(LS_LIB_PATH=$E LD_PRELOAD=$E PROBE_WINDOWS=5 timeout -k 10 300 python tools/variance_probe.py 2 $m 2>&1 | grep medians)"
    done
  done
done 2>&1 | tee $OUT/ab.txt
