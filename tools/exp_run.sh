#!/bin/bash
# GPU box: for each variant "<dir>[:ENV=VAL[,ENV=VAL...]]" under build/exp/: quick parity subset, then rocprofv3 kernel
# averages of the headline frame (one frame in flight on one stream).  usage: bash tools/exp_run.sh spec1 spec2 ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/exp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  tag=$(echo "$spec" | tr ':=,' '___')
  (
  export LS_LIB_PATH=$REPO/build/exp/$v/liblidarshooter_hip.so
  for e in ${envs//,/ }; do export "$e"; done
  echo "=== $spec"
  timeout -k 10 300 python3 -m pytest $REPO/tests/test_gpu_parity.py -x -q -k "full_size or big_ or random_soup or xt32 or many_geometries or footprints_never" > $REPO/gpurun_out/exp/$tag.pytest.log 2>&1
  rc=$?
  tail -1 $REPO/gpurun_out/exp/$tag.pytest.log
  if [ $rc -ne 0 ]; then echo "PARITY FAILED for $spec"; exit 0; fi
  timeout -k 10 300 bash $REPO/tools/rocprof_kernels.sh exp_$tag tools/shard_cost.py ${EXP_CULL:-0} ${EXP_WORLD:-1} | grep -v "^$" | tee $REPO/gpurun_out/exp/$tag.kernels.log
  grep "world" $REPO/gpurun_out/rp_exp_$tag/stdout.log
  )
done
