#!/bin/bash
# GPU box: rocprofv3 kernel averages of the headline frame (one frame in flight) for variants "<dir>[:ENV=VAL,...]" WITHOUT
# the parity subset (timing probes that leave results wrong on purpose).  usage: bash tools/exp_kernels.sh spec1 spec2 ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/exp
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  tag=$(echo "$spec" | tr ':=,' '___')
  (
  export LS_LIB_PATH=$REPO/build/exp/$v/liblidarshooter_hip.so
  for e in ${envs//,/ }; do export "$e"; done
  echo "=== $spec"
  timeout -k 10 300 bash $REPO/tools/rocprof_kernels.sh exp_$tag tools/shard_cost.py ${EXP_CULL:-0} ${EXP_WORLD:-1} | grep -v "^$" | grep "k_project"
  )
done
