#!/bin/bash
# GPU box: parity subset for the group-culling path, then rocprofv3 kernel averages of SYN-10M (auto cull) and SYN-1M with
# culling forced on; optional experimental library (built by tools/exp_build.sh base "") with spread runs off.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
python -m pytest tests/test_gpu_cull.py tests/test_gpu_parity.py -x -q -k "cull or config5 or syn_10m or full_size" > gpurun_out/cull_check_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/cull_check_pytest.log
echo "--- 10M"; W=syn128x10m bash tools/rocprof_kernels.sh cc_10m tools/shard_cost.py 2 1 2>&1 | grep "k_project<\|k_cull\|world"
if [ -f build/exp/base/liblidarshooter_hip.so ]; then
  echo "--- 10M spread 0"; LS_LIB_PATH=$REPO/build/exp/base/liblidarshooter_hip.so LS_PROJECT_SPREAD=0 W=syn128x10m bash tools/rocprof_kernels.sh cc_10m_s0 tools/shard_cost.py 2 1 2>&1 | grep "k_project<\|k_cull\|world"
fi
echo "--- 1M cull on"; bash tools/rocprof_kernels.sh cc_1m tools/shard_cost.py 1 1 2>&1 | grep "k_project<\|k_cull\|world"
