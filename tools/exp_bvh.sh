#!/bin/bash
# GPU box: BVH-engine headline frame for library variants "<dir>[:ENV=VAL,...]": parity subset, then bench.py --engine bvh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/exp $REPO/gpurun_out/final
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  tag=$(echo "$spec" | tr ':=,' '___')
  (
  export LS_LIB_PATH=$REPO/build/exp/$v/liblidarshooter_hip.so
  for e in ${envs//,/ }; do export "$e"; done
  echo "=== $spec"
  timeout -k 10 300 python3 -m pytest $REPO/tests/test_gpu_parity.py -x -q -k "full_size or random_soup or xt32 or leaf_sizes or bvh_structure or many_geometries or edge_cases" > $REPO/gpurun_out/exp/bvh_$tag.pytest.log 2>&1
  rc=$?
  tail -1 $REPO/gpurun_out/exp/bvh_$tag.pytest.log
  if [ $rc -ne 0 ] && [ -z "$EXP_NOPARITY" ]; then echo "PARITY FAILED for $spec"; exit 0; fi
  timeout -k 10 300 python3 $REPO/bench.py --engine bvh --no-dropin --no-cpu-baseline > $REPO/gpurun_out/exp/bvh_$tag.json 2> $REPO/gpurun_out/exp/bvh_$tag.err
  python3 - $REPO/gpurun_out/exp/bvh_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step %.4f  k_trace %.4f ms  nodes/ray %.2f tris/ray %.2f trips mean %.1f max %d" % (d["ms_per_step"], r["kernel_ms"], r["nodes_per_ray"], r["tris_per_ray"], r["wave_trips_mean"], r["wave_trips_max"]))
PY
  )
done
