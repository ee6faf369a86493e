#!/bin/bash
# GPU box: three frames in flight on CU-masked streams (LS_CU_MASK_MODE, experimental build build/exp/base) against the
# plain streams: bench.py's headline step time, three processes per setting (the step time varies from process to process).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export LS_LIB_PATH=$REPO/build/exp/base/liblidarshooter_hip.so
for mode in 0 1 2 0 1 2 0 1 2; do
  LS_CU_MASK_MODE=$mode python bench.py --no-cpu-baseline --no-dropin --steps 200 > /tmp/o.json 2>/tmp/o.err
  python3 - "$mode" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/o.json")); r = d["roofline"]
    print("mask mode", sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "windows", [round(x, 3) for x in d["timing"]["window_ms_min_median_max"]],
          "one in flight %.4f" % d["ms_per_step_one_frame_in_flight"], "k_project %.4f" % r["kernel_ms"], flush=True)
except Exception as e:
    print("mask mode", sys.argv[1], "failed", e, open("/tmp/o.err").read()[-400:])
PY
done
