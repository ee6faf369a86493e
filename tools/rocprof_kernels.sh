#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python tool; prints the ls:: kernels' call counts and average durations.
# usage (GPU box, repo root): bash tools/rocprof_kernels.sh <tag> <script.py> [args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python3 "$REPO/$1" "${@:2}" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "ls::" in n:
            n = n[n.index("k_"):].split("(")[0]
            print(f"{n:45s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:7.2f} max {float(r['MaxNs'])/1e3:7.2f}")
PY
