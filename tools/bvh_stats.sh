#!/bin/bash
# rocprofv3 --kernel-trace --stats of the BVH-engine headline frame -> gpurun_out/final/<tag>_bvh_kernel_stats.csv (GPU box, repo root)
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_bvh
mkdir -p "$OUT" "$REPO/gpurun_out/final"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- python3 "$REPO/bench.py" --engine bvh --steps 50 --warmup 10 --no-cpu-baseline --no-dropin > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || echo "stats pass failed"
cd "$REPO"
python3 - "$OUT" "gpurun_out/final/${TAG}_bvh_kernel_stats.csv" <<'PY'
import csv, glob, os, sys
sys.path.insert(0, "tools")
from prof_summary import short
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "stats", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(rows)
for r in rows[:12]:
    print(r[0], r[1], r[3], r[4])
PY
