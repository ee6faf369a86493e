#!/bin/bash
# Kernel durations of an azimuth shard under rocprofv3 (one mode of tools/shard_cost.py per run, so that the averages mean
# something).  usage (GPU box, repo root): bash tools/shard_profile.sh <tag> ; output: gpurun_out/shard_prof_<tag>/summary.txt
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/shard_prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/summary.txt"
run() {   # name, workload, cull, mode
    local name=$1
    W=$2 MODES=$4 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o s -- python3 "$REPO/tools/shard_cost.py" $3 8 two > "$OUT/$name.txt" 2> "$OUT/$name.err" || echo "$name failed" >> "$OUT/summary.txt"
    echo "== $name (W=$2 cull=$3 mode=$4)" >> "$OUT/summary.txt"
    grep "^world" "$OUT/$name.txt" >> "$OUT/summary.txt"
    local f=$(find "$OUT/$name" -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" >> "$OUT/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    for c in "(":
        if c in n: n = n[:n.index(c)]
    print("  %-60s calls %7s avg %9.2f us  min %9.2f  max %9.2f  %5.1f %%" % (n[-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
    find "$OUT/$name" -name "*kernel_trace.csv" -delete   # (tens of MB each; gpurun copies back 64 MiB at most)
}
DEFAULT_SPECS="1m_cull0_one:syn128x1m:0:one 1m_cull1_one:syn128x1m:1:one 1m_cull1_graph:syn128x1m:1:graph 10m_one:syn128x10m:2:one 10m_graph:syn128x10m:2:graph"
for spec in ${SPECS:-$DEFAULT_SPECS}; do
    IFS=: read name w c m <<< "$spec"
    run "$name" "$w" "$c" "$m"
done
cat "$OUT/summary.txt"
