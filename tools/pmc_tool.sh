#!/bin/bash
# rocprofv3 PMC passes (each counter set in its own run, kernel-trace only) over a python tool; prints per-kernel averages.
# usage (GPU box, repo root): bash tools/pmc_tool.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- script.py [args...]
TAG=$1; shift
PASSES=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/$1" "${@:2}" > "$OUT/p$i.out" 2> "$OUT/p$i.err" || echo "pass $i failed: $C"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k: continue
        name = k[k.index("k_"):].split("(")[0]
        a = acc[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/summary.txt", "w") as fh:
    for (n, c), (s, k) in sorted(acc.items()):
        fh.write(f"{n:44s} {c:28s} {s/k:18.1f} (n={k})\n")
print(open(sys.argv[1] + "/summary.txt").read())
PY
