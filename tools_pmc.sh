#!/bin/bash
# PMC passes over bench.py (counters in their own runs, kernel-trace only).
# Usage: bash tools_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-also ${BENCH_ARGS:-} > "$OUT/p$i.json" 2> "$OUT/p$i.err" || echo "pass $i failed: $C"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k: continue
        name = k[k.index("k_"):].split("(")[0]
        a = acc[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(sys.argv[1] + "/summary.txt", "w") as fh:
    for (n, c), (s, k) in sorted(acc.items()):
        fh.write(f"{n:28s} {c:40s} {s/k:18.1f} (n={k})\n")
print(open(sys.argv[1] + "/summary.txt").read())
PY
