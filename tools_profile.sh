#!/bin/bash
# Profiling recipe used for profiles/: kernel trace + stats, then HBM counters in separate passes.
# Usage (on the GPU box, from the repo root):  bash tools_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r06}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 10 --no-cpu-baseline --no-dropin --no-also $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || echo "stats pass failed"
# the same frames with ONE frame in flight: every dispatch runs alone, the stats average is the isolated kernel's duration
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_one" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS --no-pipeline > "$OUT/bench_stats_one.json" 2> "$OUT/stats_one.err" || echo "one-in-flight stats pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || echo "fetch pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err" || echo "write pass failed"
find "$OUT" -name "*.csv" | head -20
