#!/bin/bash
# A / B of the four-wide walk (LS_OPT_BVH_WIDE) on the BVH engine: headline scene and configs[4]'s, poses only; kernel times by rocprofv3
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for WIDE in 1 0; do
  for WL in syn128x1m cfg5; do
    LS_BENCH_BVH_WIDE=$WIDE timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${WL}_w$WIDE -o p -- python3 $REPO/bench.py --engine bvh --workload $WL --no-also --no-cpu-baseline --no-dropin --steps 50 --warmup 10 > $OUT/bench_${WL}_w$WIDE.json 2> $OUT/bench_${WL}_w$WIDE.err
    f=$(find $OUT/prof_${WL}_w$WIDE -name "*kernel_stats.csv" | head -1)
    echo "== $WL wide=$WIDE"; python3 -c "
import csv,sys
rows=list(csv.DictReader(open('$f')))
for r in rows[:8]: print('%-60.60s calls %6s avg %10.2f us  total %6.1f %%' % (r['Name'], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
"
    python3 -c "
import json; r=json.load(open('$OUT/bench_${WL}_w$WIDE.json')); print('ms_per_step', r['ms_per_step'], 'kernel_ms', r['roofline']['kernel_ms'], 'nodes/ray', r['roofline'].get('nodes_per_ray'), 'tris/ray', r['roofline'].get('tris_per_ray'), 'sha', r.get('points_sha256'))"
  done
done
