#!/bin/bash
# Everything profiles/ holds for a round (GPU box, repo root):  bash tools/final_measure.sh r05 [a|b|c|all]
#   a: bench lines of the other workloads and modes, lsbench, the group through a one-rank communicator, shard costs, BVH at configs[4]
#   b: the profiles (kernel stats, HBM bytes, SQ / TCC counters) and, behind them, the two headline lines that quote them
#      (b1: kernel stats + HBM bytes; b2: the counters, the headline lines, SYN-10M, rebuild / refit -- two calls of at most 20 minutes)
#   c: soaks
TAG=${1:-r06}
PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out/final
: > gpurun_out/final/bench.err
# (the profiler's passes write their files when they end: a line a minute tells the box's watchdog that this script is alive; every
# pass has its own time limit inside the scripts it calls, and the call as a whole gpurun's)
( while true; do date +"%T still measuring" >> gpurun_out/final/heartbeat.txt; sleep 60; done ) &
HEARTBEAT=$!
trap "kill $HEARTBEAT 2>/dev/null" EXIT
if [ "$PART" = a ] || [ "$PART" = all ]; then
python bench.py --no-cpu-baseline --no-dropin --no-also --pipeline 1 > gpurun_out/final/bench_${TAG}_projection_mode1.json 2>> gpurun_out/final/bench.err
python bench.py --no-cpu-baseline --no-dropin --workload syn128x10m > gpurun_out/final/bench_${TAG}_projection_10m.json 2>> gpurun_out/final/bench.err
python bench.py --no-cpu-baseline --no-dropin --workload syn128x10m --no-cull > gpurun_out/final/bench_${TAG}_projection_10m_nocull.json 2>> gpurun_out/final/bench.err
python bench.py --no-cpu-baseline --no-dropin --workload cfg5 > gpurun_out/final/bench_${TAG}_projection_cfg5.json 2>> gpurun_out/final/bench.err
python bench.py --no-cpu-baseline --workload xt32 > gpurun_out/final/bench_${TAG}_projection_xt32.json 2>> gpurun_out/final/bench.err
python tools/dump_mesh.py syn1m /tmp/syn1m.lsmesh   # lsbench traces bench.py's SYN-1M bit for bit (its --grid is a look-alike)
for p in 0 1 2; do
  lidarshooter_amd/lsbench --config tests/golden/data/config/hesai-pandar-XT-32-lidar_0000.json --syn 128 4096 --mesh-raw ground=/tmp/syn1m.lsmesh --frames 2000 --warmup 200 --pipeline $p
done > gpurun_out/final/lsbench_${TAG}.jsonl 2>> gpurun_out/final/bench.err
lidarshooter_amd/lsbench --config tests/golden/data/config/hesai-pandar-XT-32-lidar_0000.json --mesh ground=tests/golden/data/mesh/ground.stl --mesh face=tests/golden/data/mesh/ben.stl --frames 5000 --warmup 200 --pipeline 1 >> gpurun_out/final/lsbench_${TAG}.jsonl 2>> gpurun_out/final/bench.err
# the group through a one-rank communicator: per-set communicators + one captured graph per frame (flags 0, the default), the
# same without the graph (2), round 3's arrangement (1: one communicator on a collective stream, events), the sized gather (4); then interleaved
for fl in 0 2 1 4; do
  lidarshooter_amd/lsbench --config tests/golden/data/config/hesai-pandar-XT-32-lidar_0000.json --syn 128 4096 --mesh-raw ground=/tmp/syn1m.lsmesh --frames 2000 --warmup 200 --ranks 1 --group sharded --group-flags $fl
done >> gpurun_out/final/lsbench_${TAG}.jsonl 2>> gpurun_out/final/bench.err
lidarshooter_amd/lsbench --config tests/golden/data/config/hesai-pandar-XT-32-lidar_0000.json --syn 128 4096 --mesh-raw ground=/tmp/syn1m.lsmesh --frames 2000 --warmup 200 --ranks 1 --group interleaved >> gpurun_out/final/lsbench_${TAG}.jsonl 2>> gpurun_out/final/bench.err
lidarshooter_amd/lsbench --config tests/golden/data/config/hesai-pandar-XT-32-lidar_0000.json --syn 128 4096 --mesh-raw ground=/tmp/syn1m.lsmesh --frames 2000 --warmup 200 --pipeline 2 --graph 1 >> gpurun_out/final/lsbench_${TAG}.jsonl 2>> gpurun_out/final/bench.err
# the N > 1 driver of bench.py on one GPU (every line of it: C group, one-rank communicator), and the host-cost micro-benchmark
LS_BENCH_FORCE_GROUP=1 python bench.py --no-cpu-baseline --no-dropin > gpurun_out/final/bench_${TAG}_force_group.json 2>> gpurun_out/final/bench.err
LS_BENCH_FORCE_GROUP=1 python bench.py --no-cpu-baseline --no-dropin --group-flags 1 > gpurun_out/final/bench_${TAG}_force_group_one_communicator.json 2>> gpurun_out/final/bench.err
(cd tools/micro && ./graph_launch --rccl) > gpurun_out/final/${TAG}_graph_launch_micro.txt 2>&1
(cd tools/micro && ./chunked_h2d) > gpurun_out/final/${TAG}_chunked_h2d_micro.txt 2>&1
(cd tools/micro && ./loads_probe) > gpurun_out/final/${TAG}_loads_probe_micro.txt 2>&1
(cd tools/micro && ./atomic_rate) > gpurun_out/final/${TAG}_atomic_rate_micro.txt 2>&1
(cd tools/micro && ./strided_h2d) > gpurun_out/final/${TAG}_strided_h2d_micro.txt 2>&1
(cd tools/micro && timeout -k 10 200 ./node_fetch 400) > gpurun_out/final/${TAG}_node_fetch_micro.txt 2>&1   # (second form of the probe; profiles/r06_node_fetch_micro.txt holds both)
# what ONE of eight ranks does per frame (no collective): every rank of an eighth-of-a-turn split, SYN-1M and SYN-10M, streamed
# through the C++ loop (one / three frames in flight / three as frame graphs); then the shard's kernels under rocprofv3
python tools/shard_cost.py 2 1,8 all > gpurun_out/final/${TAG}_shard_cost_1m.txt 2>> gpurun_out/final/bench.err
W=syn128x10m python tools/shard_cost.py 2 1,8 all > gpurun_out/final/${TAG}_shard_cost_10m.txt 2>> gpurun_out/final/bench.err
SPECS="1m_one:syn128x1m:2:one 1m_graph:syn128x1m:2:graph 10m_one:syn128x10m:2:one 10m_graph:syn128x10m:2:graph" bash tools/shard_profile.sh ${TAG} > /dev/null 2>&1
grep -v "calls       [0-9] \|calls      [1-9][0-9] " gpurun_out/shard_prof_${TAG}/summary.txt > gpurun_out/final/${TAG}_shard_kernels.txt
python tools/dropin_bench.py 50 > gpurun_out/final/dropin_${TAG}.json 2>> gpurun_out/final/bench.err
# BASELINE configs[4] on the BVH engine at its own size: build, poses, refit of the moving instance, classic refit of all 10 M
python tools/bvh_cfg5_cost.py 60 > gpurun_out/final/${TAG}_bvh_cfg5_cost.json 2>> gpurun_out/final/bench.err
for ph in build poses refit_ben classic; do echo "== $ph"; PHASE=$ph bash tools/rocprof_kernels.sh cfg5_$ph tools/bvh_cfg5_cost.py 30 | sort -k5 -n -r | head -16; tail -1 gpurun_out/rp_cfg5_$ph/stdout.log; done > gpurun_out/final/${TAG}_bvh_cfg5_kernels.txt 2>&1
find gpurun_out/rp_cfg5_* -name "*kernel_trace.csv" -delete
python bench.py --workload cfg5 --engine bvh --no-cpu-baseline --no-dropin > gpurun_out/final/bench_${TAG}_bvh_cfg5.json 2>> gpurun_out/final/bench.err
fi
if [ "$PART" = b ] || [ "$PART" = b1 ] || [ "$PART" = all ]; then
bash tools_profile.sh ${TAG} > gpurun_out/final/profile.log 2>&1
python tools/prof_summary.py gpurun_out/prof_${TAG} gpurun_out/final/${TAG}_projection > gpurun_out/final/prof_summary.log 2>&1
python3 - gpurun_out/prof_${TAG}/stats_one gpurun_out/final/${TAG}_projection_kernel_stats_one_in_flight.csv <<'PY'
import csv, glob, os, sys
sys.path.insert(0, "tools")
from prof_summary import short
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ls::" in r["Name"]:
            rows.append([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    w.writerows(rows)
PY
cp gpurun_out/prof_${TAG}/bench_stats.json gpurun_out/final/${TAG}_projection_bench_under_rocprof.json   # prof_summary also wrote ${TAG}_projection_kernel_isolated.json (dispatches that ran alone vs overlapped)
BENCH_ARGS="--engine bvh --no-dropin" bash tools_pmc.sh ${TAG}_bvh "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/final/pmc_bvh.log 2>&1
bash tools/bvh_stats.sh ${TAG} > gpurun_out/final/bvh_stats.log 2>&1
python tools/pmc_to_hbm.py gpurun_out/pmc_${TAG}_bvh/summary.txt gpurun_out/final/${TAG}_bvh_hbm.json "bench.py --engine bvh (instanced hierarchies: nothing built per frame)" >> gpurun_out/final/pmc_bvh.log 2>&1
fi
if [ "$PART" = b ] || [ "$PART" = b2 ] || [ "$PART" = all ]; then
# what binds the kernels: SQ / TCC counters of the three per-frame paths and of an eighth-of-a-turn shard
bash tools/sq_profile.sh ${TAG} > gpurun_out/final/sq_profile.log 2>&1
cp gpurun_out/sq_${TAG}/*_sq.txt gpurun_out/sq_${TAG}/*_sq.json gpurun_out/final/ 2>/dev/null
for f in projection projection_10m bvh shard_1m shard_10m; do [ -f gpurun_out/final/${f}_sq.txt ] && mv gpurun_out/final/${f}_sq.txt gpurun_out/final/${TAG}_${f}_sq.txt; [ -f gpurun_out/final/${f}_sq.json ] && mv gpurun_out/final/${f}_sq.json gpurun_out/final/${TAG}_${f}_sq.json; done
# the two headline lines last: they read `roofline.traffic` and `roofline.secondary` from the profiles just taken (same kernel sources, same box)
cp gpurun_out/final/${TAG}_projection_hbm.json gpurun_out/final/${TAG}_bvh_hbm.json gpurun_out/final/${TAG}_projection_sq.json gpurun_out/final/${TAG}_bvh_sq.json profiles/
python bench.py > gpurun_out/final/${TAG}_projection_bench.json 2>> gpurun_out/final/bench.err
python bench.py --engine bvh --no-dropin > gpurun_out/final/${TAG}_bvh_bench.json 2>> gpurun_out/final/bench.err
python bench.py --engine bvh --no-dropin --no-cpu-baseline --classic-bvh > gpurun_out/final/bench_${TAG}_bvh_classic.json 2>> gpurun_out/final/bench.err
# SYN-10M: kernel averages of the culled stage (k_cull + k_project), what it reads and keeps, its HBM-side traffic
W=syn128x10m MODES=one bash tools/rocprof_kernels.sh ${TAG}_10m tools/shard_cost.py 2 1 > gpurun_out/final/${TAG}_projection_10m_kernels.txt 2>&1   # (one frame in flight: every dispatch alone)
cp gpurun_out/rp_${TAG}_10m/${TAG}_10m_kernel_stats.csv gpurun_out/final/${TAG}_projection_10m_kernel_stats.csv
python tools/cull_stats.py syn128x10m 1 >> gpurun_out/final/${TAG}_projection_10m_kernels.txt 2>> gpurun_out/final/bench.err
W=syn128x10m MODES=one bash tools/pmc_tool.sh ${TAG}_10m "FETCH_SIZE" "WRITE_SIZE" -- tools/shard_cost.py 2 1 > gpurun_out/final/${TAG}_projection_10m_pmc.txt 2>&1
# BVH engine: a full rebuild of SYN-1M every frame (instanced and classic): the build kernels, the hand-written sort among them
bash tools/rocprof_kernels.sh ${TAG}_rebuild tools/rebuild_cost.py > gpurun_out/final/${TAG}_bvh_rebuild_kernels.txt 2>&1
tail -2 gpurun_out/rp_${TAG}_rebuild/stdout.log >> gpurun_out/final/${TAG}_bvh_rebuild_kernels.txt
# and a refit per frame (classic hierarchy, poses restated): transform, leaves + range tree, k_refit_nodes
bash tools/rocprof_kernels.sh ${TAG}_refit tools/refit_cost.py 50 > gpurun_out/final/${TAG}_bvh_refit_kernels.txt 2>&1
tail -2 gpurun_out/rp_${TAG}_refit/stdout.log >> gpurun_out/final/${TAG}_bvh_refit_kernels.txt
fi
if [ "$PART" = c ] || [ "$PART" = all ]; then
{ python tests/analysis/soak.py 60; python tests/analysis/soak_bvh.py 60; python tests/analysis/soak_shard.py 120; python tests/analysis/soak_wide.py 60
  for fl in 0 4 2; do python tests/analysis/soak_group.py 40 $fl; done; } > gpurun_out/final/${TAG}_soak.txt 2>&1
fi
# (gpurun copies back 64 MiB at most: the raw traces and counter dumps have been condensed into gpurun_out/final by now)
find gpurun_out -type f \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.rocpd" -o -name "*.db" \) -not -path "*/final/*" -delete
find gpurun_out -type f -size +4M -not -path "*/final/*" -delete
du -sh gpurun_out
ls gpurun_out/final
tail -3 gpurun_out/final/bench.err
