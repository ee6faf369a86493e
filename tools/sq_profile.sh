#!/bin/bash
# What binds the kernels, from counters: SQ instruction / wait / lane-occupancy counters and the L2's
# hit rate for the three per-frame paths -- the projection engine at SYN-1M (the headline) and at SYN-10M (culled), the BVH
# engine -- each with ONE frame in flight (every dispatch alone).  Counters are collected in their own passes with
# --kernel-trace only (8 SQ slots, 4 TCC slots per pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (GPU box, repo root): bash tools/sq_profile.sh <tag>   ->  gpurun_out/sq_<tag>/{projection,projection_10m,bvh}_sq.txt
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"
PASSES=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
        "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "TCC_HIT_sum TCC_MISS_sum")
sha=$(cd "$REPO" && python3 -c "import bench; print(bench.kernel_source_sha())")
run() {   # name, then the program and its arguments (python3 directly behind --: the profiler's library initialises the GPU first)
    local name=$1; shift
    local D=$OUT/$name; mkdir -p "$D"
    local i=0
    for C in "${PASSES[@]}"; do
        i=$((i+1))
        ( cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D/p$i" -o pmc -- python3 "$@" > "$D/p$i.out" 2> "$D/p$i.err" ) || echo "$name: pass $i failed ($C)"
    done
    python3 - "$D" "$name" "$sha" "$OUT/${name}_sq.json" > "$OUT/${name}_sq.txt" <<'PY'
import csv, glob, json, sys, collections
dur = collections.defaultdict(lambda: [0.0, 0])   # kernel durations under the counter passes (the dispatches' own timestamps)
for f in glob.glob(sys.argv[1] + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ls::" not in k or "k_spin" in k: continue
        a = dur[k[k.index("k_"):].split("(")[0]]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ls::" not in k or "k_spin" in k: continue
        name = k[k.index("k_"):].split("(")[0]
        a = acc[(name, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
kern = collections.defaultdict(dict)
out_json = {}
for (n, c), (s, k) in acc.items():
    kern[n][c] = (s / k, k)
print("# %s: per-dispatch averages of rocprofv3 counters (tools/sq_profile.sh), one frame in flight; kernel_source_sha %s" % (sys.argv[2], sys.argv[3]))
print("# derived: issue_us = wave-instructions issued x 2 cycles / (1024 SIMDs x 2.3 GHz) [a wave64 instruction occupies its SIMD-32 for 2 cycles];")
print("#          lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU [active lanes per VALU instruction, of 64]; wait / stall / issuing = shares of SQ_WAVE_CYCLES")
for n in sorted(kern, key=lambda n: -kern[n].get("SQ_WAVE_CYCLES", (0, 0))[0]):
    c = {k: v[0] for k, v in kern[n].items()}
    calls = max(v[1] for v in kern[n].values())
    if calls < 20: continue
    print("\n%s   (%d dispatches per counter pass)" % (n, calls))
    for k in sorted(c): print("  %-26s %16.1f" % (k, c[k]))
    d = []
    if c.get("SQ_INSTS_VALU"):
        tot = sum(c.get(x, 0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        d.append("valu_issue_us %.2f" % (c["SQ_INSTS_VALU"] * 2 / (1024 * 2.3e3)))
        d.append("all_issue_us %.2f" % (tot * 2 / (1024 * 2.3e3)))
        if c.get("SQ_WAVES"): d.append("valu_per_wave %.0f" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        d.append("lanes_active_per_valu %.1f" % (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]))
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        d.append("wait %.3f stall %.3f" % (c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w))
        if c.get("SQ_WAVES"): d.append("wave_life_us %.2f" % (w / c["SQ_WAVES"] * 4 / 2.3e3))   # SQ_WAVE_CYCLES counts in units of 4 cycles
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d.append("l2_hit %.3f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
    if c.get("SQ_LDS_IDX_ACTIVE"): d.append("lds_bank_conflict_share %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
    if n in dur and dur[n][1]:
        k_us = dur[n][0] / dur[n][1] / 1e3
        d.append("kernel_us_under_counters %.2f" % k_us)
        if c.get("SQ_INSTS_VALU"): d.append("valu_issue_frac %.3f" % (c["SQ_INSTS_VALU"] * 2 / (1024 * 2.3e3) / k_us))
    print("  derived: " + ", ".join(d))
    out_json.setdefault("kernels", {})[n] = {"dispatches_per_pass": calls, "counters": c, "derived": {x.rsplit(" ", 1)[0]: float(x.rsplit(" ", 1)[1]) for y in d for x in ([y] if y.count(" ") == 1 else [" ".join(y.split(" ")[i:i + 2]) for i in range(0, len(y.split(" ")), 2)])}}
out_json["kernel_source_sha"] = sys.argv[3]
out_json["what"] = "rocprofv3 --pmc per-dispatch averages, one frame in flight (tools/sq_profile.sh); valu_issue_frac = VALU wave-instructions x 2 cycles / (1024 SIMDs x 2.3 GHz) / the kernel's duration under the counter passes"
json.dump(out_json, open(sys.argv[4], "w"), indent=1, sort_keys=True)
PY
    find "$D" -name "*kernel_trace.csv" -delete; find "$D" -name "*counter_collection.csv" -size +20M -delete
}
cd "$REPO"
run projection "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-dropin --no-also --no-pipeline
W=syn128x10m MODES=one run projection_10m "$REPO/tools/shard_cost.py" 2 1
run bvh "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-dropin --engine bvh
W=syn128x1m MODES=one RANKS=4 run shard_1m "$REPO/tools/shard_cost.py" 2 8
W=syn128x10m MODES=one RANKS=4 run shard_10m "$REPO/tools/shard_cost.py" 2 8
head -50 "$OUT/projection_sq.txt"
