#!/usr/bin/env python3
"""Condense a tools_profile.sh output directory into the small summaries committed under profiles/.

  python tools/prof_summary.py gpurun_out/prof_<tag> profiles/<tag>

Writes <out>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, kernel names shortened) and
<out>_hbm.json: per kernel, mean FETCH_SIZE / WRITE_SIZE per launch and the HBM bytes derived as
MI355X_MICROARCH.md section HBM prescribes: counters are in KiB, collected in separate --pmc passes, and on
gfx950 FETCH_SIZE reports half of the bytes fetched -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r"wrapped_(\w+?)_config", name)
    if m:
        return "rocprim::" + m.group(1)
    return name.split("(")[0][-60:]


def pmc(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirname, "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def overlap_excerpt(src, out):
    """A short excerpt of the kernel trace (start / end timestamps, hardware queue) around a moment at which frames on
    different queues overlap: the evidence that the reported step time is a throughput with several frames in flight,
    while `roofline.kernel_ms` is one kernel on its own."""
    rows = []
    for f in glob.glob(os.path.join(src, "stats", "*_kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"]),
                         row.get("Queue_Id", ""), row.get("Stream_Id", "")))
    rows.sort()
    proj = [i for i, r in enumerate(rows) if r[2].startswith("k_project<false")]
    best = None
    for n, i in enumerate(proj[len(proj) // 2:]):          # steady state: second half of the run
        live = [j for j in range(max(0, i - 12), i + 1) if rows[j][0] <= rows[i][0] < rows[j][1]]
        if len({rows[j][3] for j in live}) >= 2 and sum(1 for j in live if rows[j][2].startswith("k_project<false")) >= 2:
            best = i
            break
    if best is None:
        return
    lo, hi = max(0, best - 6), min(len(rows), best + 10)
    t0 = rows[lo][0]
    with open(out + "_overlap_excerpt.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "queue", "stream", "start_us", "end_us", "duration_us", "kernels_in_flight_at_start"])
        for j in range(lo, hi):
            st, en, name, q, sid = rows[j]
            live = sum(1 for k in range(max(0, j - 12), j + 1) if rows[k][0] <= st < rows[k][1])
            w.writerow([name, q, sid, f"{(st - t0) / 1e3:.2f}", f"{(en - t0) / 1e3:.2f}", f"{(en - st) / 1e3:.2f}", live])


def isolated_split(src, out):
    """Per kernel: the dispatches of the trace that ran ALONE (no other dispatch's [start, end) meets theirs) against those
    that shared the chip with another one (three frames in flight on three streams).  bench.py's `roofline.kernel_ms`
    is the first kind -- timed with one frame in flight -- so this is the figure of the same command it must agree with;
    the averages in *_kernel_stats.csv mix both kinds."""
    rows = []
    for f in glob.glob(os.path.join(src, "stats", "*_kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])))
    rows.sort()
    res = defaultdict(lambda: {"isolated": [], "overlapped": []})
    max_end_before = 0
    for i, (st, en, name) in enumerate(rows):
        alone = st >= max_end_before and (i + 1 == len(rows) or rows[i + 1][0] >= en)
        res[name]["isolated" if alone else "overlapped"].append((en - st) / 1e3)
        max_end_before = max(max_end_before, en)
    summary = {}
    for name, d in res.items():
        if not name.startswith("k_"):
            continue
        summary[name] = {k + "_calls": len(v) for k, v in d.items()}
        for k, v in d.items():
            if v:
                v.sort()
                summary[name][k + "_avg_us"] = round(sum(v) / len(v), 3)
                summary[name][k + "_median_us"] = round(v[len(v) // 2], 3)
    json.dump({"note": "from the rocprofv3 --kernel-trace of `bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-dropin`: "
                       "dispatches that ran alone vs dispatches that overlapped another one (durations in microseconds)",
               "kernels": summary}, open(out + "_kernel_isolated.json", "w"), indent=1)
    for k in sorted(summary):
        print("isolated/overlapped", k, summary[k])


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    rows = []
    for f in glob.glob(os.path.join(src, "stats", "*_kernel_stats.csv")):
        for row in csv.DictReader(open(f)):
            rows.append([short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"],
                         row["MinNs"], row["MaxNs"], row["StdDev"]])
    with open(out + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerows(rows)
    fetch = pmc(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = pmc(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    hbm = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0.0, 0))
        wv = write.get(k, (0.0, 0))
        hbm[k] = {"FETCH_SIZE_KiB_per_launch": f[0], "WRITE_SIZE_KiB_per_launch": wv[0], "launches": max(f[1], wv[1]),
                  "hbm_bytes_per_launch": (2.0 * f[0] + wv[0]) * 1024.0}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE reads 1/2)",
               "kernel_source_sha": bench.kernel_source_sha(),   # bench.py reports `traffic` only while the sources still hash to this
               "kernels": hbm}, open(out + "_hbm.json", "w"), indent=1)
    overlap_excerpt(src, out)
    isolated_split(src, out)
    for r in rows[:12]:
        print(r[0], r[1], r[3], r[4])
    for k, v in hbm.items():
        print(k, v)


if __name__ == "__main__":
    main()
