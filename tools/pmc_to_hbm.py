#!/usr/bin/env python3
"""tools_pmc.sh summary (FETCH_SIZE / WRITE_SIZE per kernel, KiB) -> the *_hbm.json layout bench.py reads.
usage: pmc_to_hbm.py gpurun_out/pmc_<tag>/summary.txt profiles/<tag>_hbm.json "<what was profiled>" """
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
acc = {}
for line in open(sys.argv[1]):
    m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
    if not m:
        continue
    k = acc.setdefault(m.group(1).strip(), {"FETCH_SIZE_KiB_per_launch": 0.0, "WRITE_SIZE_KiB_per_launch": 0.0, "launches": 0})
    k[m.group(2) + "_KiB_per_launch"] = float(m.group(3))
    k["launches"] = max(k["launches"], int(m.group(4)))
for k in acc.values():
    k["hbm_bytes_per_launch"] = (2.0 * k["FETCH_SIZE_KiB_per_launch"] + k["WRITE_SIZE_KiB_per_launch"]) * 1024.0
json.dump({"note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE reads 1/2); " + sys.argv[3],
           "kernel_source_sha": bench.kernel_source_sha(), "kernels": acc}, open(sys.argv[2], "w"), indent=1)
print(sys.argv[2], len(acc), "kernels, sha", bench.kernel_source_sha())
