"""VALU / all instructions of one kernel attributed to source lines (hipcc -gline-tables-only -S listing).
usage: python tools/isa_lines.py listing.s <mangled-name substring> [min count]"""
import re, sys
from collections import Counter
src = open(sys.argv[1]).read().split("\n")
flt, mn = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5
files = {}
for l in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
i = next(k for k, l in enumerate(src) if re.match(r"^_Z\w+:", l) and flt in l)
j = next(k for k in range(i, len(src)) if src[k].startswith(".Lfunc_end"))
cur, cnt, cntv = None, Counter(), Counter()
for l in src[i:j]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    cnt[cur] += 1
    if t.startswith("v_"):
        cntv[cur] += 1
print("total valu", sum(cntv.values()), "all", sum(cnt.values()))
cache = {}
for (f, ln), c in sorted(cnt.items(), key=lambda x: (x[0][0], x[0][1])):
    if c >= mn:
        txt = ""
        if f.endswith((".hip", ".h", ".cpp")):
            import glob
            if f not in cache:
                g = glob.glob(f"/root/repo/lidarshooter_amd/csrc/{f}")
                cache[f] = open(g[0]).read().split("\n") if g else []
            txt = cache[f][ln - 1].strip()[:100] if 0 < ln <= len(cache[f]) else ""
        print(f"{f}:{ln:5d} valu {cntv[(f, ln)]:4d} all {c:4d}  {txt}")
