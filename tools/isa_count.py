"""Static instruction counts per kernel of a hipcc -S listing (tools: which section of a kernel is the VALU going to).
usage: hipcc --offload-arch=gfx950 <flags> -S --cuda-device-only -o x.s file.hip; python tools/isa_count.py x.s [name filter]"""
import re, sys
from collections import Counter
src = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
for (i, name), nxt in zip(starts, starts[1:] + [(len(src), "")]):
    if flt not in name:
        continue
    end = next((j for j in range(i, nxt[0]) if src[j].startswith(".Lfunc_end")), nxt[0])
    c = Counter()
    for l in src[i + 1:end]:
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        m = t.split()[0]
        kind = ("valu" if m.startswith("v_") else "salu" if m.startswith("s_") else "lds" if m.startswith("ds_")
                else "vmem" if m.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        c[kind] += 1
        if m in ("s_waitcnt", "s_barrier"):
            c[m] += 1
    short = re.sub(r"_ZN2ls12_GLOBAL__N_1\d+", "", name)[:60]
    print(f"{short:62s} {dict(c)}")
