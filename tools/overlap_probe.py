"""From a rocprofv3 --kernel-trace csv of bench.py: for every pair of hardware queues that carried the frames' kernels,
how much of the time both had a kernel running (three frames in flight on three streams: all three pairs should overlap).
usage: overlap_probe.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ls::" in n and ("k_project" in n or "k_pack" in n):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], "P" if "k_project<" in n or "k_project_finish" not in n and "k_project" in n else "K"))
rows.sort()
# the last 3000 dispatches: the long windows at the end of bench.py
rows = rows[-9000:]
qs = collections.Counter(r[2] for r in rows)
print("queues:", dict(qs))
iv = collections.defaultdict(list)
for s, e, q, _ in rows: iv[q].append((s, e))
def busy(a):
    return sum(e - s for s, e in a)
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: t += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
ks = sorted(iv)
span = rows[-1][1] - rows[0][0]
for q in ks: print("queue", q, "busy %.2f of the span" % (busy(iv[q]) / span))
for x in range(len(ks)):
    for y in range(x + 1, len(ks)):
        print("queues", ks[x], ks[y], "both busy %.2f of the span" % (inter(iv[ks[x]], iv[ks[y]]) / span))
