"""Which CPUs does this box's GPU sit next to, and where does the process run before / after bench.py's pinning?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libc = ctypes.CDLL(None)
print("cpus allowed at start:", len(os.sched_getaffinity(0)), "on cpu", libc.sched_getcpu())
for n in sorted(os.listdir("/sys/devices/system/node")):
    if n.startswith("node"):
        print(n, open(f"/sys/devices/system/node/{n}/cpulist").read().strip())
import torch
import bench
print(bench.pin_to_gpu_numa_node(0))
aff = sorted(os.sched_getaffinity(0))
print("allowed now:", len(aff), aff[:4], "...", aff[-4:], "on cpu", libc.sched_getcpu())
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
import threading
print("threads:", threading.active_count(), "on cpu", libc.sched_getcpu())
for t in sorted(os.listdir("/proc/self/task"))[:40]:
    try:
        st = open(f"/proc/self/task/{t}/status").read()
        al = [l for l in st.split("\n") if l.startswith("Cpus_allowed_list")][0]
        nm = open(f"/proc/self/task/{t}/comm").read().strip()
        print(" task", t, nm, al)
    except Exception as e:
        print(" task", t, e)
