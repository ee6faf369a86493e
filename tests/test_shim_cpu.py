"""tests/shim/librccl_shim.so without a GPU: what of it needs none -- the rendezvous of ncclCommInitRank, ncclCommSplit (one colour,
one colour per rank, a split that fails on one rank only), ncclCommCount / UserRank -- between two real processes.  The shim is
test infrastructure (tests/shim/rccl_shim.cpp says what it is for); its collectives move device memory and are exercised by
tests/test_gpu_group_shim.py on the GPU box."""
import ctypes as C
import json
import os
import subprocess
import sys

from conftest import ROOT

SHIM = os.path.join(ROOT, "tests", "shim", "librccl_shim.so")

PROG = r'''
import ctypes as C, json, os, sys, time
L = C.CDLL(sys.argv[1]); rank = int(sys.argv[2]); idf = sys.argv[3]; world = int(sys.argv[4]) if len(sys.argv) > 4 else 2
class Id(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Id, C.c_int]
L.ncclCommSplit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
L.ncclCommUserRank.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
L.ncclCommDestroy.argtypes = [C.c_void_p]
u = Id()
if rank == 0:
    assert L.ncclGetUniqueId(C.byref(u)) == 0
    open(idf + ".tmp", "wb").write(bytes(u)); os.rename(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    C.memmove(C.byref(u), open(idf, "rb").read(), 128)
def facts(c):
    n, r = C.c_int(), C.c_int()
    assert L.ncclCommCount(c, C.byref(n)) == 0 and L.ncclCommUserRank(c, C.byref(r)) == 0
    return [n.value, r.value]
out = {}
c = C.c_void_p()
out["init"] = L.ncclCommInitRank(C.byref(c), world, u, rank)
out["world"] = facts(c)
d = C.c_void_p()
out["split_same"] = L.ncclCommSplit(c, 0, world - 1 - rank, C.byref(d), None)       # one colour, keys reversed: the ranks swap
out["same"] = facts(d) if d else None
e = C.c_void_p()
out["split_own"] = L.ncclCommSplit(c, 7 + rank, 0, C.byref(e), None)        # a colour per rank: two communicators of one
out["own"] = facts(e) if e else None
f = C.c_void_p()
out["split_none"] = L.ncclCommSplit(c, -1, 0, C.byref(f), None)            # NCCL_SPLIT_NOCOLOR
out["none_is_null"] = not f
v = C.c_int()
L.ncclGetVersion(C.byref(v)); out["version"] = v.value
for x in (d, e, c):
    if x: L.ncclCommDestroy(x)
print(json.dumps(out))
'''


def _two(tmp_path, env_extra=None, world=2):
    assert os.path.exists(SHIM), "make -C tests/shim (build() does it)"
    env = dict(os.environ, LS_SHIM_TIMEOUT_S="30")
    env.update(env_extra or {})
    idf = str(tmp_path / "id")
    ps = [subprocess.Popen([sys.executable, "-c", PROG, SHIM, str(r), idf, str(world)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    res = []
    for p in ps:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    return res


def test_shim_rendezvous_and_splits_between_two_processes(tmp_path):
    r0, r1 = _two(tmp_path)
    for r, v in enumerate((r0, r1)):
        assert v["init"] == 0 and v["world"] == [2, r] and v["version"] == 1
        assert v["split_same"] == 0 and v["same"] == [2, 1 - r]
        assert v["split_own"] == 0 and v["own"] == [1, 0]
        assert v["split_none"] == 0 and v["none_is_null"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("lsshim-")]   # every segment's name is gone once its ranks have met


def test_shim_split_that_fails_on_one_rank_only(tmp_path):
    r0, r1 = _two(tmp_path, {"LS_SHIM_FAIL_SPLIT_RANK": "1"})
    assert r0["split_same"] == 0 and r0["same"] == [2, 1]          # the peer's call succeeds ...
    assert r1["split_same"] != 0 and r1["same"] is None            # ... the injected rank's fails after the collective part
    assert r0["split_none"] == 0 and r1["split_none"] == 0         # and the parent communicator goes on working for both


def test_shim_split_refused_everywhere(tmp_path):
    r0, r1 = _two(tmp_path, {"LS_SHIM_NO_SPLIT": "1"})
    assert r0["split_same"] != 0 and r1["split_same"] != 0 and r0["world"] == [2, 0] and r1["world"] == [2, 1]


def test_shim_eight_ranks(tmp_path):
    """the world a node has: eight processes meet, split with reversed keys (rank r becomes 7 - r), and split into singletons"""
    res = _two(tmp_path, world=8)
    for r, v in enumerate(res):
        assert v["init"] == 0 and v["world"] == [8, r]
        assert v["split_same"] == 0 and v["same"] == [8, 7 - r]
        assert v["split_own"] == 0 and v["own"] == [1, 0] and v["none_is_null"]
