"""Names for the '(unknown)' frames of a glog stack trace, from a map of the same program taken in another run.

  python tools/resolve_stack.py gpurun_out/r5e_graph_fuse512/out.txt gpurun_out/r6b/maps.txt

Address-space randomisation moves a process's mappings as a whole: libraries loaded in the same order keep their distances.
The record's own anchor: glog prints the signal trampoline's frame, __restore_rt = libc + 0x42520 (glibc 2.35, this image), which
gives libc's base in the crashed process; every other library's base follows from its distance to libc in the probe's map
(tools/maps_probe.py, run under the same profiler).  A frame whose address falls inside a library that way is looked up in that
file's dynamic symbols (nm -D; the nearest symbol below).  Frames that fall in no library are printed as such: the method claims
nothing it cannot check -- the check is the frames whose names glog DID print (lsi::frame_graph_close ... in liblidarshooter_hip.so,
lsh_stream_frames in liblidarshooter_host.so): they must land inside those libraries.
"""
import bisect
import re
import subprocess
import sys

RESTORE_RT = 0x42520   # __restore_rt in /usr/lib/x86_64-linux-gnu/libc.so.6 of this image (nm -D does not list it: readelf -s / objdump)


def load_maps(path):
    libs = {}
    for ln in open(path):
        if ln.startswith("#"):
            continue
        a, b, off, name = ln.split(None, 3)
        name = name.strip()
        a, b, off = int(a, 16), int(b, 16), int(off, 16)
        lo, hi, o = libs.get(name, (a - off, b, off))
        libs[name] = (min(lo, a - off), max(hi, b), 0)
    return {n: (lo, hi) for n, (lo, hi, _) in libs.items()}


def symbols(path):
    try:
        out = subprocess.run(["nm", "-D", "--defined-only", "-C", path], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return []
    syms = []
    for ln in out.splitlines():
        p = ln.split(None, 2)
        if len(p) == 3 and p[1] in "TtWwiV":
            syms.append((int(p[0], 16), p[2]))
    return sorted(syms)


def main():
    trace, maps = sys.argv[1], sys.argv[2]
    frames = []
    for ln in open(trace):
        m = re.match(r"\s*(?:PC: )?@\s+(0x[0-9a-f]+)\s+(.*)", ln)
        if m:
            frames.append((int(m.group(1), 16), m.group(2).strip()))
    libs = load_maps(maps)
    libc = next(n for n in libs if n.endswith("/libc.so.6"))
    # the anchor: the frame after the signal handler's own is the trampoline
    tramp = [a for a, _ in frames if (a & 0xFFF) == (RESTORE_RT & 0xFFF)]
    if not tramp:
        sys.exit("no frame ends in 0x%03x: cannot anchor libc" % (RESTORE_RT & 0xFFF))
    libc_crash = tramp[0] - RESTORE_RT
    shift = libc_crash - libs[libc][0]
    print(f"anchor: __restore_rt at {tramp[0]:#x} -> libc base {libc_crash:#x} in the record, {libs[libc][0]:#x} in the probe (shift {shift:+#x})")
    spans = sorted((lo + shift, hi + shift, n) for n, (lo, hi) in libs.items())
    starts = [s[0] for s in spans]
    cache = {}
    for addr, said in frames:
        i = bisect.bisect_right(starts, addr) - 1
        where = "in no library of the probe's map"
        if i >= 0 and spans[i][0] <= addr < spans[i][1]:
            lo, hi, name = spans[i]
            off = addr - lo
            if name not in cache:
                cache[name] = symbols(name if not name.startswith("/tmp/code/") else name[name.index("/repo/") + 6:])
            syms = cache[name]
            j = bisect.bisect_right([s[0] for s in syms], off) - 1
            sym = f"{syms[j][1]} + {off - syms[j][0]:#x}" if j >= 0 else "?"
            where = f"{name.rsplit('/', 1)[-1]} + {off:#x}  [{sym}]"
        print(f"{addr:#x}  {said:55.55s} -> {where}")


if __name__ == "__main__":
    main()
