#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: build tests/emul/libark355_emul.so -- the library's own sources compiled with
g++ -DARK_EMUL against the single-threaded HIP emulator (hip_emul.h).  Used only by
`pytest -m "not gpu"` to exercise kernel/orchestration logic without a GPU; never loaded by snark_amd."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "snark_amd", "csrc")
OUT = os.path.join(HERE, "libark355_emul.so")
SRCS = [os.path.join(CSRC, f) for f in ("capi.hip", "ark355_bls.hip", "ark355_bn.hip")] + [
    os.path.join(HERE, "hip_emul.cpp"), os.path.join(HERE, "rccl_emul.cpp")]


def newest_src():
    t = 0
    for d in (CSRC, HERE, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".cuh", ".hip", ".cpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force=False, extra_flags=(), tag=""):
    """tag / extra_flags: a variant build (e.g. -DARK_LAZY_FLUSH=1) next to the default one: libark355_emul_<tag>.so"""
    out = OUT if not tag else OUT.replace(".so", "_%s.so" % tag)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest_src():
        return out
    # several processes may get here at once (the ranks of a multi-process test after a source edit): one builds, the others
    # wait for the lock and find the library fresh; the library appears atomically (os.replace), never half written
    import fcntl
    with open(out + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest_src():
            return out
        return _build_locked(out, extra_flags, tag)


def _build_locked(out, extra_flags, tag):
    objs = []
    flags = list(extra_flags) + ["-O2", "-std=c++17", "-fPIC", "-DARK_EMUL", "-DARK_MSM_HEAVY_SPAN=2", "-DARK_MSM_HEAVY_GRID=3u", "-DARK_MSM_TWO_LEVEL_MIN=64u", "-I", HERE, "-I", CSRC, "-w"]

    def cc(src):
        obj = os.path.join(HERE, os.path.basename(src) + (".%s" % tag if tag else "") + ".emul.o")
        subprocess.check_call(["g++", "-x", "c++", *flags, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, SRCS))
    tmp = out + ".tmp.%d" % os.getpid()
    subprocess.check_call(["g++", "-shared", "-Wl,-Bsymbolic", "-o", tmp, *objs, "-lpthread", "-lrt"])
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
