"""Build libark355.so (hipcc, gfx950 only) in-tree: snark_amd/libark355.so.

`python -m snark_amd.build [--force]`.  hipcc cross-compiles without a GPU; the .so is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
OUT = os.path.join(HERE, "libark355.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["capi.hip", "ark355_bls.hip", "ark355_bn.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DNDEBUG", "-Wno-unused-result"]
FLAGS += os.environ.get("ARK355_EXTRA_FLAGS", "").split()      # dev: e.g. -DARK_MSM_SEG=64


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libark355 is built for gfx950 only and has no CPU fallback")
    return exe


def _newest_source():
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith((".h", ".cuh", ".hip", ".cpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def is_fresh():
    return os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_source()


def build(force=False, verbose=True, resource_log=False):
    if not force and is_fresh():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src + ".o")
        cmd = [cc, *FLAGS, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if resource_log:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print("[snark_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if resource_log:
            with open(os.path.join(OBJDIR, src + ".resources.log"), "w") as f:
                f.write(r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # RCCL carries the cross-GPU exchange of the sharded prover (csrc/comm_impl.cuh)
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", OUT, *objs,
           "-L/opt/rocm/lib", "-lrccl"]
    if verbose:
        print("[snark_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    write_stats(OUT, verbose)
    return OUT


def write_stats(lib_path, verbose=False):
    """Instruction counts of the shipped kernels, read from the code objects inside the library just linked
    (tools/code_object_stats.py): libark355.stats.json next to it.  bench.py prices its integer roofline with the
    multiply-adds per mixed addition found there; the file also records the scratch accesses of the hot loops."""
    import json
    tools = os.path.normpath(os.path.join(HERE, "..", "tools"))
    path = os.path.splitext(lib_path)[0] + ".stats.json"
    try:
        sys.path.insert(0, tools)
        import code_object_stats
        st = code_object_stats.library_stats(lib_path)
        with open(path, "w") as f:
            json.dump(st, f, indent=1, sort_keys=True)
        if verbose:
            print("[snark_amd.build] kernel statistics ->", path, flush=True)
    except Exception as e:                                    # noqa: BLE001 -- statistics are not part of the product
        if verbose:
            print("[snark_amd.build] no kernel statistics (%s)" % e, flush=True)
    finally:
        if tools in sys.path:
            sys.path.remove(tools)


def build_host_mirror_exe(verbose=False):
    """tests/cpp/test_host_mirror: the C++ stand-in for a Rust host (host_mirror/*.hpp) linked against libark355.so --
    the parity harness of tests/test_host_mirror_cpp.py and the synthesis-in-the-loop reading of bench.py (`e2e`).
    Rebuilt whenever the sources differ from what the binary was built from (content hash, not mtime: a stale binary
    from another checkout must never run in place of the current sources)."""
    import hashlib
    root = os.path.normpath(os.path.join(HERE, ".."))
    cpp = os.path.join(root, "tests", "cpp")
    exe = os.path.join(cpp, "test_host_mirror")
    src = os.path.join(cpp, "test_host_mirror.cpp")
    deps = [src] + [os.path.join(root, "host_mirror", f) for f in ("relations.hpp", "snark.hpp")]
    h = hashlib.sha256()
    for d in deps + [os.path.join(INCLUDE, "ark355.h")]:
        h.update(open(d, "rb").read())
    stamp = exe + ".srchash"
    if not os.path.exists(exe) or not os.path.exists(stamp) or open(stamp).read() != h.hexdigest():
        cmd = ["g++", "-O2", "-pthread", "-std=c++17", src, "-o", exe, "-L" + HERE, "-lark355", "-Wl,-rpath," + HERE]
        if verbose:
            print("[snark_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(stamp, "w").write(h.hexdigest())
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, resource_log="--resources" in sys.argv))
