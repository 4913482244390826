#!/usr/bin/env python3
"""Where does a bucket-accumulation launch spend its time?  Dev tool for an EXPERIMENT build of the library (a patched copy
with per-wave timestamps in msm_accumulate28_kernel and the reader `ark355_debug_acc_trace`):
one stand-alone G1 MSM over resident tables, then per wave {start, end, hardware id, iterations with a bucket boundary}.
Prints the launch span, the distribution of wave durations, the start / end ramps and per-XCD figures.
usage: ARK355_LIB=variants/lib_exp_trace.so python tools/acc_trace.py [--log-n 20] [--group 1]"""
import argparse
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from snark_amd import lib as load_lib, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381")
    args = ap.parse_args()
    cv = params.CURVES[args.curve]
    L = load_lib()
    ctx = L.ctx_create(0)
    sz = L.sizes(cv.curve_id)
    rnd = random.Random(0x355)
    n = 1 << args.log_n
    psz = sz["g1"]
    nd = min(n, 1 << 18)
    ss = [rnd.getrandbits(60) + 1 for _ in range(nd)]
    distinct = np.frombuffer(L.fixed_base_mul(ctx, cv.curve_id, 1, cv.g1_gen_raw(), b"".join(cv.fr_canon(s) for s in ss), nd, psz),
                             dtype=np.uint8).reshape(nd, psz)
    bases = np.tile(distinct, (n // nd, 1))
    h = L.bases_load(ctx, cv.curve_id, 1, np.ascontiguousarray(bases).reshape(-1), n)
    g = np.random.default_rng(7)
    ks = g.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64, endpoint=False)
    ks[:, 3] &= np.uint64((1 << (cv.r.bit_length() - 1 - 192)) - 1)
    kd = torch.from_numpy(ks.view(np.uint8).reshape(-1).copy()).cuda()
    for _ in range(3):
        L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)
    torch.cuda.synchronize()
    print("accumulation kernel: %.3f ms (library's own event timing)" % L.kernel_stats(ctx)["accumulate_ms"])
    try:
        fn = L.dll.ark355_debug_acc_trace
    except AttributeError:
        print("this library has no trace (not an experiment build)")
        return
    words = 1 << 16
    buf = (C.c_uint64 * words)()
    fn.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
    rc = fn(buf, words)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
    a = a[a[:, 1] > 0]
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    base = t0.min()
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds (100 MHz counter)
    dur = en - st
    print("waves traced: %d   launch span %.1f us   first start .. last start %.1f us   first end %.1f us" % (
        len(a), en.max(), st.max(), en.min()))
    q = np.percentile(dur, [0, 5, 25, 50, 75, 95, 100])
    print("wave duration us: min %.1f  p5 %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  max %.1f" % tuple(q))
    bnd = a[:, 3].astype(np.int64)
    print("iterations with a boundary per wave: mean %.1f (min %d, max %d); corr(duration, boundaries) = %.2f" % (
        bnd.mean(), bnd.min(), bnd.max(), np.corrcoef(dur, bnd)[0, 1] if bnd.std() > 0 else 0.0))
    # rounds: waves that started after the first wave ended
    second = st > en.min() - 1.0
    print("waves started at t=0 (+-20 us): %d; started later: %d (mean start %.1f us)" % (
        int((st < 20).sum()), int((st >= 20).sum()), st[st >= 20].mean() if (st >= 20).any() else 0))
    for name, sel in (("first round", st < 20), ("later rounds", st >= 20)):
        if sel.any():
            print("  %-12s duration: mean %.1f us, p5 %.1f, p95 %.1f" % (name, dur[sel].mean(), np.percentile(dur[sel], 5), np.percentile(dur[sel], 95)))
    xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = a[:, 2].astype(np.int64) & 0xFFFFFFFF
    for x in sorted(set(xcc.tolist())):
        s = xcc == x
        print("  XCC %d: %4d waves, mean duration %.1f us, last end %.1f us" % (x, int(s.sum()), dur[s].mean(), en[s].max()))
    # occupancy over time: how many traced waves are alive in each 50 us bin
    edges = np.arange(0, en.max() + 50, 50)
    alive = [(int(((st <= e) & (en > e)).sum())) for e in edges]
    print("alive waves per 50 us: " + " ".join(str(v) for v in alive))
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 1
    simd = (hw >> 4) & 3
    slot = hw & 0xF
    print("distinct (xcc, se, cu) seen: %d" % len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))
    # the two first-round waves of every SIMD: who finishes first?
    first = st < 20
    groups = {}
    for i in np.nonzero(first)[0]:
        groups.setdefault((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])), []).append(i)
    pairs = [g for g in groups.values() if len(g) == 2]
    if pairs:
        lo = np.array([min(dur[g[0]], dur[g[1]]) for g in pairs])
        hi = np.array([max(dur[g[0]], dur[g[1]]) for g in pairs])
        older_first = np.mean([1.0 if (dur[g[0]] < dur[g[1]]) == (g[0] < g[1]) else 0.0 for g in pairs])
        print("first round, per SIMD (%d pairs; group sizes %s): faster wave mean %.1f us (p5 %.1f, p95 %.1f), slower wave mean %.1f us (p5 %.1f, p95 %.1f);"
              " the wave of the LOWER workgroup index is the faster one in %.0f %% of the pairs" % (
                  len(pairs), sorted(set(len(g) for g in groups.values())), lo.mean(), np.percentile(lo, 5), np.percentile(lo, 95), hi.mean(),
                  np.percentile(hi, 5), np.percentile(hi, 95), 100 * older_first))
        # per CU: sum over its 8 first-round waves
        cus = {}
        for k, g in groups.items():
            cus.setdefault(k[:4], []).extend(g)
        cu_mean = np.array([dur[g].mean() for g in cus.values()])
        print("first round, mean wave duration per CU: min %.1f  median %.1f  max %.1f us" % (cu_mean.min(), np.median(cu_mean), cu_mean.max()))
    for name, sel in (("wave slot id", slot), ("simd", simd)):
        vals = sorted(set(sel[first].tolist()))
        print("first round, mean duration by %s: " % name + "  ".join("%d: %.0f (%d)" % (v, dur[first & (sel == v)].mean(), int((first & (sel == v)).sum())) for v in vals))
    out = os.environ.get("ARK355_TRACE_NPZ")
    if out:
        np.savez_compressed(out, trace=a)
    L.dll.ark355_bases_free(h)
    L.ctx_destroy(ctx)


if __name__ == "__main__":
    main()
