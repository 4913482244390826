#!/usr/bin/env python3
"""MSM micro-benchmark (BASELINE.json's second metric "MSM Mscalar-mul/s"): resident bases
(ark355_bases_load: window tables) + ark355_msm_dev, sizes 2^16..2^24, G1 and G2, BLS12-381 (or --curve bn254), three
scalar distributions:
  uniform  uniform mod r (primary);
  equal    every scalar the same (the reference's DummyCircuit witness, relations/src/sr1cs/mod.rs:306-313: one bucket
           per window takes every term -- the heavy-merge path);
  boolean  90 % in {0, 1}, 10 % uniform ("boolean witness").
Bases: P_i = k_i * G made on the device for the first min(n, 2^20) rows, tiled beyond that (the distribution of the
bases does not change Pippenger's cost; byte-exactness at 2^20 / 2^22 is tests/test_gpu_o3_large.py's job).  Every
result up to 2^20 is checked against (sum k_i s_i) * G.  The time is the latency of one ark355_msm_dev call: digit
sort, bucket accumulation, bucket reduction, normalisation and the D2H of the result.
Prints one line per (group, size, distribution) and, with --json, a JSON list.  Dev tool; run on an MI355X."""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from snark_amd import lib as load_lib, params

DISTINCT = 1 << 20


def scalars(cv, n, dist, rnd):
    top = (1 << (cv.r.bit_length() - 1 - 192)) - 1
    if dist == "equal":
        return np.tile(np.frombuffer(cv.fr_canon(rnd.randrange(cv.r)), dtype="<u8"), n).reshape(n, 4)
    g = np.random.default_rng(rnd.getrandbits(63))
    uni = g.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64, endpoint=False)
    uni[:, 3] &= np.uint64(top)
    if dist == "uniform":
        return uni
    raw = np.zeros((n, 4), dtype="<u8")
    kind = g.integers(0, 256, size=n, dtype=np.uint8)
    raw[:, 0] = kind & 1
    sel = kind >= 230
    raw[sel] = uni[sel]
    return raw


def to_ints(a):
    return [int(x[0]) | (int(x[1]) << 64) | (int(x[2]) << 128) | (int(x[3]) << 192) for x in a.tolist()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--min-log", type=int, default=16)
    ap.add_argument("--max-log", type=int, default=24)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--groups", default="1,2")
    ap.add_argument("--dists", default="uniform,equal,boolean")
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-check", action="store_true", help="timing experiments with libraries that compute wrong sums on purpose")
    args = ap.parse_args()
    cv = params.CURVES[args.curve]
    L = load_lib()
    ctx = L.ctx_create(0)
    sz = L.sizes(cv.curve_id)
    rnd = random.Random(0x355)
    rows = []
    for group in [int(g) for g in args.groups.split(",")]:
        psz = sz["g1"] if group == 1 else sz["g2"]
        gen = cv.g1_gen_raw() if group == 1 else cv.g2_gen_raw()
        nd = min(DISTINCT, 1 << args.max_log)
        ss = [rnd.getrandbits(60) + 1 for _ in range(nd)]
        distinct = np.frombuffer(L.fixed_base_mul(ctx, cv.curve_id, group, gen, b"".join(cv.fr_canon(s) for s in ss), nd, psz),
                                 dtype=np.uint8).reshape(nd, psz)
        for lg in range(args.min_log, args.max_log + 1, args.step):
            n = 1 << lg
            bases = distinct[:n] if n <= nd else np.tile(distinct, (n // nd, 1))
            h = L.bases_load(ctx, cv.curve_id, group, np.ascontiguousarray(bases).reshape(-1), n)
            del bases
            for dist in args.dists.split(","):
                ks = scalars(cv, n, dist, rnd)
                kd = torch.from_numpy(ks.view(np.uint8).reshape(-1).copy()).cuda()
                torch.cuda.synchronize()
                out = L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)                 # warm-up (+ correctness)
                checked = ""
                if n <= nd and not args.no_check:
                    ki = to_ints(ks)
                    expect = L.fixed_base_mul(ctx, cv.curve_id, group, gen,
                                              cv.fr_canon(sum(k * s for k, s in zip(ki, ss)) % cv.r), 1, psz)
                    assert out == expect, "MSM result differs from (sum k_i s_i) * G"
                    checked = "  (checked)"
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.reps * 1e3
                acc = L.kernel_stats(ctx)["accumulate_ms"]
                print("%s G%d n=2^%d %-8s %9.3f ms per MSM (accumulation kernel %7.3f ms) %8.1f M scalar-mul/s%s" % (
                    args.curve, group, lg, dist, ms, acc, n / ms / 1e3, checked), flush=True)
                rows.append({"curve": args.curve, "group": group, "log_n": lg, "dist": dist, "ms": ms,
                             "accumulate_ms": acc, "mscalar_mul_per_s": n / ms / 1e3, "checked": bool(checked)})
                del kd
            L.dll.ark355_bases_free(h)
    L.ctx_destroy(ctx)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
