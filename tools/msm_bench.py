#!/usr/bin/env python3
"""MSM micro-benchmark: resident bases (ark355_bases_load: window tables) + ark355_msm_dev over uniform
scalars, sizes 2^16..2^20 (default), G1 and G2, BLS12-381 (or --curve bn254).  Bases are s_i*G made on the device
(ark355_fixed_base_mul), so every result is checked against (sum k_i s_i)*G computed by the product's own host code
path (one fixed-base multiplication).  Prints one line per (group, size): ms per MSM and M scalar-mul/s.
Dev tool; run on an MI355X."""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from snark_amd import lib as load_lib, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--min-log", type=int, default=16)
    ap.add_argument("--max-log", type=int, default=20)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    cv = params.CURVES[args.curve]
    L = load_lib()
    ctx = L.ctx_create(0)
    sz = L.sizes(cv.curve_id)
    rnd = random.Random(0x355)
    for group in (1, 2):
        psz = sz["g1"] if group == 1 else sz["g2"]
        gen = cv.g1_gen_raw() if group == 1 else cv.g2_gen_raw()
        for lg in range(args.min_log, args.max_log + 1, args.step):
            n = 1 << lg
            ss = [rnd.getrandbits(60) + 1 for _ in range(n)]
            ks = [rnd.getrandbits(255) % cv.r for _ in range(n)]
            bases = L.fixed_base_mul(ctx, cv.curve_id, group, gen, b"".join(cv.fr_canon(s) for s in ss), n, psz)
            h = L.bases_load(ctx, cv.curve_id, group, bases, n)
            kd = torch.from_numpy(np.frombuffer(b"".join(cv.fr_canon(k) for k in ks), dtype=np.uint8).copy()).cuda()
            out = L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)                 # warm-up + correctness
            expect = L.fixed_base_mul(ctx, cv.curve_id, group, gen,
                                      cv.fr_canon(sum(k * s for k, s in zip(ks, ss)) % cv.r), 1, psz)
            assert out == expect, "MSM result differs from (sum k_i s_i) * G"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                L.msm_dev(ctx, h, kd.data_ptr(), n, 0, psz)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            print("%s G%d n=2^%d: %8.3f ms per MSM  %8.1f M scalar-mul/s  (checked)" % (
                args.curve, group, lg, ms, n / ms / 1e3), flush=True)
            L.dll.ark355_bases_free(h)
    L.ctx_destroy(ctx)


if __name__ == "__main__":
    main()
