#!/usr/bin/env python3
"""Per-rank critical path of ONE sharded proof, measured on ONE GPU (VERDICT r3 "missing" #3 / "next" #4).

BASELINE configs[2] shards the MSM terms of a 2^22-constraint proof over 8 GPUs.  What bounds the speed-up is what a
single rank still has to do: its 1/G share of the five MSMs plus everything that is not sharded.  That per-rank time
needs no second GPU to measure: load shard g of G of the key (ark355_pk_load_shard) and call ark355_prove_shard, which
returns the rank's five partial sums without any collective.  The exchange that follows on a real node is one 960-byte
all-gather (or G-1 ring steps over <= 0.8 MB) plus the O(1) host tail, measured separately at world size 1.

  python tools/shard_rank_bench.py --log-n 22 --world 8 --ranks 0,7 [--steps 6]

Prints one JSON line per rank: median wall time of the C call, the library's phase timers, the shard's table layout.
Dev / measurement tool (run on an MI355X).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0,7")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--curve", default="bls12_381")
    ap.add_argument("--whole", action="store_true", help="also time the unsharded proof (shard 0 of 1) for the ratio")
    ap.add_argument("--wm", default="both", choices=["replicated", "dist", "both"],
                    help="witness map of the rank: replicated (rounds 1-3: every rank repeats the whole map), dist (the rank's 1/G "
                         "of the distributed map, exchanges as LOCAL copies of the same size: timing only, the sums are "
                         "meaningless), both")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available()
    from snark_amd import params, synthetic
    from snark_amd.groth16 import Groth16
    cv = params.CURVES[args.curve]
    n = 1 << args.log_n
    g = Groth16(cv, device=0)
    L = g.lib
    # the schedule ark355_prove_sharded runs every rank's proof as: the five-stream pipeline, unmeasured (a collective cannot
    # depend on one rank's measurements; prove_run, groth16_impl.cuh) -- ark355_prove_shard has no communicator and would
    # otherwise take the one-stream default of a lone proof
    L.ctx_set_policy(g.ctx, "SCHED", 1)
    r1, z = synthetic.mulchain(cv, n, seed=0x355)
    rnd = random.Random(1)
    t0 = time.perf_counter()
    pk, _ = g.circuit_specific_setup(r1, lambda: rnd.randrange(1, cv.r), keep_trapdoor=False)
    rh = g.load_r1cs(r1)
    zb = synthetic.z_to_mont_bytes(cv, z)
    ptr = ctypes.c_void_p()
    assert L.dll.ark355_host_alloc(len(zb), ctypes.byref(ptr)) == 0
    z_pinned = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(len(zb),))
    z_pinned[:] = np.frombuffer(zb, dtype=np.uint8)
    sys.stderr.write("[shard_rank_bench] statement + key ready in %.1f s\n" % (time.perf_counter() - t0))
    todo = [(int(x), args.world) for x in args.ranks.split(",")]
    if args.whole:
        todo.append((0, 1))
    variants = []
    for rank, world in todo:
        for wm in (("replicated", "dist") if args.wm == "both" else (args.wm,)):
            if wm == "dist" and world == 1:
                continue
            variants.append((rank, world, wm))
    for rank, world, wm in variants:
        # replicated: the key shard in the contiguous layout of rounds 1-3; dist: the layout of the distributed map and the
        # diagnostic loopback exchange (policy DWM_LOOPBACK: ark355_prove_shard has no communicator)
        L.ctx_set_policy(g.ctx, "SHARD_DIST_WM", 1 if wm == "dist" else 0)
        L.ctx_set_policy(g.ctx, "DWM_LOOPBACK", 1 if wm == "dist" else 0)
        h = L.pk_load(g.ctx, cv.curve_id, pk.ell, pk.w, pk.N, pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query,
                      pk.l_query, pk.vk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.vk.beta_g2, pk.vk.delta_g2,
                      shard=(rank, world))
        try:
            rs = [(cv.fr_canon(rnd.randrange(cv.r)), cv.fr_canon(rnd.randrange(cv.r))) for _ in range(args.steps + 2)]
            ts, tims = [], []
            for i, (r_, s_) in enumerate(rs):
                torch.cuda.synchronize()
                ta = time.perf_counter()
                L.prove_shard(g.ctx, cv.curve_id, h, rh, z_pinned, r1.m, r_, s_)
                dt = (time.perf_counter() - ta) * 1e3
                if i >= 2:
                    ts.append(dt)
                    tims.append(L.timings(g.ctx))
            ts.sort()
            med = ts[len(ts) // 2]
            tm = {k: round(sorted(t[k] for t in tims)[len(tims) // 2], 3) for k in tims[0]}
            print(json.dumps({"what": "ark355_prove_shard, host-pinned z -> partial sums, no collective",
                              "witness_map": wm + (" (exchanges as local copies of the same size: TIMING ONLY)" if wm == "dist" else ""),
                              "curve": args.curve, "n": n, "N": r1.domain_size, "shard": "%d/%d" % (rank, world),
                              "ms_median": round(med, 3), "ms_all": [round(x, 3) for x in ts],
                              "accumulate_ms": round(L.kernel_stats(g.ctx)["accumulate_ms"], 3),
                              "phases_overlapping_stream_segments_ms": tm, "tables": L.pk_table_info(h)}), flush=True)
        finally:
            L.dll.ark355_pk_free(h)
    L.dll.ark355_host_free(ptr)
    g.close()


if __name__ == "__main__":
    main()
