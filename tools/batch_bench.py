#!/usr/bin/env python3
"""Throughput of the C-ABI batch entry point (ark355_prove_batch): `count` proofs of one 2^log_n-constraint circuit
with host-resident assignments (H2D included), `inflight` in flight.  First and last proof are checked against the
trapdoor closed form.  Dev tool; run on an MI355X."""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from snark_amd import params, synthetic
from snark_amd.groth16 import Groth16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--count", type=int, default=16)
    ap.add_argument("--inflight", type=int, default=4)
    ap.add_argument("--curve", default="bls12_381")
    args = ap.parse_args()
    cv = params.CURVES[args.curve]
    n = 1 << args.log_n
    g = Groth16(cv, device=0)
    r1, z = synthetic.mulchain(cv, n)
    rnd = random.Random(7)
    pk, vk = g.circuit_specific_setup(r1, lambda: rnd.randrange(1, cv.r), keep_trapdoor=True)
    zb = synthetic.z_to_mont_bytes(cv, z)
    rs = [(rnd.randrange(cv.r), rnd.randrange(cv.r)) for _ in range(args.count)]
    g.prove_batch(pk, r1, [zb] * args.inflight, rs=rs[:args.inflight], inflight=args.inflight)      # warm-up
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    proofs = g.prove_batch(pk, r1, [zb] * args.count, rs=rs, inflight=args.inflight)
    dt = time.perf_counter() - t0
    for k in (0, args.count - 1):
        assert proofs[k] == g.prove_closed_form(pk, z, rs[k][0], rs[k][1]), k
    print("ark355_prove_batch %s n=2^%d: %d proofs, %d in flight, host z (H2D included): %.2f ms per proof, "
          "%.2f M constraints/s (first and last proof == closed form)" % (
              args.curve, args.log_n, args.count, args.inflight, dt / args.count * 1e3, n * args.count / dt / 1e6))
    g.close()


if __name__ == "__main__":
    main()
