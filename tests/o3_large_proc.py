"""Helper process of tests/test_gpu_o3_large.py::test_s2_2p22_*: one BASELINE configs[2]-size statement, key from the
oracle's generator, `ark355_prove` AND `ark355_prove_sharded` (real RCCL, world size 1, both exchange modes) against
`cbase.prove`, every proof through the Groth16 equation.  Runs in its own process -- like tests/rccl_single_rank.py --
so that the communicator, its proxy threads and the 60 GB of window tables live and die with a short-lived process
instead of the pytest session."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    n = int(sys.argv[1])
    import torch
    assert torch.cuda.is_available()
    import snark_amd
    import o3_cases as O
    from oracle import synthetic as S
    from oracle.fields import BLS12_381 as C
    lib = snark_amd.lib()
    ctx = lib.ctx_create(0)
    tm = {}
    try:
        O.check_instance(lib, ctx, C, S.mulchain_csr(C.r, n), [(0xC0FFEE, C.r - 0x22)], sharded=True, timing=tm)
    finally:
        lib.ctx_destroy(ctx)
    print("o3_large_ok n=%d oracle_setup_s=%.1f key_load_s=%.1f prove_and_oracle_s=%.1f" % (
        n, tm["oracle_setup_s"], tm["key_load_s"], tm["prove_and_oracle_s"]))


if __name__ == "__main__":
    main()
