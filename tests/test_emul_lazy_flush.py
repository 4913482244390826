"""The lazy (raw-limb) bucket flush of the radix-2^28 accumulation kernels -- compile-time variant -DARK_LAZY_FLUSH=1
(msm28_impl.cuh: runs are stored as they stand, msm_unlazy28_kernel converts every slot once in front of the merge) with
the key / row-index prefetch knobs on -- must give the same bytes as the default build: resident-table MSM edge cases
(P + P, P - P, infinity inside buckets) for both groups and curves, whole proofs, and a large window over the two-level
bucket reduction (the c = 20 configuration it exists for, scaled down)."""
import os
import sys

import numpy as np
import pytest

import parity_cases as pc
from conftest import ROOT
from oracle import synthetic as S
from oracle.fields import BLS12_381, BN254


@pytest.fixture(scope="module")
def lazy_lib():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    from snark_amd._binding import Lib
    return Lib(build_emul.build(extra_flags=["-DARK_LAZY_FLUSH=1", "-DARK_ACC_PREFETCH_KEY=1", "-DARK_G2L28_PREFETCH=1"], tag="lazy"))


@pytest.fixture(scope="module")
def lazy_ctx(lazy_lib):
    ctx = lazy_lib.ctx_create(0)
    yield ctx
    lazy_lib.ctx_destroy(ctx)


def _to_dev(b):
    a = np.frombuffer(b, dtype=np.uint8).copy()
    return a.ctypes.data, a


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("group", [1, 2])
def test_lazy_flush_resident_msm_edge_cases(lazy_lib, lazy_ctx, C, group):
    pc.resident_msm_edge_case(lazy_lib, lazy_ctx, C, group, 24, _to_dev)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_lazy_flush_prove_bytes_equal_oracle(lazy_lib, lazy_ctx, C):
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 13)
    pc.prove_case(lazy_lib, lazy_ctx, C, A, B, Cm, z, ell, rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd), (0, 5)))


def test_lazy_flush_large_window_two_level_reduction(lazy_lib, lazy_ctx):
    import o3_cases as O
    with lazy_lib.policy(lazy_ctx, MSM_C=13, MSM_SEG=37):
        O.check_resident_msm(lazy_lib, lazy_ctx, BLS12_381, 1, 1100, _to_dev, seed=5)
        O.check_resident_msm(lazy_lib, lazy_ctx, BLS12_381, 2, 1050, _to_dev, seed=6)


def test_lazy_flush_one_stream_batched_tails(lazy_lib, lazy_ctx):
    """A one-stream proof of the lazy-flush build: the raw runs are converted in front of the BATCHED G1 tails
    (msm_reduce_phase_batch) -- same bytes as the oracle, incl. the heavy buckets of the all-equal DummyCircuit."""
    C = BLS12_381
    with lazy_lib.policy(lazy_ctx, SCHED=0):
        pc.prove_case(lazy_lib, lazy_ctx, C, *S.mulchain_direct(C.r, 29), rs=((7, 9),))
        pc.prove_case(lazy_lib, lazy_ctx, C, *S.cs_to_instance(S.dummy_cs(C.r, 40)), rs=((3, 4),))
