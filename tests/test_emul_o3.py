"""The O3 checks of tests/o3_cases.py at tiny sizes over the CPU emulator build of the library's sources (checker
only): keeps the logic of the BASELINE-size GPU tests (tests/test_gpu_o3_large.py) exercised on a machine without a GPU."""
import numpy as np
import pytest

import o3_cases as O
from oracle import synthetic as S
from oracle.fields import BLS12_381, BN254


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_instances_vs_o3_on_the_emulator(emul_lib, emul_ctx, C):
    # (the Groth16 equation -- emulated MSMs + the library's host pairing -- once per curve: it is the slow part here)
    O.check_instance(emul_lib, emul_ctx, C, S.mulchain_csr(C.r, 40), [(5, 7)], batch=2 if C is BLS12_381 else 0, inflight=1,
                     equation=False)
    if C is BLS12_381:
        O.check_instance(emul_lib, emul_ctx, C, S.dummy_csr(C.r, 64), [(0, 1)], equation=False)
    O.check_instance(emul_lib, emul_ctx, C, S.bench_lc_csr(C.r, 24), [(C.r - 1, C.r - 1)])


@pytest.mark.parametrize("group", [1, 2])
def test_resident_msm_vs_o3_on_the_emulator(emul_lib, emul_ctx, group):
    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a

    O.check_resident_msm(emul_lib, emul_ctx, BLS12_381, group, 200, to_dev)


def test_large_window_two_level_reduction_and_odd_segments(emul_lib, emul_ctx, emul_policy):
    """Window size 13 over window tables (4096 buckets: the two-level bucket reduction, level 1 with 16 buckets per
    lane) and a segment length that is not a power of two, against oracle/c -- the configuration the GPU reaches with
    ARK355_MSM_C=20 at 2^20 terms."""
    emul_policy.setenv("ARK355_MSM_C", "13")
    emul_policy.setenv("ARK355_MSM_SEG", "37")

    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a

    O.check_resident_msm(emul_lib, emul_ctx, BLS12_381, 1, 1024, to_dev, seed=7)
    O.check_resident_msm(emul_lib, emul_ctx, BN254, 2, 1024, to_dev, seed=8)


@pytest.mark.parametrize("c", ["5", "15"])
def test_window_sizes_that_negate_high_scalars(emul_lib, emul_ctx, emul_policy, c):
    """MsmPlan::negate_high: when the window size divides the scalar width (5, 15, 17 for BLS12-381's 255 bits), scalars
    above (r - 1) / 2 are replaced by r - k with flipped digit signs and a whole window disappears (c = 17: 15 instead
    of 16).  Scalars around the threshold against the known discrete log; the three distributions against oracle/c;
    a whole proof."""
    import parity_cases as pc
    from oracle import synthetic as S
    emul_policy.setenv("ARK355_MSM_C", c)

    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a

    pc.resident_known_dlog_case(emul_lib, emul_ctx, BLS12_381, 1, 320, to_dev)
    if c == "5":
        pc.resident_known_dlog_case(emul_lib, emul_ctx, BLS12_381, 2, 256, to_dev)
        O.check_resident_msm(emul_lib, emul_ctx, BLS12_381, 1, 512, to_dev, seed=9)
        A, B, Cm, z, ell = S.mulchain_direct(BLS12_381.r, 300)
        pc.prove_case(emul_lib, emul_ctx, BLS12_381, A, B, Cm, z, ell, rs=((5, 7),))


def test_large_size_checks_on_the_emulator(emul_lib, emul_ctx):
    """The checks the GPU tier runs at 2^21..2^23 (tests/test_gpu_o3_large.py): whole-vector NTT and witness map against
    oracle/c, a proof through ark355_prove AND ark355_prove_sharded (world size 1, both exchange modes), every proof
    through the Groth16 equation -- here at sizes the emulator finishes in seconds."""
    O.check_ntt_full(emul_lib, emul_ctx, BLS12_381, 9)
    O.check_ntt_full(emul_lib, emul_ctx, BN254, 4)
    O.check_witness_map_full(emul_lib, emul_ctx, BLS12_381, S.mulchain_csr(BLS12_381.r, 61))
    # N = 2^10 = 2^5 x 2^5: the smallest domain on the fused inverse -> coset path (seam kernel; c leaves after its inverse)
    O.check_witness_map_full(emul_lib, emul_ctx, BLS12_381, S.mulchain_csr(BLS12_381.r, 1000))
    # (the oracle's Python pairing on an O3-keyed proof runs in the GPU tier at 2^20; here the library's own pairing only)
    O.check_instance(emul_lib, emul_ctx, BLS12_381, S.mulchain_csr(BLS12_381.r, 33), [(9, 11)], sharded=True)


@pytest.mark.parametrize("stride", ["2", "3", "16"])
def test_strided_window_tables_on_the_emulator(emul_lib, emul_ctx, emul_policy, stride):
    """MsmPlan::wstride (the fallback for keys whose full window tables do not fit HBM): tables for every s-th window,
    s bucket sets combined as sum_j 2^(c j) S_j.  s = 2, 3 (does not divide the window count) and 16 >= windows (no
    table beyond the bases themselves); resident MSMs against oracle/c for both groups and whole proofs -- incl.
    ark355_prove_sharded with the bucket-level ring -- against cbase.prove."""
    emul_policy.setenv("ARK355_TABLE_STRIDE", stride)

    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a

    # c = 8: 32 windows (stride 16: a shorter vector at c = 4 -- 64 windows, 4 table blocks, 16 bucket sets -- the plain sums over
    # 16 sets of 128 buckets are what takes the emulator's time)
    O.check_resident_msm(emul_lib, emul_ctx, BLS12_381, 1, 1100 if stride != "16" else 300, to_dev, seed=21)
    O.check_resident_msm(emul_lib, emul_ctx, BN254, 2, 150, to_dev, seed=22)          # c = 4: 64 windows
    if stride == "3":
        O.check_instance(emul_lib, emul_ctx, BLS12_381, S.mulchain_csr(BLS12_381.r, 140), [(5, 7)], sharded=True,
                         equation=False, self_exchange=False)     # (the self exchange: test_large_size_checks_on_the_emulator)


def test_table_budget_picks_a_stride_or_reports_enomem(emul_lib, emul_ctx, emul_policy):
    """The planner behind ark355_pk_load: with a budget below the full tables the key still loads with the smallest
    stride that fits (ark355_pk_table_info shows it) and proves the same bytes; with a budget below the bare vectors the
    load fails with ARK355_ENOMEM and a message, not a HIP error."""
    from oracle.c import cbase
    from snark_amd._binding import Ark355Error, ENOMEM
    C = BLS12_381
    inst = S.mulchain_csr(C.r, 600)
    n, ell, w, mats, z = inst
    pk, _ = cbase.setup_raw_c(C, n, ell, w, mats, O.TD)
    pkh, rh = O.load(emul_lib, emul_ctx, C, inst, pk)
    full = emul_lib.pk_table_info(pkh)
    O.free(emul_lib, pkh, rh)
    assert full["table_stride"] == 1 and full["windows"] * full["window_bits"] >= 256
    # a budget the unpacked BLS12-381 rows (128 B) miss and the packed ones (96 B) meet: the planner packs the rows before it
    # gives up windows (round 5) -- stride 1, three quarters of the bytes, same proof
    emul_policy.setenv("ARK355_HBM_BUDGET_MB", str(max(1, (full["table_bytes"] * 85 // 100) >> 20)))
    pkh, rh = O.load(emul_lib, emul_ctx, C, inst, pk)
    try:
        info = emul_lib.pk_table_info(pkh)
        assert info["table_stride"] == 1 and info["table_bytes"] * 100 < full["table_bytes"] * 80, (info, full)
        zb = S._mont_bytes(C.r, z)
        got = emul_lib.prove(emul_ctx, pkh, rh, zb, len(z), O.Z.fr_canon(C, 3), O.Z.fr_canon(C, 4), emul_lib.sizes(C.curve_id))
        assert got == cbase.prove(C, n, ell, w, mats, zb, pk, 3, 4)
    finally:
        O.free(emul_lib, pkh, rh)
    emul_policy.setenv("ARK355_HBM_BUDGET_MB", str(max(1, (full["table_bytes"] // 3) >> 20)))
    pkh, rh = O.load(emul_lib, emul_ctx, C, inst, pk)
    try:
        info = emul_lib.pk_table_info(pkh)
        assert info["table_stride"] > 1 and info["table_bytes"] < full["table_bytes"] // 2
        zb = S._mont_bytes(C.r, z)
        got = emul_lib.prove(emul_ctx, pkh, rh, zb, len(z), O.Z.fr_canon(C, 3), O.Z.fr_canon(C, 4), emul_lib.sizes(C.curve_id))
        assert got == cbase.prove(C, n, ell, w, mats, zb, pk, 3, 4)
    finally:
        O.free(emul_lib, pkh, rh)
    emul_policy.setenv("ARK355_HBM_BUDGET_MB", "1")
    big = S.mulchain_csr(C.r, 5000)
    n2, ell2, w2, mats2, z2 = big
    pk2, _ = cbase.setup_raw_c(C, n2, ell2, w2, mats2, O.TD)
    with pytest.raises(Ark355Error) as e:
        O.load(emul_lib, emul_ctx, C, big, pk2)
    assert e.value.code == ENOMEM and "do not fit" in str(e.value)


def test_bn254_window_17_fits_254_bit_scalars_exactly(emul_lib, emul_ctx, emul_policy):
    """Round 6: the planner picks c = 17 for BN254 tables of 2^18 terms and more: 15 windows x 17 bits = 255 bits hold a 254-bit
    scalar and the carry of its signed digits with nothing to spare, and without the negation trick (17 does not divide 254).
    Forced here at emulator sizes: scalars around r - 1 and the thresholds against the known discrete log (the digits do not
    depend on the group: G1 only, the 2^16 buckets of this window are slow on the emulator)."""
    import parity_cases as pc
    emul_policy.setenv("ARK355_MSM_C", "17")

    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a

    pc.resident_known_dlog_case(emul_lib, emul_ctx, BN254, 1, 1100, to_dev)
