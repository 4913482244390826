"""CPU tier: the library's own kernel + orchestration sources, compiled against the HIP emulator in
tests/emul (test infrastructure), checked against the oracle through the same C ABI the GPU tier uses.
This does NOT stand in for the GPU parity tests (tests/test_gpu_parity.py); it catches indexing /
barrier / buffer-size bugs before GPU minutes are spent."""
import pytest

import parity_cases as pc
from oracle import groth16 as G, synthetic as S
from oracle.fields import BLS12_381, BN254


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [1, 4, 9, 11])
def test_ntt_all_modes(emul_lib, emul_ctx, C, log_n):
    pc.ntt_case(emul_lib, emul_ctx, C, log_n)


def test_ntt_two_adicity_error(emul_lib, emul_ctx):
    import numpy as np
    with pytest.raises(Exception) as e:
        emul_lib.ntt(emul_ctx, BN254.curve_id, bytes(32), 41, 0, 0)
    assert e.value.code == -18


@pytest.mark.parametrize("C,group,n", [(BLS12_381, 1, 0), (BLS12_381, 1, 1), (BLS12_381, 1, 40), (BLS12_381, 2, 40),
                                       (BN254, 1, 40), (BN254, 2, 17), (BLS12_381, 1, 600)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_msm_vs_naive(emul_lib, emul_ctx, C, group, n):
    pc.msm_case(emul_lib, emul_ctx, C, group, n)


@pytest.mark.parametrize("C,group", [(BLS12_381, 1), (BN254, 1), (BLS12_381, 2)], ids=lambda v: getattr(v, "name", str(v)))
def test_resident_tables_exceptional_additions(emul_lib, emul_ctx, C, group):
    import numpy as np

    def to_dev(b):      # emulator: "device" pointers are host pointers
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a
    pc.resident_msm_edge_case(emul_lib, emul_ctx, C, group, 24, to_dev)


@pytest.mark.parametrize("pack", ["1", "0"], ids=["packed", "unpacked"])
def test_resident_tables_both_row_formats(emul_lib, emul_ctx, emul_policy, pack):
    """Policy PACK_ROWS: the 28-bit window tables as bit-packed rows (96 / 64 B; msm_accumulate28p_kernel and its lane-pair
    twin: parked flush, two-deep index prefetch) and as one word per limb (128 / 80 B; the plain walk) -- each forced on BOTH
    curves and groups (the default packs BN254 only), MSMs with P + P / P - P / infinity inside buckets, then whole proofs."""
    import numpy as np
    emul_policy.setenv("ARK355_PACK_ROWS", pack)

    def to_dev(b):
        a = np.frombuffer(b, dtype=np.uint8).copy()
        return a.ctypes.data, a
    for C in (BLS12_381, BN254):
        for group in (1, 2):
            pc.resident_msm_edge_case(emul_lib, emul_ctx, C, group, 24, to_dev)
        A, B, Cm, z, ell = S.mulchain_direct(C.r, 13)
        pc.prove_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd),))


@pytest.mark.parametrize("skew", ["equal", "boolean"])
def test_msm_skewed_scalars(emul_lib, emul_ctx, skew):
    # all-equal scalars: every term of a window lands in ONE bucket (long straddling runs);
    # boolean witnesses: mostly 0/1 scalars (arkworks skips zeros)
    pc.msm_known_dlog_case(emul_lib, emul_ctx, BLS12_381, 1, 200, skew=skew)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_r1cs_ops_and_witness_map(emul_lib, emul_ctx, C):
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, 20))     # general coefficients, repeated columns
    pc.r1cs_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell)
    A, B, Cm, z, ell = S.cs_to_instance(S.dummy_cs(C.r, 16))         # empty rows, degenerate values
    pc.r1cs_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_distributed_witness_map_all_ranks_on_one_device(emul_lib, emul_ctx, C):
    """witness_dist_impl.cuh through ark355_witness_map_dist_sim: world sizes 2, 4 (and 8 refused: 8 G^2 > N) at N = 2^7 / 2^8,
    general coefficients and the degenerate DummyCircuit, against the oracle and the replicated map."""
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, 100))         # N = 128
    pc.witness_map_dist_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, worlds=(2, 4, 8))
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 200)                       # N = 256
    pc.witness_map_dist_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, worlds=(2, 4))


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_prove_bytes_equal_oracle(emul_lib, emul_ctx, C):
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 13)
    pc.prove_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, rs=((0x1234567890abcdef, 0xfedcba0987654321aabbccdd), (0, 5)))


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_check_satisfied_policy(emul_lib, emul_ctx, emul_policy, C):
    pc.check_satisfied_case(emul_lib, emul_ctx, C, emul_policy)


@pytest.mark.parametrize("sched", ["SERIAL=1", "SERIAL=0", "SCHED=2", "SCHED=3"])
def test_prove_schedules_give_the_same_bytes(emul_lib, emul_ctx, emul_policy, sched):
    """prove_run's schedules (policy SCHED; SERIAL=1 / 0 are the legacy spellings of 0 / 1): one stream, the five-stream
    pipeline with either epilogue, one stream with the runtime's own wait.  Same bytes as the oracle every way."""
    name, value = sched.split("=")
    emul_policy.setenv("ARK355_" + name, value)
    C = BLS12_381
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 21)
    pc.prove_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, rs=((7, 9),))
    assert emul_lib.ctx_get_policy(emul_ctx, "SCHED") == {"SERIAL=1": 0, "SERIAL=0": 1, "SCHED=2": 2, "SCHED=3": 3}[sched]


@pytest.mark.parametrize("batch", [1, 0, "aside"])
@pytest.mark.parametrize("circuit", ["mulchain", "dummy", "bench_lc"])
def test_one_stream_batched_g1_tails(emul_lib, emul_ctx, emul_policy, circuit, batch):
    """One-stream proofs run the merge / heavy merge / reduction / combination of the four G1 MSMs as one launch each
    (msm_reduce_phase_batch; policy BATCH_TAILS = 0 keeps one launch per MSM) and clear their scratch with one fill kernel:
    same bytes as the oracle either way, incl. the all-equal DummyCircuit (heavy buckets: every term of a window in ONE bucket)."""
    emul_policy.setenv("ARK355_SCHED", "0")
    if batch == "aside":
        # (round 6) a lone proof's tails of A, B1, L' as a batch of three on the second stream, H's own at the end
        emul_policy.setenv("ARK355_BATCH_TAILS", "1")
        emul_policy.setenv("ARK355_SIDE_G1_TAILS", "1")
    else:
        emul_policy.setenv("ARK355_BATCH_TAILS", str(batch))
    C = BLS12_381
    if circuit == "mulchain":
        inst = S.mulchain_direct(C.r, 37)
    elif circuit == "dummy":
        inst = S.cs_to_instance(S.dummy_cs(C.r, 40))
    else:
        inst = S.cs_to_instance(S.bench_lc_cs(C.r, 24))
    pc.prove_case(emul_lib, emul_ctx, C, *inst, rs=((3, 0x1234567), ))


@pytest.mark.parametrize("side", ["1", "0"])
def test_pipeline_last_msm_tails_on_either_stream(emul_lib, emul_ctx, emul_policy, side):
    """Five-stream pipeline: the tails of the H MSM on the sort stream (policy SIDE_H_TAILS, default) / on the reduction stream."""
    emul_policy.setenv("ARK355_SCHED", "1")
    emul_policy.setenv("ARK355_SIDE_H_TAILS", side)
    C = BLS12_381
    pc.prove_case(emul_lib, emul_ctx, C, *S.mulchain_direct(C.r, 37), rs=((5, 0x7654321), ))


def test_measured_schedule_choice_explores_then_latches(emul_lib):
    """Policy SCHED = -1 (default): the first warm proofs of a class run the candidate schedules in turn (SCHED_EXPLORE samples
    each), every one of them gives the oracle's bytes, and the class then keeps one schedule (ark355_sched_info)."""
    from oracle import groth16 as G, serialize as Z
    C = BLS12_381
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 9)
    m = len(z)
    ctx = emul_lib.ctx_create(0)
    try:
        emul_lib.sched_reset(ctx)
        assert emul_lib.ctx_get_policy(ctx, "SCHED") == -1
        emul_lib.ctx_set_policy(ctx, "SCHED_EXPLORE", 2)
        sz = emul_lib.sizes(C.curve_id)
        pk = G.setup(C, A, B, Cm, ell, m, G.Trapdoor(tau=987654321, alpha=5, beta=7, gamma=11, delta=13))
        rh = pc.r1cs_load_from_rows(emul_lib, ctx, C, A, B, Cm, ell, m - ell)
        pkh = pc.pk_load_from_oracle(emul_lib, ctx, C, pk, ell, m - ell, 1 << pk.domain_log)
        zb = pc.z_bytes(C, z)
        exp = G.prove_closed_form(C, pk, z, ell, 5, 6)
        seen = []
        for i in range(1 + 3 * 2 + 2):                    # one cold proof, 3 candidates x 2 samples, 2 latched
            a, b, c = emul_lib.prove(ctx, pkh, rh, zb, m, Z.fr_canon(C, 5), Z.fr_canon(C, 6), sz)
            assert G.Proof(Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c)) == exp
            info = emul_lib.sched_info(ctx, pkh, False)
            seen.append(info["last"])
        assert info["latched"] in ("pipeline", "pipeline_sync", "one_stream")
        assert set(seen[1:7]) == {"pipeline", "pipeline_sync", "one_stream"} and seen[-1] == seen[-2] == info["latched"]
        assert all(n == 2 for n in info["samples"].values()) and len(info["samples"]) == 3
        emul_lib.dll.ark355_pk_free(pkh)
        emul_lib.dll.ark355_r1cs_free(rh)
    finally:
        emul_lib.ctx_destroy(ctx)


def test_resident_bases_partial_and_sum(emul_lib, emul_ctx):
    """ark355_bases_load (window tables) + msm over a PREFIX of the rows + XYZZ partial / ark355_xyzz_sum:
    the pieces the multi-GPU sharded MSM is made of."""
    import numpy as np
    from oracle import serialize as Z
    from oracle.curves import g1
    C = BLS12_381
    G1 = g1(C)
    import random
    rnd = random.Random(9)
    n = 50
    pts = G1.fixed_base_muls(G1.gen, [rnd.randrange(C.r) for _ in range(n)])
    ks = [rnd.randrange(C.r) for _ in range(n)]
    raw = b"".join(Z.g1_raw(C, p) for p in pts)
    sz = emul_lib.sizes(C.curve_id)
    halves = []
    for lo, hi in ((0, 20), (20, 50)):
        bh = emul_lib.bases_load(emul_ctx, C.curve_id, 1, raw[lo * 96:hi * 96], hi - lo)
        sc = np.frombuffer(b"".join(Z.fr_canon(C, k) for k in ks[lo:hi]), dtype=np.uint8).copy()
        # emulator: "device" pointers are host pointers
        halves.append(emul_lib.msm_dev(emul_ctx, bh, sc.ctypes.data, hi - lo, 0, 4 * 48, partial=True))
        if lo == 0:   # prefix of the rows with the same handle
            pre = emul_lib.msm_dev(emul_ctx, bh, sc.ctypes.data, 7, 0, sz["g1"])
            assert Z.g1_from_raw(C, pre) == G1.msm(pts[:7], ks[:7])
        emul_lib.dll.ark355_bases_free(bh)
    out = emul_lib.xyzz_sum(emul_ctx, C.curve_id, 1, b"".join(halves), 2, sz["g1"])
    assert Z.g1_from_raw(C, out) == G1.msm(pts, ks)


def test_host_mirror_setup_prove_over_emulator(emul_lib):
    """snark_amd.groth16.Groth16 (product host logic: setup scalars, key layout, closed form) driven over the
    emulator build; the resulting proof must verify under the oracle's pairing check."""
    from snark_amd.groth16 import Groth16
    from snark_amd import synthetic, params
    from oracle import serialize as Z
    C = BLS12_381
    cv = params.BLS12_381
    r1, z = synthetic.mulchain(cv, 6)
    A, B, Cm, z2, ell = S.mulchain_direct(C.r, 6)
    assert z == z2
    g = Groth16(cv, lib=emul_lib)
    try:
        seq = iter([11, 22, 33, 44, 55])
        pk, vk = g.circuit_specific_setup(r1, lambda: next(seq), keep_trapdoor=True)
        proof = g.prove(pk, r1, z, r=777, s=888)
        assert proof == g.prove_closed_form(pk, z, 777, 888)
        opk = G.setup(C, A, B, Cm, ell, len(z), G.Trapdoor(11, 22, 33, 44, 55))
        assert Z.g1_from_raw(C, pk.a_query[:96]) == opk.a_query[0]
        got = G.Proof(Z.g1_from_raw(C, proof.a), Z.g2_from_raw(C, proof.b), Z.g1_from_raw(C, proof.c))
        assert got == G.prove_closed_form(C, opk, z, ell, 777, 888)
        assert G.verify(C, opk.vk, z[1:ell], got)
        assert g.is_satisfied(r1, z) is None
        zb = list(z)
        zb[4] += 1
        assert g.is_satisfied(r1, zb) is not None
    finally:
        g.close()


def test_prove_batch(emul_lib, emul_ctx):
    """ark355_prove_batch over the emulator (runs its proofs one after another there)."""
    pc.prove_batch_case(emul_lib, emul_ctx, BN254, count=3, n=6)


def test_reference_example_circuit_on_the_emulator(emul_lib, emul_ctx):
    """relations/examples/satisfiable.rs / non_satisfiable.rs through the C ABI (GPU twin: test_gpu_parity.py)."""
    from helpers import r1cs_load_from_rows, z_bytes
    from oracle import r1cs as R
    C = BLS12_381
    cs = R.ConstraintSystem(C.r)
    R.example_circuit(cs, satisfiable=True)
    A, B, Cm, z, ell = S.cs_to_instance(cs)
    pc.prove_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell, verify=True)
    bad = R.ConstraintSystem(C.r)
    R.example_circuit(bad, satisfiable=False)
    A2, B2, C2, z2, _ = S.cs_to_instance(bad)
    rh = r1cs_load_from_rows(emul_lib, emul_ctx, C, A2, B2, C2, ell, len(z2) - ell)
    assert emul_lib.is_satisfied(emul_ctx, rh, z_bytes(C, z2), len(z2)) == 1
    emul_lib.dll.ark355_r1cs_free(rh)


@pytest.mark.parametrize("rmax", ["9", "3", "2"])
def test_ntt_every_radix_and_pass_count(emul_lib, emul_ctx, rmax, emul_policy):
    """Register-resident radix-8 groups, LDS re-deals, direct inter-pass tables, the fused inverse->coset seam and the
    pointwise fusion, for every pass radix 2^1..2^9 and for 1..5 passes (ARK355_NTT_RMAX shrinks the largest radix so
    that small vectors take many passes), all four transform modes against the oracle; then the witness map."""
    emul_policy.setenv("ARK355_NTT_RMAX", rmax)
    C = BLS12_381
    for log_n in (range(1, 11) if rmax != "9" else (2, 3, 5, 6, 7, 8, 10, 12)):
        pc.ntt_case(emul_lib, emul_ctx, C, log_n, seed=int(rmax))
    # 3000 / 9000 constraints: domains of 2^12 (radices 6, 6) and 2^14 (7, 7) -- the fused inverse -> coset seam kernel
    for n in (5, 60, 250, 900) + ((3000, 9000) if rmax == "9" else ()):
        A, B, Cm, z, ell = S.mulchain_direct(C.r, n)
        pc.r1cs_case(emul_lib, emul_ctx, C, A, B, Cm, z, ell)
    pc.ntt_case(emul_lib, emul_ctx, BN254, 7, seed=3)


def test_ntt_without_direct_twiddle_tables(emul_lib, emul_ctx, monkeypatch):
    """Domains beyond 2^23 points have no direct inter-pass table for their first pass: the composed twiddle is applied
    by an elementwise kernel after the pass.  Forced here on small vectors (fresh context: tables are cached)."""
    monkeypatch.setenv("ARK355_NTT_DIRECT_MAX", "4")
    monkeypatch.setenv("ARK355_NTT_RMAX", "3")
    ctx = emul_lib.ctx_create(0)
    try:
        for log_n in (7, 9, 10):
            pc.ntt_case(emul_lib, ctx, BLS12_381, log_n, seed=5)
        A, B, Cm, z, ell = S.mulchain_direct(BLS12_381.r, 300)
        pc.r1cs_case(emul_lib, ctx, BLS12_381, A, B, Cm, z, ell)
    finally:
        emul_lib.ctx_destroy(ctx)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_mul_vs_oracle(emul_lib, emul_ctx, C, group):
    pc.fixed_base_case(emul_lib, emul_ctx, C, group)


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_batch_verification_vs_oracle_pairing(emul_lib, emul_ctx, C):
    # the oracle's (slow, pure-Python) pairing is consulted for BLS12-381 here and for both curves in the GPU tier
    pc.verify_batch_case(emul_lib, emul_ctx, C, oracle_pairing=C is BLS12_381, light=True)


def test_setup_scalars_vs_oracle_incl_tau_in_the_domain_and_bad_csr(emul_lib):
    """ark355_setup_scalars (host code of the library) against the oracle's generator scalars -- also for a tau INSIDE the
    evaluation domain (Lagrange coefficients degenerate to an indicator, h = 0, as upstream's
    evaluate_all_lagrange_coefficients handles it) -- and its input validation (column indices and row
    pointers of the caller's CSR are checked before three threads write through them)."""
    import numpy as np
    from helpers import csr_from_rows
    from oracle import groth16 as G, serialize as Z
    from oracle.ntt import Domain
    from snark_amd._binding import Ark355Error, EINVAL
    C = BLS12_381
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, 11))
    n, m = len(A), len(z)
    w = m - ell
    mats = [csr_from_rows(C, M) for M in (A, B, Cm)]
    dom = Domain.for_size(C, n + ell)
    for tau in (0xABCDEF0123456789, pow(dom.omega, 3, C.r)):
        alpha, beta, gamma, delta = 3, 5, 7, 11
        td = b"".join(Z.fr_canon(C, x) for x in (tau, alpha, beta, gamma, delta))
        sc = emul_lib.setup_scalars(C.curve_id, n, ell, w, mats, td)
        u, v, ww, zt, _ = G.qap_scalars(C, A, B, Cm, n, ell, m, tau)
        canon = lambda xs: b"".join(Z.fr_canon(C, x) for x in xs)          # noqa: E731
        assert sc["u"].tobytes() == canon(u) and sc["v"].tobytes() == canon(v) and sc["w"].tobytes() == canon(ww)
        di, gi = pow(delta, -1, C.r), pow(gamma, -1, C.r)
        abc = [(beta * u[i] + alpha * v[i] + ww[i]) % C.r for i in range(m)]
        assert sc["l"].tobytes() == canon([abc[i] * di % C.r for i in range(ell, m)])
        assert sc["gamma_abc"].tobytes() == canon([abc[i] * gi % C.r for i in range(ell)])
        assert sc["h"].tobytes() == canon([zt * di % C.r * pow(tau, i, C.r) % C.r for i in range(dom.n - 1)])
        if zt == 0:
            assert sc["h"].tobytes() == bytes(32 * (dom.n - 1))
    bad = [(rp, col.copy(), cf) for rp, col, cf in mats]
    bad[1][1][0] = m                                                        # column index out of range
    with pytest.raises(Ark355Error) as e:
        emul_lib.setup_scalars(C.curve_id, n, ell, w, bad, td)
    assert e.value.code == EINVAL
    bad = [(rp.copy(), col, cf) for rp, col, cf in mats]
    bad[0][0][2] = bad[0][0][1] - 1 if bad[0][0][1] > 0 else 5             # row pointers not monotone
    bad[0][0][1] = bad[0][0][2] + 2
    with pytest.raises(Ark355Error) as e:
        emul_lib.setup_scalars(C.curve_id, n, ell, w, bad, td)
    assert e.value.code == EINVAL
