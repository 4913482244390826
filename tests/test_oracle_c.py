"""O3: the oracle's C restatement of the arkworks CPU algorithms (oracle/c) against the Python oracle (O1/O2).
It is the checker at sizes Python cannot reach and the timed `cpu_baseline` of bench.py."""
import random

import pytest

from helpers import csr_from_rows, fr_vec_from_mont, g1_vec_raw, g2_vec_raw, z_bytes
from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.c import cbase
from oracle.curves import g1, g2
from oracle.fields import BLS12_381, BN254
from oracle.ntt import Domain

CURVES = [BLS12_381, BN254]


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_c_ntt_matches_python(C):
    rnd = random.Random(1)
    for lg in (0, 1, 4, 10):
        xs = [rnd.randrange(C.r) for _ in range(1 << lg)]
        d = Domain(C, lg)
        for inv, cos, ref in ((0, 0, d.fft), (1, 0, d.ifft), (0, 1, d.coset_fft), (1, 1, d.coset_ifft)):
            assert fr_vec_from_mont(C, cbase.ntt(C, z_bytes(C, xs), lg, inv, cos)) == ref(xs)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("group", [1, 2])
def test_c_pippenger_and_fixed_base(C, group):
    rnd = random.Random(2)
    G_ = g1(C) if group == 1 else g2(C)
    raw = Z.g1_raw if group == 1 else Z.g2_raw
    fromraw = Z.g1_from_raw if group == 1 else Z.g2_from_raw
    psz = len(raw(C, G_.gen))
    for n in (0, 1, 31, 32, 200):
        ks = [rnd.randrange(C.r) for _ in range(n)]
        ss = [rnd.randrange(C.r) for _ in range(n)]
        if n >= 31:
            ks[0], ks[1], ks[2], ss[3] = 0, 1, C.r - 1, 0
        bases = cbase.fixed_base(C, group, raw(C, G_.gen), b"".join(Z.fr_canon(C, s) for s in ss), n)
        if n:
            assert fromraw(C, bases[:psz]) == G_.mul(G_.gen, ss[0])
        out = cbase.msm(C, group, bases, b"".join(Z.fr_canon(C, k) for k in ks), n)
        assert fromraw(C, out) == G_.mul(G_.gen, sum(k * s for k, s in zip(ks, ss)) % C.r)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_c_prover_equals_closed_form(C):
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, 60))
    m, w = len(z), len(z) - ell
    td = G.Trapdoor(tau=1234567, alpha=3, beta=5, gamma=7, delta=11)
    pk = G.setup(C, A, B, Cm, ell, m, td)
    mats = [csr_from_rows(C, M) for M in (A, B, Cm)]
    h = cbase.witness_map(C, len(A), ell, w, mats, z_bytes(C, z))
    assert fr_vec_from_mont(C, h) == G.witness_map(C, A, B, Cm, z, ell)
    raw = dict(a_query=g1_vec_raw(C, pk.a_query), b_g1_query=g1_vec_raw(C, pk.b_g1_query),
               b_g2_query=g2_vec_raw(C, pk.b_g2_query), h_query=g1_vec_raw(C, pk.h_query),
               l_query=g1_vec_raw(C, pk.l_query), alpha_g1=Z.g1_raw(C, pk.vk.alpha_g1), beta_g1=Z.g1_raw(C, pk.beta_g1),
               delta_g1=Z.g1_raw(C, pk.delta_g1), beta_g2=Z.g2_raw(C, pk.vk.beta_g2), delta_g2=Z.g2_raw(C, pk.vk.delta_g2))
    for r_, s_ in ((123456789, 987654321), (0, 7), (C.r - 1, 1)):
        a, b, c = cbase.prove(C, len(A), ell, w, mats, z_bytes(C, z), raw, r_, s_)
        exp = G.prove_closed_form(C, pk, z, ell, r_, s_)
        assert (Z.g1_from_raw(C, a), Z.g2_from_raw(C, b), Z.g1_from_raw(C, c)) == (exp.a, exp.b, exp.c)
    pk2, _ = cbase.setup_raw(C, A, B, Cm, ell, m, td)
    for k in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query", "alpha_g1", "delta_g2"):
        assert pk2[k] == raw[k], k


def test_cpu_baseline_record_shape():
    rec = cbase.bench_prove("bls12_381", log_n=8, budget_s=2.0)
    assert rec["kind"] == "port" and rec["unit"] == "constraints/s" and rec["value"] > 0 and rec["cores"] >= 1


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_direct_csr_builders_match_the_constraint_system_path(C):
    """The numpy builders used by the large GPU-vs-O3 checks produce exactly what the ConstraintSystem restatement
    (oracle/r1cs.py) produces for the same circuits."""
    for n in (5, 64):
        for direct, via_cs in ((S.mulchain_csr(C.r, n), S.cs_to_instance(S.mulchain_cs(C.r, n))),
                               (S.dummy_csr(C.r, max(n, 8)), S.cs_to_instance(S.dummy_cs(C.r, max(n, 8)))),
                               (S.bench_lc_csr(C.r, n), S.cs_to_instance(S.bench_lc_cs(C.r, n)))):
            nn, ell, w, mats, z = direct
            A, B, Cm, z2, ell2 = via_cs
            assert (nn, ell, z) == (len(A), ell2, z2) and w == len(z2) - ell2
            for got, M in zip(mats, (A, B, Cm)):
                exp = csr_from_rows(C, M)
                assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2] == exp[2]


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_c_setup_scalars_equal_python_setup(C):
    n, ell, w, mats, z = S.bench_lc_csr(C.r, 50)
    A, B, Cm, z2, _ = S.cs_to_instance(S.bench_lc_cs(C.r, 50))
    td = G.Trapdoor(tau=987654321987654321, alpha=3, beta=5, gamma=7, delta=11)
    pk_py, sc_py = cbase.setup_raw(C, A, B, Cm, ell, ell + w, td)
    pk_c, sc_c = cbase.setup_raw_c(C, n, ell, w, mats, td)
    assert set(pk_py) == set(pk_c)
    for k in pk_py:
        assert pk_py[k] == pk_c[k], k
    for k in "uvw":
        assert sc_c[k] == b"".join(Z.fr_canon(C, x) for x in sc_py[k])


def test_c_msm_is_independent_of_the_thread_partition():
    C = BLS12_381
    rnd = random.Random(5)
    n = 3000
    ss = b"".join(Z.fr_canon(C, rnd.randrange(C.r)) for _ in range(n))
    ks = b"".join(Z.fr_canon(C, rnd.randrange(C.r)) for _ in range(n))
    bases = cbase.fixed_base(C, 1, Z.g1_raw(C, C.g1_gen), ss, n)
    L = cbase.lib()
    nt = L.cb_num_threads()
    try:
        outs = []
        for t in (1, 3, 64):
            L.cb_set_threads(t)
            outs.append(cbase.msm(C, 1, bases, ks, n))
    finally:
        L.cb_set_threads(nt)
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("C", [BLS12_381, BN254], ids=lambda c: c.name)
def test_c_point_serializer_equals_python_serializer(C):
    """cb_points_serialize (the encoder behind the BASELINE-size key streams of tests/test_gpu_wire_large.py) against
    oracle/serialize.py: both groups, compressed / uncompressed, incl. infinity and both signs of y."""
    import random
    from oracle import serialize as Z
    from oracle.c import cbase
    from oracle.curves import g1, g2
    rnd = random.Random(5)
    for group, Gp, raw_enc, comp, unc in ((1, g1(C), Z.g1_raw, Z.g1_compressed, Z.g1_uncompressed),
                                          (2, g2(C), Z.g2_raw, Z.g2_compressed, Z.g2_uncompressed)):
        pts = Gp.fixed_base_muls(Gp.gen, [rnd.randrange(C.r) for _ in range(12)])
        pts += [Gp.neg(p) for p in pts[:6]] + [None]
        raw = b"".join(raw_enc(C, p) for p in pts)
        assert cbase.points_serialize(C, group, raw, True) == b"".join(comp(C, p) for p in pts)
        assert cbase.points_serialize(C, group, raw, False) == b"".join(unc(C, p) for p in pts)


def test_c_fixed_base_window_sizes_agree():
    """cb_fixed_base switches from 8-bit to 16-bit windows at 2^16 scalars: the long call and two short calls over the same
    scalars give the same points (G1 and G2), incl. zero scalars."""
    import numpy as np
    C = BLS12_381
    n = 1 << 16
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    sc[5] = 0
    sc[n - 1] = 0
    sb = sc.astype("<u8").tobytes()
    for group, gen in ((1, Z.g1_raw(C, C.g1_gen)), (2, Z.g2_raw(C, C.g2_gen))):
        whole = cbase.fixed_base(C, group, gen, sb, n)
        h = n // 2
        halves = cbase.fixed_base(C, group, gen, sb[:h * 32], h) + cbase.fixed_base(C, group, gen, sb[h * 32:], h)
        assert whole == halves
