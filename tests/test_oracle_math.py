"""Pins of the oracle's un-vendored arithmetic: public constants, cross-derivations, pairing."""
import random

import pytest

from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.curves import g1, g2
from oracle.fields import BLS12_381, BN254
from oracle.ntt import Domain
from oracle.pairing import Pairing

CURVES = [BLS12_381, BN254]


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_generators_on_curve_and_order(C):
    for G_ in (g1(C), g2(C)):
        assert G_.is_on_curve(G_.gen)
        assert G_.mul(G_.gen, C.r) is None
        assert G_.mul(G_.gen, C.r + 5) == G_.mul(G_.gen, 5)


def test_public_constants_bls12_381():
    # SURVEY.md Appendix B (public BLS12-381 parameters)
    C = BLS12_381
    assert C.q.bit_length() == 381 and C.r.bit_length() == 255
    assert (C.r - 1) % (1 << 32) == 0 and (C.r - 1) % (1 << 33) != 0
    assert hex(C.root_of_unity(10)) == "0x325db5c3debf77a18f4de02c0f776af3ea437f9626fc085e3c28d666a5c2d854"
    assert hex(C.root_of_unity(21)) == "0x47c8b5817018af4fc70d0874b0691d4e46b3105f04db5844cd3979122d3ea03a"
    assert pow(C.root_of_unity(32), 1 << 31, C.r) == C.r - 1
    assert hex(C.fr_R()) == "0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe"
    # zcash-format encodings of the standard generators
    assert Z.g1_compressed(C, C.g1_gen).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    assert Z.g2_compressed(C, C.g2_gen).hex() == (
        "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
    two_g = g1(C).add(C.g1_gen, C.g1_gen)
    assert hex(two_g[0]) == ("0x572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e")
    k = 0x0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef
    kg = g1(C).mul(C.g1_gen, k)
    assert hex(kg[0]) == ("0x6b50179774296419b7e8375118823ddb06940d9a28ea045ab418c7ecbe6da84d416cb55406eec6393db97ac26e38bd4")


def test_public_constants_bn254():
    C = BN254
    assert C.q.bit_length() == 254 and C.r.bit_length() == 254
    assert C.root_of_unity(28) == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(C.root_of_unity(28), 1 << 27, C.r) == C.r - 1


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_ntt_convention_and_roundtrips(C):
    rnd = random.Random(3)
    for lg in (0, 1, 2, 5):
        d = Domain(C, lg)
        xs = [rnd.randrange(C.r) for _ in range(d.n)]
        assert d.fft(xs) == d.naive_dft(xs)
        assert d.ifft(d.fft(xs)) == xs
        assert d.coset_ifft(d.coset_fft(xs)) == xs
    d = Domain(C, 4)
    tau = rnd.randrange(C.r)
    L = d.lagrange_at(tau)
    # sum_k L_k(tau) * omega^(k*j) == tau^j  (interpolation of x^j)
    for j in (0, 1, 7):
        assert sum(L[k] * pow(d.omega, k * j, C.r) for k in range(d.n)) % C.r == pow(tau, j, C.r)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_group_laws(C):
    rnd = random.Random(4)
    for G_ in (g1(C), g2(C)):
        a, b = rnd.randrange(C.r), rnd.randrange(C.r)
        P, Q = G_.mul(G_.gen, a), G_.mul(G_.gen, b)
        assert G_.add(P, Q) == G_.mul(G_.gen, a + b)
        assert G_.add(P, G_.neg(P)) is None
        assert G_.msm([P, Q, None, P], [3, 5, 7, 0]) == G_.mul(G_.gen, 3 * a + 5 * b)
        pts = G_.fixed_base_muls(G_.gen, [a, b, 0, 1])
        assert pts == [P, Q, None, G_.gen]


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_pairing_bilinear(C):
    P = Pairing(C)
    G1, G2 = g1(C), g2(C)
    e1 = P.pairing(G1.mul(C.g1_gen, 6), G2.mul(C.g2_gen, 5))
    e2 = P.pairing(G1.mul(C.g1_gen, 30), C.g2_gen)
    assert P.F.eq(e1, e2) and not P.F.eq(e1, P.F.one)
    assert P.F.eq(P.F.pow(e2, C.r), P.F.one)


@pytest.mark.parametrize("C", CURVES, ids=lambda c: c.name)
def test_groth16_three_derivations_agree_and_verify(C):
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 11)
    td = G.Trapdoor(tau=0xabcdef123, alpha=3, beta=5, gamma=7, delta=11, g1_k=2, g2_k=3)
    pk = G.setup(C, A, B, Cm, ell, len(z), td)
    h = G.witness_map(C, A, B, Cm, z, ell)
    assert h[-1] == 0                                  # deg h <= N-2 for a satisfying z
    r_, s_ = 0x1111111111111111, 0x2222222222222222
    o1 = G.prove(C, pk, A, B, Cm, z, ell, r_, s_)      # step by step (MSM + NTT)
    o2 = G.prove_closed_form(C, pk, z, ell, r_, s_)    # trapdoor closed form (no MSM, no NTT)
    assert o1 == o2
    assert G.verify(C, pk.vk, z[1:ell], o1)
    assert not G.verify(C, pk.vk, [(z[1] + 1) % C.r], o1)
    assert len(Z.proof_bytes(C, o1)) == (128 if C.bn_like else 192)
    # r = 0 path (upstream skips B1)
    assert G.prove(C, pk, A, B, Cm, z, ell, 0, s_) == G.prove_closed_form(C, pk, z, ell, 0, s_)


def test_dummy_circuit_config1_proves_and_verifies():
    """BASELINE.json configs[0] flavour (DummyCircuit shape, CPU reference only), reduced size."""
    C = BLS12_381
    cs = S.dummy_cs(C.r, 32)
    A, B, Cm, z, ell = S.cs_to_instance(cs)
    pk = G.setup(C, A, B, Cm, ell, len(z), G.Trapdoor(99, 2, 3, 4, 5))
    pr = G.prove(C, pk, A, B, Cm, z, ell, 12345, 67890)
    assert pr == G.prove_closed_form(C, pk, z, ell, 12345, 67890)
    assert G.verify(C, pk.vk, z[1:ell], pr)
