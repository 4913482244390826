"""Product-side ark-serialize formats (snark_amd/serialize.py) against the oracle's encoders, plus round trips
through compressed point decoding (square roots in Fq / Fq2)."""
import random

import pytest

from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.curves import g1, g2
from oracle.fields import BLS12_381, BN254
from snark_amd import params, serialize as PS
from snark_amd.groth16 import Proof, ProvingKey, VerifyingKey


@pytest.mark.parametrize("C,cv", [(BLS12_381, params.BLS12_381), (BN254, params.BN254)], ids=["bls12_381", "bn254"])
def test_point_encodings_match_oracle_and_round_trip(C, cv):
    rnd = random.Random(11)
    G1, G2 = g1(C), g2(C)
    pts1 = [None, C.g1_gen] + [G1.mul(C.g1_gen, rnd.randrange(C.r)) for _ in range(6)]
    pts2 = [None, C.g2_gen] + [G2.mul(C.g2_gen, rnd.randrange(C.r)) for _ in range(6)]
    for P in pts1:
        for comp, enc in ((True, Z.g1_compressed), (False, Z.g1_uncompressed)):
            b = PS.g1_serialize(cv, P, comp)
            assert b == enc(C, P)
            assert PS.g1_deserialize(cv, b, comp) == P
        assert PS.g1_from_raw(cv, Z.g1_raw(C, P)) == P and PS.g1_to_raw(cv, P) == Z.g1_raw(C, P)
    for P in pts2:
        for comp, enc in ((True, Z.g2_compressed), (False, Z.g2_uncompressed)):
            b = PS.g2_serialize(cv, P, comp)
            assert b == enc(C, P)
            assert PS.g2_deserialize(cv, b, comp) == P
        assert PS.g2_from_raw(cv, Z.g2_raw(C, P)) == P


def test_bls_generator_known_encoding():
    cv = params.BLS12_381
    assert PS.g1_serialize(cv, cv.g1_gen).hex().startswith("97f1d3a73197d794")
    assert PS.g2_serialize(cv, cv.g2_gen).hex().startswith("93e02b6052719f60")


@pytest.mark.parametrize("C,cv", [(BLS12_381, params.BLS12_381), (BN254, params.BN254)], ids=["bls12_381", "bn254"])
def test_proof_and_key_bytes_match_oracle_and_round_trip(C, cv):
    A, B, Cm, z, ell = S.mulchain_direct(C.r, 5)
    opk = G.setup(C, A, B, Cm, ell, len(z), G.Trapdoor(77, 2, 3, 5, 7))
    opr = G.prove_closed_form(C, opk, z, ell, 11, 13)
    proof = Proof(Z.g1_raw(C, opr.a), Z.g2_raw(C, opr.b), Z.g1_raw(C, opr.c))
    for comp in (True, False):
        b = PS.proof_to_bytes(cv, proof, comp)
        assert b == Z.proof_bytes(C, opr, comp)
        assert PS.proof_from_bytes(cv, b, comp) == proof
    assert len(PS.proof_to_bytes(cv, proof)) == (192 if C is BLS12_381 else 128)
    j1 = lambda pts: b"".join(Z.g1_raw(C, p) for p in pts)
    j2 = lambda pts: b"".join(Z.g2_raw(C, p) for p in pts)
    vk = VerifyingKey(Z.g1_raw(C, opk.vk.alpha_g1), Z.g2_raw(C, opk.vk.beta_g2), Z.g2_raw(C, opk.vk.gamma_g2),
                      Z.g2_raw(C, opk.vk.delta_g2), j1(opk.vk.gamma_abc_g1))
    pk = ProvingKey(vk=vk, beta_g1=Z.g1_raw(C, opk.beta_g1), delta_g1=Z.g1_raw(C, opk.delta_g1),
                    a_query=j1(opk.a_query), b_g1_query=j1(opk.b_g1_query), b_g2_query=j2(opk.b_g2_query),
                    h_query=j1(opk.h_query), l_query=j1(opk.l_query), ell=ell, w=len(z) - ell, N=1 << opk.domain_log)
    for comp in (True, False):
        assert PS.vk_to_bytes(cv, vk, comp) == Z.vk_bytes(C, opk.vk, comp)
        b = PS.pk_to_bytes(cv, pk, comp)
        assert b == Z.pk_bytes(C, opk, comp)
        back = PS.pk_from_bytes(cv, b, comp)
        assert (back.a_query, back.b_g2_query, back.h_query, back.l_query, back.ell, back.w, back.N) == (
            pk.a_query, pk.b_g2_query, pk.h_query, pk.l_query, pk.ell, pk.w, pk.N)
        assert back.vk == vk and back.beta_g1 == pk.beta_g1 and back.delta_g1 == pk.delta_g1


def test_malformed_proving_keys_are_rejected_before_the_abi():
    """pk vectors reach ark355_pk_load as bare pointers, so inconsistent lengths, truncated streams and
    off-curve uncompressed points must fail in the loader (ark-serialize Validate::Yes behaviour)."""
    import pytest
    from snark_amd import params, serialize as PS
    from snark_amd.groth16 import ProvingKey, VerifyingKey
    cv = params.BLS12_381
    g1, g2 = cv.g1_gen_raw(), cv.g2_gen_raw()
    vk = VerifyingKey(g1, g2, g2, g2, g1 * 2)
    pk = ProvingKey(vk=vk, beta_g1=g1, delta_g1=g1, a_query=g1 * 5, b_g1_query=g1 * 5, b_g2_query=g2 * 5,
                    h_query=g1 * 7, l_query=g1 * 3, ell=2, w=3, N=8)
    good = PS.pk_to_bytes(cv, pk)
    back = PS.pk_from_bytes(cv, good)
    assert (back.ell, back.w, back.N, back.a_query) == (2, 3, 8, pk.a_query)
    back.check_lengths({"g1": cv.g1_bytes, "g2": cv.g2_bytes})
    with pytest.raises(ValueError):
        PS.pk_from_bytes(cv, good[:-5])
    short = ProvingKey(vk=vk, beta_g1=g1, delta_g1=g1, a_query=g1 * 5, b_g1_query=g1 * 4, b_g2_query=g2 * 5,
                       h_query=g1 * 7, l_query=g1 * 3, ell=2, w=3, N=8)
    with pytest.raises(ValueError):
        PS.pk_from_bytes(cv, PS.pk_to_bytes(cv, short))
    with pytest.raises(ValueError):
        short.check_lengths({"g1": cv.g1_bytes, "g2": cv.g2_bytes})
    bad = bytearray(good)
    bad[cv.fq_bytes + 3] ^= 1                     # y of alpha_g1: no longer on the curve
    with pytest.raises(ValueError):
        PS.pk_from_bytes(cv, bytes(bad))
