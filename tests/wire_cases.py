"""ark-serialize wire formats THROUGH the C ABI against the oracle's encoders (oracle/serialize.py), shared by the
CPU-emulator tier and the GPU tier: point codecs (ark355_points_decode/_encode), proof bytes, and the end-to-end
path bytes of an ark_groth16::ProvingKey -> ark355_pk_load_bytes -> ark355_prove -> proof bytes == oracle."""
import random

import pytest

from helpers import csr_from_rows, z_bytes
from oracle import groth16 as G, serialize as Z, synthetic as S
from oracle.curves import g1, g2


def points_case(lib, ctx, C, n=9, seed=3):
    rnd = random.Random(seed)
    sz = lib.sizes(C.curve_id)
    for group, Gp, gen, raw, encs in ((1, g1(C), C.g1_gen, Z.g1_raw, (Z.g1_compressed, Z.g1_uncompressed)),
                                      (2, g2(C), C.g2_gen, Z.g2_raw, (Z.g2_compressed, Z.g2_uncompressed))):
        pts = [None, gen] + [Gp.mul(gen, rnd.randrange(C.r)) for _ in range(n - 2)]
        pts += [Gp.neg(p) for p in pts[1:4]]                       # both signs of y
        raws = b"".join(raw(C, p) for p in pts)
        rsz = sz["g1"] if group == 1 else sz["g2"]
        for comp, enc in ((True, encs[0]), (False, encs[1])):
            wire = b"".join(enc(C, p) for p in pts)
            assert lib.point_size(C.curve_id, group, comp) * len(pts) == len(wire)
            assert lib.points_encode(ctx, C.curve_id, group, raws, len(pts), comp) == wire
            assert lib.points_decode(ctx, C.curve_id, group, wire, len(pts), comp, True, rsz) == raws
        # an off-curve uncompressed point and a non-residue compressed x must be rejected with the index
        bad = bytearray(b"".join(encs[1](C, p) for p in pts))
        psz = lib.point_size(C.curve_id, group, False)
        bad[2 * psz + psz // 2 + 1] ^= 1
        with pytest.raises(Exception) as ei:
            lib.points_decode(ctx, C.curve_id, group, bytes(bad), len(pts), False, True, rsz)
        assert "[2]" in str(ei.value) and "curve" in str(ei.value)
        assert len(lib.points_decode(ctx, C.curve_id, group, bytes(bad), len(pts), False, False, rsz)) == len(raws)


def key_stream_case(lib, ctx, C, n=24, compressed=False):
    A, B, Cm, z, ell = S.cs_to_instance(S.bench_lc_cs(C.r, n))
    m, w = len(z), len(z) - ell
    opk = G.setup(C, A, B, Cm, ell, m, G.Trapdoor(tau=0xABCDEF123, alpha=3, beta=5, gamma=7, delta=11))
    stream = Z.pk_bytes(C, opk, compressed)
    pkh = lib.pk_load_bytes(ctx, C.curve_id, stream, compressed=compressed, validate=True)
    rh = lib.r1cs_load(ctx, C.curve_id, len(A), ell, w, [csr_from_rows(C, M) for M in (A, B, Cm)])
    try:
        assert lib.pk_dims(pkh) == (ell, w, 1 << opk.domain_log)
        sizes = lib.sizes(C.curve_id)
        r_, s_ = 0x1234567, 0x7654321
        a, b, c = lib.prove(ctx, pkh, rh, z_bytes(C, z), m, Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes)
        exp = G.prove_closed_form(C, opk, z, ell, r_, s_)
        for comp in (True, False):
            wire = lib.proof_to_bytes(C.curve_id, a, b, c, comp)
            assert wire == Z.proof_bytes(C, exp, comp)
            assert lib.proof_from_bytes(C.curve_id, wire, sizes, comp) == (a, b, c)
    finally:
        lib.dll.ark355_pk_free(pkh)
        lib.dll.ark355_r1cs_free(rh)
    # malformed streams fail in the loader
    for broken in (stream[:-3], stream + b"\0", stream[:40]):
        with pytest.raises(Exception):
            lib.pk_load_bytes(ctx, C.curve_id, broken, compressed=compressed)


VALIDATE_NONE, VALIDATE_FULL, VALIDATE_CURVE = 0, 1, 2


def validation_case(lib, ctx, C, seed=17):
    """What ark-serialize's Validate::Yes refuses and the on-curve-only mode lets through: points on
    the curve but outside the prime-order subgroup, infinity flags with a payload, flag combinations upstream rejects in
    every mode; and the same on a proof (three points, host path of ark355_proof_from_bytes)."""
    rnd = random.Random(seed)
    sz = lib.sizes(C.curve_id)
    q = C.q
    stray = {}
    for group, Gp, comp_enc, unc_enc, from_raw, rsz in ((1, g1(C), Z.g1_compressed, Z.g1_uncompressed, Z.g1_from_raw, sz["g1"]),
                                                        (2, g2(C), Z.g2_compressed, Z.g2_uncompressed, Z.g2_from_raw, sz["g2"])):
        # a random x whose curve equation has a root: the compressed decoder (curve-only mode) hands back the point
        P = None
        for _ in range(64):
            x = rnd.randrange(q) if group == 1 else (rnd.randrange(q), rnd.randrange(q))
            wire = comp_enc(C, (x, 1 if group == 1 else (1, 0)))
            try:
                raw = lib.points_decode(ctx, C.curve_id, group, wire, 1, True, VALIDATE_CURVE, rsz)
            except Exception:
                continue                                   # x^3 + b is not a square
            P = from_raw(C, raw)
            break
        assert P is not None and Gp.is_on_curve(P)
        in_subgroup = Gp.add(Gp.mul(P, C.r - 1), P) is None         # [r]P == O  (Group.mul reduces its scalar mod r)
        cofactor_one = C.bn_like and group == 1
        assert in_subgroup == cofactor_one, "a random curve point lies outside the subgroup unless the cofactor is 1"
        for comp, enc in ((True, comp_enc), (False, unc_enc)):
            w = enc(C, P)
            assert lib.points_decode(ctx, C.curve_id, group, w, 1, comp, VALIDATE_CURVE, rsz) == raw
            assert lib.points_decode(ctx, C.curve_id, group, w, 1, comp, VALIDATE_NONE, rsz) == raw
            if cofactor_one:
                assert lib.points_decode(ctx, C.curve_id, group, w, 1, comp, VALIDATE_FULL, rsz) == raw
            else:
                with pytest.raises(Exception) as ei:
                    lib.points_decode(ctx, C.curve_id, group, w, 1, comp, VALIDATE_FULL, rsz)
                assert "subgroup" in str(ei.value)
        stray[group] = P
        # subgroup members pass the full check
        good = Gp.mul(Gp.gen, rnd.randrange(1, C.r))
        for comp, enc in ((True, comp_enc), (False, unc_enc)):
            lib.points_decode(ctx, C.curve_id, group, enc(C, good), 1, comp, VALIDATE_FULL, rsz)
        # infinity flag with a payload
        for comp, enc in ((True, comp_enc), (False, unc_enc)):
            inf = bytearray(enc(C, None))
            assert from_raw(C, lib.points_decode(ctx, C.curve_id, group, bytes(inf), 1, comp, VALIDATE_FULL, rsz)) is None
            inf[len(inf) // 2] ^= 0x10
            for mode in (VALIDATE_FULL, VALIDATE_CURVE, VALIDATE_NONE):
                if C.bn_like:
                    # ark-ec's generic reader: the coordinates under an infinity flag are parsed (reduced) and ignored
                    assert from_raw(C, lib.points_decode(ctx, C.curve_id, group, bytes(inf), 1, comp, mode, rsz)) is None
                else:
                    # ark-bls12-381's readers refuse a payload under the infinity flag, whatever the validation mode
                    with pytest.raises(Exception) as ei:
                        lib.points_decode(ctx, C.curve_id, group, bytes(inf), 1, comp, mode, rsz)
                    assert "flag" in str(ei.value)
            if C.bn_like:
                big = bytearray(enc(C, None))
                for i in range(C.fq_bytes - 1):
                    big[i] = 0xFF                                          # first coordinate >= q under the infinity flag
                big[C.fq_bytes - 1] |= 0x3F if (group == 1 and comp) else 0xFF
                with pytest.raises(Exception):
                    lib.points_decode(ctx, C.curve_id, group, bytes(big), 1, comp, VALIDATE_NONE, rsz)
        # flag combinations upstream refuses in every mode
        if C.bn_like:
            both = bytearray(comp_enc(C, good))
            both[-1] |= 0xC0                                               # SWFlags::from_u8(0b11) is None
            bad_flags = [(True, bytes(both))]
        else:
            unc = bytearray(unc_enc(C, good))
            unc[0] |= 0x20                                                 # sort bit without the compressed bit
            infs = bytearray(comp_enc(C, None))
            infs[0] |= 0x20                                                # sort bit with the infinity bit
            mism = bytearray(unc_enc(C, good))
            mism[0] |= 0x80                                                # compressed bit on an uncompressed encoding
            bad_flags = [(False, bytes(unc)), (True, bytes(infs)), (False, bytes(mism))]
        for comp, w in bad_flags:
            for mode in (VALIDATE_NONE, VALIDATE_FULL, VALIDATE_CURVE):
                with pytest.raises(Exception) as ei:
                    lib.points_decode(ctx, C.curve_id, group, w, 1, comp, mode, rsz)
                assert "flag" in str(ei.value)
    # a proof whose B (and, where G1 has a cofactor, A) left the subgroup: refused by the default mode of
    # ark355_proof_from_bytes, accepted only by the explicit curve-only mode
    G1, G2 = g1(C), g2(C)
    a, b, c = G1.mul(G1.gen, 5), G2.mul(G2.gen, 7), G1.mul(G1.gen, 11)
    for comp in (True, False):
        e1, e2 = (Z.g1_compressed, Z.g2_compressed) if comp else (Z.g1_uncompressed, Z.g2_uncompressed)
        ok = e1(C, a) + e2(C, b) + e1(C, c)
        assert lib.proof_from_bytes(C.curve_id, ok, sz, comp) == (Z.g1_raw(C, a), Z.g2_raw(C, b), Z.g1_raw(C, c))
        evil = [e1(C, a) + e2(C, stray[2]) + e1(C, c)]
        if not C.bn_like:
            evil.append(e1(C, stray[1]) + e2(C, b) + e1(C, c))
        for w in evil:
            with pytest.raises(Exception):
                lib.proof_from_bytes(C.curve_id, w, sz, comp)
            lib.proof_from_bytes(C.curve_id, w, sz, comp, validate=VALIDATE_CURVE)
