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
