"""Wire formats at the BASELINE key size: the 2^20-constraint BLS12-381 proving key of the
ORACLE's generator, serialised by the oracle's C encoder (oracle/c cb_points_serialize, itself checked against
oracle/serialize.py) as an ark-serialize `ProvingKey` stream -- ~0.6 GB uncompressed / ~0.3 GB compressed, 5.2 M points
-- through `ark355_pk_load_bytes` (device decoders: a Fermat square root per compressed point, a 255-bit subgroup
multiplication per point under VALIDATE_FULL) and on to `ark355_prove`: proof bytes == `cbase.prove` on the raw key.
Malformed streams of that size: truncated inside the h query, one coordinate of one point in the middle of a query damaged.

Reference anchors: CanonicalSerialize / CanonicalDeserialize on SNARK::ProvingKey (/root/reference/snark/src/lib.rs:25-36),
Validate::Yes semantics (un-vendored ark-serialize)."""
import json
import os
import time

import pytest

import o3_cases as O
from oracle import serialize as Z, synthetic as S
from oracle.c import cbase
from oracle.fields import BLS12_381

pytestmark = pytest.mark.gpu

VALIDATE_NONE, VALIDATE_FULL, VALIDATE_CURVE = 0, 1, 2
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.mark.parametrize("compressed,validate", [(False, VALIDATE_CURVE), (True, VALIDATE_CURVE), (False, VALIDATE_FULL),
                                                 (True, VALIDATE_FULL)],
                         ids=["uncompressed-curve", "compressed-curve", "uncompressed-full", "compressed-full"])
def test_key_stream_2p20_to_proof_bytes(gpu_lib, gpu_ctx, compressed, validate):
    C = BLS12_381
    lib, ctx = gpu_lib, gpu_ctx
    inst = S.mulchain_csr(C.r, 1 << 20)
    n, ell, w, mats, z = inst
    pk = O.oracle_key(C, inst)
    stream = cbase.pk_stream(C, pk, compressed)
    points = 3 * (ell + w) + w + ((1 << 21) - 1) + ell + 6       # a, b1, b2 (m each), l (w), h (N - 1), gamma_abc (ell), six single points
    t0 = time.perf_counter()
    pkh = lib.pk_load_bytes(ctx, C.curve_id, stream, compressed=compressed, validate=validate)
    dt = time.perf_counter() - t0
    rh = lib.r1cs_load(ctx, C.curve_id, n, ell, w, mats)
    try:
        assert lib.pk_dims(pkh) == (ell, w, 1 << 21)
        sizes = lib.sizes(C.curve_id)
        zb = S._mont_bytes(C.r, z)
        r_, s_ = 0x1234567, 0x89ABCDE                                           # (the pair of test_s2_2p20_bls12_381_vs_o3: one CPU proof)
        a, b, c = lib.prove(ctx, pkh, rh, zb, len(z), Z.fr_canon(C, r_), Z.fr_canon(C, s_), sizes)
        exp = O.oracle_prove(C, inst, zb, pk, r_, s_)
        assert (a, b, c) == exp
        for comp in (True, False):
            wire = lib.proof_to_bytes(C.curve_id, a, b, c, comp)
            assert wire == cbase.points_serialize(C, 1, exp[0], comp) + cbase.points_serialize(C, 2, exp[1], comp) + \
                cbase.points_serialize(C, 1, exp[2], comp)
    finally:
        lib.dll.ark355_pk_free(pkh)
        lib.dll.ark355_r1cs_free(rh)
    rec = {"what": "ark355_pk_load_bytes, 2^20-constraint BLS12-381 key (stream in host memory -> resident key incl. window tables)",
           "compressed": compressed, "validate": {VALIDATE_CURVE: "curve", VALIDATE_FULL: "full"}[validate],
           "stream_bytes": len(stream), "points": points, "seconds": round(dt, 3),
           "stream_GB_per_s": round(len(stream) / dt / 1e9, 3), "points_per_s": round(points / dt)}
    print(json.dumps(rec))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "wire_large.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


def test_key_stream_2p20_malformed(gpu_lib, gpu_ctx):
    """A stream cut inside the h query, and one damaged coordinate at index 2^19 of the a query (uncompressed: the point
    leaves the curve; compressed: x is no longer reduced), both refused by the loader at full size."""
    C = BLS12_381
    lib, ctx = gpu_lib, gpu_ctx
    inst = S.mulchain_csr(C.r, 1 << 20)
    pk = O.oracle_key(C, inst)
    for compressed in (False, True):
        stream = cbase.pk_stream(C, pk, compressed)
        psz = 48 * (1 if compressed else 2)
        with pytest.raises(Exception):
            lib.pk_load_bytes(ctx, C.curve_id, stream[:len(stream) - 5 * psz - 7], compressed=compressed, validate=VALIDATE_CURVE)
        # offset of a_query[k]: vk (alpha_g1, beta_g2, gamma_g2, delta_g2, len + gamma_abc) + beta_g1 + delta_g1 + len
        ell = inst[1]
        off = psz + 3 * 2 * psz + 8 + ell * psz + 2 * psz + 8
        k = 1 << 19
        bad = bytearray(stream)
        if compressed:
            bad[off + k * psz] |= 0x1F                       # x >= q: not reduced (the flag bits stay as they are)
            bad[off + k * psz + 1] = 0xFF
        else:
            bad[off + k * psz + psz - 1] ^= 0x01             # lowest byte of y
        with pytest.raises(Exception) as e:
            lib.pk_load_bytes(ctx, C.curve_id, bytes(bad), compressed=compressed, validate=VALIDATE_CURVE)
        assert "a_query" in str(e.value) or getattr(e.value, "code", 0) != 0
