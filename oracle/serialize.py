"""ark-serialize compatible encodings (oracle; restates the un-vendored ``ark-serialize`` +
``ark-bls12-381/src/curves/util.rs`` + ``ark-ec`` SW flags; SURVEY.md Appendix A "Serialisation").

* field elements: canonical (non-Montgomery) little-endian bytes;
* BLS12-381 points: zcash/IETF format -- big-endian x (G2: x.c1 || x.c0); top byte flags
  bit7 = compressed, bit6 = infinity, bit5 = y lexicographically largest;
* BN254 points: little-endian x (G2: c0 || c1), flags in the two top bits of the LAST byte:
  bit7 = y is "negative" (y > -y), bit6 = infinity;
* ``Vec<T>`` = u64 LE length then the elements; ``Proof`` = a || b || c.
"""
from __future__ import annotations

from .fields import CurveParams


def _fq_gt_neg(y, q):
    return y > (q - y) % q


def _fq2_gt_neg(y, q):
    """Fq2 ordering: compare c1 first, then c0 (ark-ff QuadExtField Ord)."""
    n = ((-y[0]) % q, (-y[1]) % q)
    if y[1] != n[1]:
        return y[1] > n[1]
    return y[0] > n[0]


def g1_compressed(curve: CurveParams, P) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    if curve.bn_like:
        if P is None:
            b = bytearray(nb)
            b[-1] |= 1 << 6
            return bytes(b)
        b = bytearray(P[0].to_bytes(nb, "little"))
        if _fq_gt_neg(P[1], q):
            b[-1] |= 1 << 7
        return bytes(b)
    if P is None:
        b = bytearray(nb)
        b[0] |= 0xC0
        return bytes(b)
    b = bytearray(P[0].to_bytes(nb, "big"))
    b[0] |= 0x80
    if _fq_gt_neg(P[1], q):
        b[0] |= 0x20
    return bytes(b)


def g2_compressed(curve: CurveParams, P) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    if curve.bn_like:
        if P is None:
            b = bytearray(2 * nb)
            b[-1] |= 1 << 6
            return bytes(b)
        b = bytearray(P[0][0].to_bytes(nb, "little") + P[0][1].to_bytes(nb, "little"))
        if _fq2_gt_neg(P[1], q):
            b[-1] |= 1 << 7
        return bytes(b)
    if P is None:
        b = bytearray(2 * nb)
        b[0] |= 0xC0
        return bytes(b)
    b = bytearray(P[0][1].to_bytes(nb, "big") + P[0][0].to_bytes(nb, "big"))
    b[0] |= 0x80
    if _fq2_gt_neg(P[1], q):
        b[0] |= 0x20
    return bytes(b)


def g1_uncompressed(curve: CurveParams, P) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    if curve.bn_like:
        if P is None:
            b = bytearray(2 * nb)
            b[-1] |= 1 << 6
            return bytes(b)
        b = bytearray(P[0].to_bytes(nb, "little") + P[1].to_bytes(nb, "little"))
        if _fq_gt_neg(P[1], q):
            b[-1] |= 1 << 7
        return bytes(b)
    if P is None:
        b = bytearray(2 * nb)
        b[0] |= 0x40
        return bytes(b)
    return P[0].to_bytes(nb, "big") + P[1].to_bytes(nb, "big")


def g2_uncompressed(curve: CurveParams, P) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    if curve.bn_like:
        if P is None:
            b = bytearray(4 * nb)
            b[-1] |= 1 << 6
            return bytes(b)
        b = bytearray(b"".join(v.to_bytes(nb, "little") for v in (P[0][0], P[0][1], P[1][0], P[1][1])))
        if _fq2_gt_neg(P[1], q):
            b[-1] |= 1 << 7
        return bytes(b)
    if P is None:
        b = bytearray(4 * nb)
        b[0] |= 0x40
        return bytes(b)
    return b"".join(v.to_bytes(nb, "big") for v in (P[0][1], P[0][0], P[1][1], P[1][0]))


def proof_bytes(curve: CurveParams, proof, compressed=True) -> bytes:
    if compressed:
        return g1_compressed(curve, proof.a) + g2_compressed(curve, proof.b) + g1_compressed(curve, proof.c)
    return g1_uncompressed(curve, proof.a) + g2_uncompressed(curve, proof.b) + g1_uncompressed(curve, proof.c)


def _vec(items, enc):
    return len(items).to_bytes(8, "little") + b"".join(enc(x) for x in items)


def vk_bytes(curve, vk, compressed=True) -> bytes:
    e1 = (lambda P: g1_compressed(curve, P)) if compressed else (lambda P: g1_uncompressed(curve, P))
    e2 = (lambda P: g2_compressed(curve, P)) if compressed else (lambda P: g2_uncompressed(curve, P))
    return e1(vk.alpha_g1) + e2(vk.beta_g2) + e2(vk.gamma_g2) + e2(vk.delta_g2) + _vec(vk.gamma_abc_g1, e1)


def pk_bytes(curve, pk, compressed=False) -> bytes:
    e1 = (lambda P: g1_compressed(curve, P)) if compressed else (lambda P: g1_uncompressed(curve, P))
    e2 = (lambda P: g2_compressed(curve, P)) if compressed else (lambda P: g2_uncompressed(curve, P))
    return (vk_bytes(curve, pk.vk, compressed) + e1(pk.beta_g1) + e1(pk.delta_g1)
            + _vec(pk.a_query, e1) + _vec(pk.b_g1_query, e1) + _vec(pk.b_g2_query, e2)
            + _vec(pk.h_query, e1) + _vec(pk.l_query, e1))


# ---- raw memory images handed across the C ABI (include/ark355.h) --------------------------------
def g1_raw(curve: CurveParams, P) -> bytes:
    """x || y, LE u64 limbs, Montgomery form; infinity = all zero."""
    nb, q = curve.fq_bytes, curve.q
    if P is None:
        return bytes(2 * nb)
    R = 1 << (8 * nb)
    return (P[0] * R % q).to_bytes(nb, "little") + (P[1] * R % q).to_bytes(nb, "little")


def g2_raw(curve: CurveParams, P) -> bytes:
    """x.c0 || x.c1 || y.c0 || y.c1, Montgomery; infinity = all zero."""
    nb, q = curve.fq_bytes, curve.q
    if P is None:
        return bytes(4 * nb)
    R = 1 << (8 * nb)
    return b"".join((v * R % q).to_bytes(nb, "little") for v in (P[0][0], P[0][1], P[1][0], P[1][1]))


def g1_from_raw(curve: CurveParams, b: bytes):
    nb, q = curve.fq_bytes, curve.q
    if not any(b):
        return None
    Ri = pow(1 << (8 * nb), -1, q)
    return (int.from_bytes(b[:nb], "little") * Ri % q, int.from_bytes(b[nb:2 * nb], "little") * Ri % q)


def g2_from_raw(curve: CurveParams, b: bytes):
    nb, q = curve.fq_bytes, curve.q
    if not any(b):
        return None
    Ri = pow(1 << (8 * nb), -1, q)
    v = [int.from_bytes(b[i * nb:(i + 1) * nb], "little") * Ri % q for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def fr_mont(curve: CurveParams, v) -> bytes:
    nb = curve.fr_bytes
    return (v % curve.r * (1 << (8 * nb)) % curve.r).to_bytes(nb, "little")


def fr_canon(curve: CurveParams, v) -> bytes:
    return (v % curve.r).to_bytes(curve.fr_bytes, "little")


def fr_from_mont(curve: CurveParams, b: bytes) -> int:
    nb = curve.fr_bytes
    return int.from_bytes(b, "little") * pow(1 << (8 * nb), -1, curve.r) % curve.r
