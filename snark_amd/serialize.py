"""ark-serialize compatible wire formats for keys and proofs.

`SNARK` requires `CanonicalSerialize + CanonicalDeserialize` on `ProvingKey`, `VerifyingKey`, `Proof`
(/root/reference/snark/src/lib.rs:25-36).  This module converts between those byte strings and the raw
memory images of the C ABI (include/ark355.h), so that the backend can ingest real arkworks keys and emit
interchangeable proofs.  Conventions (un-vendored `ark-serialize`, `ark-bls12-381/src/curves/util.rs`, ark-ec SW
flags; SURVEY.md Appendix A "Serialisation"):

* field elements: canonical little-endian bytes;
* BLS12-381 points: zcash/IETF format -- big-endian x (G2: x.c1 || x.c0), flags in the top three bits of the
  first byte: bit7 compressed, bit6 infinity, bit5 "y is lexicographically largest";
* BN254 points: little-endian x (G2: c0 || c1), flags in the two top bits of the LAST byte: bit7 "y is
  negative" (y > -y), bit6 infinity;
* `Vec<T>`: u64 LE length, then the elements;
* `Proof` = a || b || c;  `VerifyingKey` = alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1;
  `ProvingKey` = vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query.
"""
from __future__ import annotations

from typing import Optional, Tuple

from .groth16 import Proof, ProvingKey, VerifyingKey
from .params import Curve


# ---- raw image <-> integer coordinates -----------------------------------------------------------------------
def _from_mont(curve: Curve, b: bytes) -> int:
    nb = curve.fq_bytes
    return int.from_bytes(b, "little") * pow(1 << (8 * nb), -1, curve.q) % curve.q


def g1_from_raw(curve: Curve, raw: bytes) -> Optional[Tuple[int, int]]:
    nb = curve.fq_bytes
    if not any(raw):
        return None
    return (_from_mont(curve, raw[:nb]), _from_mont(curve, raw[nb:2 * nb]))


def g2_from_raw(curve: Curve, raw: bytes):
    nb = curve.fq_bytes
    if not any(raw):
        return None
    v = [_from_mont(curve, raw[i * nb:(i + 1) * nb]) for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def g1_to_raw(curve: Curve, P) -> bytes:
    if P is None:
        return bytes(2 * curve.fq_bytes)
    return curve.fq_mont(P[0]) + curve.fq_mont(P[1])


def g2_to_raw(curve: Curve, P) -> bytes:
    if P is None:
        return bytes(4 * curve.fq_bytes)
    return b"".join(curve.fq_mont(v) for v in (P[0][0], P[0][1], P[1][0], P[1][1]))


# ---- field helpers (q = 3 mod 4 for both curves) ---------------------------------------------------------------
def _fq_sqrt(q, a):
    r = pow(a, (q + 1) // 4, q)
    return r if r * r % q == a % q else None


def _fq2_mul(q, a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)


def _fq2_sqrt(q, a):
    """Square root in Fq[u]/(u^2+1) via the norm (complex method)."""
    a0, a1 = a[0] % q, a[1] % q
    if a1 == 0:
        r = _fq_sqrt(q, a0)
        if r is not None:
            return (r, 0)
        r = _fq_sqrt(q, (-a0) % q)
        return (0, r) if r is not None else None
    n = _fq_sqrt(q, (a0 * a0 + a1 * a1) % q)
    if n is None:
        return None
    inv2 = pow(2, -1, q)
    for cand in ((a0 + n) * inv2 % q, (a0 - n) * inv2 % q):
        x0 = _fq_sqrt(q, cand)
        if x0 is not None and x0 != 0:
            x1 = a1 * pow(2 * x0, -1, q) % q
            if _fq2_mul(q, (x0, x1), (x0, x1)) == (a0, a1):
                return (x0, x1)
    return None


def _gt_neg_fq(q, y):
    return y > (q - y) % q


def _gt_neg_fq2(q, y):
    n = ((-y[0]) % q, (-y[1]) % q)
    return y[1] > n[1] if y[1] != n[1] else y[0] > n[0]


_G1_B = {"bls12_381": 4, "bn254": 3}


def _g2_b(curve: Curve):
    if curve.name == "bls12_381":
        return (4, 4)
    i82 = pow(82, -1, curve.q)
    return (27 * i82 % curve.q, (-3 * i82) % curve.q)


# ---- points ---------------------------------------------------------------------------------------------------
def g1_serialize(curve: Curve, P, compressed=True) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    bls = curve.name == "bls12_381"
    size = nb if compressed else 2 * nb
    if P is None:
        b = bytearray(size)
        if bls:
            b[0] |= 0xC0 if compressed else 0x40
        else:
            b[-1] |= 1 << 6
        return bytes(b)
    if bls:
        b = bytearray(P[0].to_bytes(nb, "big") + (b"" if compressed else P[1].to_bytes(nb, "big")))
        if compressed:
            b[0] |= 0x80 | (0x20 if _gt_neg_fq(q, P[1]) else 0)
        return bytes(b)
    b = bytearray(P[0].to_bytes(nb, "little") + (b"" if compressed else P[1].to_bytes(nb, "little")))
    if _gt_neg_fq(q, P[1]):
        b[-1] |= 1 << 7
    return bytes(b)


def g2_serialize(curve: Curve, P, compressed=True) -> bytes:
    q, nb = curve.q, curve.fq_bytes
    bls = curve.name == "bls12_381"
    size = 2 * nb if compressed else 4 * nb
    if P is None:
        b = bytearray(size)
        if bls:
            b[0] |= 0xC0 if compressed else 0x40
        else:
            b[-1] |= 1 << 6
        return bytes(b)
    (x0, x1), (y0, y1) = P
    if bls:
        parts = [x1, x0] + ([] if compressed else [y1, y0])
        b = bytearray(b"".join(v.to_bytes(nb, "big") for v in parts))
        if compressed:
            b[0] |= 0x80 | (0x20 if _gt_neg_fq2(q, P[1]) else 0)
        return bytes(b)
    parts = [x0, x1] + ([] if compressed else [y0, y1])
    b = bytearray(b"".join(v.to_bytes(nb, "little") for v in parts))
    if _gt_neg_fq2(q, P[1]):
        b[-1] |= 1 << 7
    return bytes(b)


def _g1_checked(curve: Curve, x, y):
    """Uncompressed points get the on-curve test ark-serialize performs under Validate::Yes.  (Its prime-subgroup test
    is a scalar multiplication per point; keys come from a trusted setup, proofs are checked by the verifier.)"""
    q = curve.q
    if x >= q or y >= q:
        raise ValueError("coordinate not reduced")
    if (y * y - x * x * x - _G1_B[curve.name]) % q:
        raise ValueError("point not on curve")
    return (x, y)


def _g2_checked(curve: Curve, x, y):
    q = curve.q
    if max(x[0], x[1], y[0], y[1]) >= q:
        raise ValueError("coordinate not reduced")
    x3 = _fq2_mul(q, _fq2_mul(q, x, x), x)
    bb = _g2_b(curve)
    y2 = _fq2_mul(q, y, y)
    if (y2[0] - x3[0] - bb[0]) % q or (y2[1] - x3[1] - bb[1]) % q:
        raise ValueError("point not on curve")
    return (x, y)


def g1_deserialize(curve: Curve, b: bytes, compressed=True):
    q, nb = curve.q, curve.fq_bytes
    bls = curve.name == "bls12_381"
    b = bytearray(b)
    if bls:
        flags = b[0] & 0xE0
        b[0] &= 0x1F
        if flags & 0x40:
            return None
        x = int.from_bytes(b[:nb], "big")
        if not compressed:
            return _g1_checked(curve, x, int.from_bytes(b[nb:2 * nb], "big"))
        want_largest = bool(flags & 0x20)
    else:
        flags = b[-1] & 0xC0
        b[-1] &= 0x3F
        if flags & 0x40:
            return None
        x = int.from_bytes(b[:nb], "little")
        if not compressed:
            return _g1_checked(curve, x, int.from_bytes(b[nb:2 * nb], "little"))
        want_largest = bool(flags & 0x80)
    if x >= q:
        raise ValueError("coordinate not reduced")
    y = _fq_sqrt(q, (x * x * x + _G1_B[curve.name]) % q)
    if y is None:
        raise ValueError("point not on curve")
    if _gt_neg_fq(q, y) != want_largest:
        y = (q - y) % q
    return (x, y)


def g2_deserialize(curve: Curve, b: bytes, compressed=True):
    q, nb = curve.q, curve.fq_bytes
    bls = curve.name == "bls12_381"
    b = bytearray(b)
    if bls:
        flags = b[0] & 0xE0
        b[0] &= 0x1F
        if flags & 0x40:
            return None
        x = (int.from_bytes(b[nb:2 * nb], "big"), int.from_bytes(b[:nb], "big"))
        if not compressed:
            return _g2_checked(curve, x, (int.from_bytes(b[3 * nb:4 * nb], "big"), int.from_bytes(b[2 * nb:3 * nb], "big")))
        want_largest = bool(flags & 0x20)
    else:
        flags = b[-1] & 0xC0
        b[-1] &= 0x3F
        if flags & 0x40:
            return None
        n = 2 if compressed else 4
        v = [int.from_bytes(b[i * nb:(i + 1) * nb], "little") for i in range(n)]
        x = (v[0], v[1])
        if not compressed:
            return _g2_checked(curve, x, (v[2], v[3]))
        want_largest = bool(flags & 0x80)
    x3 = _fq2_mul(q, _fq2_mul(q, x, x), x)
    bb = _g2_b(curve)
    y = _fq2_sqrt(q, ((x3[0] + bb[0]) % q, (x3[1] + bb[1]) % q))
    if y is None:
        raise ValueError("point not on curve")
    if _gt_neg_fq2(q, y) != want_largest:
        y = ((-y[0]) % q, (-y[1]) % q)
    return (x, y)


def g1_size(curve, compressed):
    return curve.fq_bytes * (1 if compressed else 2)


def g2_size(curve, compressed):
    return curve.fq_bytes * (2 if compressed else 4)


# ---- Proof / keys ------------------------------------------------------------------------------------------------
def proof_to_bytes(curve: Curve, proof: Proof, compressed=True) -> bytes:
    return (g1_serialize(curve, g1_from_raw(curve, proof.a), compressed)
            + g2_serialize(curve, g2_from_raw(curve, proof.b), compressed)
            + g1_serialize(curve, g1_from_raw(curve, proof.c), compressed))


def proof_from_bytes(curve: Curve, b: bytes, compressed=True) -> Proof:
    s1, s2 = g1_size(curve, compressed), g2_size(curve, compressed)
    if len(b) != 2 * s1 + s2:
        raise ValueError("bad proof length")
    return Proof(g1_to_raw(curve, g1_deserialize(curve, b[:s1], compressed)),
                 g2_to_raw(curve, g2_deserialize(curve, b[s1:s1 + s2], compressed)),
                 g1_to_raw(curve, g1_deserialize(curve, b[s1 + s2:], compressed)))


def _vec_g1(curve, raw, compressed):
    n = len(raw) // curve.g1_bytes
    return n.to_bytes(8, "little") + b"".join(
        g1_serialize(curve, g1_from_raw(curve, raw[i * curve.g1_bytes:(i + 1) * curve.g1_bytes]), compressed) for i in range(n))


def _vec_g2(curve, raw, compressed):
    n = len(raw) // curve.g2_bytes
    return n.to_bytes(8, "little") + b"".join(
        g2_serialize(curve, g2_from_raw(curve, raw[i * curve.g2_bytes:(i + 1) * curve.g2_bytes]), compressed) for i in range(n))


def vk_to_bytes(curve: Curve, vk: VerifyingKey, compressed=True) -> bytes:
    return (g1_serialize(curve, g1_from_raw(curve, vk.alpha_g1), compressed)
            + g2_serialize(curve, g2_from_raw(curve, vk.beta_g2), compressed)
            + g2_serialize(curve, g2_from_raw(curve, vk.gamma_g2), compressed)
            + g2_serialize(curve, g2_from_raw(curve, vk.delta_g2), compressed)
            + _vec_g1(curve, vk.gamma_abc_g1, compressed))


def pk_to_bytes(curve: Curve, pk: ProvingKey, compressed=False) -> bytes:
    return (vk_to_bytes(curve, pk.vk, compressed)
            + g1_serialize(curve, g1_from_raw(curve, pk.beta_g1), compressed)
            + g1_serialize(curve, g1_from_raw(curve, pk.delta_g1), compressed)
            + _vec_g1(curve, pk.a_query, compressed) + _vec_g1(curve, pk.b_g1_query, compressed)
            + _vec_g2(curve, pk.b_g2_query, compressed) + _vec_g1(curve, pk.h_query, compressed)
            + _vec_g1(curve, pk.l_query, compressed))


class _Reader:
    def __init__(self, curve, b, compressed):
        self.c, self.b, self.o, self.cmp = curve, b, 0, compressed

    def g1(self):
        n = g1_size(self.c, self.cmp)
        if self.o + n > len(self.b):
            raise ValueError("truncated proving key")
        p = g1_deserialize(self.c, self.b[self.o:self.o + n], self.cmp)
        self.o += n
        return g1_to_raw(self.c, p)

    def g2(self):
        n = g2_size(self.c, self.cmp)
        if self.o + n > len(self.b):
            raise ValueError("truncated proving key")
        p = g2_deserialize(self.c, self.b[self.o:self.o + n], self.cmp)
        self.o += n
        return g2_to_raw(self.c, p)

    def vec(self, fn):
        if self.o + 8 > len(self.b):
            raise ValueError("truncated proving key")
        n = int.from_bytes(self.b[self.o:self.o + 8], "little")
        self.o += 8
        if n > len(self.b):                       # every element takes at least one byte
            raise ValueError("vector length exceeds the stream")
        return b"".join(fn() for _ in range(n)), n


def pk_from_bytes(curve: Curve, b: bytes, compressed=False) -> ProvingKey:
    """Inverse of pk_to_bytes: an ark-serialize `ProvingKey` -> raw images ready for `ark355_pk_load`."""
    r = _Reader(curve, b, compressed)
    alpha, beta2, gamma2, delta2 = r.g1(), r.g2(), r.g2(), r.g2()
    gabc, ell = r.vec(r.g1)
    vk = VerifyingKey(alpha, beta2, gamma2, delta2, gabc)
    beta1, delta1 = r.g1(), r.g1()
    a, m = r.vec(r.g1)
    b1, _ = r.vec(r.g1)
    b2, _ = r.vec(r.g2)
    h, hn = r.vec(r.g1)
    l, w = r.vec(r.g1)
    if r.o != len(b):
        raise ValueError("trailing bytes in proving key")
    # ark355_pk_load copies (ell+w), (N-1) and w points from bare pointers: the five vectors must agree
    if not (m == ell + w and len(b1) == len(a) and len(b2) // curve.g2_bytes == m and ell >= 1):
        raise ValueError("proving key: inconsistent query lengths (a=%d, b_g1=%d, b_g2=%d, gamma_abc=%d, l=%d)"
                         % (m, len(b1) // curve.g1_bytes, len(b2) // curve.g2_bytes, ell, w))
    if (hn + 1) & hn:
        raise ValueError("proving key: h_query length + 1 is not a power of two")
    return ProvingKey(vk=vk, beta_g1=beta1, delta_g1=delta1, a_query=a, b_g1_query=b1, b_g2_query=b2, h_query=h,
                      l_query=l, ell=ell, w=w, N=hn + 1)
