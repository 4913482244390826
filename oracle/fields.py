"""Prime fields and quadratic extensions as plain Python ints (oracle; test infrastructure).

Restates the *mathematics* of ark-ff ``Fp<MontBackend<_, N>>`` (un-vendored crate,
``ark-ff/src/fields/models/fp/montgomery_backend.rs``): an element is an integer mod p; its
in-memory image is ``N`` little-endian u64 limbs of ``a * R mod p`` with ``R = 2^(64 N)``.
Curve parameters are the public BLS12-381 / BN254 constants.
"""
from __future__ import annotations


class CurveParams:
    """Public constants of one pairing-friendly curve family."""

    def __init__(self, name, q, r, fr_generator, two_adicity, g1_b, g2_b, g1_gen, g2_gen,
                 fq_limbs64, fr_limbs64, ate_loop_count, bn_like, curve_id):
        self.name = name
        self.q = q                    # base field modulus
        self.r = r                    # scalar field modulus (group order)
        self.fr_generator = fr_generator
        self.two_adicity = two_adicity
        self.g1_b = g1_b              # y^2 = x^3 + g1_b
        self.g2_b = g2_b              # (c0, c1) in Fq2 = Fq[u]/(u^2+1)
        self.g1_gen = g1_gen
        self.g2_gen = g2_gen
        self.fq_limbs64 = fq_limbs64
        self.fr_limbs64 = fr_limbs64
        self.ate_loop_count = ate_loop_count
        self.bn_like = bn_like
        self.curve_id = curve_id

    @property
    def fq_bytes(self):
        return 8 * self.fq_limbs64

    @property
    def fr_bytes(self):
        return 8 * self.fr_limbs64

    def fq_R(self):
        return pow(2, 64 * self.fq_limbs64, self.q)

    def fr_R(self):
        return pow(2, 64 * self.fr_limbs64, self.r)

    def root_of_unity(self, log_n):
        """omega for the radix-2 domain of size 2^log_n (ark-poly Radix2EvaluationDomain::new)."""
        if log_n > self.two_adicity:
            raise ValueError("PolynomialDegreeTooLarge")
        rho = pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)
        return pow(rho, 1 << (self.two_adicity - log_n), self.r)


_BLS_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

BLS12_381 = CurveParams(
    name="bls12_381", q=_BLS_Q, r=_BLS_R, fr_generator=7, two_adicity=32,
    g1_b=4, g2_b=(4, 4),
    g1_gen=(
        0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
    ),
    g2_gen=(
        (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
         0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
        (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
         0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
    ),
    fq_limbs64=6, fr_limbs64=4,
    ate_loop_count=15132376222941642752, bn_like=False, curve_id=0,
)

_BN_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
_BN_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _bn_g2_b():
    # b' = 3 / (9 + u)
    num = 3
    # (9+u)^-1 = (9 - u) / (81 + 1)
    inv82 = pow(82, -1, _BN_Q)
    return ((num * 9 * inv82) % _BN_Q, (-num * inv82) % _BN_Q)


BN254 = CurveParams(
    name="bn254", q=_BN_Q, r=_BN_R, fr_generator=5, two_adicity=28,
    g1_b=3, g2_b=_bn_g2_b(),
    g1_gen=(1, 2),
    g2_gen=(
        (10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531),
    ),
    fq_limbs64=4, fr_limbs64=4,
    ate_loop_count=29793968203157093288, bn_like=True, curve_id=1,
)

CURVES = {"bls12_381": BLS12_381, "bn254": BN254, 0: BLS12_381, 1: BN254}


# ----------------------------------------------------------------------------------------------
# field-operation namespaces used by the generic curve code
# ----------------------------------------------------------------------------------------------
class FpOps:
    """Arithmetic in F_p on Python ints."""

    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def mul(self, a, b):
        return (a * b) % self.p

    def sqr(self, a):
        return (a * a) % self.p

    def inv(self, a):
        if a % self.p == 0:
            raise ZeroDivisionError
        return pow(a, -1, self.p)

    def is_zero(self, a):
        return a % self.p == 0

    def eq(self, a, b):
        return (a - b) % self.p == 0

    def from_int(self, v):
        return v % self.p

    def mul_small(self, a, k):
        return (a * k) % self.p


class Fp2Ops:
    """Arithmetic in F_p[u]/(u^2+1) on (c0, c1) tuples."""

    def __init__(self, p):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def neg(self, a):
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        p = self.p
        n = (a[0] * a[0] + a[1] * a[1]) % p
        if n == 0:
            raise ZeroDivisionError
        ni = pow(n, -1, p)
        return ((a[0] * ni) % p, (-a[1] * ni) % p)

    def is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def eq(self, a, b):
        return (a[0] - b[0]) % self.p == 0 and (a[1] - b[1]) % self.p == 0

    def from_int(self, v):
        return (v % self.p, 0)

    def mul_small(self, a, k):
        return ((a[0] * k) % self.p, (a[1] * k) % self.p)


# ----------------------------------------------------------------------------------------------
# memory images (ark-ff BigInt<N>([u64; N]) little-endian limbs, Montgomery form)
# ----------------------------------------------------------------------------------------------
def to_mont_bytes(v, p, nlimbs64):
    """Canonical int -> bytes of the Montgomery representation (LE u64 limbs)."""
    R = 1 << (64 * nlimbs64)
    return ((v % p) * R % p).to_bytes(8 * nlimbs64, "little")


def from_mont_bytes(b, p, nlimbs64):
    R = 1 << (64 * nlimbs64)
    x = int.from_bytes(b, "little")
    return x * pow(R, -1, p) % p


def to_canon_bytes(v, p, nlimbs64):
    return (v % p).to_bytes(8 * nlimbs64, "little")


def from_canon_bytes(b):
    return int.from_bytes(b, "little")
