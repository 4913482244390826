"""Public curve constants used by the host side (scalar-field arithmetic of setup, raw images of the
generators).  Same numbers as tools/gen_params.py feeds to the device headers."""
from __future__ import annotations


class Curve:
    def __init__(self, curve_id, name, q, r, fr_generator, two_adicity, fq_bytes, g1_gen, g2_gen):
        self.curve_id = curve_id
        self.name = name
        self.q = q
        self.r = r
        self.fr_generator = fr_generator
        self.two_adicity = two_adicity
        self.fq_bytes = fq_bytes
        self.fr_bytes = 32
        self.g1_gen = g1_gen
        self.g2_gen = g2_gen
        self.g1_bytes = 2 * fq_bytes
        self.g2_bytes = 4 * fq_bytes

    # ---- raw memory images (include/ark355.h conventions) ---------------------------------------
    def fq_mont(self, v):
        return (v % self.q * (1 << (8 * self.fq_bytes)) % self.q).to_bytes(self.fq_bytes, "little")

    def fr_mont(self, v):
        return (v % self.r * (1 << 256) % self.r).to_bytes(32, "little")

    def fr_canon(self, v):
        return (v % self.r).to_bytes(32, "little")

    def g1_gen_raw(self):
        return self.fq_mont(self.g1_gen[0]) + self.fq_mont(self.g1_gen[1])

    def g2_gen_raw(self):
        (x0, x1), (y0, y1) = self.g2_gen
        return b"".join(self.fq_mont(v) for v in (x0, x1, y0, y1))

    def root_of_unity(self, log_n):
        if log_n > self.two_adicity:
            raise ValueError("PolynomialDegreeTooLarge")
        rho = pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)
        return pow(rho, 1 << (self.two_adicity - log_n), self.r)


BLS12_381 = Curve(
    0, "bls12_381",
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    7, 32, 48,
    (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
      0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
     (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
      0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)))

BN254 = Curve(
    1, "bn254",
    0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    5, 28, 32,
    (1, 2),
    ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
      11559732032986387107991004021392285783925812861821192530917403151452391805634),
     (8495653923123431417604973247489272438418190587263600148770280649306958101930,
      4082367875863433681332203403145435568316851327593401208105741076214120093531)))

CURVES = {0: BLS12_381, 1: BN254, "bls12_381": BLS12_381, "bn254": BN254}
