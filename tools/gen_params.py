#!/usr/bin/env python3
"""Generate snark_amd/csrc/curve_params.h from the public curve constants.

Developer tool (run by hand; its output is committed).  The constants themselves are the public
BLS12-381 / BN254 parameters; everything else (R, R^2, -p^-1 mod 2^32,
roots of unity, Montgomery images) is derived here with Python integers.
"""
import os
import sys



class _Curve:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# Public constants (kept here so that no product/dev-tool file imports the test oracle).
BLS12_381 = _Curve(
    name="bls12_381",
    q=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    r=0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    fr_generator=7, two_adicity=32, g1_b=4, g2_b=(4, 4), fq_limbs64=6, fr_limbs64=4,
    g1_gen=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
            0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    g2_gen=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
             0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
            (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
             0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)))
_BN_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
_i82 = pow(82, -1, _BN_Q)
BN254 = _Curve(
    name="bn254", q=_BN_Q,
    r=0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    fr_generator=5, two_adicity=28, g1_b=3, g2_b=((27 * _i82) % _BN_Q, (-3 * _i82) % _BN_Q),
    fq_limbs64=4, fr_limbs64=4, g1_gen=(1, 2),
    g2_gen=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
             11559732032986387107991004021392285783925812861821192530917403151452391805634),
            (8495653923123431417604973247489272438418190587263600148770280649306958101930,
             4082367875863433681332203403145435568316851327593401208105741076214120093531)))


def limbs32(v, n):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(v, n):
    return "{" + ", ".join("0x%08xu" % x for x in limbs32(v, n)) + "}"


def field_struct(name, p, n32, extra=""):
    R = 1 << (32 * n32)
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    out = []
    out.append("struct %s {" % name)
    out.append("  static constexpr int N = %d;" % n32)
    out.append("  static constexpr int BITS = %d;" % p.bit_length())
    out.append("  static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^32" % inv)
    for fn, val in (("mod", p), ("one", R % p), ("r2", R * R % p), ("pm2", p - 2)):
        out.append("  ARK_HD static constexpr uint32_t %s(int i) {" % fn)
        out.append("    constexpr uint32_t v[%d] = %s;" % (n32, arr(val, n32)))
        out.append("    return v[i];")
        out.append("  }")
    out.append(extra)
    out.append("};")
    return "\n".join(out)


def fr_extra(curve):
    p = curve.r
    n32 = curve.fr_limbs64 * 2
    R = 1 << (32 * n32)
    rho = pow(curve.fr_generator, (p - 1) >> curve.two_adicity, p)
    lines = []
    lines.append("  static constexpr int TWO_ADICITY = %d;" % curve.two_adicity)
    for fn, val in (("root", rho * R % p), ("root_inv", pow(rho, -1, p) * R % p),
                    ("gen", curve.fr_generator * R % p),
                    ("gen_inv", pow(curve.fr_generator, -1, p) * R % p)):
        lines.append("  // Montgomery image")
        lines.append("  ARK_HD static constexpr uint32_t %s(int i) {" % fn)
        lines.append("    constexpr uint32_t v[%d] = %s;" % (n32, arr(val, n32)))
        lines.append("    return v[i];")
        lines.append("  }")
    return "\n".join(lines)


def curve_block(tag, curve):
    nq = curve.fq_limbs64 * 2
    nr = curve.fr_limbs64 * 2
    q = curve.q
    Rq = 1 << (32 * nq)
    b1 = curve.g1_b * Rq % q
    b2 = (curve.g2_b[0] * Rq % q, curve.g2_b[1] * Rq % q)
    s = []
    s.append("// ---- %s ----" % curve.name)
    s.append(field_struct("%sFqParams" % tag, q, nq))
    s.append(field_struct("%sFrParams" % tag, curve.r, nr, fr_extra(curve)))
    s.append("struct %sCurveConsts {" % tag)
    for fn, val in (("g1_b", b1), ("g2_b_c0", b2[0]), ("g2_b_c1", b2[1]),
                    ("g1_gen_x", curve.g1_gen[0] * Rq % q), ("g1_gen_y", curve.g1_gen[1] * Rq % q),
                    ("g2_gen_x0", curve.g2_gen[0][0] * Rq % q), ("g2_gen_x1", curve.g2_gen[0][1] * Rq % q),
                    ("g2_gen_y0", curve.g2_gen[1][0] * Rq % q), ("g2_gen_y1", curve.g2_gen[1][1] * Rq % q)):
        s.append("  ARK_HD static constexpr uint32_t %s(int i) {" % fn)
        s.append("    constexpr uint32_t v[%d] = %s;" % (nq, arr(val, nq)))
        s.append("    return v[i];")
        s.append("  }")
    s.append("};")
    return "\n".join(s)


def main():
    out = []
    out.append("// GENERATED by tools/gen_params.py -- do not edit by hand.")
    out.append("// Public BLS12-381 / BN254 constants, 32-bit little-endian limbs, Montgomery R = 2^(32N).")
    out.append("#pragma once")
    out.append("#include <stdint.h>")
    out.append('#include "hd.h"')
    out.append("namespace ark355 {")
    out.append(curve_block("Bls", BLS12_381))
    out.append(curve_block("Bn", BN254))
    out.append("}  // namespace ark355")
    path = os.path.join(os.path.dirname(__file__), "..", "snark_amd", "csrc", "curve_params.h")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
