// ark-serialize wire formats of curve points, on the device (key vectors) and on the host (the three points of a proof).
//
// `SNARK` requires CanonicalSerialize + CanonicalDeserialize on ProvingKey / VerifyingKey / Proof
// (/root/reference/snark/src/lib.rs:25-36).  A Groth16 proving key at n = 2^20 holds ~6.3 M points: turning its byte
// stream into the Montgomery images the prover consumes means one byte-order change, one Montgomery conversion and
// -- compressed form -- one square root (a 380-bit exponentiation) per point.  Here every point is decoded by one GPU
// lane straight from the uploaded byte stream (wire_decode_kernel); the host only walks the stream's structure.
//
// Formats (un-vendored ark-serialize / ark-ec SWFlags / ark-bls12-381 curves/util.rs):
//   BLS12-381  zcash / IETF: big-endian coordinates, G2 as c1 || c0; flags in the top bits of the FIRST byte:
//              bit7 compressed, bit6 infinity, bit5 "y is lexicographically largest" (compressed only);
//   BN254      little-endian coordinates, G2 as c0 || c1; flags in the top bits of the LAST byte:
//              bit7 "y is negative" (y > -y), bit6 infinity.
// "y > -y" compares canonical integers; in Fq2 the c1 components decide first.
#pragma once
#include "common.h"
#include "curve.cuh"

namespace ark355 {

enum WireStatus : int { WIRE_OK = 0, WIRE_NOT_REDUCED = 1, WIRE_NOT_ON_CURVE = 2, WIRE_BAD_FLAGS = 3, WIRE_NOT_IN_SUBGROUP = 4 };
// `validate` of the decoders (include/ark355.h ARK355_VALIDATE_*): 0 = none, 1 = everything ark-serialize's Validate::Yes
// checks (canonical flags, on the curve, in the prime-order subgroup), 2 = flags and curve only -- the explicit opt-out
// for key material from a trusted source (the subgroup test is a 255-bit scalar multiplication per point)
constexpr int WIRE_VALIDATE_NONE = 0, WIRE_VALIDATE_FULL = 1, WIRE_VALIDATE_CURVE = 2;

template <class Curve>
struct Wire {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using Fr = typename Curve::Fr;
  using K = typename Curve::Consts;
  static constexpr bool BLS = Curve::ID == ARK355_BLS12_381;
  static constexpr bool G1_HAS_COFACTOR = BLS;             // BN254's G1 is the whole curve group (cofactor 1)

  // [r] P == O: membership in the prime-order subgroup (`is_in_correct_subgroup_assuming_on_curve` upstream).  Points of
  // small order pair to 1, so a proof carrying one would still verify: many encodings per proof for callers that key
  // on proof bytes.
  template <class F>
  ARK_HD static bool in_subgroup(const Affine<F>& p) {
    uint32_t r[Fr::N];
    for (int i = 0; i < Fr::N; i++) r[i] = Fr::Params::mod(i);
    return xyzz_mul_scalar(XYZZ<F>::from_affine(p), r, Fr::N).is_inf();
  }
  // every payload bit of an encoding whose infinity flag is set must be zero (flag bits masked)
  ARK_HD static bool payload_is_zero(const uint8_t* in, size_t size) {
    uint8_t acc = 0;
    for (size_t i = 0; i < size; i++) {
      uint8_t v = in[i];
      if (i == 0) v &= FIRST_MASK;
      if (i == size - 1) v &= LAST_MASK;
      acc |= v;
    }
    return acc == 0;
  }
  // Flag combinations upstream refuses whatever the validation mode: ark-bls12-381 EncodingFlags::get_flags (the sort
  // bit needs the compressed bit and excludes infinity; the compressed bit must match the requested form),
  // ark-ec SWFlags::from_u8 (both bits set is no flag value).
  ARK_HD static bool flags_ok(bool compressed, bool inf, bool sign, bool cbit) {
    if (BLS) return cbit == compressed && !(sign && (!compressed || inf));
    return !(inf && sign);
  }
  static constexpr int NB = Fq::N * 4;                     // bytes per base-field element

  ARK_HD static Fq g1_b() {
    Fq r;
    for (int i = 0; i < Fq::N; i++) r.l[i] = K::g1_b(i);
    return r;
  }
  ARK_HD static Fq2 g2_b() {
    Fq2 r;
    for (int i = 0; i < Fq::N; i++) {
      r.c0.l[i] = K::g2_b_c0(i);
      r.c1.l[i] = K::g2_b_c1(i);
    }
    return r;
  }

  // canonical integer (32-bit limbs) from NB bytes; `mask` clears the flag bits of the flagged byte
  ARK_HD static Fq limbs_from_bytes(const uint8_t* b, uint8_t first_mask, uint8_t last_mask) {
    Fq r = Fq::zero();
    for (int i = 0; i < NB; i++) {
      uint8_t v = b[i];
      if (i == 0) v &= first_mask;
      if (i == NB - 1) v &= last_mask;
      const int pos = BLS ? (NB - 1 - i) : i;               // significance of byte i
      r.l[pos >> 2] |= (uint32_t)v << (8 * (pos & 3));
    }
    return r;
  }
  ARK_HD static void limbs_to_bytes(const Fq& c, uint8_t* b) {
    for (int i = 0; i < NB; i++) {
      const int pos = BLS ? (NB - 1 - i) : i;
      b[i] = (uint8_t)(c.l[pos >> 2] >> (8 * (pos & 3)));
    }
  }
  ARK_HD static bool is_reduced(const Fq& c) {
    for (int i = Fq::N - 1; i >= 0; i--) {
      const uint32_t m = Fq::Params::mod(i);
      if (c.l[i] < m) return true;
      if (c.l[i] > m) return false;
    }
    return false;                                             // == q
  }
  // canonical y > canonical (q - y)
  ARK_HD static int cmp_neg(const Fq& y_mont) {               // +1: y > -y, 0: equal (y == 0), -1: y < -y
    const Fq y = Fq::from_mont(y_mont), n = Fq::from_mont(Fq::neg(y_mont));
    for (int i = Fq::N - 1; i >= 0; i--) {
      if (y.l[i] > n.l[i]) return 1;
      if (y.l[i] < n.l[i]) return -1;
    }
    return 0;
  }
  ARK_HD static bool gt_neg(const Fq& y) { return cmp_neg(y) > 0; }
  ARK_HD static bool gt_neg(const Fq2& y) {
    const int c = cmp_neg(y.c1);
    return c != 0 ? c > 0 : cmp_neg(y.c0) > 0;
  }

  // a^((q+1)/4): the square root for q = 3 mod 4 (both curves) when a is a residue
  ARK_HD static Fq pow_qp1_4(const Fq& a) {
    uint32_t e[Fq::N];
    uint64_t c = 1;
    for (int i = 0; i < Fq::N; i++) {
      c += Fq::Params::mod(i);
      e[i] = (uint32_t)c;
      c >>= 32;
    }
    for (int i = 0; i < Fq::N; i++) e[i] = (e[i] >> 2) | ((i + 1 < Fq::N ? e[i + 1] : (uint32_t)c) << 30);
    Fq result = Fq::one();
    bool started = false;
    for (int i = Fq::N - 1; i >= 0; i--) {
      for (int b = 31; b >= 0; b--) {
        if (started) result = Fq::mul_ni(result, result);
        if ((e[i] >> b) & 1) {
          result = started ? Fq::mul_ni(result, a) : a;
          started = true;
        }
      }
    }
    return started ? result : Fq::one();
  }
  ARK_HD static bool sqrt(const Fq& a, Fq* out) {
    const Fq r = pow_qp1_4(a);
    *out = r;
    return Fq::sqr_ni(r) == a;
  }
  // Fq[u]/(u^2 + 1), complex method: |a| = sqrt(a0^2 + a1^2), x0 = sqrt((a0 +- |a|)/2), x1 = a1 / (2 x0)
  ARK_HD static bool sqrt(const Fq2& a, Fq2* out) {
    if (a.c1.is_zero()) {
      Fq r;
      if (sqrt(a.c0, &r)) {
        *out = Fq2{r, Fq::zero()};
        return true;
      }
      if (sqrt(Fq::neg(a.c0), &r)) {
        *out = Fq2{Fq::zero(), r};
        return true;
      }
      return false;
    }
    Fq n;
    if (!sqrt(Fq::add(Fq::sqr_ni(a.c0), Fq::sqr_ni(a.c1)), &n)) return false;
    Fq two = Fq::add(Fq::one(), Fq::one());
    const Fq inv2 = Fq::inv(two);
    const Fq cands[2] = {Fq::mul_ni(Fq::add(a.c0, n), inv2), Fq::mul_ni(Fq::sub(a.c0, n), inv2)};
    for (int k = 0; k < 2; k++) {
      Fq x0;
      if (!sqrt(cands[k], &x0) || x0.is_zero()) continue;
      const Fq x1 = Fq::mul_ni(a.c1, Fq::inv(Fq::add(x0, x0)));
      const Fq2 x{x0, x1};
      if (Fq2::sqr_ni(x) == a) {
        *out = x;
        return true;
      }
    }
    return false;
  }

  ARK_HD static Fq curve_rhs(const Fq& x) { return Fq::add(Fq::mul_ni(Fq::sqr_ni(x), x), g1_b()); }
  ARK_HD static Fq2 curve_rhs(const Fq2& x) { return Fq2::add(Fq2::mul_ni(Fq2::sqr_ni(x), x), g2_b()); }

  static constexpr uint8_t FIRST_MASK = BLS ? 0x1F : 0xFF;
  static constexpr uint8_t LAST_MASK = BLS ? 0xFF : 0x3F;
  ARK_HD static size_t g1_size(bool compressed) { return (size_t)NB * (compressed ? 1 : 2); }
  ARK_HD static size_t g2_size(bool compressed) { return (size_t)NB * (compressed ? 2 : 4); }

  // flags of an encoded point of `size` bytes
  ARK_HD static void read_flags(const uint8_t* in, size_t size, bool* infinity, bool* sign, bool* comp_bit) {
    if (BLS) {
      *comp_bit = (in[0] & 0x80) != 0;
      *infinity = (in[0] & 0x40) != 0;
      *sign = (in[0] & 0x20) != 0;
    } else {
      *comp_bit = false;
      *infinity = (in[size - 1] & 0x40) != 0;
      *sign = (in[size - 1] & 0x80) != 0;
    }
  }

  ARK_HD static int g1_decode(const uint8_t* in, bool compressed, int validate, Affine<Fq>* out) {
    const size_t size = g1_size(compressed);
    bool inf, sign, cbit;
    read_flags(in, size, &inf, &sign, &cbit);
    if (!flags_ok(compressed, inf, sign, cbit)) return WIRE_BAD_FLAGS;
    if (inf) {
      // What upstream does with the bytes under an infinity flag differs by curve and does NOT depend on the validation mode:
      // ark-bls12-381's own point readers (curves/util.rs read_g1_* / read_g2_*) refuse a non-zero payload; ark-ec's generic
      // short-Weierstrass reader (BN254) parses the coordinates as field elements -- they must be reduced -- and then returns
      // the identity whatever they hold.
      if (BLS) {
        if (!payload_is_zero(in, size)) return WIRE_BAD_FLAGS;
      } else {
        if (!is_reduced(limbs_from_bytes(in, FIRST_MASK, compressed ? LAST_MASK : 0xFF))) return WIRE_NOT_REDUCED;
        if (!compressed && !is_reduced(limbs_from_bytes(in + NB, 0xFF, LAST_MASK))) return WIRE_NOT_REDUCED;
      }
      *out = Affine<Fq>::inf();
      return WIRE_OK;
    }
    // the flagged byte: first byte of the encoding (BLS) / last byte of the encoding (BN)
    const Fq xc = limbs_from_bytes(in, FIRST_MASK, compressed ? LAST_MASK : 0xFF);
    if (!is_reduced(xc)) return WIRE_NOT_REDUCED;
    const Fq x = Fq::to_mont(xc);
    Fq y;
    if (compressed) {
      if (!sqrt(curve_rhs(x), &y)) return WIRE_NOT_ON_CURVE;
      if (gt_neg(y) != sign) y = Fq::neg(y);
    } else {
      const Fq yc = limbs_from_bytes(in + NB, 0xFF, LAST_MASK);
      if (!is_reduced(yc)) return WIRE_NOT_REDUCED;
      y = Fq::to_mont(yc);
      if (validate && !(Fq::sqr_ni(y) == curve_rhs(x))) return WIRE_NOT_ON_CURVE;
    }
    *out = Affine<Fq>{x, y};
    if (G1_HAS_COFACTOR && validate == WIRE_VALIDATE_FULL && !in_subgroup(*out)) return WIRE_NOT_IN_SUBGROUP;
    return WIRE_OK;
  }

  ARK_HD static int g2_decode(const uint8_t* in, bool compressed, int validate, Affine<Fq2>* out) {
    const size_t size = g2_size(compressed);
    bool inf, sign, cbit;
    read_flags(in, size, &inf, &sign, &cbit);
    if (!flags_ok(compressed, inf, sign, cbit)) return WIRE_BAD_FLAGS;
    if (inf) {
      if (BLS) {                     // (see g1_decode)
        if (!payload_is_zero(in, size)) return WIRE_BAD_FLAGS;
      } else {
        const int nc = compressed ? 2 : 4;
        for (int k = 0; k < nc; k++)
          if (!is_reduced(limbs_from_bytes(in + (size_t)k * NB, (k == 0) ? FIRST_MASK : 0xFF, (k == nc - 1) ? LAST_MASK : 0xFF)))
            return WIRE_NOT_REDUCED;
      }
      *out = Affine<Fq2>::inf();
      return WIRE_OK;
    }
    // element order on the wire: BLS c1 || c0 (flags on the very first byte), BN c0 || c1 (flags on the very last byte)
    const int ncoord = compressed ? 2 : 4;
    Fq c[4];
    for (int k = 0; k < ncoord; k++) {
      const uint8_t fm = (k == 0) ? FIRST_MASK : 0xFF;
      const uint8_t lm = (k == ncoord - 1) ? LAST_MASK : 0xFF;
      c[k] = limbs_from_bytes(in + (size_t)k * NB, fm, lm);
      if (!is_reduced(c[k])) return WIRE_NOT_REDUCED;
      c[k] = Fq::to_mont(c[k]);
    }
    const Fq2 x = BLS ? Fq2{c[1], c[0]} : Fq2{c[0], c[1]};
    Fq2 y;
    if (compressed) {
      if (!sqrt(curve_rhs(x), &y)) return WIRE_NOT_ON_CURVE;
      if (gt_neg(y) != sign) y = Fq2::neg(y);
    } else {
      y = BLS ? Fq2{c[3], c[2]} : Fq2{c[2], c[3]};
      if (validate && !(Fq2::sqr_ni(y) == curve_rhs(x))) return WIRE_NOT_ON_CURVE;
    }
    *out = Affine<Fq2>{x, y};
    if (validate == WIRE_VALIDATE_FULL && !in_subgroup(*out)) return WIRE_NOT_IN_SUBGROUP;
    return WIRE_OK;
  }

  ARK_HD static void g1_encode(const Affine<Fq>& p, bool compressed, uint8_t* out) {
    const size_t size = g1_size(compressed);
    for (size_t i = 0; i < size; i++) out[i] = 0;
    if (p.is_inf()) {
      if (BLS) out[0] = compressed ? 0xC0 : 0x40;
      else out[size - 1] = 0x40;
      return;
    }
    limbs_to_bytes(Fq::from_mont(p.x), out);
    if (!compressed) limbs_to_bytes(Fq::from_mont(p.y), out + NB);
    const bool big = gt_neg(p.y);
    if (BLS) {
      if (compressed) out[0] |= 0x80 | (big ? 0x20 : 0);
    } else if (big) {
      out[size - 1] |= 0x80;
    }
  }

  ARK_HD static void g2_encode(const Affine<Fq2>& p, bool compressed, uint8_t* out) {
    const size_t size = g2_size(compressed);
    for (size_t i = 0; i < size; i++) out[i] = 0;
    if (p.is_inf()) {
      if (BLS) out[0] = compressed ? 0xC0 : 0x40;
      else out[size - 1] = 0x40;
      return;
    }
    const Fq x0 = Fq::from_mont(p.x.c0), x1 = Fq::from_mont(p.x.c1);
    limbs_to_bytes(BLS ? x1 : x0, out);
    limbs_to_bytes(BLS ? x0 : x1, out + NB);
    if (!compressed) {
      const Fq y0 = Fq::from_mont(p.y.c0), y1 = Fq::from_mont(p.y.c1);
      limbs_to_bytes(BLS ? y1 : y0, out + 2 * NB);
      limbs_to_bytes(BLS ? y0 : y1, out + 3 * NB);
    }
    const bool big = gt_neg(p.y);
    if (BLS) {
      if (compressed) out[0] |= 0x80 | (big ? 0x20 : 0);
    } else if (big) {
      out[size - 1] |= 0x80;
    }
  }
};

// one lane per point; *err receives the smallest (index + 1) << 4 | status of a failing point (0 = all good)
template <class Curve, int GROUP>
__global__ void __launch_bounds__(128)
wire_decode_kernel(const uint8_t* __restrict__ in, uint64_t n, int compressed, int validate, void* __restrict__ out,
                   unsigned long long* __restrict__ err) {
  using W = Wire<Curve>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int st;
  if constexpr (GROUP == 1) {
    Affine<typename Curve::Fq> p = Affine<typename Curve::Fq>::inf();
    st = W::g1_decode(in + i * W::g1_size(compressed != 0), compressed != 0, validate, &p);
    reinterpret_cast<Affine<typename Curve::Fq>*>(out)[i] = p;
  } else {
    Affine<typename Curve::Fq2> p = Affine<typename Curve::Fq2>::inf();
    st = W::g2_decode(in + i * W::g2_size(compressed != 0), compressed != 0, validate, &p);
    reinterpret_cast<Affine<typename Curve::Fq2>*>(out)[i] = p;
  }
  if (st != WIRE_OK) {
    const unsigned long long code = ((i + 1) << 4) | (unsigned long long)st;
    // keep the failure with the lowest index (0 means "none yet")
    unsigned long long cur = *err;
    while (cur == 0 || code < cur) {
      const unsigned long long prev = atomicCAS(err, cur, code);
      if (prev == cur) break;
      cur = prev;
    }
  }
}

template <class Curve, int GROUP>
__global__ void __launch_bounds__(128)
wire_encode_kernel(const void* __restrict__ in, uint64_t n, int compressed, uint8_t* __restrict__ out) {
  using W = Wire<Curve>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (GROUP == 1)
    W::g1_encode(reinterpret_cast<const Affine<typename Curve::Fq>*>(in)[i], compressed != 0,
                 out + i * W::g1_size(compressed != 0));
  else
    W::g2_encode(reinterpret_cast<const Affine<typename Curve::Fq2>*>(in)[i], compressed != 0,
                 out + i * W::g2_size(compressed != 0));
}

static inline const char* wire_status_name(int st) {
  switch (st) {
    case WIRE_NOT_REDUCED: return "coordinate not reduced";
    case WIRE_NOT_ON_CURVE: return "point not on curve";
    case WIRE_BAD_FLAGS: return "bad flag bits";
    case WIRE_NOT_IN_SUBGROUP: return "point not in the prime-order subgroup";
    default: return "ok";
  }
}

}  // namespace ark355
