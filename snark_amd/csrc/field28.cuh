// Radix-2^28 Montgomery arithmetic for the bucket-accumulation kernels (device + host/emulator).
//
// Why a second limb size.  gfx950 has a 32x32+64 multiply-add (v_mad_u64_u32) but no carry-IN on it, so the
// saturated 32-bit-limb product scanning of field.cuh pays one v_addc_co_u32 per multiply-add: 35% of the bucket
// kernel's VALU instructions are carry bookkeeping (round-1 ISA listing).  With 28-bit limbs a column of a
// 14-limb product-scanning pass (<= 14 a*b + 14 m*p products of < 2^60) fits a 64-bit accumulator with room to
// spare: NO carry instruction at all, no per-product reduce_once, and additions/subtractions become limb-wise
// (lazy) operations.  (hipcc starts each column on a second accumulator and joins the two with one 64-bit add;
// forcing a single dependent chain was measured: no difference.)  BLS12-381 Fq: 392 mad + 68 other VALU instead of 300 mad + 300 addc + ~50.
//
// Representation.  N = 14 (BLS12-381 Fq) / 10 (BN254 Fq) limbs, value = sum l[i] 2^(28 i), Montgomery radix
// R' = 2^(28 N) (R' = R * 2^SHIFT with R = 2^(32 NB) of field.cuh).  Values are NOT kept canonical:
//   * "N"  (normalised): every limb < 2^28 except possibly the top one; value < 2^11 p.
//   * "L1" (lazy):       limbs < 2^30 (sums / biased differences of N values).
//   mul/mul2sum accept operands whose limb bounds 2^x, 2^y satisfy x + y <= 60 (see the column bound in mul) AND,
//   since the Karatsuba level of round 5 is the default (mulsum -> mulsum_kara), whose EVERY limb is below 2^31: the
//   level multiplies signed differences of limbs (v_mad_i64_i32), so a 2^32 limb against a 2^28 limb -- legal for the
//   schoolbook pass -- is not.  Every caller in this library stays below 2^30.6; the emulator build traps on a limb
//   >= 2^31 (ARK_F28_TRAP in mulsum_kara), the device build does not check.  They
//   return an N value < 1.05 p as long as the product of the operand VALUES is < 2^10 p^2 -- far above anything
//   the mixed addition produces.  Only the kernel boundary converts: window tables are converted to this form
//   once per key (from_fp), flushed bucket partials are converted back to the canonical 32-bit Montgomery form
//   (to_fp), so nothing outside the accumulation kernels sees a non-canonical value and no output byte changes.
#pragma once
#include "field.cuh"

namespace ark355 {

#if defined(ARK_EMUL)
// emulator builds check every 64-bit column accumulator against a 128-bit shadow and every lazy limb operation
// against wrap-around: the bounds analysis written next to the formulas is enforced, not assumed
#define ARK_F28_CHECK 1
#include <stdio.h>
#include <stdlib.h>
#define ARK_F28_TRAP() (fprintf(stderr, "field28.cuh:%d: limb / column bound violated\n", __LINE__), abort())
#endif

template <class P>
struct Fp28 {
  using Base = Fp<P>;
  using Params = P;
  static constexpr int NB = Base::N;
  static constexpr int N = (P::BITS + 8 + 27) / 28;
  static constexpr int RBITS = 28 * N;
  static constexpr int SHIFT = RBITS - 32 * NB;
  static constexpr uint32_t MASK = (1u << 28) - 1u;
  static constexpr uint32_t INV = P::INV & MASK;        // -p^-1 mod 2^28
  static_assert(SHIFT >= 0 && SHIFT < 32, "radix mismatch");
  static_assert(RBITS >= P::BITS + 8, "needs 8 bits of head-room");
  uint32_t l[N];

  struct Limbs {
    uint32_t v[N];
  };
  // k*p in the normalised 28-bit form (k < 2^10); compile-time only
  ARK_HD static constexpr Limbs make_kp(uint32_t k) {
    uint32_t w[NB + 2] = {};
    uint64_t c = 0;
    for (int j = 0; j < NB; j++) {
      c += (uint64_t)P::mod(j) * k;
      w[j] = (uint32_t)c;
      c >>= 32;
    }
    w[NB] = (uint32_t)c;
    w[NB + 1] = 0;
    Limbs out = {};
    for (int i = 0; i < N; i++) {
      const int bit = 28 * i, q = bit / 32, r = bit % 32;
      uint64_t v = 0;
      if (q < NB + 2) v = w[q];
      if (q + 1 < NB + 2) v |= (uint64_t)w[q + 1] << 32;
      out.v[i] = (uint32_t)(v >> r) & MASK;
    }
    return out;
  }
  // k*p in "borrow-friendly" form: every limb except the top one is raised by BETA*2^28 (the borrow is taken
  // from the next limb), so that bias(i) - x.l[i] >= 0 for any x with limbs <= BETA*(2^28 - 1) and value
  // < (k-1) p.  Same integer value k*p.
  ARK_HD static constexpr Limbs make_bias(uint32_t k, uint32_t beta) {
    Limbs d = make_kp(k);
    for (int i = 0; i < N; i++) {
      if (i == 0) d.v[i] += beta << 28;
      else if (i == N - 1) d.v[i] -= beta;
      else d.v[i] += (beta << 28) - beta;
    }
    return d;
  }
  template <uint32_t K>
  ARK_HD static constexpr uint32_t kp(int i) {
    constexpr Limbs t = make_kp(K);
    return t.v[i];
  }
  template <uint32_t K, uint32_t BETA>
  ARK_HD static constexpr uint32_t bias(int i) {
    constexpr Limbs t = make_bias(K, BETA);
    return t.v[i];
  }

  // Register image for values RETURNED by out-of-line (cold) functions of the bucket kernels: a vector type comes back
  // in VGPRs, a struct of N limbs through a hidden pointer -- and `acc.x = cold_call()` then hands the address of the
  // hot loop's accumulator to the callee, which keeps the whole accumulator in scratch memory.
  typedef uint32_t Vec __attribute__((vector_size(64)));
  static_assert(N <= 16, "Vec holds 16 limbs");
  ARK_HD Vec to_vec() const {
    Vec v = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = l[i];
    return v;
  }
  ARK_HD static Fp28 from_vec(const Vec& v) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = v[i];
    return r;
  }

  ARK_HD static Fp28 zero() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  ARK_HD bool limbs_all_zero() const {
    uint32_t a = 0;
#pragma unroll
    for (int i = 0; i < N; i++) a |= l[i];
    return a == 0;
  }

  // ---- lazy limb-wise operations ---------------------------------------------------------------------------
  ARK_HD static Fp28 add(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) {
#if ARK_F28_CHECK
      if ((uint64_t)a.l[i] + b.l[i] > 0xFFFFFFFFull) ARK_F28_TRAP();
#endif
      r.l[i] = a.l[i] + b.l[i];
    }
    return r;
  }
  // a - b + K p   (b: limbs <= BETA (2^28 - 1), value < (K - 1) p)
  template <uint32_t K, uint32_t BETA>
  ARK_HD static Fp28 sub(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) {
#if ARK_F28_CHECK
      if (bias<K, BETA>(i) < b.l[i] || (uint64_t)a.l[i] + bias<K, BETA>(i) - b.l[i] > 0xFFFFFFFFull) ARK_F28_TRAP();
#endif
      r.l[i] = a.l[i] + (bias<K, BETA>(i) - b.l[i]);
    }
    return r;
  }
  // K p - b
  template <uint32_t K, uint32_t BETA>
  ARK_HD static Fp28 neg(const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) {
#if ARK_F28_CHECK
      if (bias<K, BETA>(i) < b.l[i]) ARK_F28_TRAP();      // a limb would wrap: the bias class is too small
#endif
      r.l[i] = bias<K, BETA>(i) - b.l[i];
    }
    return r;
  }
  // carry propagation: limbs < 2^28 except the top one (which keeps whatever is left)
  ARK_HD static Fp28 norm(const Fp28& a) {
    Fp28 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      const uint32_t t = a.l[i] + c;
      r.l[i] = t & MASK;
      c = t >> 28;
    }
    r.l[N - 1] = a.l[N - 1] + c;
    return r;
  }

  // ---- Montgomery products ---------------------------------------------------------------------------------
  struct Col {
    uint64_t acc;
#if ARK_F28_CHECK
    unsigned __int128 shadow;
#endif
    ARK_HD void init() {
      acc = 0;
#if ARK_F28_CHECK
      shadow = 0;
#endif
    }
    ARK_HD void mad(uint32_t x, uint32_t y) {
      acc += (uint64_t)x * y;
#if ARK_F28_CHECK
      shadow += (unsigned __int128)x * y;
      if ((shadow >> 64) != 0) ARK_F28_TRAP();       // a column overflowed: the limb-bound analysis is wrong
#endif
    }
    ARK_HD void shift() {
      acc >>= 28;
#if ARK_F28_CHECK
      shadow >>= 28;
#endif
    }
  };

  // column K of sum_j x_j*y_j (J operand pairs), low half: also produces m[K]
  template <int K, int J>
  ARK_HD static void col_lo(Col& c, const Fp28* const (&x)[J], const Fp28* const (&y)[J], uint32_t* m) {
    if constexpr (K < N) {
#pragma unroll
      for (int j = 0; j < J; j++) {
#pragma unroll
        for (int i = 0; i <= K; i++) c.mad(x[j]->l[i], y[j]->l[K - i]);
      }
#pragma unroll
      for (int i = 0; i < K; i++) c.mad(m[i], kp<1>(K - i));
      m[K] = ((uint32_t)c.acc * INV) & MASK;
      c.mad(m[K], kp<1>(0));
      c.shift();
      col_lo<K + 1, J>(c, x, y, m);
    }
  }
  template <int K, int J>
  ARK_HD static void col_hi(Col& c, const Fp28* const (&x)[J], const Fp28* const (&y)[J], const uint32_t* m,
                            Fp28& r) {
    if constexpr (K < 2 * N - 1) {
#pragma unroll
      for (int j = 0; j < J; j++) {
#pragma unroll
        for (int i = K - N + 1; i < N; i++) c.mad(x[j]->l[i], y[j]->l[K - i]);
      }
#pragma unroll
      for (int i = K - N + 1; i < N; i++) c.mad(m[i], kp<1>(K - i));
      r.l[K - N] = (uint32_t)c.acc & MASK;
      c.shift();
      col_hi<K + 1, J>(c, x, y, m, r);
    }
  }
  // (sum_j x_j * y_j) / R' mod p, one reduction.  Column bound: J*N products of < 2^(x+y) plus N of < 2^56 plus a
  // carry-in of < 2^37 must stay below 2^64: J = 1: x + y <= 60, J = 2: x + y <= 59 (or 59.6 + 57), J = 4: 58.
  template <int J>
  ARK_HD static Fp28 mulsum_school(const Fp28* const (&x)[J], const Fp28* const (&y)[J]) {
    uint32_t m[N];
    Fp28 r;
    Col c;
    c.init();
    col_lo<0, J>(c, x, y, m);
    col_hi<N, J>(c, x, y, m, r);
    r.l[N - 1] = (uint32_t)c.acc;
    return r;
  }

  // ---- one Karatsuba level over the operand products (round 5) --------------------------------------------
  // x = x0 + x1 B^H, y = y0 + y1 B^H (H = N / 2, B = 2^28):
  //     x y = (1 + B^H)(x0 y0 + x1 y1 B^H) + (x1 - x0)(y0 - y1) B^H
  // With G_k = column k of x0 y0 + x1 y1 B^H (k <= 3H - 2; the two half products share ONE accumulator chain per
  // column) and D_k = column k of the SIGNED product (x1 - x0)(y0 - y1) (v_mad_i64_i32; k <= 2H - 2):
  //     column k of x y  =  G_k + G_{k-H} + D_{k-H}
  // 3 H^2 = 147 operand multiply-adds per product instead of 4 H^2 = 196 (N = 14), paid with 2H subtractions per
  // product and H-1 more 64-bit additions per REDUCTION (shared by the J products of a mulsum: the dual / quad
  // product passes of the lane-pair G2 kernel save 98 / 196 multiply-adds for the same 13 additions).  Every
  // column's true value is the schoolbook column (same bounds, same result limb for limb); partial sums of the
  // signed chain may wrap modulo 2^64, which is harmless -- the emulator build checks each column when it closes.
  static constexpr int H = N / 2;
  static_assert(N % 2 == 0, "Karatsuba split needs an even limb count");
  struct KCol {
    uint64_t acc;
#if ARK_F28_CHECK
    __int128 shadow;
#endif
    ARK_HD void init() {
      acc = 0;
#if ARK_F28_CHECK
      shadow = 0;
#endif
    }
    ARK_HD void mad(uint32_t a, uint32_t b) {
      acc += (uint64_t)a * b;
#if ARK_F28_CHECK
      shadow += (__int128)((unsigned __int128)a * b);
#endif
    }
    ARK_HD void mads(int32_t a, int32_t b) {
      acc += (uint64_t)((int64_t)a * (int64_t)b);
#if ARK_F28_CHECK
      shadow += (__int128)a * (__int128)b;
#endif
    }
    ARK_HD void add(const KCol& o) {
      acc += o.acc;
#if ARK_F28_CHECK
      shadow += o.shadow;
#endif
    }
    // the column is complete: its true value must be a non-negative 64-bit number
    ARK_HD void close() const {
#if ARK_F28_CHECK
      if (shadow < 0 || (shadow >> 64) != 0 || (uint64_t)shadow != acc) ARK_F28_TRAP();
#endif
    }
    ARK_HD void shift() {
      acc >>= 28;
#if ARK_F28_CHECK
      shadow >>= 28;
#endif
    }
  };
  // G_K: column K of x0 y0 + x1 y1 B^H, summed over the J operand pairs
  template <int K, int J>
  ARK_HD static KCol kara_g(const Fp28* const (&x)[J], const Fp28* const (&y)[J]) {
    KCol g;
    g.init();
#pragma unroll
    for (int j = 0; j < J; j++) {
      if constexpr (K <= 2 * H - 2) {
        constexpr int lo = (K - H + 1 > 0) ? K - H + 1 : 0, hi = (K < H - 1) ? K : H - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) g.mad(x[j]->l[i], y[j]->l[K - i]);
      }
      if constexpr (K >= H && K <= 3 * H - 2) {
        constexpr int K1 = K - H;
        constexpr int lo = (K1 - H + 1 > 0) ? K1 - H + 1 : 0, hi = (K1 < H - 1) ? K1 : H - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) g.mad(x[j]->l[H + i], y[j]->l[H + K1 - i]);
      }
    }
    return g;
  }
  // operand part of column K: G_K + G_{K-H} + D_{K-H}; g[] is a ring of the last H values of G
  template <int K, int J>
  ARK_HD static void kara_col(KCol& c, KCol (&g)[H], const Fp28* const (&x)[J], const Fp28* const (&y)[J],
                              const int32_t (&dx)[J][H], const int32_t (&dy)[J][H]) {
    if constexpr (K >= H && K <= 4 * H - 2) {
      KCol t = g[K % H];                                   // G_{K-H}: last use
      constexpr int K1 = K - H;
      if constexpr (K1 <= 2 * H - 2) {
        constexpr int lo = (K1 - H + 1 > 0) ? K1 - H + 1 : 0, hi = (K1 < H - 1) ? K1 : H - 1;
#pragma unroll
        for (int j = 0; j < J; j++) {
#pragma unroll
          for (int i = lo; i <= hi; i++) t.mads(dx[j][i], dy[j][K1 - i]);
        }
      }
      c.add(t);
    }
    if constexpr (K <= 3 * H - 2) {
      g[K % H] = kara_g<K, J>(x, y);
      c.add(g[K % H]);
    }
  }
  template <int K, int J>
  ARK_HD static void kcol_lo(KCol& c, KCol (&g)[H], const Fp28* const (&x)[J], const Fp28* const (&y)[J],
                             const int32_t (&dx)[J][H], const int32_t (&dy)[J][H], uint32_t* m) {
    if constexpr (K < N) {
      kara_col<K, J>(c, g, x, y, dx, dy);
#pragma unroll
      for (int i = 0; i < K; i++) c.mad(m[i], kp<1>(K - i));
      m[K] = ((uint32_t)c.acc * INV) & MASK;
      c.mad(m[K], kp<1>(0));
      c.close();
      c.shift();
      kcol_lo<K + 1, J>(c, g, x, y, dx, dy, m);
    }
  }
  template <int K, int J>
  ARK_HD static void kcol_hi(KCol& c, KCol (&g)[H], const Fp28* const (&x)[J], const Fp28* const (&y)[J],
                             const int32_t (&dx)[J][H], const int32_t (&dy)[J][H], const uint32_t* m, Fp28& r) {
    if constexpr (K < 2 * N - 1) {
      kara_col<K, J>(c, g, x, y, dx, dy);
#pragma unroll
      for (int i = K - N + 1; i < N; i++) c.mad(m[i], kp<1>(K - i));
      c.close();
      r.l[K - N] = (uint32_t)c.acc & MASK;
      c.shift();
      kcol_hi<K + 1, J>(c, g, x, y, dx, dy, m, r);
    }
  }
  // CONTRACT (beyond the column bound of mulsum_school): every limb of every operand < 2^31 -- the half differences below
  // must fit a signed 32-bit multiplier operand.  Checked by the emulator build only.
  template <int J>
  ARK_HD static Fp28 mulsum_kara(const Fp28* const (&x)[J], const Fp28* const (&y)[J]) {
    int32_t dx[J][H], dy[J][H];
#pragma unroll
    for (int j = 0; j < J; j++) {
#pragma unroll
      for (int i = 0; i < H; i++) {
        dx[j][i] = (int32_t)(x[j]->l[H + i] - x[j]->l[i]);       // limbs < 2^31: the difference fits
        dy[j][i] = (int32_t)(y[j]->l[i] - y[j]->l[H + i]);
#if ARK_F28_CHECK
        if ((x[j]->l[H + i] | x[j]->l[i] | y[j]->l[i] | y[j]->l[H + i]) >> 31) ARK_F28_TRAP();
#endif
      }
    }
    uint32_t m[N];
    Fp28 r;
    KCol c, g[H];
    c.init();
#pragma unroll
    for (int i = 0; i < H; i++) g[i].init();
    kcol_lo<0, J>(c, g, x, y, dx, dy, m);
    kcol_hi<N, J>(c, g, x, y, dx, dy, m, r);
    r.l[N - 1] = (uint32_t)c.acc;
    return r;
  }
#ifndef ARK_F28_KARATSUBA
#define ARK_F28_KARATSUBA 1
#endif
  template <int J>
  ARK_HD static Fp28 mulsum(const Fp28* const (&x)[J], const Fp28* const (&y)[J]) {
#if ARK_F28_KARATSUBA
    return mulsum_kara<J>(x, y);
#else
    return mulsum_school<J>(x, y);
#endif
  }
  ARK_HD static Fp28 mul(const Fp28& a, const Fp28& b) {
    const Fp28* const x[1] = {&a};
    const Fp28* const y[1] = {&b};
    return mulsum<1>(x, y);
  }
  // a^2 / R': the cross products are taken once against the doubled operand (limb-wise doubling is free of
  // carries here), 105 instead of 196 operand products for N = 14.  Limbs of a < 2^29.6: the widest column is
  // 7 * 2^60.2 + 14 * 2^56 < 2^63.2.
  template <int K>
  ARK_HD static void sqr_col(Col& c, const Fp28& a, const Fp28& a2) {
    constexpr int lo = (K - N + 1 > 0) ? K - N + 1 : 0;
#pragma unroll
    for (int i = lo; 2 * i < K; i++) c.mad(a2.l[i], a.l[K - i]);
    if constexpr (K % 2 == 0) c.mad(a.l[K / 2], a.l[K / 2]);
  }
  template <int K>
  ARK_HD static void sqr_lo(Col& c, const Fp28& a, const Fp28& a2, uint32_t* m) {
    if constexpr (K < N) {
      sqr_col<K>(c, a, a2);
#pragma unroll
      for (int i = 0; i < K; i++) c.mad(m[i], kp<1>(K - i));
      m[K] = ((uint32_t)c.acc * INV) & MASK;
      c.mad(m[K], kp<1>(0));
      c.shift();
      sqr_lo<K + 1>(c, a, a2, m);
    }
  }
  template <int K>
  ARK_HD static void sqr_hi(Col& c, const Fp28& a, const Fp28& a2, const uint32_t* m, Fp28& r) {
    if constexpr (K < 2 * N - 1) {
      sqr_col<K>(c, a, a2);
#pragma unroll
      for (int i = K - N + 1; i < N; i++) c.mad(m[i], kp<1>(K - i));
      r.l[K - N] = (uint32_t)c.acc & MASK;
      c.shift();
      sqr_hi<K + 1>(c, a, a2, m, r);
    }
  }
  ARK_HD static Fp28 sqr(const Fp28& a) {
    Fp28 a2;
#pragma unroll
    for (int i = 0; i < N; i++) a2.l[i] = a.l[i] << 1;
    uint32_t m[N];
    Fp28 r;
    Col c;
    c.init();
    sqr_lo<0>(c, a, a2, m);
    sqr_hi<N>(c, a, a2, m, r);
    r.l[N - 1] = (uint32_t)c.acc;
    return r;
  }
  ARK_HD static Fp28 mul2sum(const Fp28& a, const Fp28& b, const Fp28& c, const Fp28& d) {
    const Fp28* const x[2] = {&a, &c};
    const Fp28* const y[2] = {&b, &d};
    return mulsum<2>(x, y);
  }
  ARK_HD static Fp28 mul4sum(const Fp28& a, const Fp28& b, const Fp28& c, const Fp28& d, const Fp28& e,
                             const Fp28& f, const Fp28& g, const Fp28& h) {
    const Fp28* const x[4] = {&a, &c, &e, &g};
    const Fp28* const y[4] = {&b, &d, &f, &h};
    return mulsum<4>(x, y);
  }

  // ---- the cheap "is this a multiple of p" filter ------------------------------------------------------------
  // v = k p  =>  k = v p^-1 mod 2^28 = -(v INV) mod 2^28.  For a value that is NOT a multiple of p the result is
  // uniform in [0, 2^28), so "k < bound" passes a non-multiple with probability bound / 2^28; callers confirm a
  // hit with is_zero_mod_p() on a cold path.
  ARK_HD uint32_t multiple_hint() const { return (0u - (l[0] & MASK) * INV) & MASK; }

  // ---- boundary conversions (cold) ---------------------------------------------------------------------------
  // a >= k p ? a - k p : a     (a normalised)
  template <uint32_t K>
  ARK_HD static Fp28 cond_sub(const Fp28& a) {
    Fp28 d;
    int32_t br = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      const int32_t t = (int32_t)a.l[i] - (int32_t)kp<K>(i) + br;
      d.l[i] = (uint32_t)t & MASK;
      br = t >> 28;                      // arithmetic: 0 or -1
    }
    const int32_t t = (int32_t)a.l[N - 1] - (int32_t)kp<K>(N - 1) + br;
    d.l[N - 1] = (uint32_t)t;
    return t < 0 ? a : d;
  }
  // canonical representative in [0, p), normalised limbs.  Input: any lazy value < 32 p.
  ARK_HD_NOINLINE static Fp28 canon(Fp28 a) {
    Fp28 r = norm(a);
    r = cond_sub<16>(r);
    r = cond_sub<8>(r);
    r = cond_sub<4>(r);
    r = cond_sub<2>(r);
    r = cond_sub<1>(r);
    return r;
  }
  ARK_HD_NOINLINE static bool is_zero_mod_p(Fp28 a) { return canon(a).limbs_all_zero(); }
  // Inlined flavour for the bucket kernels: a CALL with the accumulator live across it forces those values into the
  // few callee-saved VGPRs; they did not fit, and hipcc kept one accumulator coordinate in scratch for the whole loop.
  ARK_HD static bool is_zero_mod_p_inl(const Fp28& a) {
    Fp28 r = norm(a);
    r = cond_sub<16>(r);
    r = cond_sub<8>(r);
    r = cond_sub<4>(r);
    r = cond_sub<2>(r);
    r = cond_sub<1>(r);
    return r.limbs_all_zero();
  }

  // x R mod p (32-bit Montgomery form, canonical)  ->  x R' mod p, normalised 28-bit limbs, canonical
  ARK_HD_NOINLINE static Fp28 from_fp(Base xin) {
    Base x = xin;
    for (int s = 0; s < SHIFT; s++) x = Base::dbl(x);
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int bit = 28 * i, q = bit / 32, sh = bit % 32;
      uint64_t v = (q < NB) ? x.l[q] : 0u;
      if (q + 1 < NB) v |= (uint64_t)x.l[q + 1] << 32;
      r.l[i] = (uint32_t)(v >> sh) & MASK;
    }
    return r;
  }
  // canonical limbs (value < p) -> the 32-bit Montgomery form: repack, then divide by 2^SHIFT in ONE Montgomery-style
  // step: t = -x p^-1 mod 2^SHIFT makes x + t p divisible by 2^SHIFT, and (x + t p) / 2^SHIFT < 2p.  (The first
  // version halved SHIFT times: 8 / 24 carry chains per coordinate, most of the cost of a bucket flush on BN254.)
  ARK_HD static Base repack_and_unshift(const Fp28& c) {
    Base x = Base::zero();
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int bit = 28 * i, q = bit / 32, sh = bit % 32;
      const uint64_t v = (uint64_t)c.l[i] << sh;
      if (q < NB) x.l[q] |= (uint32_t)v;
      if (q + 1 < NB) x.l[q + 1] |= (uint32_t)(v >> 32);
    }
    if constexpr (SHIFT > 0) {
      const uint32_t t = (x.l[0] * P::INV) & ((1u << SHIFT) - 1u);
      uint32_t w[NB + 1];
      uint64_t cy = 0;
#pragma unroll
      for (int i = 0; i < NB; i++) {
        cy += (uint64_t)t * P::mod(i) + x.l[i];
        w[i] = (uint32_t)cy;
        cy >>= 32;
      }
      w[NB] = (uint32_t)cy;
      Base r;
#pragma unroll
      for (int i = 0; i < NB; i++) r.l[i] = (w[i] >> SHIFT) | (w[i + 1] << (32 - SHIFT));
      return Base::reduce_once(r, 0u);
    } else {
      return x;
    }
  }
  // inverse of from_fp for any lazy value < 32 p
  ARK_HD_NOINLINE static Base to_fp(Fp28 a) { return repack_and_unshift(canon(a)); }
  // the same for values < 8 p (everything a bucket accumulator holds): three conditional subtractions instead of five
  ARK_HD_NOINLINE static Base to_fp_lt8(Fp28 a) {
    Fp28 r = norm(a);
    r = cond_sub<4>(r);
    r = cond_sub<2>(r);
    r = cond_sub<1>(r);
    return repack_and_unshift(r);
  }
};

using BlsFq28 = Fp28<BlsFqParams>;
using BnFq28 = Fp28<BnFqParams>;

}  // namespace ark355
