// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and the host side of the library).
//
// Replaces, on the device, what the un-vendored ark-ff `Fp<MontBackend<_, N>>` does on the CPU
// (ark-ff/src/fields/models/fp/montgomery_backend.rs); the in-memory image is identical:
// little-endian limbs of a*R mod p with R = 2^(64*N64) = 2^(32*N32), so buffers produced by the
// arkworks host (SURVEY.md 8b "Numeric conventions") are consumed without conversion.
//
// CDNA4 has no 64x64 multiplier: the inner product is v_mad_u64_u32 (32x32+64 -> 64), hence
// 32-bit limbs.  Everything is fully unrolled so limbs live in VGPRs.
#pragma once
#include <stdint.h>
#include "hd.h"
#include "curve_params.h"

namespace ark355 {

template <class P>
struct Fp {
  static constexpr int N = P::N;
  using Params = P;
  uint32_t l[N];

  ARK_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  ARK_HD static Fp one() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::one(i);
    return r;
  }
  ARK_HD static Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::r2(i);
    return r;
  }
  ARK_HD bool is_zero() const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= l[i];
    return acc == 0;
  }
  ARK_HD bool operator==(const Fp& o) const {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) acc |= (l[i] ^ o.l[i]);
    return acc == 0;
  }
  ARK_HD bool operator!=(const Fp& o) const { return !(*this == o); }

  // r = a - p if a >= p else a   (a < 2p)
  // Carry chains.  hipcc (clang) lowers __builtin_addc/__builtin_subc to v_add_co/v_addc_co/v_subb_co chains -- one
  // VALU instruction per limb; the portable 64-bit formulation compiles to ~6 instructions per limb on gfx950
  // (v_lshl_add_u64 + sign-extension moves; round-1 ISA listing: 13% of the bucket kernel).  g++ (the emulator
  // build of the test tier) takes the portable path; both produce identical values.
#if defined(__clang__)
  ARK_HD static uint32_t adc(uint32_t a, uint32_t b, uint32_t& carry) {
    unsigned c;
    uint32_t r = __builtin_addc(a, b, carry, &c);
    carry = c;
    return r;
  }
  ARK_HD static uint32_t sbb(uint32_t a, uint32_t b, uint32_t& borrow) {
    unsigned c;
    uint32_t r = __builtin_subc(a, b, borrow, &c);
    borrow = c;
    return r;
  }
#else
  ARK_HD static uint32_t adc(uint32_t a, uint32_t b, uint32_t& carry) {
    uint64_t t = (uint64_t)a + b + carry;
    carry = (uint32_t)(t >> 32);
    return (uint32_t)t;
  }
  ARK_HD static uint32_t sbb(uint32_t a, uint32_t b, uint32_t& borrow) {
    uint64_t t = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(t >> 32) & 1u;
    return (uint32_t)t;
  }
#endif

  ARK_HD static Fp reduce_once(const Fp& a, uint32_t top) {
    Fp d;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.l[i] = sbb(a.l[i], P::mod(i), br);
    // a >= p  <=>  no final borrow, or the (N+1)-th limb `top` absorbs it
    bool ge = (top != 0) || (br == 0);
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = ge ? d.l[i] : a.l[i];
    return r;
  }

  ARK_HD static Fp add(const Fp& a, const Fp& b) {
    Fp s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s.l[i] = adc(a.l[i], b.l[i], c);
    return reduce_once(s, c);
  }

  ARK_HD static Fp sub(const Fp& a, const Fp& b) {
    Fp d;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d.l[i] = sbb(a.l[i], b.l[i], br);
    uint32_t mask = (uint32_t)0 - br;
    uint32_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = adc(d.l[i], P::mod(i) & mask, c);
    return r;
  }

  ARK_HD static Fp neg(const Fp& a) {
    if (a.is_zero()) return a;
    Fp r;
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = sbb(P::mod(i), a.l[i], br);
    return r;
  }

  ARK_HD static Fp dbl(const Fp& a) { return add(a, a); }

  // Montgomery product a*b*R^-1 mod p.
  //
  // Host (and ARK_NO_ASM_MUL) flavour: CIOS, operand scanning, plain C.  The moduli used here all leave at
  // least one spare bit in the top limb, so the running value fits in N+1 limbs.
  ARK_HD static Fp mul_c(const Fp& a, const Fp& b) {
    static_assert(P::BITS <= 32 * N - 1, "needs a spare top bit");
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint64_t c = 0;
      const uint32_t bi = b.l[i];
#pragma unroll
      for (int j = 0; j < N; j++) {
        uint64_t x = (uint64_t)a.l[j] * bi + t[j] + c;
        t[j] = (uint32_t)x;
        c = x >> 32;
      }
      uint64_t top = (uint64_t)t[N] + c;
      const uint32_t m = t[0] * P::INV;
      uint64_t x = (uint64_t)m * P::mod(0) + t[0];
      c = x >> 32;
#pragma unroll
      for (int j = 1; j < N; j++) {
        x = (uint64_t)m * P::mod(j) + t[j] + c;
        t[j - 1] = (uint32_t)x;
        c = x >> 32;
      }
      top += c;
      t[N - 1] = (uint32_t)top;
      t[N] = (uint32_t)(top >> 32);
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    return reduce_once(r, t[N]);
  }

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
  // Host flavour: the same Montgomery product on 64-bit limbs (CIOS, 128-bit partial products); ~3x the speed of the
  // 32-bit loop above on the EPYC host.  Used by the O(1) proof tail (two 255-bit scalar multiplications, three
  // inversions) and by the host-side setup scalars; identical values (R = 2^(32N) = 2^(64 N/2)).
  static Fp mul_h64(const Fp& a, const Fp& b) {
    static_assert(N % 2 == 0, "even limb count");
    constexpr int M = N / 2;
    typedef unsigned __int128 u128;
    uint64_t x[M], y[M], p[M], t[M + 2];
    for (int i = 0; i < M; i++) {
      x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
      y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
      p[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step: inv64 = inv32 * (2 + p0 * inv32)
    const uint64_t inv32 = (uint64_t)P::INV;
    const uint64_t inv64 = inv32 * (2 + p[0] * inv32);
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
      u128 c = 0;
      for (int j = 0; j < M; j++) {
        c += (u128)x[j] * y[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[M];
      t[M] = (uint64_t)c;
      t[M + 1] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * inv64;
      c = (u128)m * p[0] + t[0];
      c >>= 64;
      for (int j = 1; j < M; j++) {
        c += (u128)m * p[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[M];
      t[M - 1] = (uint64_t)c;
      t[M] = t[M + 1] + (uint64_t)(c >> 64);
    }
    Fp r;
    for (int i = 0; i < M; i++) {
      r.l[2 * i] = (uint32_t)t[i];
      r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    return reduce_once(r, (uint32_t)t[M]);
  }
#endif

#if defined(__HIP_DEVICE_COMPILE__) && !defined(ARK_NO_ASM_MUL)
  // gfx950 flavour: product scanning (FIPS) over a 96-bit column accumulator {top : acc}.  Every partial
  // product is ONE v_mad_u64_u32 (32x32 + 64 -> 64, carry-out in VCC) plus ONE v_addc_co_u32 that banks the
  // carry; hipcc's own lowering of the C version spends ~5 instructions per partial product on moves and
  // 64-bit adds (measured: 1420 vs ~740 instructions per BLS12-381 Fq multiplication).  The two-instruction
  // sequences are inline asm because the compiler does not use the MAD's carry-out; they contain no memory
  // operation and no hazard pair (VALU VCC producer -> VALU carry-in consumer needs no wait state).
  ARK_D static void macc_vv(uint64_t& acc, uint32_t& top, uint32_t x, uint32_t y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(top) : "v"(x), "v"(y) : "vcc");
  }
  // second factor is a compile-time constant kept in an SGPR (VOP3 on gfx9 takes no 32-bit literal)
  ARK_D static void macc_vs(uint64_t& acc, uint32_t& top, uint32_t x, uint32_t y_const) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(top) : "v"(x), "s"(y_const) : "vcc");
  }
  ARK_D static void macc_pair(uint64_t& acc, uint32_t& top, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1_const) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(top) : "v"(x0), "v"(y0), "v"(x1), "s"(y1_const) : "vcc");
  }
#include "mont_asm_chunks.inc"
  // pairs i = I0 .. I0+CNT-1 of column K: a[i]*b[K-i] + m[i]*p[K-i], at most 6 pairs per asm statement (hipcc pads
  // every asm statement whose result feeds the next instruction with an s_nop: fewer statements, fewer bubbles)
  template <int CNT, int I0, int K, bool FIRST = false>
  ARK_D static void col_pairs(uint64_t& acc, uint32_t& top, const Fp& a, const Fp& b, const uint32_t* m) {
#define ARK_PAIR(j) a.l[I0 + j], b.l[K - I0 - j], m[I0 + j], P::mod(K - I0 - j)
#define ARK_CH(n, ...) do { if constexpr (FIRST) macc_chunk##n##_f(acc, top, __VA_ARGS__); else macc_chunk##n(acc, top, __VA_ARGS__); } while (0)
    if constexpr (CNT >= 6) {
      ARK_CH(6, ARK_PAIR(0), ARK_PAIR(1), ARK_PAIR(2), ARK_PAIR(3), ARK_PAIR(4), ARK_PAIR(5));
      col_pairs<CNT - 6, I0 + 6, K, false>(acc, top, a, b, m);
    } else if constexpr (CNT == 5) {
      ARK_CH(5, ARK_PAIR(0), ARK_PAIR(1), ARK_PAIR(2), ARK_PAIR(3), ARK_PAIR(4));
    } else if constexpr (CNT == 4) {
      ARK_CH(4, ARK_PAIR(0), ARK_PAIR(1), ARK_PAIR(2), ARK_PAIR(3));
    } else if constexpr (CNT == 3) {
      ARK_CH(3, ARK_PAIR(0), ARK_PAIR(1), ARK_PAIR(2));
    } else if constexpr (CNT == 2) {
      ARK_CH(2, ARK_PAIR(0), ARK_PAIR(1));
    } else if constexpr (CNT == 1) {
      ARK_CH(1, ARK_PAIR(0));
    }
#undef ARK_CH
#undef ARK_PAIR
  }
  template <int K>
  ARK_D static void mul_col_lo(uint64_t& acc, uint32_t& top, const Fp& a, const Fp& b, uint32_t* m, Fp& r) {
    if constexpr (K < N) {
      if constexpr (K == 0) top = 0;
      col_pairs<K, 0, K, true>(acc, top, a, b, m);            // starts the column: writes `top`
      macc_vv(acc, top, a.l[K], b.l[0]);
      m[K] = (uint32_t)acc * P::INV;
      macc_vs(acc, top, m[K], P::mod(0));
      acc = (acc >> 32) | ((uint64_t)top << 32);
      mul_col_lo<K + 1>(acc, top, a, b, m, r);
    }
  }
  template <int K>
  ARK_D static void mul_col_hi(uint64_t& acc, uint32_t& top, const Fp& a, const Fp& b, const uint32_t* m, Fp& r) {
    if constexpr (K < 2 * N - 1) {
      col_pairs<2 * N - 1 - K, K - N + 1, K, true>(acc, top, a, b, m);
      r.l[K - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)top << 32);
      mul_col_hi<K + 1>(acc, top, a, b, m, r);
    }
  }
  // (A two-chain variant -- a*b and m*p products on separate accumulators with SGPR-pair carries -- was
  // measured on MI355X and is SLOWER: 46.0 vs 51.9 Gmul/s; the MAD chain is issue-bound, not latency-bound.)
  ARK_D static Fp mul(const Fp& a, const Fp& b) {
    static_assert(P::BITS <= 32 * N - 1, "needs a spare top bit");
    uint32_t m[N];
    Fp r;
    uint64_t acc = 0;
    uint32_t top = 0;
    mul_col_lo<0>(acc, top, a, b, m, r);
    mul_col_hi<N>(acc, top, a, b, m, r);
    r.l[N - 1] = (uint32_t)acc;
    return reduce_once(r, (uint32_t)(acc >> 32));
  }
  // (x1*y1 + x2*y2) * R^-1 mod p in ONE product-scanning pass (one Montgomery reduction for two products):
  // the lane-split Fq2 multiplication of the G2 bucket kernel.  (x1 y1 + x2 y2 + m p)/R < p (1 + 2p/R) < 2p.
  ARK_D static void macc_pair_vv(uint64_t& acc, uint32_t& top, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(top) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
  }
  // dual-product pairs i = I0 .. I0+CNT-1 of column K (x1[i]*y1[K-i] + x2[i]*y2[K-i]) and the reduction products
  // m[i]*p[K-i], chunked like col_pairs
  template <int CNT, int I0, int K, bool FIRST = false>
  ARK_D static void col_dual(uint64_t& acc, uint32_t& top, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2) {
#define ARK_D4(j) x1.l[I0 + j], y1.l[K - I0 - j], x2.l[I0 + j], y2.l[K - I0 - j]
#define ARK_CH(n, ...) do { if constexpr (FIRST) macc_dchunk##n##_f(acc, top, __VA_ARGS__); else macc_dchunk##n(acc, top, __VA_ARGS__); } while (0)
    if constexpr (CNT >= 6) {
      ARK_CH(6, ARK_D4(0), ARK_D4(1), ARK_D4(2), ARK_D4(3), ARK_D4(4), ARK_D4(5));
      col_dual<CNT - 6, I0 + 6, K, false>(acc, top, x1, y1, x2, y2);
    } else if constexpr (CNT == 5) {
      ARK_CH(5, ARK_D4(0), ARK_D4(1), ARK_D4(2), ARK_D4(3), ARK_D4(4));
    } else if constexpr (CNT == 4) {
      ARK_CH(4, ARK_D4(0), ARK_D4(1), ARK_D4(2), ARK_D4(3));
    } else if constexpr (CNT == 3) {
      ARK_CH(3, ARK_D4(0), ARK_D4(1), ARK_D4(2));
    } else if constexpr (CNT == 2) {
      ARK_CH(2, ARK_D4(0), ARK_D4(1));
    } else if constexpr (CNT == 1) {
      ARK_CH(1, ARK_D4(0));
    }
#undef ARK_CH
#undef ARK_D4
  }
  template <int CNT, int I0, int K, bool FIRST = false>
  ARK_D static void col_red(uint64_t& acc, uint32_t& top, const uint32_t* m) {
#define ARK_S2(j) m[I0 + j], P::mod(K - I0 - j)
#define ARK_CH(n, ...) do { if constexpr (FIRST) macc_schunk##n##_f(acc, top, __VA_ARGS__); else macc_schunk##n(acc, top, __VA_ARGS__); } while (0)
    if constexpr (CNT >= 6) {
      ARK_CH(6, ARK_S2(0), ARK_S2(1), ARK_S2(2), ARK_S2(3), ARK_S2(4), ARK_S2(5));
      col_red<CNT - 6, I0 + 6, K, false>(acc, top, m);
    } else if constexpr (CNT == 5) {
      ARK_CH(5, ARK_S2(0), ARK_S2(1), ARK_S2(2), ARK_S2(3), ARK_S2(4));
    } else if constexpr (CNT == 4) {
      ARK_CH(4, ARK_S2(0), ARK_S2(1), ARK_S2(2), ARK_S2(3));
    } else if constexpr (CNT == 3) {
      ARK_CH(3, ARK_S2(0), ARK_S2(1), ARK_S2(2));
    } else if constexpr (CNT == 2) {
      ARK_CH(2, ARK_S2(0), ARK_S2(1));
    } else if constexpr (CNT == 1) {
      ARK_CH(1, ARK_S2(0));
    }
#undef ARK_CH
#undef ARK_S2
  }
  template <int K>
  ARK_D static void m2_col_lo(uint64_t& acc, uint32_t& top, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2,
                              uint32_t* m) {
    if constexpr (K < N) {
      col_dual<K + 1, 0, K, true>(acc, top, x1, y1, x2, y2);
      col_red<K, 0, K>(acc, top, m);
      m[K] = (uint32_t)acc * P::INV;
      macc_vs(acc, top, m[K], P::mod(0));
      acc = (acc >> 32) | ((uint64_t)top << 32);
      m2_col_lo<K + 1>(acc, top, x1, y1, x2, y2, m);
    }
  }
  template <int K>
  ARK_D static void m2_col_hi(uint64_t& acc, uint32_t& top, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2,
                              const uint32_t* m, Fp& r) {
    if constexpr (K < 2 * N - 1) {
      col_dual<2 * N - 1 - K, K - N + 1, K, true>(acc, top, x1, y1, x2, y2);
      col_red<2 * N - 1 - K, K - N + 1, K>(acc, top, m);
      r.l[K - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)top << 32);
      m2_col_hi<K + 1>(acc, top, x1, y1, x2, y2, m, r);
    }
  }
  ARK_D static Fp mul2sum(const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2) {
    static_assert(P::BITS <= 32 * N - 2, "needs two spare top bits");
    uint32_t m[N];
    Fp r;
    uint64_t acc = 0;
    uint32_t top = 0;
    m2_col_lo<0>(acc, top, x1, y1, x2, y2, m);
    m2_col_hi<N>(acc, top, x1, y1, x2, y2, m, r);
    r.l[N - 1] = (uint32_t)acc;
    return reduce_once(r, (uint32_t)(acc >> 32));
  }
  // (x1*y1 + x2*y2 + x3*y3 + x4*y4) * R^-1 mod p, one reduction for four products (the fused
  // Y3 = R*(Q - X3) - Y1*PPP of the lane-split G2 mixed addition).  (4 p^2 + m p)/R < p (1 + 4p/R) < 2p needs
  // 4p < R, i.e. two spare top bits.
  template <int K>
  ARK_D static void m4_col_lo(uint64_t& acc, uint32_t& top, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2,
                              const Fp& x3, const Fp& y3, const Fp& x4, const Fp& y4, uint32_t* m) {
    if constexpr (K < N) {
      col_dual<K + 1, 0, K, true>(acc, top, x1, y1, x2, y2);
      col_dual<K + 1, 0, K>(acc, top, x3, y3, x4, y4);
      col_red<K, 0, K>(acc, top, m);
      m[K] = (uint32_t)acc * P::INV;
      macc_vs(acc, top, m[K], P::mod(0));
      acc = (acc >> 32) | ((uint64_t)top << 32);
      m4_col_lo<K + 1>(acc, top, x1, y1, x2, y2, x3, y3, x4, y4, m);
    }
  }
  template <int K>
  ARK_D static void m4_col_hi(uint64_t& acc, uint32_t& top, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2,
                              const Fp& x3, const Fp& y3, const Fp& x4, const Fp& y4, const uint32_t* m, Fp& r) {
    if constexpr (K < 2 * N - 1) {
      col_dual<2 * N - 1 - K, K - N + 1, K, true>(acc, top, x1, y1, x2, y2);
      col_dual<2 * N - 1 - K, K - N + 1, K>(acc, top, x3, y3, x4, y4);
      col_red<2 * N - 1 - K, K - N + 1, K>(acc, top, m);
      r.l[K - N] = (uint32_t)acc;
      acc = (acc >> 32) | ((uint64_t)top << 32);
      m4_col_hi<K + 1>(acc, top, x1, y1, x2, y2, x3, y3, x4, y4, m, r);
    }
  }
  ARK_D static Fp mul4sum(const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2, const Fp& x3, const Fp& y3,
                          const Fp& x4, const Fp& y4) {
    static_assert(P::BITS <= 32 * N - 2, "needs two spare top bits");
    uint32_t m[N];
    Fp r;
    uint64_t acc = 0;
    uint32_t top = 0;
    m4_col_lo<0>(acc, top, x1, y1, x2, y2, x3, y3, x4, y4, m);
    m4_col_hi<N>(acc, top, x1, y1, x2, y2, x3, y3, x4, y4, m, r);
    r.l[N - 1] = (uint32_t)acc;
    return reduce_once(r, (uint32_t)(acc >> 32));
  }
#else
  ARK_HD static Fp mul(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    return mul_h64(a, b);
#else
    return mul_c(a, b);
#endif
  }
  ARK_HD static Fp mul2sum(const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2) {
    return add(mul_c(x1, y1), mul_c(x2, y2));
  }
  ARK_HD static Fp mul4sum(const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2, const Fp& x3, const Fp& y3,
                           const Fp& x4, const Fp& y4) {
    return add(add(mul_c(x1, y1), mul_c(x2, y2)), add(mul_c(x3, y3), mul_c(x4, y4)));
  }
#endif
  // a*b - c*d with ONE Montgomery reduction (the Y3 of the mixed addition)
  static constexpr bool FUSED_MUL_SUB = true;
  static constexpr bool COLD_INLINE_MUL = true;      // curve.cuh: out-of-line group operations inline their multiplications
  ARK_HD static Fp mul_sub(const Fp& a, const Fp& b, const Fp& c, const Fp& d) { return mul2sum(a, b, neg(c), d); }

  ARK_HD static Fp sqr(const Fp& a) { return mul(a, a); }

  // Out-of-line copies for the cold kernels (reduction tails, normalisation, setup): one ~1.5k
  // instruction body per field instead of one per call site keeps code size and compile time sane.
  ARK_HD_NOINLINE static Fp mul_ni(const Fp& a, const Fp& b) { return mul(a, b); }
  ARK_HD static Fp sqr_ni(const Fp& a) { return mul_ni(a, a); }

  // a^(p-2): Fermat inversion (tails only; a hot path never inverts per element).
  ARK_HD_NOINLINE static Fp inv(const Fp& a) {
    Fp result = one();
    bool started = false;
    for (int i = N - 1; i >= 0; i--) {
      uint32_t w = P::pm2(i);
      for (int b = 31; b >= 0; b--) {
        if (started) result = mul_ni(result, result);
        if ((w >> b) & 1) {
          result = started ? mul_ni(result, a) : a;
          started = true;
        }
      }
    }
    return result;
  }

  // canonical integer <-> Montgomery image
  ARK_HD static Fp to_mont(const Fp& canon) { return mul_ni(canon, r2()); }
  ARK_HD static Fp from_mont(const Fp& m) {
    Fp o = zero();
    o.l[0] = 1;
    return mul_ni(m, o);
  }

  // small-constant helpers
  ARK_HD static Fp mul2(const Fp& a) { return add(a, a); }
  ARK_HD static Fp mul3(const Fp& a) { return add(add(a, a), a); }
};

// Quadratic extension Fp[u]/(u^2 + 1) (BLS12-381 and BN254 both use non-residue -1).
template <class P>
struct Fp2 {
  using Base = Fp<P>;
  Base c0, c1;

  ARK_HD static Fp2 zero() { return Fp2{Base::zero(), Base::zero()}; }
  ARK_HD static Fp2 one() { return Fp2{Base::one(), Base::zero()}; }
  ARK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ARK_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
  ARK_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
  ARK_HD static Fp2 add(const Fp2& a, const Fp2& b) { return Fp2{Base::add(a.c0, b.c0), Base::add(a.c1, b.c1)}; }
  ARK_HD static Fp2 sub(const Fp2& a, const Fp2& b) { return Fp2{Base::sub(a.c0, b.c0), Base::sub(a.c1, b.c1)}; }
  ARK_HD static Fp2 neg(const Fp2& a) { return Fp2{Base::neg(a.c0), Base::neg(a.c1)}; }
  ARK_HD static Fp2 dbl(const Fp2& a) { return add(a, a); }
  ARK_HD static Fp2 mul2(const Fp2& a) { return add(a, a); }
  ARK_HD static Fp2 mul3(const Fp2& a) { return add(add(a, a), a); }
  // Karatsuba: 3 base multiplications
  ARK_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    Base v0 = Base::mul(a.c0, b.c0);
    Base v1 = Base::mul(a.c1, b.c1);
    Base s = Base::mul(Base::add(a.c0, a.c1), Base::add(b.c0, b.c1));
    return Fp2{Base::sub(v0, v1), Base::sub(Base::sub(s, v0), v1)};
  }
  // (a0+a1)(a0-a1), 2 a0 a1
  ARK_HD static Fp2 sqr(const Fp2& a) {
    Base t = Base::mul(Base::add(a.c0, a.c1), Base::sub(a.c0, a.c1));
    Base m = Base::mul(a.c0, a.c1);
    return Fp2{t, Base::add(m, m)};
  }
  static constexpr bool FUSED_MUL_SUB = false;
  static constexpr bool COLD_INLINE_MUL = false;     // curve.cuh: out-of-line group operations keep calling mul_ni
  ARK_HD static Fp2 mul_sub(const Fp2& a, const Fp2& b, const Fp2& c, const Fp2& d) { return sub(mul(a, b), mul(c, d)); }
  ARK_HD_NOINLINE static Fp2 mul_ni(const Fp2& a, const Fp2& b) {
    Base v0 = Base::mul_ni(a.c0, b.c0);
    Base v1 = Base::mul_ni(a.c1, b.c1);
    Base s = Base::mul_ni(Base::add(a.c0, a.c1), Base::add(b.c0, b.c1));
    return Fp2{Base::sub(v0, v1), Base::sub(Base::sub(s, v0), v1)};
  }
  ARK_HD_NOINLINE static Fp2 sqr_ni(const Fp2& a) {
    Base t = Base::mul_ni(Base::add(a.c0, a.c1), Base::sub(a.c0, a.c1));
    Base m = Base::mul_ni(a.c0, a.c1);
    return Fp2{t, Base::add(m, m)};
  }
  ARK_HD_NOINLINE static Fp2 inv(const Fp2& a) {
    Base n = Base::add(Base::sqr_ni(a.c0), Base::sqr_ni(a.c1));
    Base ni = Base::inv(n);
    return Fp2{Base::mul_ni(a.c0, ni), Base::neg(Base::mul_ni(a.c1, ni))};
  }
};

#ifndef ARK_PLAIN_HOST
// Lane-split Fq2: the two lanes of a pair (lane ^ 1) hold c0 and c1 of the SAME element.  Used by the G2 bucket
// accumulation so that a G2 mixed addition needs G1-like registers per lane (the whole-element version sits at
// 256 VGPR + 253 AGPR, one wave per SIMD).  Additions are component-wise; a multiplication exchanges the two
// operands with the partner lane (DPP) and is ONE fused dual-product Montgomery pass per lane:
//   lane 0:  c0 = a0*b0 + (-a1)*b1        lane 1:  c1 = a0*b1 + a1*b0
// Every predicate (is_zero, ==) is pair-wide, so control flow stays uniform inside a pair.
template <class P>
struct Fp2L {
  using Base = Fp<P>;
  using Params = P;
  static constexpr int N = Base::N;
  Base c;     // c0 on even lanes, c1 on odd lanes

  ARK_D static uint32_t parity() { return threadIdx.x & 1u; }
  ARK_D static Base xchg(const Base& v) {
    Base r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = ark_pair_xchg(v.l[i]);
    return r;
  }
  ARK_D static Fp2L zero() { return Fp2L{Base::zero()}; }
  ARK_D static Fp2L one() { return Fp2L{parity() ? Base::zero() : Base::one()}; }
  ARK_D bool is_zero() const {
    uint32_t accw = 0;
#pragma unroll
    for (int i = 0; i < N; i++) accw |= c.l[i];
    const uint32_t other = ark_pair_xchg(accw);
    return (accw | other) == 0;
  }
  ARK_D bool operator==(const Fp2L& o) const {
    uint32_t accw = 0;
#pragma unroll
    for (int i = 0; i < N; i++) accw |= (c.l[i] ^ o.c.l[i]);
    const uint32_t other = ark_pair_xchg(accw);
    return (accw | other) == 0;
  }
  ARK_D bool operator!=(const Fp2L& o) const { return !(*this == o); }
  ARK_D static Fp2L add(const Fp2L& a, const Fp2L& b) { return Fp2L{Base::add(a.c, b.c)}; }
  ARK_D static Fp2L sub(const Fp2L& a, const Fp2L& b) { return Fp2L{Base::sub(a.c, b.c)}; }
  ARK_D static Fp2L neg(const Fp2L& a) { return Fp2L{Base::neg(a.c)}; }
  ARK_D static Fp2L dbl(const Fp2L& a) { return add(a, a); }
  ARK_D static Fp2L mul2(const Fp2L& a) { return add(a, a); }
  ARK_D static Fp2L mul3(const Fp2L& a) { return add(add(a, a), a); }
  ARK_D static Fp2L mul(const Fp2L& a, const Fp2L& b) {
    const Base pa = xchg(a.c), pb = xchg(b.c);
    const Base npa = Base::neg(pa);
    const bool odd = parity() != 0;
    Base x1, x2;
#pragma unroll
    for (int i = 0; i < N; i++) {
      x1.l[i] = odd ? pa.l[i] : a.c.l[i];
      x2.l[i] = odd ? a.c.l[i] : npa.l[i];
    }
    return Fp2L{Base::mul2sum(x1, b.c, x2, pb)};
  }
  // (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u: ONE single-product Montgomery pass per lane
  //   even lane: (a0 + a1) * (a0 - a1)        odd lane: (a0 + a0) * a1
  ARK_D static Fp2L sqr(const Fp2L& a) {
    const Base pa = xchg(a.c);
    const bool odd = parity() != 0;
    Base t;
#pragma unroll
    for (int i = 0; i < N; i++) t.l[i] = odd ? pa.l[i] : a.c.l[i];
    const Base u = Base::add(pa, t);                 // even: a1 + a0      odd: a0 + a0
    const Base d = Base::sub(a.c, pa);               // even: a0 - a1      (odd: unused)
    Base v;
#pragma unroll
    for (int i = 0; i < N; i++) v.l[i] = odd ? a.c.l[i] : d.l[i];
    return Fp2L{Base::mul(u, v)};
  }
  // a*b - c*d: four products, one reduction per lane
#ifndef ARK_G2L_FUSE_Y3
#define ARK_G2L_FUSE_Y3 1
#endif
  static constexpr bool FUSED_MUL_SUB = ARK_G2L_FUSE_Y3 != 0;
  ARK_D static Fp2L mul_sub(const Fp2L& a, const Fp2L& b, const Fp2L& c, const Fp2L& d) {
    const bool odd = parity() != 0;
    const Base pa = xchg(a.c), pb = xchg(b.c), pc = xchg(c.c), pd = xchg(d.c);
    const Base npa = Base::neg(pa), nc = Base::neg(c.c);
    Base x1, x2, x3, x4;
#pragma unroll
    for (int i = 0; i < N; i++) {
      x1.l[i] = odd ? pa.l[i] : a.c.l[i];            // even: a0*b0 - a1*b1        odd: a0*b1 + a1*b0
      x2.l[i] = odd ? a.c.l[i] : npa.l[i];
      x4.l[i] = odd ? nc.l[i] : pc.l[i];             // even: -c0*d0 + c1*d1       odd: -c0*d1 - c1*d0
    }
    const Base npc = Base::neg(pc);
#pragma unroll
    for (int i = 0; i < N; i++) x3.l[i] = odd ? npc.l[i] : nc.l[i];
    return Fp2L{Base::mul4sum(x1, b.c, x2, pb, x3, d.c, x4, pd)};
  }
  ARK_D static Fp2L mul_ni(const Fp2L& a, const Fp2L& b) { return mul(a, b); }
  ARK_D static Fp2L sqr_ni(const Fp2L& a) { return mul(a, a); }
  static constexpr bool COLD_INLINE_MUL = true;     // curve.cuh: a lane's half of an Fq2 operation has Fq-like registers
};

#endif  // ARK_PLAIN_HOST

using BlsFq = Fp<BlsFqParams>;
using BlsFr = Fp<BlsFrParams>;
using BlsFq2 = Fp2<BlsFqParams>;
using BnFq = Fp<BnFqParams>;
using BnFr = Fp<BnFrParams>;
using BnFq2 = Fp2<BnFqParams>;

}  // namespace ark355
