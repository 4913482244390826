// Bucket accumulation over radix-2^28 window tables (field28.cuh): the hot kernel of every resident-key G1 MSM.
//
// Same algorithm, same segment/run/flush protocol and the same outputs as msm_accumulate_kernel (msm_impl.cuh); the
// mixed addition runs on lazily reduced 28-bit limbs so that a Montgomery product needs no carry instructions.
// Window tables are converted to Affine28 rows once per key (table_to28_kernel); bucket partials leave the kernel
// in the canonical 32-bit form every other kernel uses.
#pragma once
#include "field28.cuh"
#include "curve.cuh"

namespace ark355 {

// One table row: x, y in the canonical 28-bit Montgomery form (value < p, radix R' = 2^(28 N)), BIT-PACKED: each
// coordinate is the plain little-endian binary image of its value in NB 32-bit words (48 B for BLS12-381, 32 B for
// BN254), so a G1 row is 96 B / 64 B instead of the 128 B / 80 B of one word per limb (round 5: -25 % / -20 % of the
// G1 table bytes; a BN254 row is exactly one 64-byte sector).  The accumulation kernels cut the words into 28-bit limbs in
// registers (one v_alignbit_b32 + one v_and_b32 per limb: ~50 of ~4 500 instructions per addition).  Infinity is the
// all-zero row.
template <class P>
struct alignas(16) Affine28 {
  using F = Fp28<P>;
  static constexpr int NB = F::NB;                   // words per coordinate
  static constexpr int WORDS = 2 * NB;
  static constexpr int Q = WORDS / 4;                // 16-byte loads per row
  static_assert(WORDS % 4 == 0, "row must be a multiple of 16 bytes");
  uint32_t w[WORDS];
  // limbs of one coordinate from its NB words (compile-time shifts)
  ARK_HD static F unpack(const uint32_t* c) {
    F r;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      const int bit = 28 * i, q = bit / 32, sh = bit % 32;
      uint32_t v = 0;
      if (q < NB) {
        v = c[q] >> sh;
        if (sh + 28 > 32 && q + 1 < NB) v |= c[q + 1] << (32 - sh);
      }
      r.l[i] = v & F::MASK;
    }
    return r;
  }
  // canonical limbs (< 2^28 each, value < p < 2^(32 NB)) -> NB words
  ARK_HD static void pack(const F& a, uint32_t* c) {
#pragma unroll
    for (int q = 0; q < NB; q++) c[q] = 0;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      const int bit = 28 * i, q = bit / 32, sh = bit % 32;
      if (q < NB) c[q] |= a.l[i] << sh;
      if (sh + 28 > 32 && q + 1 < NB) c[q + 1] |= a.l[i] >> (32 - sh);
    }
  }
};

// The UNPACKED row of rounds 2-4: one word per limb, padded to a multiple of 16 B (128 B for BLS12-381, 80 B for BN254).
// Which of the two a table uses is decided per key (PrecompTable::packed, msm_impl.cuh): packed rows when they are whole
// 64-byte sectors (BN254: faster AND smaller) or when the key would not fit HBM otherwise; unpacked rows else -- a 96-byte
// row straddles 128-byte lines, which costs BLS12-381 proofs 1.5 % of throughput with other proofs in flight (runs B, F)
// and the lane-pair kernel its spill-free hot loop.
template <class P>
struct alignas(16) Affine28U {
  using F = Fp28<P>;
  static constexpr int USED = 2 * F::N;
  static constexpr int WORDS = (USED * 4 > 96) ? 32 : ((USED + 3) / 4) * 4;
  static constexpr int Q = USED / 4;                 // 16-byte loads that carry data
  static_assert(USED % 4 == 0, "limb count must be even");
  uint32_t w[WORDS];
};

#define ARK_KEY_NONE 0xFFFFFFFFu

template <class P>
struct Acc28 {
  Fp28<P> x, y, zz, zzz;
};

// ---- bucket slots in the accumulation kernels' own form (round 6) ------------------------------------------------------
// What the accumulation kernels write a finished run into, and what the tail kernels (tails28_impl.cuh) read and write:
// x | y | zz | zzz, N limbs each, normalised (every limb < 2^28, the top one whatever is left); x < 6.1 p, y < 2 p,
// zz, zzz < 1.05 p -- exactly what the accumulator holds, so a flush is 4 N stores and no arithmetic (rounds 2-5 converted
// to the canonical 32-bit form here: ~2 000 instructions per run, and the tails converted nothing back because they ran on
// the 32-bit out-of-line code).  The all-zero slot is the point at infinity (zz of a finite point is a unit of the field).
template <class P>
struct alignas(16) Slot28 {
  static constexpr int N = Fp28<P>::N;
  static constexpr int WORDS = 4 * N;
  static_assert(WORDS % 4 == 0, "slot must be a whole number of 16-byte words");
  uint32_t w[WORDS];
};
// G2: one Slot28 per Fq2 component (even lane of a pair: c0, odd lane: c1)
template <class P>
struct alignas(16) Slot28G2 {
  Slot28<P> half[2];
};
template <class P, int COORDS>
using Msm28Slot = typename std::conditional<COORDS == 4, Slot28<P>, Slot28G2<P>>::type;

template <class P>
ARK_HD void slot28_put(Slot28<P>* dst, const Acc28<P>& a, bool empty) {
  constexpr int N = Fp28<P>::N;
  uint4* d = reinterpret_cast<uint4*>(dst);
  uint32_t v[4 * N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    v[0 * N + i] = empty ? 0u : a.x.l[i];
    v[1 * N + i] = empty ? 0u : a.y.l[i];
    v[2 * N + i] = empty ? 0u : a.zz.l[i];
    v[3 * N + i] = empty ? 0u : a.zzz.l[i];
  }
#pragma unroll
  for (int k = 0; k < N; k++) d[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}
// returns "this (half) slot is the all-zero one"
template <class P>
ARK_HD bool slot28_get(const Slot28<P>* src, Acc28<P>& a) {
  constexpr int N = Fp28<P>::N;
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint32_t v[4 * N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const uint4 t = s[k];
    v[4 * k] = t.x;
    v[4 * k + 1] = t.y;
    v[4 * k + 2] = t.z;
    v[4 * k + 3] = t.w;
  }
  uint32_t any = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    a.x.l[i] = v[0 * N + i];
    a.y.l[i] = v[1 * N + i];
    a.zz.l[i] = v[2 * N + i];
    a.zzz.l[i] = v[3 * N + i];
    any |= v[2 * N + i];
  }
  return any == 0;
}

// Where a finished run goes (same protocol as msm_flush_run, msm_impl.cuh): buckets[key] when it is the bucket's only run,
// head[seg] when it opened the segment, tail[seg] otherwise.  Three explicit branches, NOT a select among the pointers:
// hipcc turns such a select into an indexed load from the caller's closure object, which then cannot be scalarised -- and
// every captured variable, the accumulator included, lives in scratch memory for the whole loop.
template <class P>
ARK_D void msm_flush_slot28(Slot28<P>* buckets_half, Slot28<P>* head_half, Slot28<P>* tail_half, uint32_t* head_key,
                            uint32_t* tail_key, bool write_key, uint32_t key, const Acc28<P>& acc, bool empty, bool first_run,
                            uint32_t run_start, uint32_t run_end, uint32_t seg, const uint32_t* offsets, const uint32_t* counts,
                            size_t slot_stride) {
  const uint32_t o = offsets[key], cnt = counts[key];
  const bool complete = (run_start == o) && (run_end == o + cnt);
  if (complete) {
    slot28_put<P>(reinterpret_cast<Slot28<P>*>(reinterpret_cast<uint8_t*>(buckets_half) + (size_t)key * slot_stride), acc, empty);
  } else if (first_run) {
    slot28_put<P>(reinterpret_cast<Slot28<P>*>(reinterpret_cast<uint8_t*>(head_half) + (size_t)seg * slot_stride), acc, empty);
    if (write_key) head_key[seg] = key;
  } else {
    slot28_put<P>(reinterpret_cast<Slot28<P>*>(reinterpret_cast<uint8_t*>(tail_half) + (size_t)seg * slot_stride), acc, empty);
    if (write_key) tail_key[seg] = key;
  }
}

template <class P, bool PACKED>
using Row28 = typename std::conditional<PACKED, Affine28<P>, Affine28U<P>>::type;

template <class P, bool PACKED>
__global__ void __launch_bounds__(256)
table_to28_kernel(const Affine<Fp<P>>* __restrict__ src, Row28<P, PACKED>* __restrict__ dst, uint64_t rows) {
  using F = Fp28<P>;
  using Row = Row28<P, PACKED>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const Affine<Fp<P>> a = src[i];
  Row o;
#pragma unroll
  for (int k = 0; k < Row::WORDS; k++) o.w[k] = 0;
  if (!a.is_inf()) {
    const F x = F::from_fp(a.x), y = F::from_fp(a.y);
    if constexpr (PACKED) {
      Affine28<P>::pack(x, o.w);
      Affine28<P>::pack(y, o.w + Affine28<P>::NB);
    } else {
#pragma unroll
      for (int k = 0; k < F::N; k++) {
        o.w[k] = x.l[k];
        o.w[F::N + k] = y.l[k];
      }
    }
  }
  dst[i] = o;
}

// ---- cold paths of the mixed addition (out of line, arguments BY VALUE: see xyzz_dbl_affine_ni) ----------------
// 0: Pd is not a multiple of p (false alarm of the cheap filter), 1: P == acc (double), 2: P == -acc (infinity)
template <class P>
ARK_HD_NOINLINE int madd28_classify(Fp28<P> pd, Fp28<P> r) {
  if (!Fp28<P>::is_zero_mod_p(pd)) return 0;
  return Fp28<P>::is_zero_mod_p(r) ? 1 : 2;
}
// One coordinate (0: x, 1: y, 2: zz, 3: zzz) of 2 * (px, py), through the canonical 32-bit formulas.  Cold path;
// called once per coordinate so that arguments and result travel in registers: a by-value / sret Acc28 made hipcc
// keep the hot loop's accumulator in scratch memory (round-1 ISA listing: ~90 scratch accesses per mixed addition).
template <class P>
ARK_HD_NOINLINE typename Fp28<P>::Vec dbl28_coord_ni(Fp28<P> px, Fp28<P> py, int which) {
  using F = Fp28<P>;
  const Affine<Fp<P>> a{F::to_fp(px), F::to_fp(py)};
  const XYZZ<Fp<P>> d = xyzz_dbl_affine_t<true>(a);
  return F::from_fp(which == 0 ? d.x : which == 1 ? d.y : which == 2 ? d.zz : d.zzz).to_vec();
}
template <class P>
ARK_HD_NOINLINE typename Fp28<P>::Vec one28_ni() {
  return Fp28<P>::from_fp(Fp<P>::one()).to_vec();
}

// Where zz / zzz of the pair's accumulator live between their uses.  ZzRegs: in the accumulator's own registers.  ZzLds
// (round 6): in LDS.  The lane-pair addition over BLS12-381 does not fit 256 registers -- the code object of the first
// round-6 library shows the allocator writing three 14-limb values (new zz, new zzz and one more) to scratch memory while
// the fused Y3 pass runs and fetching them back at the end of every addition: 53 scratch loads + 41 stores per iteration
// (tools/code_object_stats.py), and the kernel 11 % slower than round 5's (7.17 against 6.47 ms per 2^20-term launch), whose
// spills happened to sit in the flush path.  zz and zzz are each read twice and written once per addition, so they are the
// cheapest state to keep elsewhere, and so are x and y (the new x is dead during the fused Y3 pass, where the pressure peaks;
// with all four there the listing shows 5 scratch loads + 4 stores per addition): 4 x 4 b128 slots per lane (256 B; 64 KiB for
// 256 lanes, two workgroups per CU), every lane on its own consecutive 16-byte column (hardware counters, run Q: bank-conflict cycles 5.7 % of the
// LDS-active cycles, LDS waits 0.06 % of the wave cycles): 32 ds_read_b128 +
// 16 ds_write_b128 per addition of ~7 600 instructions.
template <class P>
struct ZzRegs {
  Acc28<P>& a;
  ARK_D Fp28<P> zz() const { return a.zz; }
  ARK_D Fp28<P> zzz() const { return a.zzz; }
  ARK_D Fp28<P> x() const { return a.x; }
  ARK_D Fp28<P> y() const { return a.y; }
  ARK_D void set_y(const Fp28<P>& v) const { a.y = v; }
  ARK_D void set_zz(const Fp28<P>& v) const { a.zz = v; }
  ARK_D void set_zzz(const Fp28<P>& v) const { a.zzz = v; }
  ARK_D void set_x(const Fp28<P>& v) const { a.x = v; }
};
#ifndef ARK_G2L28_LDS_VALUES
#define ARK_G2L28_LDS_VALUES 4      // how many of zz, zzz, x, y (in that order) live in LDS: 0 (none: rounds 2-5), 2, 3, 4
#endif
template <class P, int NV = ARK_G2L28_LDS_VALUES>
struct ZzLds {
  using F = Fp28<P>;
  static constexpr int N = F::N, QN = (N + 3) / 4;
  static constexpr int VALUES = NV < 2 ? 2 : NV;
  Acc28<P>& a;        // the coordinates that stay in registers
  uint4* mine;        // quad q of value s (0: zz, 1: zzz, 2: x, 3: y) sits at mine[(s * QN + q) * stride]
  uint32_t stride;    // lanes of the workgroup
  static size_t bytes(uint32_t threads) { return (size_t)VALUES * QN * threads * sizeof(uint4); }
  // (the barrier keeps the compiler from serving a later read out of registers it loaded earlier: the point of the exercise
  // is that the value is NOT live in between)
  ARK_D F get(int s) const {
    asm volatile("" ::: "memory");
    F r;
#pragma unroll
    for (int q = 0; q < QN; q++) {
      const uint4 t = mine[(size_t)(s * QN + q) * stride];
      r.l[4 * q] = t.x;
      if (4 * q + 1 < N) r.l[4 * q + 1] = t.y;
      if (4 * q + 2 < N) r.l[4 * q + 2] = t.z;
      if (4 * q + 3 < N) r.l[4 * q + 3] = t.w;
    }
    return r;
  }
  ARK_D void put(int s, const F& v) const {
#pragma unroll
    for (int q = 0; q < QN; q++)
      mine[(size_t)(s * QN + q) * stride] = make_uint4(v.l[4 * q], 4 * q + 1 < N ? v.l[4 * q + 1] : 0u, 4 * q + 2 < N ? v.l[4 * q + 2] : 0u,
                                                         4 * q + 3 < N ? v.l[4 * q + 3] : 0u);
  }
  ARK_D F zz() const { return get(0); }
  ARK_D F zzz() const { return get(1); }
  ARK_D void set_zz(const F& v) const { put(0, v); }
  ARK_D void set_zzz(const F& v) const { put(1, v); }
  ARK_D F x() const {
    if constexpr (VALUES > 2) return get(2);
    else return a.x;
  }
  ARK_D void set_x(const F& v) const {
    if constexpr (VALUES > 2) put(2, v);
    else a.x = v;
  }
  ARK_D F y() const {
    if constexpr (VALUES > 3) return get(3);
    else return a.y;
  }
  ARK_D void set_y(const F& v) const {
    if constexpr (VALUES > 3) put(3, v);
    else a.y = v;
  }
};

// acc += (px, +-py).  Value/limb classes (field28.cuh): table coordinates are canonical; acc.x is normalised and
// < 6.1 p; acc.y, acc.zz, acc.zzz are products (< 1.05 p, normalised) -- or, right after a bucket was opened with a
// negated point, acc.y = norm(2p - py) < 2p.
//   U2 = px zz            S2 = py' zzz           (py' = py or 2p - py: limbs < 2^29)
//   Pd = U2 + 8p - X1     R  = S2 + 3p - Y1      (limbs < 2^29.6, values < 9.1 p / < 4.1 p)
//   PP = Pd^2, PPP = Pd PP, Q = X1 PP, ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP
//   X3 = norm(R^2 + 5p - (PPP + 2Q))             (< 6.1 p)
//   Y3 = R (Q + 8p - X3) + (3p - Y1) PPP         (one fused pass; operand limbs < 2^29.6 each)
// Largest column: 14 * 2^59.2 + 14 * 2^57.6 + 14 * 2^56 < 2^63.5.
template <class P, class Z>
ARK_D void madd28z(Acc28<P>& acc, const Z& z, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  using F = Fp28<P>;
  F pys;
  {
    const F n = F::template neg<2, 1>(py);
#pragma unroll
    for (int i = 0; i < F::N; i++) pys.l[i] = negate ? n.l[i] : py.l[i];
  }
  if (empty) {
    const F one = F::from_vec(one28_ni<P>());
    z.set_x(px);
    z.set_y(F::norm(pys));
    z.set_zz(one);
    z.set_zzz(one);
    empty = false;
    return;
  }
  const F U2 = F::mul(px, z.zz());
  const F S2 = F::mul(pys, z.zzz());
  const F Pd = F::template sub<8, 1>(U2, z.x());
  const F R = F::template sub<3, 1>(S2, z.y());
  if (Pd.multiple_hint() < 10u) {
    const int cls = madd28_classify<P>(Pd, R);
    if (cls == 1) {
      const F yn = F::norm(pys);
      z.set_x(F::from_vec(dbl28_coord_ni<P>(px, yn, 0)));
      z.set_y(F::from_vec(dbl28_coord_ni<P>(px, yn, 1)));
      const F nzz = F::from_vec(dbl28_coord_ni<P>(px, yn, 2));
      z.set_zz(nzz);
      z.set_zzz(F::from_vec(dbl28_coord_ni<P>(px, yn, 3)));
      if (nzz.limbs_all_zero()) empty = true;       // 2P = infinity (no such point on these curves)
      return;
    }
    if (cls == 2) {
      empty = true;
      return;
    }
  }
  const F PP = F::sqr(Pd);
  const F PPP = F::mul(Pd, PP);
  const F Q = F::mul(z.x(), PP);
  z.set_zz(F::mul(z.zz(), PP));
  z.set_zzz(F::mul(z.zzz(), PPP));
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(F::sqr(R), F::template neg<5, 4>(W)));
  const F T = F::template sub<8, 1>(Q, X3);
  const F NY = F::template neg<3, 1>(z.y());
  z.set_x(X3);
  z.set_y(F::mul2sum(R, T, NY, PPP));
}
template <class P>
ARK_D void madd28(Acc28<P>& acc, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  const ZzRegs<P> z{acc};
  madd28z<P, ZzRegs<P>>(acc, z, empty, px, py, negate);
}

// (Round 6, run O: this walk at THREE waves per SIMD -- 168 registers, x / zz / zzz of the accumulator in LDS through ZzLds, no
// row prefetch, `msm_accumulate28w3_kernel` of commit 2a07047 -- measured 4.7 % slower per launch and 3.7 % per proof in
// flight, profiles/r06_runO_g1_three_waves.txt: a third wave does not take the slots the relay of two leaves.)
// Same contract as msm_accumulate_kernel<Fp<P>, false>; `bases` holds UNPACKED rows (Affine28U): the plain segment walk of
// rounds 2-4 -- the row of the next entry prefetched into registers, a finished run flushed where it ends.
//
// What bounds this kernel (round 5; tools/ubench5, tools/acc_trace.py on a tracing build, rocprofv3 --pmc; DESIGN.md
// section 11): VALU issue.  A lane executes ~4 560 VALU instructions per addition (3 155 multiply-adds); the SIMD
// arbitrates its two waves oldest-first, the older one runs at the pace of a wave alone (one instruction per ~5.3 cycles:
// 10.9 us per addition), the younger one advances ~20 % as fast in the slots the older leaves, and takes over when the
// older one retires -- a launch is a relay of waves, 1.2 additions per 24 k cycles and SIMD against the 17.6 k cycles of
// VALU time they need.  The instruction cache holds the 36 KB loop (2 725 misses in 281 M fetches), waits for memory are
// 8 % of a wave's cycles, and neither the bucket flushes (run C: -2 % without any), nor bucket boundaries (-4 %), nor the
// table gathers (-3 % with every row in cache) explain the rest: what moves the kernel is the instruction count.
template <class P>
__global__ void __launch_bounds__(MSM_THREADS)
msm_accumulate28_kernel(const Affine28U<P>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                        const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                        Msm28Slot<P, 4>* __restrict__ buckets, Msm28Slot<P, 4>* __restrict__ head,
                        uint32_t* __restrict__ head_key, Msm28Slot<P, 4>* __restrict__ tail,
                        uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Row = Affine28U<P>;
  constexpr int Q = Row::Q;
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint64_t start64 = (uint64_t)seg * seg_len;
  if (start64 >= total) return;
  const uint32_t start = (uint32_t)start64;
  const uint32_t end = (start + seg_len < total) ? start + seg_len : total;
  uint32_t cur_key = sorted_keys[start];
  uint32_t run_start = start;
  bool first_run = true;
  bool empty = true;
  Acc28<P> acc;
  acc.x = F::zero();
  acc.y = F::zero();
  acc.zz = F::zero();
  acc.zzz = F::zero();
  // always_inline: a closure that is inlined late keeps every captured variable (the accumulator!) in scratch memory
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    msm_flush_slot28<P>(buckets, head, tail, head_key, tail_key, true, key, acc, empty, first_run, run_start, run_end, seg, offsets,
                        counts, sizeof(Slot28<P>));
  };
  // software prefetch of the next row into explicit 16-byte registers (see msm_accumulate_kernel)
  uint4 nx[Q];
  uint32_t v_next = sorted_vals[start];
  {
    const uint4* src = reinterpret_cast<const uint4*>(bases + (v_next & ARK_TBL_MASK));
#pragma unroll
    for (int k = 0; k < Q; k++) nx[k] = src[k];
  }
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = sorted_keys[e];
    const uint32_t v = v_next;
    uint32_t d[4 * Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
      d[4 * k + 0] = nx[k].x;
      d[4 * k + 1] = nx[k].y;
      d[4 * k + 2] = nx[k].z;
      d[4 * k + 3] = nx[k].w;
    }
    const uint32_t en = (e + 1 < end) ? e + 1 : e;       // clamp: the last iteration re-reads its own entry
    v_next = sorted_vals[en];
    {
      const uint4* src = reinterpret_cast<const uint4*>(bases + (v_next & ARK_TBL_MASK));
#pragma unroll
      for (int k = 0; k < Q; k++) nx[k] = src[k];
    }
    if (key != cur_key) {
      flush(cur_key, e);
      cur_key = key;
      run_start = e;
      first_run = false;
      empty = true;
    }
    F px, py;
#pragma unroll
    for (int k = 0; k < F::N; k++) {
      px.l[k] = d[k];
      py.l[k] = d[F::N + k];
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < F::N; k++) any |= px.l[k] | py.l[k];
    if (any == 0) continue;                              // base at infinity
    madd28<P>(acc, empty, px, py, (v >> 31) != 0);
  }
  flush(cur_key, end);
}


// ---- the segment walk of the kernels over PACKED rows (msm_accumulate28p_kernel, msm_accumulate_g2l28p_kernel) ------------
// Run / flush protocol as in msm_accumulate_kernel (msm_impl.cuh): a lane walks ONE segment of the sorted entry list; a
// maximal stretch of equal keys is a run; a finished run goes to buckets[key] when it is the bucket's only run, to
// head[seg] when it opened the segment, to tail[seg] otherwise.
//
// Two differences to the plain walk above, built in round 5 (runs B, E, F):
//   * deferred flush.  Converting a finished run to the canonical form (4 x to_fp_lt8 + stores, ~2 000 instructions) used
//     to happen where the run ended: in ~23 % of a wave's iterations SOME lane met a bucket boundary and all 64 waited
//     for it.  The first run a lane finishes inside its segment is now PARKED in LDS (4 N words per lane, word-major:
//     no bank conflicts; 56 KiB per workgroup, two workgroups per CU) and flushed at the end of the segment, right
//     behind the last run, by the same code (a two-pass loop) -- wave-uniformly.  A second interior boundary in one
//     segment (a bucket shorter than the segment) is flushed on the spot, as before.
//   * nothing the next iteration needs is waited for: the row of entry e+1 is gathered at the top of iteration e from an
//     index that arrived during iteration e-1, the index of entry e+2 and the key of entry e+1 are loaded beside it.
template <class P>
struct Park28 {
  using F = Fp28<P>;
  static constexpr int WORDS = 4 * F::N + 4;          // the run's four coordinates + key, start, end, flags
  // word-major: word k of lane t at lds[k * MSM_THREADS + t].  The run's bookkeeping lives in LDS as well: it is touched
  // twice per segment, and as registers it pushed the lane-pair kernel's hot loop into scratch memory.
  ARK_D static void store(uint32_t* lds, const Acc28<P>& a, bool empty, bool first, uint32_t key, uint32_t start, uint32_t end) {
    uint32_t* d = lds + threadIdx.x;
    if (!empty) {
#pragma unroll
      for (int i = 0; i < F::N; i++) {
        d[(0 * F::N + i) * MSM_THREADS] = a.x.l[i];
        d[(1 * F::N + i) * MSM_THREADS] = a.y.l[i];
        d[(2 * F::N + i) * MSM_THREADS] = a.zz.l[i];
        d[(3 * F::N + i) * MSM_THREADS] = a.zzz.l[i];
      }
    }
    d[(4 * F::N + 0) * MSM_THREADS] = key;
    d[(4 * F::N + 1) * MSM_THREADS] = start;
    d[(4 * F::N + 2) * MSM_THREADS] = end;
    d[(4 * F::N + 3) * MSM_THREADS] = (empty ? 1u : 0u) | (first ? 2u : 0u);
  }
  ARK_D static void load(const uint32_t* lds, Acc28<P>& a, bool& empty, bool& first, uint32_t& key, uint32_t& start, uint32_t& end) {
    const uint32_t* d = lds + threadIdx.x;
    key = d[(4 * F::N + 0) * MSM_THREADS];
    start = d[(4 * F::N + 1) * MSM_THREADS];
    end = d[(4 * F::N + 2) * MSM_THREADS];
    const uint32_t fl = d[(4 * F::N + 3) * MSM_THREADS];
    empty = (fl & 1u) != 0;
    first = (fl & 2u) != 0;
    if (!empty) {
#pragma unroll
      for (int i = 0; i < F::N; i++) {
        a.x.l[i] = d[(0 * F::N + i) * MSM_THREADS];
        a.y.l[i] = d[(1 * F::N + i) * MSM_THREADS];
        a.zz.l[i] = d[(2 * F::N + i) * MSM_THREADS];
        a.zzz.l[i] = d[(3 * F::N + i) * MSM_THREADS];
      }
    }
  }
};

// Same contract as msm_accumulate28_kernel; `bases` holds PACKED rows (Affine28).
template <class P>
__global__ void __launch_bounds__(MSM_THREADS)
msm_accumulate28p_kernel(const Affine28<P>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                        const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                        Msm28Slot<P, 4>* __restrict__ buckets, Msm28Slot<P, 4>* __restrict__ head,
                        uint32_t* __restrict__ head_key, Msm28Slot<P, 4>* __restrict__ tail,
                        uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Row = Affine28<P>;
  constexpr int Q = Row::Q;
  __shared__ uint32_t park_lds[Park28<P>::WORDS * MSM_THREADS];
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint64_t start64 = (uint64_t)seg * seg_len;
  if (start64 >= total) return;
  const uint32_t start = (uint32_t)start64;
  const uint32_t end = (start + seg_len < total) ? start + seg_len : total;
  const uint32_t last = end - 1;
  uint32_t cur_key = sorted_keys[start];
  uint32_t run_start = start, run_end = end;
  bool first_run = true;
  bool empty = true;
  Acc28<P> acc;
  acc.x = F::zero();
  acc.y = F::zero();
  acc.zz = F::zero();
  acc.zzz = F::zero();
  bool parked = false;
  // always_inline: a closure that is inlined late keeps every captured variable (the accumulator!) in scratch memory
  auto flush = [&]() __attribute__((always_inline)) {
    msm_flush_slot28<P>(buckets, head, tail, head_key, tail_key, true, cur_key, acc, empty, first_run, run_start, run_end, seg,
                        offsets, counts, sizeof(Slot28<P>));
  };
  // entry e: key k0, row index v0, row nx (explicit 16-byte registers); entry e + 1: row index v1
  uint32_t k0 = cur_key;
  uint32_t v0 = sorted_vals[start];
  uint32_t v1 = sorted_vals[start < last ? start + 1 : last];
  uint4 nx[Q];
  {
    const uint4* src = reinterpret_cast<const uint4*>(bases + (v0 & ARK_TBL_MASK));
#pragma unroll
    for (int k = 0; k < Q; k++) nx[k] = src[k];
  }
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = k0;
    const uint32_t v = v0;
    uint32_t d[4 * Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
      d[4 * k + 0] = nx[k].x;
      d[4 * k + 1] = nx[k].y;
      d[4 * k + 2] = nx[k].z;
      d[4 * k + 3] = nx[k].w;
    }
    // what the next iteration needs (clamped: the last iterations re-read the last entry)
    k0 = sorted_keys[e < last ? e + 1 : last];
    v0 = v1;
    {
      const uint4* src = reinterpret_cast<const uint4*>(bases + (v0 & ARK_TBL_MASK));
#pragma unroll
      for (int k = 0; k < Q; k++) nx[k] = src[k];
    }
    v1 = sorted_vals[e + 2 < end ? e + 2 : last];
    if (key != cur_key) {
      run_end = e;
      if (!parked) {
        Park28<P>::store(park_lds, acc, empty, first_run, cur_key, run_start, e);
        parked = true;
      } else {
        flush();
      }
      cur_key = key;
      run_start = e;
      run_end = end;
      first_run = false;
      empty = true;
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < Row::WORDS; k++) any |= d[k];
    if (any == 0) continue;                              // base at infinity
    const F px = Row::unpack(d), py = Row::unpack(d + Row::NB);
    madd28<P>(acc, empty, px, py, (v >> 31) != 0);
  }
  // the segment's last run, then the parked one: one copy of the flush code, two passes
#pragma nounroll
  for (int pass = 0; pass < 2; pass++) {
    flush();
    if (!parked) break;
    Park28<P>::load(park_lds, acc, empty, first_run, cur_key, run_start, run_end);
    parked = false;
  }
}


// ================================================================================================================
// G2: lane-split accumulation on 28-bit limbs.  Two lanes per segment (even: c0 components, odd: c1 components of
// every Fq2 value, as in msm_accumulate_g2l_kernel); a table row is two Affine28 halves, {x.c0, y.c0} then
// {x.c1, y.c1}, so that each lane gathers one aligned half.
// Measured on MI355X: 9.2 ms per 2^20-term MSM against 10.8 ms for the 32-bit lane-split kernel (same box, both
// interleaved with the witness map) -- once the accumulator really lived in registers.  The first version of this
// kernel only tied: its flush picked the destination with a select among captured pointers, hipcc turned that into an
// indexed load from the lambda's closure object, the closure could not be scalarised, and EVERY captured variable (the
// accumulator included) stayed in scratch memory: 28 x 16-byte scratch accesses per mixed addition.
template <class P, bool PACKED>
struct Affine28G2 {
  Row28<P, PACKED> half[2];
};

template <class P, bool PACKED>
__global__ void __launch_bounds__(256)
table_to28_g2_kernel(const Affine<Fp2<P>>* __restrict__ src, Affine28G2<P, PACKED>* __restrict__ dst, uint64_t rows) {
  using F = Fp28<P>;
  using Row = Row28<P, PACKED>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const Affine<Fp2<P>> a = src[i];
  Affine28G2<P, PACKED> o;
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int k = 0; k < Row::WORDS; k++) o.half[h].w[k] = 0;
  }
  if (!a.is_inf()) {
    const F x0 = F::from_fp(a.x.c0), x1 = F::from_fp(a.x.c1), y0 = F::from_fp(a.y.c0), y1 = F::from_fp(a.y.c1);
    if constexpr (PACKED) {
      constexpr int NB = Affine28<P>::NB;
      Affine28<P>::pack(x0, o.half[0].w);
      Affine28<P>::pack(y0, o.half[0].w + NB);
      Affine28<P>::pack(x1, o.half[1].w);
      Affine28<P>::pack(y1, o.half[1].w + NB);
    } else {
#pragma unroll
      for (int k = 0; k < F::N; k++) {
        o.half[0].w[k] = x0.l[k];
        o.half[0].w[F::N + k] = y0.l[k];
        o.half[1].w[k] = x1.l[k];
        o.half[1].w[F::N + k] = y1.l[k];
      }
    }
  }
  dst[i] = o;
}

// Fq2 arithmetic of a lane pair on Fp28 components.  KA / BETA describe the PARTNER component of the first
// operand (value < (KA-1) p, limbs <= BETA (2^28 - 1)): it is negated lazily on the even lane.
template <class P>
struct Pair28 {
  using F = Fp28<P>;
  static constexpr int N = F::N;
  ARK_D static bool odd() { return (threadIdx.x & 1u) != 0; }
  ARK_D static F xchg(const F& v) {
    F r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = ark_pair_xchg(v.l[i]);
    return r;
  }
  // (both lanes always exchange: a lane that skipped the exchange because its own flag is false would leave its partner
  // reading a lane that is masked off -- correct on the hardware only through DPP's bound_ctrl, and a desynchronised
  // pair in the emulator)
  ARK_D static bool both(bool b) {
    const uint32_t other = ark_pair_xchg(b ? 1u : 0u);
    return b && other != 0;
  }
  ARK_D static F sel(bool c, const F& a, const F& b) {
    F r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
  }
  //   even: a0 b0 + (-a1) b1          odd: a0 b1 + a1 b0
  template <uint32_t KA, uint32_t BETA>
  ARK_D static F mul(const F& a, const F& b) {
    const F pa = xchg(a), pb = xchg(b);
    const F npa = F::template neg<KA, BETA>(pa);
    const bool o = odd();
    return F::mul2sum(sel(o, pa, a), b, sel(o, a, npa), pb);
  }
  //   even: (a0 + a1)(a0 - a1)        odd: (2 a0) a1            (a normalised, components < (KA-1) p)
  template <uint32_t KA>
  ARK_D static F sqr(const F& a) {
    const F pa = xchg(a);
    const bool o = odd();
    const F u = F::add(pa, sel(o, pa, a));
    const F d = F::template sub<KA, 1>(a, pa);
    return F::mul(u, sel(o, a, d));
  }
  // a b + c d, four products and one reduction per lane
  template <uint32_t KA, uint32_t BA, uint32_t KC, uint32_t BC>
  ARK_D static F mul2(const F& a, const F& b, const F& c, const F& d) {
    const F pa = xchg(a), pb = xchg(b), pc = xchg(c), pd = xchg(d);
    const F npa = F::template neg<KA, BA>(pa), npc = F::template neg<KC, BC>(pc);
    const bool o = odd();
    return F::mul4sum(sel(o, pa, a), b, sel(o, a, npa), pb, sel(o, pc, c), d, sel(o, c, npc), pd);
  }
};

// One coordinate (0: x, 1: y, 2: zz, 3: zzz) of 2 * (x, y) for the pair's point (mdbl-2008-s-1), computed with the
// same lane-split 28-bit arithmetic as the mixed addition.  Cold path (P == acc), called once per coordinate so that
// arguments and result travel in registers.  Its register footprint matters: the kernel's VGPR budget is the maximum
// over its callees, and a version that went through the canonical Fq2 formulas (256 VGPRs, 56 argument registers)
// both capped the kernel at two waves per SIMD and made the allocator keep the hot loop's accumulator in scratch.
//   U = 2y, V = U^2, W = U V, S = x V, M = 3 x^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W y, ZZ = V, ZZZ = W
// x canonical, y normalised and < 2p.
template <class P>
ARK_HD_NOINLINE typename Fp28<P>::Vec dbl28_g2_coord_ni(Fp28<P> x, Fp28<P> y, int which) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F U = F::norm(F::add(y, y));                                   // < 4p
  const F V = L::template sqr<6>(U);
  if (which == 2) return V.to_vec();
  const F W = L::template mul<6, 1>(U, V);
  if (which == 3) return W.to_vec();
  const F S = L::template mul<2, 1>(x, V);
  const F X2 = L::template sqr<3>(x);
  const F M = F::norm(F::add(X2, F::add(X2, X2)));                     // < 3.6p
  const F X3 = F::norm(F::add(L::template sqr<5>(M), F::template neg<4, 2>(F::add(S, S))));   // < 5.2p
  if (which == 0) return X3.to_vec();
  const F T = F::norm(F::template sub<8, 1>(S, X3));
  const F NW = F::template neg<3, 1>(W);
  return L::template mul2<5, 1, 4, 3>(M, T, NW, y).to_vec();
}

// acc += (px, +-py) over Fq2, one component per lane.  Same formula and value classes as madd28; Pd, R and
// T = Q - X3 are normalised before they enter a multi-product pass, which keeps every column below
// 14 (2^56 + 2^57 + 2^57 + 2^58) + 14 2^56 < 2^63 (the emulator build traps on any column overflow).
// The addition comes in two halves so that the accumulation loop can issue the gather of the NEXT table row between
// them: after the first half px / py are dead, and the row's latency hides under the eight products of the second.
// madd28_g2_head returns false when the entry is already dealt with (bucket opened, doubling, cancellation).
template <class P, class Z>
ARK_D bool madd28_g2_head(Acc28<P>& acc, const Z& z, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate,
                          Fp28<P>& Pd, Fp28<P>& R) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F pys = L::sel(negate, F::template neg<2, 1>(py), py);
  if (empty) {
    const F one = L::odd() ? F::zero() : F::from_vec(one28_ni<P>());
    z.set_x(px);
    z.set_y(F::norm(pys));
    z.set_zz(one);
    z.set_zzz(one);
    empty = false;
    return false;
  }
  const F U2 = L::template mul<2, 1>(px, z.zz());
  const F S2 = L::template mul<3, 3>(pys, z.zzz());          // pys limbs <= 2^29 - 1
  Pd = F::norm(F::template sub<8, 1>(U2, z.x()));
  R = F::norm(F::template sub<3, 1>(S2, z.y()));
  if (L::both(Pd.multiple_hint() < 10u)) {
    if (L::both(F::is_zero_mod_p_inl(Pd))) {
      if (L::both(F::is_zero_mod_p_inl(R))) {
        const F yn = F::norm(pys);
        z.set_x(F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 0)));
        z.set_y(F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 1)));
        const F nzz = F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 2));
        z.set_zz(nzz);
        z.set_zzz(F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 3)));
        if (L::both(nzz.limbs_all_zero())) empty = true;
      } else {
        empty = true;
      }
      return false;
    }
  }
  return true;
}
template <class P, class Z>
ARK_D void madd28_g2_tail(Acc28<P>& acc, const Z& z, const Fp28<P>& Pd, const Fp28<P>& R) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F PP = L::template sqr<11>(Pd);
  const F PPP = L::template mul<11, 1>(Pd, PP);
  const F Q = L::template mul<8, 1>(z.x(), PP);
  z.set_zz(L::template mul<3, 1>(z.zz(), PP));
  z.set_zzz(L::template mul<3, 1>(z.zzz(), PPP));
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(L::template sqr<6>(R), F::template neg<5, 4>(W)));
  const F T = F::norm(F::template sub<8, 1>(Q, X3));
  const F NY = F::template neg<3, 1>(z.y());
  z.set_x(X3);
  z.set_y(L::template mul2<6, 1, 4, 3>(R, T, NY, PPP));
}
template <class P, class Z>
ARK_D void madd28_g2z(Acc28<P>& acc, const Z& z, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  Fp28<P> Pd, R;
  if (madd28_g2_head<P, Z>(acc, z, empty, px, py, negate, Pd, R)) madd28_g2_tail<P, Z>(acc, z, Pd, R);
}
// (zz / zzz in the accumulator's registers: the packed-row kernel, BN254, tests)
template <class P>
ARK_D void madd28_g2(Acc28<P>& acc, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  const ZzRegs<P> z{acc};
  madd28_g2z<P, ZzRegs<P>>(acc, z, empty, px, py, negate);
}

#ifndef ARK_G2L28_WAVES
#define ARK_G2L28_WAVES 2     // waves per SIMD the register budget is sized for.  3 (168 VGPRs, ~90 spills per addition):
                              // 10.5 vs 9.2 ms per 2^20-term MSM (round 2); 1 (512 registers): 23.2-23.5 vs 22.1-22.5 ms per
                              // proof (round 5, profiles/r05_runA_karatsuba_ab.txt) -- the hot loop has no scratch access at 2
#endif
// Key, row index and this lane's half row are loaded at the top of every iteration: held in registers across the lane-pair
// addition the half row (28 VGPRs) pushes the hot loop into scratch memory, and loading ahead was measured neutral in
// round 2 (an addition is 2.2x as long as in G1; waits for memory are a few percent of a wave's cycles).
// ZLDS: zz / zzz of the accumulator in LDS (ZzLds above; dynamic LDS of ZzLds<P>::bytes(blockDim.x)); the host asks for it
// where the registers do not suffice (g2l28_zz_in_lds: 14-limb fields).
template <class P>
constexpr bool g2l28_zz_in_lds() { return ARK_G2L28_LDS_VALUES > 0 && Fp28<P>::N > 12; }
template <class P, bool ZLDS>
__global__ void __launch_bounds__(MSM_THREADS, ARK_G2L28_WAVES)
msm_accumulate_g2l28_kernel(const Affine28G2<P, false>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                            const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                            Msm28Slot<P, 8>* __restrict__ buckets, Msm28Slot<P, 8>* __restrict__ head,
                            uint32_t* __restrict__ head_key, Msm28Slot<P, 8>* __restrict__ tail,
                            uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Zt = typename std::conditional<ZLDS, ZzLds<P>, ZzRegs<P>>::type;
  constexpr int Q = Affine28U<P>::Q;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t seg = gid >> 1, par = gid & 1u;       // blockDim is even: par == lane parity
  const uint32_t total = *total_ptr;
  const uint64_t start64 = (uint64_t)seg * seg_len;
  if (start64 >= total) return;                        // both lanes of a pair leave together
  const uint32_t start = (uint32_t)start64;
  const uint32_t end = (start + seg_len < total) ? start + seg_len : total;
  uint32_t cur_key = sorted_keys[start];
  uint32_t run_start = start;
  bool first_run = true;
  bool empty = true;
  Acc28<P> acc;
  acc.x = F::zero();
  acc.y = F::zero();
  acc.zz = F::zero();
  acc.zzz = F::zero();
  ARK_DYN_SMEM(uint4, zlds);
  Zt z = [&]() {
    if constexpr (ZLDS) return ZzLds<P>{acc, zlds + threadIdx.x, blockDim.x};
    else return ZzRegs<P>{acc};
  }();
  // always_inline: a closure that is inlined late keeps every captured variable (the accumulator!) in scratch memory
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    // this lane's component of the four Fq2 coordinates: half `par` of the pair's slot
    if constexpr (ZLDS) {
      if (!empty) {
        acc.x = z.x();
        acc.y = z.y();
        acc.zz = z.zz();
        acc.zzz = z.zzz();
      }
    }
    msm_flush_slot28<P>(&buckets->half[par], &head->half[par], &tail->half[par], head_key, tail_key, par == 0, key, acc, empty,
                        first_run, run_start, run_end, seg, offsets, counts, sizeof(Slot28G2<P>));
  };
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = sorted_keys[e];
    const uint32_t v = sorted_vals[e];
    F px, py;
    {
      const uint4* src = reinterpret_cast<const uint4*>(&bases[v & ARK_TBL_MASK].half[par]);
      uint32_t d[4 * Q];
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const uint4 t = src[k];
        d[4 * k + 0] = t.x;
        d[4 * k + 1] = t.y;
        d[4 * k + 2] = t.z;
        d[4 * k + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < F::N; k++) {
        px.l[k] = d[k];
        py.l[k] = d[F::N + k];
      }
    }
    if (key != cur_key) {
      flush(cur_key, e);
      cur_key = key;
      run_start = e;
      first_run = false;
      empty = true;
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < F::N; k++) any |= px.l[k] | py.l[k];
    if ((any | ark_pair_xchg(any)) == 0) continue;       // base at infinity (pair-wide)
    madd28_g2z<P, Zt>(acc, z, empty, px, py, (v >> 31) != 0);
  }
  flush(cur_key, end);
}

// The same over PACKED halves, with the parked flush of msm_accumulate28p_kernel (its hot loop spills one Fp28 value: 26 scratch
// stores + 28 loads per addition in the listing; BN254, whose halves are whole sectors, still gains 5 % per launch).
template <class P>
__global__ void __launch_bounds__(MSM_THREADS, ARK_G2L28_WAVES)
msm_accumulate_g2l28p_kernel(const Affine28G2<P, true>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                            const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                            Msm28Slot<P, 8>* __restrict__ buckets, Msm28Slot<P, 8>* __restrict__ head,
                            uint32_t* __restrict__ head_key, Msm28Slot<P, 8>* __restrict__ tail,
                            uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Row = Affine28<P>;
  constexpr int Q = Row::Q;
  __shared__ uint32_t park_lds[Park28<P>::WORDS * MSM_THREADS];      // this lane's halves of the parked run
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t seg = gid >> 1, par = gid & 1u;       // blockDim is even: par == lane parity
  const uint32_t total = *total_ptr;
  const uint64_t start64 = (uint64_t)seg * seg_len;
  if (start64 >= total) return;                        // both lanes of a pair leave together
  const uint32_t start = (uint32_t)start64;
  const uint32_t end = (start + seg_len < total) ? start + seg_len : total;
  const uint32_t last = end - 1;
  uint32_t cur_key = sorted_keys[start];
  uint32_t run_start = start, run_end = end;
  bool first_run = true;
  bool empty = true;
  Acc28<P> acc;
  acc.x = F::zero();
  acc.y = F::zero();
  acc.zz = F::zero();
  acc.zzz = F::zero();
  bool parked = false;
  // always_inline: a closure that is inlined late keeps every captured variable (the accumulator!) in scratch memory
  auto flush = [&]() __attribute__((always_inline)) {
    // this lane's component of the four Fq2 coordinates: half `par` of the pair's slot
    msm_flush_slot28<P>(&buckets->half[par], &head->half[par], &tail->half[par], head_key, tail_key, par == 0, cur_key, acc, empty,
                        first_run, run_start, run_end, seg, offsets, counts, sizeof(Slot28G2<P>));
  };
  // Key and row index of entry e + 1 are loaded during entry e.  The half row itself is gathered at the top of its own
  // iteration: held in registers across the lane-pair addition (24 VGPRs) it pushes the hot loop into scratch memory
  // (39 scratch accesses per addition in the listing), and the G2 kernel loses less to that wait than it would to the
  // spills -- an addition is 2.2x as long as in G1 and a wave holds 32 segments, not 64.
  uint32_t k0 = cur_key;
  uint32_t v0 = sorted_vals[start];
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = k0;
    const uint32_t v = v0;
    uint32_t d[4 * Q];
    {
      const uint4* src = reinterpret_cast<const uint4*>(&bases[v & ARK_TBL_MASK].half[par]);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const uint4 t = src[k];
        d[4 * k + 0] = t.x;
        d[4 * k + 1] = t.y;
        d[4 * k + 2] = t.z;
        d[4 * k + 3] = t.w;
      }
    }
    k0 = sorted_keys[e < last ? e + 1 : last];
    v0 = sorted_vals[e < last ? e + 1 : last];
    if (key != cur_key) {
      run_end = e;
      if (!parked) {
        Park28<P>::store(park_lds, acc, empty, first_run, cur_key, run_start, e);
        parked = true;
      } else {
        flush();
      }
      cur_key = key;
      run_start = e;
      run_end = end;
      first_run = false;
      empty = true;
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < Row::WORDS; k++) any |= d[k];
    if ((any | ark_pair_xchg(any)) == 0) continue;       // base at infinity (pair-wide)
    const F px = Row::unpack(d), py = Row::unpack(d + Row::NB);
    madd28_g2<P>(acc, empty, px, py, (v >> 31) != 0);
  }
#pragma nounroll
  for (int pass = 0; pass < 2; pass++) {
    flush();
    if (!parked) break;
    Park28<P>::load(park_lds, acc, empty, first_run, cur_key, run_start, run_end);
    parked = false;
  }
}

}  // namespace ark355
