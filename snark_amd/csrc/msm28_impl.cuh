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

// One table row: x, y in the canonical 28-bit Montgomery form, padded to a multiple of 16 B (128 B for
// BLS12-381: two 64 B sectors per gather).  Infinity is the all-zero row.
template <class P>
struct alignas(16) Affine28 {
  using F = Fp28<P>;
  static constexpr int USED = 2 * F::N;
  static constexpr int WORDS = (USED * 4 > 96) ? 32 : ((USED + 3) / 4) * 4;
  static constexpr int Q = USED / 4;                 // 16-byte loads that carry data
  static_assert(USED % 4 == 0, "limb count must be even");
  uint32_t w[WORDS];
};

template <class P>
struct Acc28 {
  Fp28<P> x, y, zz, zzz;
};

// ARK_LAZY_FLUSH: the accumulation kernels store a finished run as it stands -- the lazily reduced 28-bit limbs of the
// accumulator, 4 x N words (G1) or this lane's 4 x N of 8 x N (G2) -- into a raw slot array [buckets | head | tail],
// and msm_unlazy28_kernel converts every slot to the canonical 32-bit XYZZ form once, in front of the merge.  The
// conversion (carry propagation, three conditional subtractions, repacking, one Montgomery step: ~500 instructions per
// coordinate) used to sit in the flush, where ONE lane of a wave closing a run made all 64 wait for it: 12 % of the
// iterations at 512 entries per bucket.  0 keeps the conversion in the flush (A/B).
#ifndef ARK_LAZY_FLUSH
#define ARK_LAZY_FLUSH 0
#endif
#define ARK_KEY_NONE 0xFFFFFFFFu

// What the accumulation kernels write a run into: raw limbs when ARK_LAZY_FLUSH, the canonical point otherwise.
#if ARK_LAZY_FLUSH
template <class P, int COORDS>
struct alignas(8) Msm28SlotRaw {
  uint32_t w[COORDS * Fp28<P>::N];
};
template <class P, int COORDS>
using Msm28Slot = Msm28SlotRaw<P, COORDS>;
#else
template <class P, int COORDS>
using Msm28Slot = typename std::conditional<COORDS == 4, XYZZ<Fp<P>>, XYZZ<Fp2<P>>>::type;
#endif

// one lane per (slot, coordinate); COORDS = 4 (XYZZ over Fq) or 8 (over Fq2: x.c0, x.c1, y.c0, ...).  An empty run was
// stored as zeros, which convert to the all-zero XYZZ = infinity; head / tail slots without a key were never written.
template <class P, int COORDS>
__global__ void __launch_bounds__(256)
msm_unlazy28_kernel(const uint32_t* __restrict__ raw, uint32_t nb, uint32_t segs, const uint32_t* __restrict__ head_key,
                    const uint32_t* __restrict__ tail_key, Fp<P>* __restrict__ buckets, Fp<P>* __restrict__ head,
                    Fp<P>* __restrict__ tail) {
  using F = Fp28<P>;
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t slot = idx / COORDS;
  const uint32_t c = (uint32_t)(idx % COORDS);
  if (slot >= (uint64_t)nb + 2ull * segs) return;
  Fp<P>* dst;
  if (slot < nb) {
    dst = buckets + slot * COORDS + c;
  } else if (slot < (uint64_t)nb + segs) {
    const uint64_t sg = slot - nb;
    if (head_key[sg] == ARK_KEY_NONE) return;
    dst = head + sg * COORDS + c;
  } else {
    const uint64_t sg = slot - nb - segs;
    if (tail_key[sg] == ARK_KEY_NONE) return;
    dst = tail + sg * COORDS + c;
  }
  const uint32_t* src = raw + (slot * COORDS + c) * F::N;
  F v;
#pragma unroll
  for (int i = 0; i < F::N; i++) v.l[i] = src[i];
  *dst = F::to_fp_lt8(v);
}

template <class P>
__global__ void __launch_bounds__(256)
table_to28_kernel(const Affine<Fp<P>>* __restrict__ src, Affine28<P>* __restrict__ dst, uint64_t rows) {
  using F = Fp28<P>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const Affine<Fp<P>> a = src[i];
  Affine28<P> o;
#pragma unroll
  for (int k = 0; k < Affine28<P>::WORDS; k++) o.w[k] = 0;
  if (!a.is_inf()) {
    const F x = F::from_fp(a.x), y = F::from_fp(a.y);
#pragma unroll
    for (int k = 0; k < F::N; k++) {
      o.w[k] = x.l[k];
      o.w[F::N + k] = y.l[k];
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

// acc += (px, +-py).  Value/limb classes (field28.cuh): table coordinates are canonical; acc.x is normalised and
// < 6.1 p; acc.y, acc.zz, acc.zzz are products (< 1.05 p, normalised) -- or, right after a bucket was opened with a
// negated point, acc.y = norm(2p - py) < 2p.
//   U2 = px zz            S2 = py' zzz           (py' = py or 2p - py: limbs < 2^29)
//   Pd = U2 + 8p - X1     R  = S2 + 3p - Y1      (limbs < 2^29.6, values < 9.1 p / < 4.1 p)
//   PP = Pd^2, PPP = Pd PP, Q = X1 PP, ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP
//   X3 = norm(R^2 + 5p - (PPP + 2Q))             (< 6.1 p)
//   Y3 = R (Q + 8p - X3) + (3p - Y1) PPP         (one fused pass; operand limbs < 2^29.6 each)
// Largest column: 14 * 2^59.2 + 14 * 2^57.6 + 14 * 2^56 < 2^63.5.
template <class P>
ARK_D void madd28(Acc28<P>& acc, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  using F = Fp28<P>;
  F pys;
  {
    const F n = F::template neg<2, 1>(py);
#pragma unroll
    for (int i = 0; i < F::N; i++) pys.l[i] = negate ? n.l[i] : py.l[i];
  }
  if (empty) {
    const F one = F::from_vec(one28_ni<P>());
    acc.x = px;
    acc.y = F::norm(pys);
    acc.zz = one;
    acc.zzz = one;
    empty = false;
    return;
  }
  const F U2 = F::mul(px, acc.zz);
  const F S2 = F::mul(pys, acc.zzz);
  const F Pd = F::template sub<8, 1>(U2, acc.x);
  const F R = F::template sub<3, 1>(S2, acc.y);
  if (Pd.multiple_hint() < 10u) {
    const int cls = madd28_classify<P>(Pd, R);
    if (cls == 1) {
      const F yn = F::norm(pys);
      acc.x = F::from_vec(dbl28_coord_ni<P>(px, yn, 0));
      acc.y = F::from_vec(dbl28_coord_ni<P>(px, yn, 1));
      acc.zz = F::from_vec(dbl28_coord_ni<P>(px, yn, 2));
      acc.zzz = F::from_vec(dbl28_coord_ni<P>(px, yn, 3));
      if (acc.zz.limbs_all_zero()) empty = true;       // 2P = infinity (no such point on these curves)
      return;
    }
    if (cls == 2) {
      empty = true;
      return;
    }
  }
  const F PP = F::sqr(Pd);
  const F PPP = F::mul(Pd, PP);
  const F Q = F::mul(acc.x, PP);
  acc.zz = F::mul(acc.zz, PP);
  acc.zzz = F::mul(acc.zzz, PPP);
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(F::sqr(R), F::template neg<5, 4>(W)));
  const F T = F::template sub<8, 1>(Q, X3);
  const F NY = F::template neg<3, 1>(acc.y);
  acc.y = F::mul2sum(R, T, NY, PPP);
  acc.x = X3;
}

// Same contract as msm_accumulate_kernel<Fp<P>, false>; `bases` holds Affine28 rows.
#ifndef ARK_ACC_PREFETCH_KEY
#define ARK_ACC_PREFETCH_KEY 0   // the bucket key of the next entry travels with its row index, one iteration ahead
#endif
template <class P>
__global__ void __launch_bounds__(MSM_THREADS)
msm_accumulate28_kernel(const Affine28<P>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                        const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                        Msm28Slot<P, 4>* __restrict__ buckets, Msm28Slot<P, 4>* __restrict__ head,
                        uint32_t* __restrict__ head_key, Msm28Slot<P, 4>* __restrict__ tail,
                        uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Fq = Fp<P>;
  constexpr int Q = Affine28<P>::Q;
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
#if ARK_LAZY_FLUSH
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    const uint32_t o = offsets[key], cnt = counts[key];
    const bool complete = (run_start == o) && (run_end == o + cnt);
    // three explicit branches, not a select among the captured pointers (see msm_accumulate_g2l28_kernel)
    auto store = [&](uint32_t* d) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < F::N; i++) {
        d[i] = empty ? 0u : acc.x.l[i];
        d[F::N + i] = empty ? 0u : acc.y.l[i];
        d[2 * F::N + i] = empty ? 0u : acc.zz.l[i];
        d[3 * F::N + i] = empty ? 0u : acc.zzz.l[i];
      }
    };
    if (complete) {
      store(buckets[key].w);
    } else if (first_run) {
      store(head[seg].w);
      head_key[seg] = key;
    } else {
      store(tail[seg].w);
      tail_key[seg] = key;
    }
  };
#else
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    XYZZ<Fq> out = XYZZ<Fq>::inf();
    if (!empty) {
      out.x = F::to_fp_lt8(acc.x);
      out.y = F::to_fp_lt8(acc.y);
      out.zz = F::to_fp_lt8(acc.zz);
      out.zzz = F::to_fp_lt8(acc.zzz);
    }
    msm_flush_run<Fq>(key, out, first_run, run_start, run_end, seg, offsets, counts, buckets, head, head_key, tail,
                      tail_key);
  };
#endif
  // software prefetch of the next row into explicit 16-byte registers (see msm_accumulate_kernel)
  uint4 nx[Q];
  uint32_t v_next = sorted_vals[start];
#if ARK_ACC_PREFETCH_KEY
  uint32_t key_next = cur_key;
#endif
  {
    const uint4* src = reinterpret_cast<const uint4*>(bases + (v_next & ARK_TBL_MASK));
#pragma unroll
    for (int k = 0; k < Q; k++) nx[k] = src[k];
  }
  for (uint32_t e = start; e < end; e++) {
#if ARK_ACC_PREFETCH_KEY
    const uint32_t key = key_next;
#else
    const uint32_t key = sorted_keys[e];
#endif
    const uint32_t v = v_next;
    F px, py;
    {
      uint32_t d[4 * Q];
#pragma unroll
      for (int k = 0; k < Q; k++) {
        d[4 * k + 0] = nx[k].x;
        d[4 * k + 1] = nx[k].y;
        d[4 * k + 2] = nx[k].z;
        d[4 * k + 3] = nx[k].w;
      }
#pragma unroll
      for (int k = 0; k < F::N; k++) {
        px.l[k] = d[k];
        py.l[k] = d[F::N + k];
      }
    }
    const uint32_t en = (e + 1 < end) ? e + 1 : e;       // clamp: the last iteration re-reads its own entry
    v_next = sorted_vals[en];
#if ARK_ACC_PREFETCH_KEY
    key_next = sorted_keys[en];
#endif
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
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < F::N; k++) any |= px.l[k] | py.l[k];
    if (any == 0) continue;                              // base at infinity
    madd28<P>(acc, empty, px, py, (v >> 31) != 0);
  }
  flush(cur_key, end);
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
#ifndef ARK_G2L28_FUSE_Y3
#define ARK_G2L28_FUSE_Y3 1
#endif
template <class P>
struct Affine28G2 {
  Affine28<P> half[2];
};

template <class P>
__global__ void __launch_bounds__(256)
table_to28_g2_kernel(const Affine<Fp2<P>>* __restrict__ src, Affine28G2<P>* __restrict__ dst, uint64_t rows) {
  using F = Fp28<P>;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const Affine<Fp2<P>> a = src[i];
  Affine28G2<P> o;
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int k = 0; k < Affine28<P>::WORDS; k++) o.half[h].w[k] = 0;
  }
  if (!a.is_inf()) {
    const F x0 = F::from_fp(a.x.c0), x1 = F::from_fp(a.x.c1), y0 = F::from_fp(a.y.c0), y1 = F::from_fp(a.y.c1);
#pragma unroll
    for (int k = 0; k < F::N; k++) {
      o.half[0].w[k] = x0.l[k];
      o.half[0].w[F::N + k] = y0.l[k];
      o.half[1].w[k] = x1.l[k];
      o.half[1].w[F::N + k] = y1.l[k];
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
  ARK_D static bool both(bool b) { return b && (ark_pair_xchg(b ? 1u : 0u) != 0); }
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
template <class P>
ARK_D bool madd28_g2_head(Acc28<P>& acc, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate,
                          Fp28<P>& Pd, Fp28<P>& R) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F pys = L::sel(negate, F::template neg<2, 1>(py), py);
  if (empty) {
    const F one = L::odd() ? F::zero() : F::from_vec(one28_ni<P>());
    acc.x = px;
    acc.y = F::norm(pys);
    acc.zz = one;
    acc.zzz = one;
    empty = false;
    return false;
  }
  const F U2 = L::template mul<2, 1>(px, acc.zz);
  const F S2 = L::template mul<3, 3>(pys, acc.zzz);          // pys limbs <= 2^29 - 1
  Pd = F::norm(F::template sub<8, 1>(U2, acc.x));
  R = F::norm(F::template sub<3, 1>(S2, acc.y));
  if (L::both(Pd.multiple_hint() < 10u)) {
    if (L::both(F::is_zero_mod_p_inl(Pd))) {
      if (L::both(F::is_zero_mod_p_inl(R))) {
        const F yn = F::norm(pys);
        acc.x = F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 0));
        acc.y = F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 1));
        acc.zz = F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 2));
        acc.zzz = F::from_vec(dbl28_g2_coord_ni<P>(px, yn, 3));
        if (L::both(acc.zz.limbs_all_zero())) empty = true;
      } else {
        empty = true;
      }
      return false;
    }
  }
  return true;
}
template <class P>
ARK_D void madd28_g2_tail(Acc28<P>& acc, const Fp28<P>& Pd, const Fp28<P>& R) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F PP = L::template sqr<11>(Pd);
  const F PPP = L::template mul<11, 1>(Pd, PP);
  const F Q = L::template mul<8, 1>(acc.x, PP);
  acc.zz = L::template mul<3, 1>(acc.zz, PP);
  acc.zzz = L::template mul<3, 1>(acc.zzz, PPP);
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(L::template sqr<6>(R), F::template neg<5, 4>(W)));
  const F T = F::norm(F::template sub<8, 1>(Q, X3));
  const F NY = F::template neg<3, 1>(acc.y);
#if ARK_G2L28_FUSE_Y3
  acc.y = L::template mul2<6, 1, 4, 3>(R, T, NY, PPP);
#else
  // two dual-product passes and a lazy sum: 196 more multiply-adds than the fused four-product pass (A/B knob;
  // it does not change the register spills of this kernel)
  const F Y3a = L::template mul<6, 1>(R, T);
  const F Y3b = L::template mul<4, 3>(NY, PPP);
  acc.y = F::norm(F::add(Y3a, Y3b));
#endif
  acc.x = X3;
}
template <class P>
ARK_D void madd28_g2(Acc28<P>& acc, bool& empty, const Fp28<P>& px, const Fp28<P>& py, bool negate) {
  Fp28<P> Pd, R;
  if (madd28_g2_head<P>(acc, empty, px, py, negate, Pd, R)) madd28_g2_tail<P>(acc, Pd, R);
}

#ifndef ARK_G2L28_PREFETCH
#define ARK_G2L28_PREFETCH 0  // 0: load key, row index and row at the top of every iteration; 1: key and index one
                              // iteration ahead; 2: also the row, gathered between the two halves of the addition
#endif
#ifndef ARK_G2L28_WAVES
#define ARK_G2L28_WAVES 2     // waves per SIMD the register budget is sized for; 3 (168 VGPRs, ~90 spills per
                              // addition) was measured: 10.5 vs 9.2 ms per 2^20-term MSM
#endif
template <class P>
__global__ void __launch_bounds__(MSM_THREADS, ARK_G2L28_WAVES)
msm_accumulate_g2l28_kernel(const Affine28G2<P>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                            const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                            Msm28Slot<P, 8>* __restrict__ buckets, Msm28Slot<P, 8>* __restrict__ head,
                            uint32_t* __restrict__ head_key, Msm28Slot<P, 8>* __restrict__ tail,
                            uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using F = Fp28<P>;
  using Fq = Fp<P>;
  using L = Pair28<P>;
  constexpr int Q = Affine28<P>::Q;
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
  // always_inline: a closure that is inlined late keeps every captured variable (the accumulator!) in scratch memory
#if ARK_LAZY_FLUSH
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    const uint32_t o = offsets[key], cnt = counts[key];
    const bool complete = (run_start == o) && (run_end == o + cnt);
    // this lane's halves of the four Fq2 coordinates as they stand: coordinate k, component par -> words
    // [(2k + par) N, (2k + par + 1) N) of the slot (the order of the Fq values inside XYZZ<Fp2>).
    // Three explicit branches, NOT a select among the captured pointers: hipcc turns such a select into an indexed
    // load from the closure object, which then cannot be scalarised -- and every captured variable, the accumulator
    // included, lives in scratch memory for the whole loop (28 x 16-byte scratch accesses per mixed addition)
    auto store = [&](uint32_t* d) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < F::N; i++) {
        d[(0 + par) * F::N + i] = empty ? 0u : acc.x.l[i];
        d[(2 + par) * F::N + i] = empty ? 0u : acc.y.l[i];
        d[(4 + par) * F::N + i] = empty ? 0u : acc.zz.l[i];
        d[(6 + par) * F::N + i] = empty ? 0u : acc.zzz.l[i];
      }
    };
    if (complete) {
      store(buckets[key].w);
    } else if (first_run) {
      store(head[seg].w);
      if (par == 0) head_key[seg] = key;
    } else {
      store(tail[seg].w);
      if (par == 0) tail_key[seg] = key;
    }
  };
#else
  auto flush = [&](uint32_t key, uint32_t run_end) __attribute__((always_inline)) {
    // this lane's halves of the four Fq2 coordinates, canonical 32-bit form
    XYZZ<Fq> mine = XYZZ<Fq>::inf();
    if (!empty) {
      mine.x = F::to_fp_lt8(acc.x);
      mine.y = F::to_fp_lt8(acc.y);
      mine.zz = F::to_fp_lt8(acc.zz);
      mine.zzz = F::to_fp_lt8(acc.zzz);
    }
    const uint32_t o = offsets[key], cnt = counts[key];
    const bool complete = (run_start == o) && (run_end == o + cnt);
    // three explicit branches, NOT a select among the captured pointers: hipcc turns such a select into an indexed
    // load from the closure object, which then cannot be scalarised -- and every captured variable, the accumulator
    // included, lives in scratch memory for the whole loop (28 x 16-byte scratch accesses per mixed addition)
    auto store = [&](XYZZ<Fp2<P>>* dst) __attribute__((always_inline)) {
      Fq* d = reinterpret_cast<Fq*>(dst);
      d[0 + par] = mine.x;
      d[2 + par] = mine.y;
      d[4 + par] = mine.zz;
      d[6 + par] = mine.zzz;
    };
    if (complete) {
      store(&buckets[key]);
    } else if (first_run) {
      store(&head[seg]);
      if (par == 0) head_key[seg] = key;
    } else {
      store(&tail[seg]);
      if (par == 0) tail_key[seg] = key;
    }
  };
#endif
#if ARK_G2L28_PREFETCH
  // entry e + 1's key and row index are loaded during entry e (both are needed before anything else can start);
  // with ARK_G2L28_PREFETCH == 2 the row itself is gathered between the two halves of the addition
  uint32_t v_next = sorted_vals[start], key_next = cur_key;
#if ARK_G2L28_PREFETCH == 2
  uint4 nx[Q];
  {
    const uint4* src = reinterpret_cast<const uint4*>(&bases[v_next & ARK_TBL_MASK].half[par]);
#pragma unroll
    for (int k = 0; k < Q; k++) nx[k] = src[k];
  }
#endif
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = key_next;
    const uint32_t v = v_next;
    F px, py;
    {
      uint32_t d[4 * Q];
#if ARK_G2L28_PREFETCH == 2
#pragma unroll
      for (int k = 0; k < Q; k++) {
        d[4 * k + 0] = nx[k].x;
        d[4 * k + 1] = nx[k].y;
        d[4 * k + 2] = nx[k].z;
        d[4 * k + 3] = nx[k].w;
      }
#else
      const uint4* src = reinterpret_cast<const uint4*>(&bases[v & ARK_TBL_MASK].half[par]);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const uint4 t = src[k];
        d[4 * k + 0] = t.x;
        d[4 * k + 1] = t.y;
        d[4 * k + 2] = t.z;
        d[4 * k + 3] = t.w;
      }
#endif
#pragma unroll
      for (int k = 0; k < F::N; k++) {
        px.l[k] = d[k];
        py.l[k] = d[F::N + k];
      }
    }
    const uint32_t en = (e + 1 < end) ? e + 1 : e;       // clamp: the last iteration re-reads its own entry
    v_next = sorted_vals[en];
    key_next = sorted_keys[en];
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
    F Pd, R;
    bool more = false;
    if ((any | ark_pair_xchg(any)) != 0)                 // else: base at infinity (pair-wide)
      more = madd28_g2_head<P>(acc, empty, px, py, (v >> 31) != 0, Pd, R);
#if ARK_G2L28_PREFETCH == 2
    {
      const uint4* src = reinterpret_cast<const uint4*>(&bases[v_next & ARK_TBL_MASK].half[par]);
#pragma unroll
      for (int k = 0; k < Q; k++) nx[k] = src[k];
    }
#endif
    if (more) madd28_g2_tail<P>(acc, Pd, R);
  }
#else
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
    madd28_g2<P>(acc, empty, px, py, (v >> 31) != 0);
  }
#endif
  flush(cur_key, end);
}

}  // namespace ark355
