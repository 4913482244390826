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
// 2 * (px, py) through the canonical 32-bit formulas
template <class P>
ARK_HD_NOINLINE Acc28<P> dbl28_affine_ni(Fp28<P> px, Fp28<P> py) {
  using F = Fp28<P>;
  const Affine<Fp<P>> a{F::to_fp(px), F::to_fp(py)};
  const XYZZ<Fp<P>> d = xyzz_dbl_affine_t<true>(a);
  return Acc28<P>{F::from_fp(d.x), F::from_fp(d.y), F::from_fp(d.zz), F::from_fp(d.zzz)};
}
template <class P>
ARK_HD_NOINLINE Fp28<P> one28_ni() {
  return Fp28<P>::from_fp(Fp<P>::one());
}
template <class P>
ARK_HD_NOINLINE XYZZ<Fp<P>> acc28_to_xyzz_ni(Acc28<P> a) {
  using F = Fp28<P>;
  return XYZZ<Fp<P>>{F::to_fp(a.x), F::to_fp(a.y), F::to_fp(a.zz), F::to_fp(a.zzz)};
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
    const F one = one28_ni<P>();
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
      acc = dbl28_affine_ni<P>(px, F::norm(pys));
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
template <class P>
__global__ void __launch_bounds__(MSM_THREADS)
msm_accumulate28_kernel(const Affine28<P>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                        const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                        XYZZ<Fp<P>>* __restrict__ buckets, XYZZ<Fp<P>>* __restrict__ head,
                        uint32_t* __restrict__ head_key, XYZZ<Fp<P>>* __restrict__ tail,
                        uint32_t* __restrict__ tail_key, uint32_t seg_log) {
  using F = Fp28<P>;
  using Fq = Fp<P>;
  constexpr int Q = Affine28<P>::Q;
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint64_t start64 = (uint64_t)seg << seg_log;
  if (start64 >= total) return;
  const uint32_t start = (uint32_t)start64;
  const uint32_t seg_len = 1u << seg_log;
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
  auto flush = [&](uint32_t key, uint32_t run_end) {
    const XYZZ<Fq> out = empty ? XYZZ<Fq>::inf() : acc28_to_xyzz_ni<P>(acc);
    msm_flush_run<Fq>(key, out, first_run, run_start, run_end, seg, offsets, counts, buckets, head, head_key, tail,
                      tail_key);
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

}  // namespace ark355
