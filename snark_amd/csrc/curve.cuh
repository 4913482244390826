// Short-Weierstrass (a = 0) group arithmetic for G1 (over Fq) and G2 (over Fq2), host + device.
//
// Device-side replacement for the un-vendored ark-ec `short_weierstrass::{Affine, Projective}`
// used by `VariableBaseMSM`.  Accumulators are kept in extended Jacobian
// ("XYZZ": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2) because the mixed addition is the cheapest known
// inversion-free one (8M + 2S); results are normalised to affine, which is canonical, so the choice of
// coordinates cannot change any output byte.
//
// Conventions: affine infinity is (0, 0) (not on either curve since b != 0); XYZZ infinity is ZZ == 0.
//
// Every formula exists in two instantiations: NI = false inlines the field multiplications (the hot
// bucket-accumulation kernel), NI = true calls the out-of-line field routines and is itself wrapped in
// out-of-line functions (`*_ni`) for the cold kernels.
#pragma once
#include "field.cuh"

namespace ark355 {

template <bool NI, class F>
ARK_HD F fmul(const F& a, const F& b) {
  if constexpr (NI) return F::mul_ni(a, b);
  else return F::mul(a, b);
}
template <bool NI, class F>
ARK_HD F fsqr(const F& a) {
  if constexpr (NI) return F::sqr_ni(a);
  else return F::sqr(a);
}

template <class F>
struct Affine {
  F x, y;
  ARK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  ARK_HD static Affine inf() { return Affine{F::zero(), F::zero()}; }
  ARK_HD static Affine neg(const Affine& p) { return Affine{p.x, F::neg(p.y)}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ARK_HD bool is_inf() const { return zz.is_zero(); }
  ARK_HD static XYZZ inf() { return XYZZ{F::zero(), F::zero(), F::zero(), F::zero()}; }
  ARK_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return XYZZ{p.x, p.y, F::one(), F::one()};
  }
  ARK_HD static XYZZ neg(const XYZZ& p) { return XYZZ{p.x, F::neg(p.y), p.zz, p.zzz}; }
};

// 2*P for affine P (mdbl-2008-s-1)
template <bool NI, class F>
ARK_HD XYZZ<F> xyzz_dbl_affine_t(const Affine<F>& p) {
  if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::inf();
  F U = F::mul2(p.y);
  F V = fsqr<NI>(U);
  F W = fmul<NI>(U, V);
  F S = fmul<NI>(p.x, V);
  F M = F::mul3(fsqr<NI>(p.x));
  F X3 = F::sub(fsqr<NI>(M), F::mul2(S));
  F Y3 = F::sub(fmul<NI>(M, F::sub(S, X3)), fmul<NI>(W, p.y));
  return XYZZ<F>{X3, Y3, V, W};
}

// 2*P (dbl-2008-s-1, a = 0)
template <bool NI, class F>
ARK_HD XYZZ<F> xyzz_dbl_t(const XYZZ<F>& p) {
  if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::inf();
  F U = F::mul2(p.y);
  F V = fsqr<NI>(U);
  F W = fmul<NI>(U, V);
  F S = fmul<NI>(p.x, V);
  F M = F::mul3(fsqr<NI>(p.x));
  F X3 = F::sub(fsqr<NI>(M), F::mul2(S));
  F Y3 = F::sub(fmul<NI>(M, F::sub(S, X3)), fmul<NI>(W, p.y));
  return XYZZ<F>{X3, Y3, fmul<NI>(V, p.zz), fmul<NI>(W, p.zzz)};
}

// by VALUE on purpose: a reference parameter would force the caller's point into a stack slot that hipcc
// then keeps up to date in every iteration of the hot loop (6 scratch stores per mixed addition, round-1 ISA)
template <class F>
ARK_HD_NOINLINE XYZZ<F> xyzz_dbl_affine_ni(Affine<F> p) {
  return xyzz_dbl_affine_t<true>(p);
}

// acc += P, P affine (madd-2008-s); handles every special case.
template <bool NI, class F>
ARK_HD void xyzz_madd_t(XYZZ<F>& acc, const Affine<F>& p) {
  if (p.is_inf()) return;
  if (acc.is_inf()) {
    acc = XYZZ<F>{p.x, p.y, F::one(), F::one()};
    return;
  }
  F U2 = fmul<NI>(p.x, acc.zz);
  F S2 = fmul<NI>(p.y, acc.zzz);
  F Pd = F::sub(U2, acc.x);
  F R = F::sub(S2, acc.y);
  if (Pd.is_zero()) {
    if (R.is_zero()) {
      // rare (P == acc): out-of-line doubling, argument passed by value (see xyzz_dbl_affine_ni)
      acc = xyzz_dbl_affine_ni(p);
    } else {
      acc = XYZZ<F>::inf();
    }
    return;
  }
  // ordered so that values die early (Pd after PPP, PP after ZZ3, PPP after X3): the G2 instantiation lives
  // at the edge of the 512-register file
  F PP = fsqr<NI>(Pd);
  F PPP = fmul<NI>(Pd, PP);
  F Q = fmul<NI>(acc.x, PP);
  acc.zz = fmul<NI>(acc.zz, PP);
  acc.zzz = fmul<NI>(acc.zzz, PPP);
  if constexpr (!NI && F::FUSED_MUL_SUB) {
    // Y3 = R (Q - X3) - Y1 PPP as one multi-product Montgomery pass (one reduction instead of two)
    F X3 = F::sub(F::sub(fsqr<NI>(R), PPP), F::mul2(Q));
    acc.x = X3;
    acc.y = F::mul_sub(R, F::sub(Q, X3), acc.y, PPP);
  } else {
    F T = fmul<NI>(acc.y, PPP);
    F X3 = F::sub(F::sub(fsqr<NI>(R), PPP), F::mul2(Q));
    acc.x = X3;
    acc.y = F::sub(fmul<NI>(R, F::sub(Q, X3)), T);
  }
}

// a + b (add-2008-s); handles every special case.
template <bool NI, class F>
ARK_HD XYZZ<F> xyzz_add_t(const XYZZ<F>& a, const XYZZ<F>& b) {
  if (a.is_inf()) return b;
  if (b.is_inf()) return a;
  F U1 = fmul<NI>(a.x, b.zz);
  F U2 = fmul<NI>(b.x, a.zz);
  F S1 = fmul<NI>(a.y, b.zzz);
  F S2 = fmul<NI>(b.y, a.zzz);
  F Pd = F::sub(U2, U1);
  F R = F::sub(S2, S1);
  if (Pd.is_zero()) {
    if (R.is_zero()) return xyzz_dbl_t<true>(a);
    return XYZZ<F>::inf();
  }
  F PP = fsqr<NI>(Pd);
  F PPP = fmul<NI>(Pd, PP);
  F Q = fmul<NI>(U1, PP);
  F X3 = F::sub(F::sub(fsqr<NI>(R), PPP), F::mul2(Q));
  F Y3 = F::sub(fmul<NI>(R, F::sub(Q, X3)), fmul<NI>(S1, PPP));
  F ZZ3 = fmul<NI>(fmul<NI>(a.zz, b.zz), PP);
  F ZZZ3 = fmul<NI>(fmul<NI>(a.zzz, b.zzz), PPP);
  return XYZZ<F>{X3, Y3, ZZ3, ZZZ3};
}

// ---- hot (inlined) flavour: the bucket-accumulation kernel ------------------------------------------
template <class F>
ARK_HD void xyzz_madd(XYZZ<F>& acc, const Affine<F>& p) {
  xyzz_madd_t<false>(acc, p);
}

// ---- cold (out-of-line) flavours -------------------------------------------------------------------------
template <class F>
ARK_HD_NOINLINE void xyzz_madd_ni(XYZZ<F>& acc, const Affine<F>& p) {
  xyzz_madd_t<true>(acc, p);
}
// One out-of-line body per field with the multiplications INLINED: the latency-bound tails (merge, bucket reduction,
// combination: a few dozen dependent group operations on a handful of waves) spend their time in exactly these two
// functions, and a call per field multiplication (operands and result through scratch memory) doubled it.
#ifndef ARK_COLD_GROUP_OPS_INLINE_MUL
#define ARK_COLD_GROUP_OPS_INLINE_MUL 1
#endif
// Fq only: the Fq2 addition with inlined multiplications wants 456 registers, which drags every kernel that calls it down to
// one wave per SIMD -- and then the CUs a tail kernel sits on cannot host a wave of another proof's accumulation kernel.
// (A register cap cannot be put on a device function: amdgpu_num_vgpr applies to kernels only.)
template <class F>
ARK_HD_NOINLINE XYZZ<F> xyzz_add(const XYZZ<F>& a, const XYZZ<F>& b) {
  return xyzz_add_t<!(ARK_COLD_GROUP_OPS_INLINE_MUL && F::COLD_INLINE_MUL)>(a, b);
}
template <class F>
ARK_HD_NOINLINE XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
  return xyzz_dbl_t<!(ARK_COLD_GROUP_OPS_INLINE_MUL && F::COLD_INLINE_MUL)>(p);
}

// canonical affine image (one field inversion)
template <class F>
ARK_HD_NOINLINE Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
  if (p.is_inf()) return Affine<F>::inf();
  F i3 = F::inv(p.zzz);               // 1/Z^3
  F iz = F::mul_ni(p.zz, i3);         // 1/Z
  F i2 = F::sqr_ni(iz);               // 1/Z^2
  return Affine<F>{F::mul_ni(p.x, i2), F::mul_ni(p.y, i3)};
}

// k * P by double-and-add over a little-endian u32 scalar of `nlimbs` limbs (tails only).
template <class F>
ARK_HD_NOINLINE XYZZ<F> xyzz_mul_scalar(const XYZZ<F>& p, const uint32_t* k, int nlimbs) {
  XYZZ<F> acc = XYZZ<F>::inf();
  bool started = false;
  for (int i = nlimbs - 1; i >= 0; i--) {
    uint32_t w = k[i];
    for (int b = 31; b >= 0; b--) {
      if (started) acc = xyzz_dbl(acc);
      if ((w >> b) & 1) {
        acc = xyzz_add(acc, p);
        started = true;
      }
    }
  }
  return acc;
}

using BlsG1Affine = Affine<BlsFq>;
using BlsG2Affine = Affine<BlsFq2>;
using BlsG1 = XYZZ<BlsFq>;
using BlsG2 = XYZZ<BlsFq2>;
using BnG1Affine = Affine<BnFq>;
using BnG2Affine = Affine<BnFq2>;
using BnG1 = XYZZ<BnFq>;
using BnG2 = XYZZ<BnFq2>;

}  // namespace ark355
