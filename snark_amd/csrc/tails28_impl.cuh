// The MSM tails on radix-2^28 limbs (round 6): partial-run merge and bucket reduction for every MSM whose buckets were
// accumulated by the 28-bit kernels (msm28_impl.cuh), G1 and -- on lane pairs -- G2.
//
// Device replacement for the tail of ark-ec `VariableBaseMSM::msm_bigint` (un-vendored crate,
// ark-ec/src/scalar_mul/variable_base/mod.rs: the running-sum bucket reduction and the window combination), as reached
// five times per proof from `SNARK::prove` (/root/reference/snark/src/lib.rs:50-54; SURVEY.md 3.1 HOT LOOP #5).
//
// Why these kernels replaced msm_merge / msm_reduce / msm_combine (rounds 1-5, 32-bit out-of-line group operations): a
// group operation of one lane is ~5 000 dependent VALU instructions (~12 us when its wave has a SIMD to itself), so what a
// tail costs is (sequential group operations) x (that latency), and the old tails chained ~55 of them at ~31 us each
// (operands through scratch memory): 1.7 ms for the four G1 MSMs of a 2^20 proof, 5.1 ms for the G2 one -- 7 ms of kernel
// time for 0.66 M additions the accumulation loop does in 0.1 ms.  Three changes:
//   * the arithmetic is the accumulation kernels' own: lazily reduced 28-bit limbs, Karatsuba column products, everything
//     inlined, no scratch memory in any loop (add28 / add28_g2 below: add-2008-s with one fused Y3 pass, 13 reductions);
//     bucket slots stay in that form from the flush of the accumulation to the last kernel here (Slot28: no conversion);
//   * no scalar multiplication and no running sums.  With b = hi L + lo,
//         sum_b (b+1) B_b  =  sum_lo lo C_lo  +  L sum_hi hi D_hi  +  sum_hi D_hi,     C_lo = sum_hi B, D_hi = sum_lo B,
//     and for a vector X of 2^k points  sum_i i X_i = sum_j 2^j (sum of the X_i whose index has bit j set).  Stage A
//     computes the H + L row / column sums of the bucket matrix, stage B the c - 1 bit sums of those two vectors (plus one
//     complement, for the unweighted total): PLAIN sums only, each a short per-lane chain and one wave butterfly --
//     merge (~5) + A (3 + 6) + B (1 + 6) sequential additions instead of ~55 sequential group operations;
//   * the last 2 c group operations -- sum_j 2^j S_j by Horner -- leave the device: c partial sums per bucket set go to
//     the host with the proof's last copy, where the same operation costs ~1 us instead of ~12 (msm_finish_host).
#pragma once
#include "msm28_impl.cuh"

namespace ark355 {

// ---- cold path of the general addition: 2 (x, y, zz, zzz) ---------------------------------------------------------
// dbl-2008-s-1 with a = 0 never reads zz / zzz for X3, Y3: they are the coordinates of mdbl on (x, y), and
// ZZ3 = V zz, ZZZ3 = W zzz with V, W the "zz, zzz" that mdbl returns.  So the cold helpers of the mixed addition serve.
template <class P>
ARK_D void dbl28(Acc28<P>& a, bool& empty) {
  using F = Fp28<P>;
  const F x = a.x, y = a.y;
  const F X3 = F::from_vec(dbl28_coord_ni<P>(x, y, 0));
  const F Y3 = F::from_vec(dbl28_coord_ni<P>(x, y, 1));
  const F V = F::from_vec(dbl28_coord_ni<P>(x, y, 2));
  const F W = F::from_vec(dbl28_coord_ni<P>(x, y, 3));
  if (V.limbs_all_zero()) {      // y == 0: a point of order two (neither curve has one)
    empty = true;
    return;
  }
  a.x = X3;
  a.y = Y3;
  a.zz = F::mul(a.zz, V);
  a.zzz = F::mul(a.zzz, W);
}

// a += b, both in the slot classes above (add-2008-s).  Value / limb classes as in madd28:
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1                  (products: < 1.05 p, normalised)
//   Pd = U2 + 3p - U1, R = S2 + 3p - S1                                     (< 4.05 p, limbs < 2^29.6)
//   PP = Pd^2, PPP = Pd PP, Q = U1 PP, ZZ3 = ZZ1 ZZ2 PP, ZZZ3 = ZZZ1 ZZZ2 PPP
//   X3 = norm(R^2 + 5p - (PPP + 2Q))                                         (< 6.05 p)
//   Y3 = R (Q + 8p - X3) + (3p - S1) PPP                                     (one fused pass, columns as in madd28)
template <class P>
ARK_D void add28(Acc28<P>& a, bool& a_empty, const Acc28<P>& b, bool b_empty) {
  using F = Fp28<P>;
  if (b_empty) return;
  if (a_empty) {
    a = b;
    a_empty = false;
    return;
  }
  const F U1 = F::mul(a.x, b.zz);
  const F U2 = F::mul(b.x, a.zz);
  const F S1 = F::mul(a.y, b.zzz);
  const F S2 = F::mul(b.y, a.zzz);
  const F Pd = F::template sub<3, 1>(U2, U1);
  const F R = F::template sub<3, 1>(S2, S1);
  if (Pd.multiple_hint() < 10u) {
    const int cls = madd28_classify<P>(Pd, R);
    if (cls == 1) {
      dbl28<P>(a, a_empty);
      return;
    }
    if (cls == 2) {
      a_empty = true;
      return;
    }
  }
  const F PP = F::sqr(Pd);
  const F PPP = F::mul(Pd, PP);
  const F Q = F::mul(U1, PP);
  a.zz = F::mul(F::mul(a.zz, b.zz), PP);
  a.zzz = F::mul(F::mul(a.zzz, b.zzz), PPP);
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(F::sqr(R), F::template neg<5, 4>(W)));
  const F T = F::template sub<8, 1>(Q, X3);
  const F NS1 = F::template neg<3, 1>(S1);
  a.y = F::mul2sum(R, T, NS1, PPP);
  a.x = X3;
}

// The same over Fq2 on a lane pair (one component of every coordinate per lane; Pair28 of msm28_impl.cuh).  `empty` flags
// are pair-wide (both lanes hold the same value).
template <class P>
ARK_D void dbl28_g2(Acc28<P>& a, bool& empty) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  const F x = F::canon(a.x), y = a.y;           // dbl28_g2_coord_ni wants x canonical, y normalised and < 2p
  const F X3 = F::from_vec(dbl28_g2_coord_ni<P>(x, y, 0));
  const F Y3 = F::from_vec(dbl28_g2_coord_ni<P>(x, y, 1));
  const F V = F::from_vec(dbl28_g2_coord_ni<P>(x, y, 2));
  const F W = F::from_vec(dbl28_g2_coord_ni<P>(x, y, 3));
  if (L::both(F::is_zero_mod_p(V))) {
    empty = true;
    return;
  }
  a.x = X3;
  a.y = Y3;
  a.zz = L::template mul<3, 1>(a.zz, V);
  a.zzz = L::template mul<3, 1>(a.zzz, W);
}
template <class P>
ARK_D void add28_g2(Acc28<P>& a, bool& a_empty, const Acc28<P>& b, bool b_empty) {
  using F = Fp28<P>;
  using L = Pair28<P>;
  if (b_empty) return;
  if (a_empty) {
    a = b;
    a_empty = false;
    return;
  }
  const F U1 = L::template mul<8, 1>(a.x, b.zz);
  const F U2 = L::template mul<8, 1>(b.x, a.zz);
  const F S1 = L::template mul<3, 1>(a.y, b.zzz);
  const F S2 = L::template mul<3, 1>(b.y, a.zzz);
  const F Pd = F::norm(F::template sub<3, 1>(U2, U1));
  const F R = F::norm(F::template sub<3, 1>(S2, S1));
  if (L::both(Pd.multiple_hint() < 10u)) {
    if (L::both(F::is_zero_mod_p_inl(Pd))) {
      if (L::both(F::is_zero_mod_p_inl(R))) dbl28_g2<P>(a, a_empty);
      else a_empty = true;
      return;
    }
  }
  const F PP = L::template sqr<6>(Pd);
  const F PPP = L::template mul<6, 1>(Pd, PP);
  const F Q = L::template mul<3, 1>(U1, PP);
  a.zz = L::template mul<3, 1>(L::template mul<3, 1>(a.zz, b.zz), PP);
  a.zzz = L::template mul<3, 1>(L::template mul<3, 1>(a.zzz, b.zzz), PPP);
  const F W = F::add(PPP, F::add(Q, Q));
  const F X3 = F::norm(F::add(L::template sqr<6>(R), F::template neg<5, 4>(W)));
  const F T = F::norm(F::template sub<8, 1>(Q, X3));
  const F NS1 = F::template neg<3, 1>(S1);
  a.y = L::template mul2<6, 1, 4, 3>(R, T, NS1, PPP);
  a.x = X3;
}

// ---- one item (bucket / partial sum) per lane (G1) or per lane pair (G2) --------------------------------------------
template <class P, bool G2>
struct Tail28 {
  using F = Fp28<P>;
  using Fq = Fp<P>;
  using Slot = typename std::conditional<G2, Slot28G2<P>, Slot28<P>>::type;
  using Out = typename std::conditional<G2, XYZZ<Fp2<P>>, XYZZ<Fp<P>>>::type;
  static constexpr int LPI = G2 ? 2 : 1;            // lanes per item
  static constexpr int ITEMS_PER_WAVE = 64 / LPI;
  ARK_D static uint32_t par() { return G2 ? (threadIdx.x & 1u) : 0u; }
  ARK_D static const Slot28<P>* half(const Slot* s) {
    if constexpr (G2) return &s->half[threadIdx.x & 1u];
    else return s;
  }
  ARK_D static Slot28<P>* half(Slot* s) {
    if constexpr (G2) return &s->half[threadIdx.x & 1u];
    else return s;
  }
  ARK_D static void load(const Slot* s, Acc28<P>& a, bool& empty) {
    const bool z = slot28_get<P>(half(s), a);
    if constexpr (G2) empty = Pair28<P>::both(z);
    else empty = z;
  }
  ARK_D static void store(Slot* s, const Acc28<P>& a, bool empty) { slot28_put<P>(half(s), a, empty); }
  ARK_D static void add(Acc28<P>& a, bool& ae, const Acc28<P>& b, bool be) {
    if constexpr (G2) add28_g2<P>(a, ae, b, be);
    else add28<P>(a, ae, b, be);
  }
  // a += (the value of lane ^ mask); mask >= LPI keeps the lane parity of a pair
  ARK_D static void add_xor(Acc28<P>& a, bool& ae, int mask) {
    Acc28<P> o;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
      o.x.l[i] = (uint32_t)__shfl_xor((int)a.x.l[i], mask, 64);
      o.y.l[i] = (uint32_t)__shfl_xor((int)a.y.l[i], mask, 64);
      o.zz.l[i] = (uint32_t)__shfl_xor((int)a.zz.l[i], mask, 64);
      o.zzz.l[i] = (uint32_t)__shfl_xor((int)a.zzz.l[i], mask, 64);
    }
    const bool oe = __shfl_xor(ae ? 1 : 0, mask, 64) != 0;
    add(a, ae, o, oe);
  }
  // butterfly over the items of a wave: afterwards every lane (pair) holds the wave's sum
  ARK_D static void wave_sum(Acc28<P>& a, bool& ae) {
#pragma unroll 1
    for (int mask = 32; mask >= LPI; mask >>= 1) add_xor(a, ae, mask);
  }
  // the canonical 32-bit image of this lane's component(s) of a point
  ARK_D static void store_canonical(Out* dst, const Acc28<P>& a, bool empty) {
    Fq* d = reinterpret_cast<Fq*>(dst);
    const uint32_t pr = par();
    const Fq z = Fq::zero();
    d[0 * LPI + pr] = empty ? z : F::to_fp(a.x);
    d[1 * LPI + pr] = empty ? z : F::to_fp(a.y);
    d[2 * LPI + pr] = empty ? z : F::to_fp(a.zz);
    d[3 * LPI + pr] = empty ? z : F::to_fp(a.zzz);
  }
};

// ---- the jobs of one launch: the tails of up to TAIL28_MAX MSMs of one shape side by side (blockIdx.z = MSM) -----------
constexpr int TAIL28_MAX = 4;
struct TailJob28 {
  const uint32_t* offsets;
  const uint32_t* counts;
  void* buckets;            // Slot[total_buckets]
  const void* head;         // Slot[segs]
  const uint32_t* head_key;
  const void* tail;
  const uint32_t* tail_key;
  uint32_t* heavy_count;
  uint32_t* heavy_list;
  void* rc;                 // stage A: Slot[key_windows][H + L]: row sums D_hi, then column sums C_lo
  void* out;                // stage B: Out[key_windows][c]: bit sums 0 .. c-2, then the complement of the top bit
  uint32_t seg_len, heavy_span;
};
struct TailBatch28 {
  TailJob28 j[TAIL28_MAX];
};

#ifndef ARK_TAIL28_WAVES
#define ARK_TAIL28_WAVES 2      // waves per SIMD the register budget of the tail kernels is sized for
#endif

// Merge: the lane (pair) of a bucket whose entries straddle segment boundaries adds its partial runs.  Which slots those
// are follows from the flush protocol of the accumulation (msm_flush_run): segment t0 holds the bucket's first run in
// head[t0] when the bucket starts the segment and in tail[t0] otherwise; every later segment t0 < t <= t1 STARTS inside the
// bucket, so its first run -- head[t] -- is the bucket's.  One addition per loop iteration for every lane of the wave
// (the rounds-1-5 kernel tested head and tail of every segment in turn: two divergent additions per iteration).  Buckets
// over more than heavy_span segments are recorded for msm_merge_heavy28_kernel.
template <class P, bool G2>
__global__ void __launch_bounds__(MSM_THREADS, ARK_TAIL28_WAVES)
msm_merge28_kernel(TailBatch28 tb, uint32_t total_buckets) {
  using T = Tail28<P, G2>;
  using Slot = typename T::Slot;
  const TailJob28& J = tb.j[blockIdx.z];
  const uint32_t key = (blockIdx.x * blockDim.x + threadIdx.x) / T::LPI;
  if (key >= total_buckets) return;
  const uint32_t cnt = J.counts[key];
  if (cnt == 0) return;
  const uint32_t o = J.offsets[key];
  const uint32_t t0 = o / J.seg_len, t1 = (o + cnt - 1) / J.seg_len;
  if (t0 == t1) return;   // the single run was complete and already written
  if (t1 - t0 > J.heavy_span) {
    if (T::par() == 0) J.heavy_list[atomicAdd(J.heavy_count, 1u)] = key;
    return;
  }
  const Slot* head = static_cast<const Slot*>(J.head);
  const Slot* tail = static_cast<const Slot*>(J.tail);
  const bool starts_segment = (o == t0 * J.seg_len);
  Acc28<P> sum;
  bool empty;
  T::load(starts_segment ? &head[t0] : &tail[t0], sum, empty);
  if ((starts_segment ? J.head_key[t0] : J.tail_key[t0]) != key) empty = true;
#pragma unroll 1
  for (uint32_t t = t0 + 1; t <= t1; t++) {
    Acc28<P> p;
    bool pe;
    T::load(&head[t], p, pe);
    if (J.head_key[t] != key) pe = true;
    T::add(sum, empty, p, pe);
  }
  T::store(&static_cast<Slot*>(J.buckets)[key], sum, empty);
}

// one workgroup per heavy bucket: its items stride over the bucket's segments, wave butterfly, the waves meet in LDS
template <class P, bool G2>
__global__ void __launch_bounds__(MSM_THREADS, ARK_TAIL28_WAVES)
msm_merge_heavy28_kernel(TailBatch28 tb) {
  using T = Tail28<P, G2>;
  using Slot = typename T::Slot;
  __shared__ Slot wave_out[MSM_THREADS / 64];
  const TailJob28& J = tb.j[blockIdx.z];
  const Slot* head = static_cast<const Slot*>(J.head);
  const Slot* tail = static_cast<const Slot*>(J.tail);
  const uint32_t nheavy = *J.heavy_count;
  const uint32_t item = threadIdx.x / T::LPI, items = blockDim.x / T::LPI;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t key = J.heavy_list[h];
    const uint32_t o = J.offsets[key], cnt = J.counts[key];
    const uint32_t t0 = o / J.seg_len, t1 = (o + cnt - 1) / J.seg_len;
    Acc28<P> sum;
    bool empty = true;
    sum.x = sum.y = sum.zz = sum.zzz = Fp28<P>::zero();
#pragma unroll 1
    for (uint32_t t = t0 + item; t <= t1; t += items) {
      // (segment t0 may hold the bucket's run in either slot; see msm_merge28_kernel)
      const bool use_tail = (t == t0) && (o != t0 * J.seg_len);
      Acc28<P> p;
      bool pe;
      T::load(use_tail ? &tail[t] : &head[t], p, pe);
      if ((use_tail ? J.tail_key[t] : J.head_key[t]) != key) pe = true;
      T::add(sum, empty, p, pe);
    }
    T::wave_sum(sum, empty);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < (uint32_t)T::LPI) T::store(&wave_out[wave], sum, empty);
    __syncthreads();
    if (threadIdx.x < (uint32_t)T::LPI) {
      Acc28<P> tot;
      bool te;
      T::load(&wave_out[0], tot, te);
      for (uint32_t v = 1; v < blockDim.x / 64; v++) {
        Acc28<P> p;
        bool pe;
        T::load(&wave_out[v], p, pe);
        T::add(tot, te, p, pe);
      }
      T::store(&static_cast<Slot*>(J.buckets)[key], tot, te);
    }
    __syncthreads();
  }
}

// ---- selected plain sums: one wave per output ---------------------------------------------------------------------------
// Stage A (STAGE == 0), grid (H + L, bucket sets, MSMs): output o < H is the row sum D_o = sum_lo B[o L + lo], output
// H + lo the column sum C_lo = sum_hi B[hi L + lo]; written as slots to rc[set][o].
// Stage B (STAGE == 1), grid (c, bucket sets, MSMs), over the two vectors stage A left: output k < lb is the sum of the C_lo
// whose index has bit k set, output lb + k (k < hb) the sum of the D_hi with bit k set, output c - 1 the sum of the D_hi
// with the TOP bit clear (so that outputs c - 2 and c - 1 add up to the plain total); written in the canonical 32-bit form
// to out[set][k] for the host (msm_finish_host).
// An output's items are dealt to the wave's lanes (lane pairs) round-robin: ceil(count / 64) - 1 sequential additions per
// lane, then the butterfly (6 steps; 5 on pairs).
template <class P, bool G2, int STAGE>
__global__ void __launch_bounds__(64, ARK_TAIL28_WAVES)
msm_selsum28_kernel(TailBatch28 tb, uint32_t lb, uint32_t hb) {
  using T = Tail28<P, G2>;
  using Slot = typename T::Slot;
  const TailJob28& J = tb.j[blockIdx.z];
  const uint32_t L = 1u << lb, H = 1u << hb;
  const uint32_t o = blockIdx.x, set = blockIdx.y;
  const Slot* src;
  uint32_t count, k = 0;
  int mode;               // 0: base + i * stride, 1: index i with bit k inserted as 1, 2: index i
  uint32_t base = 0, stride = 1;
  if constexpr (STAGE == 0) {
    src = static_cast<const Slot*>(J.buckets) + (size_t)set * ((size_t)L << hb);
    mode = 0;
    if (o < H) {
      base = o << lb;
      stride = 1;
      count = L;
    } else {
      base = o - H;
      stride = L;
      count = H;
    }
  } else {
    const Slot* rc = static_cast<const Slot*>(J.rc) + (size_t)set * (H + L);
    if (o < lb) {
      src = rc + H;         // the column sums
      k = o;
      mode = 1;
      count = L >> 1;
    } else if (o < lb + hb) {
      src = rc;             // the row sums
      k = o - lb;
      mode = 1;
      count = H >> 1;
    } else {
      src = rc;
      mode = 2;
      count = H >> 1;
    }
  }
  const uint32_t item = threadIdx.x / T::LPI;
  Acc28<P> sum;
  bool empty = true;
  sum.x = sum.y = sum.zz = sum.zzz = Fp28<P>::zero();
#pragma unroll 1
  for (uint32_t i = item; i < count; i += T::ITEMS_PER_WAVE) {
    uint32_t idx;
    if (mode == 0) idx = base + i * stride;
    else if (mode == 1) idx = ((i >> k) << (k + 1)) | (1u << k) | (i & ((1u << k) - 1u));
    else idx = i;
    Acc28<P> p;
    bool pe;
    T::load(&src[idx], p, pe);
    T::add(sum, empty, p, pe);
  }
  T::wave_sum(sum, empty);
  if (threadIdx.x < (uint32_t)T::LPI) {
    if constexpr (STAGE == 0) {
      T::store(static_cast<Slot*>(J.rc) + (size_t)set * (H + L) + o, sum, empty);
    } else {
      using Out = typename T::Out;
      T::store_canonical(static_cast<Out*>(J.out) + (size_t)set * (lb + hb + 1) + o, sum, empty);
    }
  }
}

// How the c - 1 bucket-index bits are split into the column index (low lb bits) and the row index (high hb bits)
static inline void tails28_split(uint32_t c, uint32_t* lb, uint32_t* hb) {
  const uint32_t bits = c - 1;
  *lb = (bits + 1) / 2;
  *hb = bits - *lb;
}

// ---- the host's share: Horner over the bit sums -----------------------------------------------------------------------
// parts[set][k], k < c: what stage B wrote for `sets` bucket sets of window size c (set j carries weight 2^(c j): window
// tables with a stride; ONE set otherwise).  sum_set 2^(c set) [ sum_{k < c-1} 2^k parts[k] + parts[c-2] + parts[c-1] ].
template <class F>
static XYZZ<F> msm_finish_host(const XYZZ<F>* parts, uint32_t sets, uint32_t c) {
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t s = sets; s-- > 0;) {
    const XYZZ<F>* p = parts + (size_t)s * c;
    for (uint32_t k = c; k-- > 0;) {
      // bit position c - 1 of a set does not exist (c - 1 index bits): a plain doubling keeps the sets c bits apart
      if (!acc.is_inf()) acc = xyzz_dbl(acc);
      if (k < c - 1) acc = xyzz_add(acc, p[k]);
    }
    acc = xyzz_add(acc, xyzz_add(p[c - 2], p[c - 1]));
  }
  return acc;
}

}  // namespace ark355
