// Batch-affine bucket accumulation: the first tree levels of every bucket's sum as AFFINE additions that share field
// inversions (Montgomery's trick), the rest through the XYZZ path of msm_impl.cuh.
//
// Why it exists and why it is OFF by default.  An XYZZ mixed addition is 8M + 2S; an affine addition is 2M + 1S plus
// one inversion, and a shared inversion costs 3M per addition: 5M + 1S, 37 % fewer multiplications -- the lever the
// round-1 review named.  The price is memory: operands, prefix products and results travel through HBM (G1 ~480 B, G2
// ~960 B per addition).  Over Fq2 an addition is 3x the multiplications of G1 for 2x the bytes, so G2 was the candidate.
// Measured on MI355X (profiles/r02_batch_affine.txt): 18.3 ms against 8.2 ms for the 2^20-term G2
// accumulation -- on the canonical 32-bit field code an affine addition costs as many cycles as the lane-pair XYZZ
// addition on carry-free 28-bit limbs, the forward passes are HBM-bound and every level waits ~1 ms for its leaf
// inversions; on 28-bit limbs the traffic (~21 GB per MSM) would bound it at about the time the XYZZ kernel takes.
// ARK355_G2_BATCH_AFFINE=1 / ARK355_G1_BATCH_AFFINE=1 (read when a key is loaded: the window table of such a group
// stays in the canonical affine form) select it; parity tests keep it honest.
//
// Shape.  The sort leaves, per bucket b, cnt_0[b] entries at off_0[b] (msm_sort).  Level l -> l + 1 pairs the nodes of
// every bucket: node t of bucket b at level l + 1 is node 2t (+ node 2t + 1 when it exists) of level l, so
// cnt_{l+1} = ceil(cnt_l / 2), off_{l+1} = its exclusive scan, and the nodes of a level lie bucket by bucket in one
// array.  One level is
//     ba_forward_kernel   every lane walks B consecutive output nodes, multiplies their denominators
//                         (x2 - x1, or 2y for a doubling, or 1 when nothing is to be inverted) into a running product,
//                         stores the prefix product per node and the lane's total
//     ba_inv_*            the lane totals are inverted together: products of four, level by level, down to <= 2048
//                         values that take a field inversion each, and back
//     ba_backward_kernel  every lane walks its nodes backwards with the inverse of its total: 1/den of a node is
//                         (running inverse) * (prefix of the node before), then lambda, x3, y3
// After ARK355_BA_LEVELS levels (default 5: 97 % of the additions) the surviving nodes are entries of an ordinary
// sorted list -- key = bucket, value = node index -- and msm_accumulate_phase / msm_reduce_phase finish the job, which
// also keeps every skewed distribution (one bucket holding everything) on the code that already handles it.
// Exceptional pairs never reach the shared product: P + P inverts 2y, P + (-P) and pairs with the point at infinity
// invert nothing.
#pragma once

namespace ark355 {

constexpr uint32_t BA_INV_R = 4;            // fan-in of the inversion tree
#if defined(ARK_EMUL)
constexpr uint32_t BA_INV_LEAF = 3;         // the emulator's launches are tiny: keep the inversion tree in play
#else
constexpr uint32_t BA_INV_LEAF = 2048;      // values that take a field inversion each
#endif
constexpr uint32_t BA_MAX_LEVELS = 10;

struct BaScratch {
  DevBuf cnt[2], off[2], tot;               // per-level bucket counts / offsets (ping-pong), totals [BA_MAX_LEVELS + 1]
  DevBuf nodes[2];                          // Affine<F> of the level being read / written
  DevBuf pref;                              // F per output node
  DevBuf lane;                              // lane totals, then their inverses, then the upper levels of the inversion tree
  MsmSort tail;                             // the surviving nodes as a sorted entry list
};

enum : int { BA_ADD = 0, BA_DBL = 1, BA_COPY = 2, BA_INF = 3 };

// what the pair (p1, p2) needs inverted (one when nothing), and which formula applies
template <class F>
ARK_D F ba_den(const Affine<F>& p1, const Affine<F>& p2, int& kind) {
  if (p1.is_inf() || p2.is_inf()) {
    kind = BA_COPY;
    return F::one();
  }
  const F d = F::sub(p2.x, p1.x);
  if (!d.is_zero()) {
    kind = BA_ADD;
    return d;
  }
  if (p1.y == p2.y && !p1.y.is_zero()) {
    kind = BA_DBL;
    return F::add(p1.y, p1.y);
  }
  kind = BA_INF;                            // P + (-P), or 2P with y = 0
  return F::one();
}

// node `idx` of the level being read: a row of the window table (first level: val = row | sign << 31) or a node
template <class F, bool FIRST>
ARK_D Affine<F> ba_load(const Affine<F>* __restrict__ src, const uint32_t* __restrict__ vals, uint32_t idx) {
  if constexpr (FIRST) {
    const uint32_t v = vals[idx];
    Affine<F> p = src[v & ARK_TBL_MASK];
    if (v >> 31) p.y = F::neg(p.y);         // -(0, 0) = (0, 0): infinity stays infinity
    return p;
  } else {
    return src[idx];
  }
}

// last bucket b with off[b] <= o (o < total, so that bucket is not empty)
ARK_D uint32_t ba_bucket_of(const uint32_t* __restrict__ off, uint32_t nb, uint32_t o) {
  uint32_t lo = 0, hi = nb;                 // invariant: off[lo] <= o, (hi == nb or off[hi] > o)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= o) lo = mid;
    else hi = mid;
  }
  return lo;
}

static __global__ void __launch_bounds__(256)
ba_halve_counts_kernel(const uint32_t* __restrict__ cnt_in, uint32_t* __restrict__ cnt_out, uint32_t nb) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) cnt_out[b] = (cnt_in[b] + 1u) >> 1;
}

template <class F, bool FIRST>
__global__ void __launch_bounds__(256)
ba_forward_kernel(const Affine<F>* __restrict__ src, const uint32_t* __restrict__ vals,
                  const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ cnt_in,
                  const uint32_t* __restrict__ off_out, const uint32_t* __restrict__ cnt_out,
                  const uint32_t* __restrict__ total_out, uint32_t nb, uint32_t per_lane, uint32_t lanes,
                  F* __restrict__ pref, F* __restrict__ lane_prod) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= lanes) return;
  const uint32_t total = *total_out;
  const uint64_t lo64 = (uint64_t)g * per_lane;
  F acc = F::one();
  if (lo64 < total) {
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (lo64 + per_lane < total) ? lo + per_lane : total;
    uint32_t b = ba_bucket_of(off_out, nb, lo);
    uint32_t ob = off_out[b], cb = cnt_out[b];
    for (uint32_t o = lo; o < hi; o++) {
      while (o >= ob + cb) {                // next non-empty bucket
        b++;
        ob = off_out[b];
        cb = cnt_out[b];
      }
      const uint32_t t = o - ob;
      if (2 * t + 1 < cnt_in[b]) {
        const uint32_t left = off_in[b] + 2 * t;
        const Affine<F> p1 = ba_load<F, FIRST>(src, vals, left), p2 = ba_load<F, FIRST>(src, vals, left + 1);
        int kind;
        const F den = ba_den(p1, p2, kind);
        if (kind <= BA_DBL) acc = F::mul(acc, den);
      }
      pref[o] = acc;
    }
  }
  lane_prod[g] = acc;
}

template <class F, bool FIRST>
__global__ void __launch_bounds__(256)
ba_backward_kernel(const Affine<F>* __restrict__ src, const uint32_t* __restrict__ vals,
                   const uint32_t* __restrict__ off_in, const uint32_t* __restrict__ cnt_in,
                   const uint32_t* __restrict__ off_out, const uint32_t* __restrict__ total_out, uint32_t nb,
                   uint32_t per_lane, uint32_t lanes, const F* __restrict__ pref, const F* __restrict__ lane_inv,
                   Affine<F>* __restrict__ dst) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= lanes) return;
  const uint32_t total = *total_out;
  const uint64_t lo64 = (uint64_t)g * per_lane;
  if (lo64 >= total) return;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = (lo64 + per_lane < total) ? lo + per_lane : total;
  F inv = lane_inv[g];
  uint32_t b = ba_bucket_of(off_out, nb, hi - 1);
  uint32_t ob = off_out[b];
  for (uint32_t o = hi; o-- > lo;) {
    while (o < ob) {                        // previous bucket (an empty one shares its successor's offset: skipped)
      b--;
      ob = off_out[b];
    }
    const uint32_t t = o - ob;
    const uint32_t left = off_in[b] + 2 * t;
    const Affine<F> p1 = ba_load<F, FIRST>(src, vals, left);
    if (2 * t + 1 >= cnt_in[b]) {           // odd one out: carried to the next level as it is
      dst[o] = p1;
      continue;
    }
    const Affine<F> p2 = ba_load<F, FIRST>(src, vals, left + 1);
    int kind;
    const F den = ba_den(p1, p2, kind);
    if (kind == BA_COPY) {
      dst[o] = p1.is_inf() ? p2 : p1;
      continue;
    }
    if (kind == BA_INF) {
      dst[o] = Affine<F>::inf();
      continue;
    }
    const F idn = (o > lo) ? F::mul(inv, pref[o - 1]) : inv;      // 1 / den
    inv = F::mul(inv, den);
    F num = F::sub(p2.y, p1.y);
    if (kind == BA_DBL) num = F::mul3(F::sqr(p1.x));              // rare: kept out of the common path
    const F lam = F::mul(num, idn);
    const F x3 = F::sub(F::sub(F::sqr(lam), p1.x), p2.x);
    const F y3 = F::sub(F::mul(lam, F::sub(p1.x, x3)), p1.y);
    dst[o] = Affine<F>{x3, y3};
  }
}

// ---- inversion of the lane totals --------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256)
ba_inv_up_kernel(const F* __restrict__ in, uint32_t n, F* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = (uint64_t)t * BA_INV_R;
  if (lo >= n) return;
  F acc = in[lo];
  for (uint32_t k = 1; k < BA_INV_R && lo + k < n; k++) acc = F::mul(acc, in[lo + k]);
  out[t] = acc;
}
// v[i] <- 1 / v[i] given inv_group[t] = 1 / (v[4t] ... v[4t+3])
template <class F>
__global__ void __launch_bounds__(256)
ba_inv_down_kernel(F* __restrict__ v, uint32_t n, const F* __restrict__ inv_group) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lo = (uint64_t)t * BA_INV_R;
  if (lo >= n) return;
  const uint32_t cnt = (lo + BA_INV_R <= n) ? BA_INV_R : (uint32_t)(n - lo);
  F val[BA_INV_R], pre[BA_INV_R];
#pragma unroll
  for (uint32_t k = 0; k < BA_INV_R; k++) {
    if (k < cnt) {
      val[k] = v[lo + k];
      pre[k] = (k == 0) ? val[0] : F::mul(pre[k - 1], val[k]);
    }
  }
  F inv = inv_group[t];
#pragma unroll
  for (uint32_t kk = 0; kk < BA_INV_R; kk++) {
    const uint32_t k = BA_INV_R - 1 - kk;
    if (k < cnt) {
      v[lo + k] = (k == 0) ? inv : F::mul(inv, pre[k - 1]);
      if (k != 0) inv = F::mul(inv, val[k]);
    }
  }
}
template <class F>
__global__ void __launch_bounds__(64)
ba_inv_leaf_kernel(F* __restrict__ v, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) v[t] = F::inv(v[t]);
}

// v[0..n) <- their inverses; `tree` has room for the upper levels (n / 3 + 8 elements are plenty)
template <class F>
static void ba_invert_all(F* v, uint32_t n, F* tree, hipStream_t stream) {
  uint32_t sizes[16];
  F* level[16];
  int depth = 0;
  sizes[0] = n;
  level[0] = v;
  while (sizes[depth] > BA_INV_LEAF) {
    sizes[depth + 1] = (sizes[depth] + BA_INV_R - 1) / BA_INV_R;
    level[depth + 1] = (depth == 0) ? tree : level[depth] + sizes[depth];
    ARK_LAUNCH((ba_inv_up_kernel<F>), dim3((sizes[depth + 1] + 255) / 256), dim3(256), 0, stream,
               (const F*)level[depth], sizes[depth], level[depth + 1]);
    ARK_CHECK_LAUNCH();
    depth++;
  }
  ARK_LAUNCH((ba_inv_leaf_kernel<F>), dim3((sizes[depth] + 63) / 64), dim3(64), 0, stream, level[depth], sizes[depth]);
  ARK_CHECK_LAUNCH();
  while (depth > 0) {
    depth--;
    ARK_LAUNCH((ba_inv_down_kernel<F>), dim3((sizes[depth + 1] + 255) / 256), dim3(256), 0, stream, level[depth],
               sizes[depth], (const F*)level[depth + 1]);
    ARK_CHECK_LAUNCH();
  }
}

// the surviving nodes of bucket b become entries (key b, value = node index)
static __global__ void __launch_bounds__(256)
ba_tail_entries_kernel(const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt, uint32_t nb,
                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint32_t o = off[b], c = cnt[b];
  for (uint32_t j = 0; j < c; j++) {
    keys[o + j] = b;
    vals[o + j] = o + j;
  }
}

static inline uint32_t ba_levels(const TunePolicy& pol) {
  const int v = pol.ba_levels;
  return (v >= 1 && v <= (int)BA_MAX_LEVELS) ? (uint32_t)v : 5u;
}

// Accumulation phase of one MSM over an existing sort, batch-affine flavour.  `table`: window table in the canonical
// affine form.  Leaves b ready for msm_reduce_phase(ctx, tail_sort(b), ...): the caller must hand THAT sort to the
// reduction (its offsets / counts describe the surviving nodes).
template <class F>
static const MsmSort& msm_ba_accumulate_phase(ark355_ctx* ctx, const MsmSort& s, MsmBuckets& b, BaScratch& ba,
                                             const Affine<F>* table, hipStream_t stream, hipEvent_t ev0 = nullptr,
                                             hipEvent_t ev1 = nullptr) {
  const MsmPlan& p = s.plan;
  ARK_REQUIRE(p.precomp && p.key_windows == 1, ARK355_EINVAL, "batch-affine accumulation needs window tables");
  const uint32_t nb = p.total_buckets;
  const uint64_t entries = (uint64_t)p.windows * p.n;
  const uint32_t levels = ba_levels(ctx->policy);
  // upper bounds of the node counts per level (the exact totals live on the device)
  uint64_t bound[BA_MAX_LEVELS + 1];
  bound[0] = entries;
  for (uint32_t l = 0; l < levels; l++) bound[l + 1] = (bound[l] + nb + 1) / 2;
  for (int k = 0; k < 2; k++) {
    ba.cnt[k].ensure((size_t)nb * 4);
    ba.off[k].ensure((size_t)nb * 4);
  }
  ba.tot.ensure((BA_MAX_LEVELS + 1) * 4);
  ba.nodes[0].ensure((size_t)bound[1] * sizeof(Affine<F>));
  ba.nodes[1].ensure((size_t)(levels > 1 ? bound[2] : 1) * sizeof(Affine<F>));
  ba.pref.ensure((size_t)bound[1] * sizeof(F));
  if (ev0) ARK_CHECK_HIP(hipEventRecord(ev0, stream));
  MsmSort& tail = ba.tail;
  tail.plan = p;
  if (p.n == 0) {
    tail.plan.n = 0;
    b.prepared = false;
    if (ev1) ARK_CHECK_HIP(hipEventRecord(ev1, stream));
    return tail;
  }
#if defined(ARK_EMUL)
  const uint64_t max_lanes = 64;
#else
  const uint64_t max_lanes = 131072;        // 2 waves per SIMD on 256 CUs
#endif
  const uint32_t* off_in = s.offsets.as<uint32_t>();
  const uint32_t* cnt_in = s.counts.as<uint32_t>();
  const Affine<F>* src = table;
  for (uint32_t l = 0; l < levels; l++) {
    uint32_t* cnt_out = ba.cnt[l & 1].as<uint32_t>();
    uint32_t* off_out = ba.off[l & 1].as<uint32_t>();
    uint32_t* tot_out = ba.tot.as<uint32_t>() + l + 1;
    ARK_LAUNCH(ba_halve_counts_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, cnt_in, cnt_out, nb);
    ARK_CHECK_LAUNCH();
    ARK_LAUNCH(scan_exclusive_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, (const uint32_t*)cnt_out, off_out, nb, tot_out);
    ARK_CHECK_LAUNCH();
    // at least 8 nodes per lane so that the inversion tree stays a small fraction of the work
    uint64_t lanes = (bound[l + 1] + 7) / 8;
    if (lanes > max_lanes) lanes = max_lanes;
    if (lanes < 1) lanes = 1;
    const uint32_t per_lane = (uint32_t)((bound[l + 1] + lanes - 1) / lanes);
    ba.lane.ensure((size_t)(lanes + lanes / 2 + 64) * sizeof(F));
    F* lane = ba.lane.as<F>();
    Affine<F>* dst = ba.nodes[l & 1].as<Affine<F>>();
    const uint32_t grid = (uint32_t)((lanes + 255) / 256);
    if (l == 0) {
      ARK_LAUNCH((ba_forward_kernel<F, true>), dim3(grid), dim3(256), 0, stream, src, s.sorted_vals.as<uint32_t>(), off_in,
                 cnt_in, (const uint32_t*)off_out, (const uint32_t*)cnt_out, (const uint32_t*)tot_out, nb, per_lane,
                 (uint32_t)lanes, ba.pref.as<F>(), lane);
    } else {
      ARK_LAUNCH((ba_forward_kernel<F, false>), dim3(grid), dim3(256), 0, stream, src, (const uint32_t*)nullptr, off_in,
                 cnt_in, (const uint32_t*)off_out, (const uint32_t*)cnt_out, (const uint32_t*)tot_out, nb, per_lane,
                 (uint32_t)lanes, ba.pref.as<F>(), lane);
    }
    ARK_CHECK_LAUNCH();
    ba_invert_all<F>(lane, (uint32_t)lanes, lane + lanes, stream);
    if (l == 0) {
      ARK_LAUNCH((ba_backward_kernel<F, true>), dim3(grid), dim3(256), 0, stream, src, s.sorted_vals.as<uint32_t>(), off_in,
                 cnt_in, (const uint32_t*)off_out, (const uint32_t*)tot_out, nb, per_lane, (uint32_t)lanes,
                 (const F*)ba.pref.as<F>(), (const F*)lane, dst);
    } else {
      ARK_LAUNCH((ba_backward_kernel<F, false>), dim3(grid), dim3(256), 0, stream, src, (const uint32_t*)nullptr, off_in,
                 cnt_in, (const uint32_t*)off_out, (const uint32_t*)tot_out, nb, per_lane, (uint32_t)lanes,
                 (const F*)ba.pref.as<F>(), (const F*)lane, dst);
    }
    ARK_CHECK_LAUNCH();
    src = dst;
    off_in = off_out;
    cnt_in = cnt_out;
  }
  // the survivors as a sorted entry list for the XYZZ path
  const uint64_t tail_entries = bound[levels];
  if (ctx->policy.trace_host)
    fprintf(stderr, "[ark355] batch-affine accumulation: %u levels over <= %llu entries, <= %llu nodes to the XYZZ tail\n", levels,
            (unsigned long long)entries, (unsigned long long)tail_entries);
  tail.plan.n = (tail_entries + p.windows - 1) / p.windows;      // the phases size their segments from windows * n
  tail.sorted_keys.ensure((size_t)tail.plan.n * p.windows * 4 + 16);
  tail.sorted_vals.ensure((size_t)tail.plan.n * p.windows * 4 + 16);
  tail.offsets.ensure((size_t)nb * 4);
  tail.counts.ensure((size_t)nb * 4);
  tail.total.ensure(16);
  ARK_CHECK_HIP(hipMemcpyAsync(tail.offsets.p, off_in, (size_t)nb * 4, hipMemcpyDeviceToDevice, stream));
  ARK_CHECK_HIP(hipMemcpyAsync(tail.counts.p, cnt_in, (size_t)nb * 4, hipMemcpyDeviceToDevice, stream));
  ARK_CHECK_HIP(hipMemcpyAsync(tail.total.p, ba.tot.as<uint32_t>() + levels, 4, hipMemcpyDeviceToDevice, stream));
  ARK_LAUNCH(ba_tail_entries_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, (const uint32_t*)tail.offsets.as<uint32_t>(),
             (const uint32_t*)tail.counts.as<uint32_t>(), nb, tail.sorted_keys.as<uint32_t>(), tail.sorted_vals.as<uint32_t>());
  ARK_CHECK_LAUNCH();
  b.prepared = false;
  msm_accumulate_phase<F>(ctx, tail, b, src, stream, nullptr, nullptr, /*bases28=*/false);
  if (ev1) ARK_CHECK_HIP(hipEventRecord(ev1, stream));
  return tail;
}

}  // namespace ark355
