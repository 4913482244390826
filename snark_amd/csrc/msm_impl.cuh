// Pippenger multi-scalar multiplication for gfx950 (G1 over Fq, G2 over Fq2; BLS12-381 and BN254).
//
// Device replacement for ark-ec `VariableBaseMSM::msm_bigint` (un-vendored crate,
// ark-ec/src/scalar_mul/variable_base/mod.rs) as called five times by the Groth16 prover
// (SURVEY.md 3.1 HOT LOOP #5).  Same mathematics -- signed c-bit window digits, bucket
// accumulation, running-sum bucket reduction, window combination -- laid out for the GPU:
//
//   K2/K3 sort     signed c-bit digits of every scalar (key = bucket, value = table row | sign) and a two-level
//                  counting sort of the (key, value) entries: level 1 groups by key >> 8 with per-workgroup LDS
//                  histograms and private output ranges, level 2 places tiles of 4096 entries with LDS ranks and one
//                  global atomic per (tile, key)  ("two-level counting sort" below; the one-pass version with
//                  per-entry global atomics is kept as ARK355_SORT=legacy)
//   K4 accumulate  THE dominant kernels: the sorted entry list is cut into fixed segments of L
//                  entries, one lane per segment (two lanes for G2), so every lane performs exactly L mixed
//                  additions (XYZZ += affine) whatever the bucket-size distribution; runs that cover a whole
//                  bucket are written straight to the bucket array, the at most two partial runs
//                  per segment go to head/tail slots.  Resident keys: radix-2^28 kernels over window tables
//                  (msm28_impl.cuh); ad-hoc bases: the 32-bit kernels of this file
//      merge       one thread per bucket that straddles segments adds its partial runs (a workgroup for heavy ones)
//   K5 reduce      per window sum_b (b+1) B_b: every lane takes K consecutive buckets (running sum),
//                  adds (first index)*S via a short double-and-add, then wave-wide butterfly
//                  reduction with __shfl_xor, one partial per workgroup
//      combine     lane w sums window w's partials, doubles it c*w times, __shfl_xor tree over windows
//                  (with window tables there is ONE bucket set and no doubling)
//   The merge / reduce / combine tails are latency-bound (a few dozen dependent group operations on 32-128 workgroups).
//   G1: the out-of-line group addition inlines its multiplications (curve.cuh); G2: the *_pair_kernel flavours give every
//   bucket / chunk to a LANE PAIR (Fp2L: one component of each Fq2 coordinate per lane).
//
// Algorithmic bytes per term: 32 B scalar + affine base (G1 96 B / G2 192 B BLS12-381).
#pragma once
#include <type_traits>
#include "common.h"

namespace ark355 {

constexpr uint32_t MSM_INVALID = 0xFFFFFFFFu;
#ifdef ARK_DEBUG_SMALL_TABLE
#define ARK_TBL_MASK 0x3FFu   // experiment only: every gather hits the same 1024 rows (results are wrong)
#else
#define ARK_TBL_MASK 0x7FFFFFFFu
#endif
// entries per accumulate lane: msm_seg_len() below (32 keeps small MSMs wide enough to fill the chip; ~64 halves the
// partial runs of the 2^24-entry MSMs of a 2^20 proof -- measured on MI355X: 53.7 / 52.6 / 52.3 ms per proof for
// 32 / 64 / 128)
constexpr uint32_t MSM_RED_K = 4;         // buckets per reduce lane (latency-bound kernel: short chains, many lanes)
constexpr uint32_t MSM_THREADS = 256;
#ifndef ARK_G1_PREFETCH
#define ARK_G1_PREFETCH 1   // measured neutral on MI355X (7.75 vs 7.79 ms for A+B1); kept for small-n latency
#endif

struct MsmPlan {
  uint64_t n = 0;
  uint32_t c = 0, windows = 0, buckets_per_window = 0, total_buckets = 0;
  uint32_t scalar_bits = 0;
  // precomp: the bases come with per-window tables T[w*n + i] = 2^(c*w) * P_i (built once when a key is
  // loaded; MI355X has the HBM for it), so every window shares ONE bucket set and the c*w doublings of the
  // window combination disappear from the per-proof path.
  bool precomp = false;
  // Window tables may hold only every wstride-th window (T[q*n + i] = 2^(c*wstride*q) * P_i, q < table_windows): a key
  // whose full tables would not fit HBM trades table rows for bucket sets.  Window w = wstride*q + j then reads table
  // row q and lands in bucket set j; the sets are combined as sum_j 2^(c*j) S_j (msm_combine_kernel), i.e. c*(wstride-1)
  // doublings per MSM come back.  wstride = 1: one bucket set, no doubling at all (the default whenever it fits);
  // without tables wstride = windows (every window its own set, every value indexes the base vector itself).
  uint32_t wstride = 1, table_windows = 0;
  uint32_t key_windows = 0;     // bucket sets: wstride with precomp, `windows` without
  // Scalars above (r - 1) / 2 are replaced by r - k with every digit's sign flipped (k P = (r - k)(-P)): the values
  // that remain are one bit shorter, which saves a whole window exactly when c divides the scalar width -- c = 17 for
  // 255-bit scalars: 15 windows instead of 16.  Derived from c alone, so a table and the MSMs over it always agree.
  bool negate_high = false;
};

// Entries per accumulation lane ("segment length").  The accumulation kernels keep CUs x 2 workgroups x 256 lanes
// resident (2 waves per SIMD), so a launch runs in whole ROUNDS of that many lanes: 2^20 terms x 16 windows / 64 = 1024
// workgroups = exactly two rounds, but 13 windows (c = 20) / 64 = 832 workgroups still take two rounds -- the second
// 62 % full -- and the saved additions buy nothing (measured in round 1: -19 % entries, -5 % time).  The length is
// therefore chosen so that the segments fill a whole number of rounds: nearest round count at ~64 entries per lane,
// then ceil(entries / (slots x rounds)).  pair_lanes: the G2 kernels use two lanes per segment.  force: TunePolicy::msm_seg
// (policy MSM_SEG=<len>: A/B, tests).
static inline uint32_t msm_seg_len(uint64_t entries, bool pair_lanes, int force = 0) {
  if (force >= 1 && force <= 4096) return (uint32_t)force;
#if defined(ARK_EMUL)
  const uint64_t cus = 1;
#else
  static const uint64_t cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return (uint64_t)n;
  }();
#endif
  const uint64_t slots = cus * 2 * MSM_THREADS / (pair_lanes ? 2 : 1);
  uint64_t rounds = (entries + slots * 32) / (slots * 64);
  if (rounds < 1) rounds = 1;
  // one round: every lane takes entries / slots (a 2^18-term MSM at c = 13 has 5.2 M entries = 40 per lane; with a fixed
  // 32 it ran as 1.25 rounds, i.e. two); never below 16, so that small MSMs do not drown in partial runs
  uint64_t len = (entries + slots * rounds - 1) / (slots * rounds);
  if (len < 16) len = 16;
  if (len > 160) len = 160;
  return (uint32_t)len;
}

struct MsmPlan;
static inline int msm_digit_flags(const MsmPlan& p);
// pref_c: TunePolicy::msm_c -- the window size asked for resident tables (0: this planner's choice)
inline MsmPlan msm_plan(uint64_t n, uint32_t scalar_bits, bool precomp = false, int force_c = 0, uint32_t force_stride = 0,
                        int pref_c = 0) {
  MsmPlan p;
  p.n = n;
  p.precomp = precomp;
  p.scalar_bits = scalar_bits;
  // target window: ceil(log2 n) - 4, clamped to [4, 16]; then the nearest window size whose TOP window is still
  // well populated.  With c not dividing the scalar width the last window only sees a few bits, every term of
  // that window lands in a handful of buckets, and the sort/merge degenerate (measured: c = 15 -> 138 ms instead
  // of 62 ms per 2^20 proof).  For 255/254-bit scalars the good sizes are 4, 8, 13, 16 (19, 20).
  uint32_t lg = 0;
  while ((1ull << lg) < (n ? n : 1)) lg++;
  int target = (int)lg - 4;
  if (target < 4) target = 4;
  if (target > 16) target = 16;
  auto top_bits = [&](int cc) {
    const int w = ((int)scalar_bits + 1 + cc - 1) / cc;
    return (int)scalar_bits + 1 - (w - 1) * cc;
  };
  int c = target;
  for (int d = 0; d <= 16; d++) {
    bool found = false;
    for (int cand : {target + d, target - d}) {
      if (cand < 4 || cand > 16) continue;
      if (top_bits(cand) >= (cand < 7 ? cand : 7)) {
        c = cand;
        found = true;
        break;
      }
    }
    if (found) break;
  }
  if (precomp && !force_c) {
    // Resident window tables of 2^18 terms and more: c = 17.  Over 255-bit scalars (BLS12-381) with the scalars above (r - 1) / 2
    // negated (negate_high below) that is 15 windows instead of 16 -- 6 % fewer additions, 1/16 less table memory, 2^16 buckets;
    // over 254-bit scalars (BN254) 15 windows of 17 bits hold the value and its carry bit as they are.
    // History: round 3 measured +1 % at 2^20 (the tails of twice the buckets ate the rest); with the batched tails of round 4
    // 23.77 -> 23.27 ms per proof in flight (profiles/r04_c17_ab.txt) and c = 17 became the choice from 2^20 terms on, BLS12-381
    // only (smaller vectors: "fewer than 256 terms per bucket: flush divergence"; BN254: -0.8 % in round 3).  Round 6 made the
    // flush 14 stores and the tails plain sums, and run I (profiles/r06_runI_c17_small_bn254.txt, same box, interleaved) reads:
    // BN254 2^20 13.61 -> 13.04-13.13 ms per proof (-3.8 %), BLS12-381 2^19 10.77 -> 10.48 ms, 2^18 x 8 in flight 5.67-5.74 ->
    // 5.59-5.61 ms, the 2^19-term shards of a rank of the sharded 2^22 proof 13.5-13.6 -> 12.9-13.6 ms.
    if (c == 16 && (scalar_bits == 255 || scalar_bits == 254) && n >= (1ull << 18)) c = 17;
    // ... and c = 20 from 2^22 terms on: 13 windows instead of 15 (-13 % additions, tables 13/15 the size) against the fixed cost of
    // 2^19 buckets per MSM (fills, sorts over 8x the keys, two general additions per bucket in the row / column sums: ~1 ms per MSM,
    // whatever its length).  Run B: at 2^20 constraints it loses (23.1 against 22.4 ms per proof; the H MSM alone, 2^21 terms: 22.8);
    // run K, 2^22 constraints on one GPU, same box, interleaved: 81.4 -> 76.7 ms per proof (-5.8 %, 54.7 M constraints/s), c = 19:
    // 79.3 (profiles/r06_runK_large_key_windows.txt).
    if (c == 17 && n >= (1ull << 22)) c = 20;
    // tuning knob for resident keys (window tables): policy MSM_C=<bits>, applied when the key is loaded
    if (pref_c >= 4 && pref_c <= 24 && n >= 1024) c = pref_c;
  }
  if (force_c) c = force_c;     // an MSM over window tables must use the window size the tables were built for
  p.c = (uint32_t)c;
  // one extra bit so that the top window never produces a carry
  p.windows = (scalar_bits + 1 + p.c - 1) / p.c;
  if (precomp && (scalar_bits + p.c - 1) / p.c < p.windows) {
    p.negate_high = true;
    p.windows = (scalar_bits + p.c - 1) / p.c;
  }
  p.buckets_per_window = 1u << (p.c - 1);
  p.wstride = precomp ? (force_stride ? force_stride : 1u) : p.windows;
  if (p.wstride > p.windows) p.wstride = p.windows;
  p.table_windows = precomp ? (p.windows + p.wstride - 1) / p.wstride : 0u;
  p.key_windows = p.wstride;
  p.total_buckets = p.key_windows * p.buckets_per_window;
  return p;
}

// `precomp` argument of the digit kernels: bit 0 = window tables, bit 1 = MsmPlan::negate_high, bits 8.. = wstride
// (window w -> bucket set w % wstride, table row block w / wstride)
static inline int msm_digit_flags(const MsmPlan& p) { return (p.precomp ? 1 : 0) | (p.negate_high ? 2 : 0) | (int)(p.wstride << 8); }
constexpr int MSM_DIGITS_PRECOMP = 1, MSM_DIGITS_NEGATE_HIGH = 2;

// k (canonical) > (r - 1) / 2: replace it by r - k and report it
template <class Fr>
ARK_D bool msm_negate_if_high(Fr& k) {
  using P = typename Fr::Params;
  Fr nk;
  uint32_t borrow = 0;
#pragma unroll
  for (int j = 0; j < Fr::N; j++) {
    const uint64_t t = (uint64_t)P::mod(j) - k.l[j] - borrow;
    nk.l[j] = (uint32_t)t;
    borrow = (uint32_t)(t >> 32) & 1u;
  }
  bool greater = false, decided = false;          // k > nk ?
#pragma unroll
  for (int j = Fr::N - 1; j >= 0; j--) {
    if (!decided && k.l[j] != nk.l[j]) {
      greater = k.l[j] > nk.l[j];
      decided = true;
    }
  }
  if (greater) k = nk;
  return greater;
}


// Wave-aggregated "fetch-and-add 1" on counter[key]: returns this lane's slot, i.e. what
// atomicAdd(&counter[key], 1) would have returned, but lanes of a wave that share a key are served by ONE atomic.
// Skewed scalar distributions (boolean witnesses, the all-equal DummyCircuit, a short top window) put most lanes
// of a wave on the same bucket; un-aggregated, those same-address L2 atomics serialise (a degenerate window cost
// ~10 ms per sort at n = 2^20 in round 1).  MSM_AGG_ROUNDS leader rounds peel off the largest groups with
// __ballot/__shfl, the remaining lanes fall back to individual atomics.  Must be called by every lane of the
// wave (inactive lanes pass valid = false).
constexpr int MSM_AGG_ROUNDS = 3;
ARK_D uint32_t wave_agg_inc(uint32_t* counter, uint32_t key, bool valid) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t slot = 0;
  bool pending = valid;
  for (int round = 0; round < MSM_AGG_ROUNDS; round++) {
    const unsigned long long live = __ballot(pending);
    if (live == 0) break;                                    // wave-uniform
    const int leader = __ffsll((long long)live) - 1;
    const uint32_t lkey = (uint32_t)__shfl((int)key, leader, 64);
    const bool mine = pending && key == lkey;
    const unsigned long long group = __ballot(mine);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&counter[lkey], (uint32_t)__popcll(group));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (mine) {
      slot = base + (uint32_t)__popcll(group & ((1ull << lane) - 1ull));
      pending = false;
    }
  }
  if (pending) slot = atomicAdd(&counter[key], 1u);
  return slot;
}

// ---- K2: signed window digits + histogram ----------------------------------------------------------
template <class Fr>
__global__ void __launch_bounds__(MSM_THREADS)
msm_digits_kernel(const Fr* __restrict__ scalars, uint32_t n, int mont, uint32_t c, uint32_t windows, int precomp_flags,
                  uint32_t table_stride,
                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < n;                     // no early return: the wave-level primitives need every lane
  Fr k = in_range ? scalars[i] : Fr::zero();
  if (mont) k = Fr::from_mont(k);
  const uint32_t wstride = (uint32_t)precomp_flags >> 8;
  const uint32_t flip = ((precomp_flags & MSM_DIGITS_NEGATE_HIGH) && msm_negate_if_high(k)) ? 1u : 0u;
  const uint32_t B = 1u << (c - 1);
  const uint32_t full = 1u << c;
  uint32_t carry = 0;
  uint32_t q = 0, j = 0;                              // w = wstride * q + j
  for (uint32_t w = 0; w < windows; w++) {
    const uint32_t bit = w * c;
    const uint32_t limb = bit >> 5, off = bit & 31;
    uint32_t d = 0;
    if (limb < (uint32_t)Fr::N) {
      uint64_t v = k.l[limb];
      if (limb + 1 < (uint32_t)Fr::N) v |= (uint64_t)k.l[limb + 1] << 32;
      d = (uint32_t)(v >> off) & (full - 1);
    }
    d += carry;
    uint32_t neg = 0;
    if (d > B) {
      d = full - d;
      neg = 1;
      carry = 1;
    } else {
      carry = 0;
    }
    const uint64_t e = (uint64_t)w * n + i;
    const bool valid = in_range && d != 0;
    // bucket set j, table row block q (window tables: table_stride rows per block; none: table_stride = 0)
    const uint32_t key = valid ? (j * B + d - 1) : MSM_INVALID;
    if (in_range) {
      keys[e] = key;
      vals[e] = valid ? ((q * table_stride + i) | ((neg ^ flip) << 31)) : 0u;
    }
    (void)wave_agg_inc(counts, valid ? key : 0u, valid);
    if (++j == wstride) {
      j = 0;
      q++;
    }
  }
}

// ---- exclusive scan of `m` u32 counters by one workgroup ---------------------------------------------
// Tiles of 1024 x SCAN_EPT counters: coalesced 16-byte loads, a shuffle scan per wave, the 16 wave totals through
// LDS, a running carry across tiles.  (The first version gave every thread one contiguous chunk: uncoalesced, and
// 0.17 ms for the 2^17..2^18-entry tables of the two-level sort.)
constexpr uint32_t SCAN_THREADS = 1024;
constexpr uint32_t SCAN_EPT = 16;
// Tables of more than SCAN_SPLIT_MIN counters (2^19 buckets at c = 20; the first sort level of long vectors) are cut into spans of
// whole tiles, one workgroup each: this kernel scans its span from zero and leaves the span's total in span_total[blockIdx.x],
// scan_add_spans_kernel adds the totals of the spans in front (round 6: one workgroup took 144 us per 2^19-counter table, 0.5 ms
// per proof at c = 20 -- profiles/r06_runM_c20_fixed_cost.txt).  span = 0: one workgroup, the whole table.
static __global__ void __launch_bounds__(SCAN_THREADS)
scan_exclusive_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t m,
                      uint32_t* __restrict__ total, uint32_t span, uint32_t* __restrict__ span_total) {
  __shared__ uint32_t wave_tot[SCAN_THREADS / 64];
  __shared__ uint32_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr uint32_t TILE = SCAN_THREADS * SCAN_EPT;
  const uint64_t lo = span ? (uint64_t)blockIdx.x * span : 0;
  const uint64_t hi = span ? ((lo + span < m) ? lo + span : (uint64_t)m) : (uint64_t)m;
  for (uint64_t base = lo; base < hi; base += TILE) {
    // thread t owns SCAN_EPT consecutive counters: four aligned 16-byte pieces
    uint32_t v[SCAN_EPT];
    const uint64_t first = base + (uint64_t)tid * SCAN_EPT;
#pragma unroll
    for (uint32_t q = 0; q < SCAN_EPT / 4; q++) {
      const uint64_t i = first + 4 * q;
      if (i + 3 < m) {
        const uint4 t = *reinterpret_cast<const uint4*>(in + i);
        v[4 * q + 0] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
      } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) v[4 * q + k] = (i + k < m) ? in[i + k] : 0u;
      }
    }
    uint32_t s = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_EPT; k++) s += v[k];
    // inclusive scan of s across the wave
    uint32_t inc = s;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tile_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCAN_THREADS / 64; w++) {
      const uint32_t t = wave_tot[w];
      if (w < wave) wbase += t;
      tile_total += t;
    }
    uint32_t run = carry_s + wbase + (inc - s);
#pragma unroll
    for (uint32_t q = 0; q < SCAN_EPT / 4; q++) {
      const uint64_t i = first + 4 * q;
      uint32_t o[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        o[k] = run;
        run += v[4 * q + k];
      }
      if (i + 3 < m) {
        *reinterpret_cast<uint4*>(out + i) = make_uint4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
          if (i + k < m) out[i + k] = o[k];
      }
    }
    __syncthreads();                     // everyone has read carry_s and wave_tot
    if (tid == 0) carry_s += tile_total;
    __syncthreads();
  }
  if (tid == 0) {
    if (span) span_total[blockIdx.x] = carry_s;
    else if (total) *total = carry_s;
  }
}
#if defined(ARK_EMUL)
constexpr uint32_t SCAN_SPLIT_MIN = SCAN_THREADS * SCAN_EPT;           // (the CPU tier reaches 2^16 buckets at most: split from one tile on, so that it runs this path)
#else
constexpr uint32_t SCAN_SPLIT_MIN = 4 * SCAN_THREADS * SCAN_EPT;       // 65 536 counters
#endif
constexpr uint32_t SCAN_MAX_SPANS = 256;
static __global__ void __launch_bounds__(SCAN_THREADS)
scan_add_spans_kernel(uint32_t* __restrict__ out, uint32_t m, uint32_t* __restrict__ total, uint32_t span,
                      const uint32_t* __restrict__ span_total, uint32_t spans) {
  __shared__ uint32_t off_s;
  if (threadIdx.x == 0) {
    uint32_t o = 0;
    for (uint32_t g = 0; g < blockIdx.x; g++) o += span_total[g];
    off_s = o;
    if (blockIdx.x + 1 == spans && total) *total = o + span_total[blockIdx.x];
  }
  __syncthreads();
  const uint32_t off = off_s;
  if (off == 0) return;
  const uint64_t lo = (uint64_t)blockIdx.x * span;
  const uint64_t hi = (lo + span < m) ? lo + span : (uint64_t)m;
  for (uint64_t i = lo + 4ull * threadIdx.x; i < hi; i += 4ull * SCAN_THREADS) {       // span and lo are multiples of the tile: 16-byte aligned
    if (i + 3 < hi) {
      uint4 t = *reinterpret_cast<uint4*>(out + i);
      t.x += off; t.y += off; t.z += off; t.w += off;
      *reinterpret_cast<uint4*>(out + i) = t;
    } else {
      for (uint64_t k = i; k < hi; k++) out[k] += off;
    }
  }
}
// exclusive scan of m counters on `stream`; *total (may be null) = their sum.  aux: SCAN_MAX_SPANS words of scratch.
static inline void scan_exclusive(hipStream_t stream, const uint32_t* in, uint32_t* out, uint32_t m, uint32_t* total, DevBuf& aux) {
  constexpr uint32_t TILE = SCAN_THREADS * SCAN_EPT;
  if (m <= SCAN_SPLIT_MIN) {
    ARK_LAUNCH(scan_exclusive_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, in, out, m, total, 0u, (uint32_t*)nullptr);
    ARK_CHECK_LAUNCH();
    return;
  }
  uint32_t tiles = (m + TILE - 1) / TILE;
  uint32_t tiles_per_span = (tiles + SCAN_MAX_SPANS - 1) / SCAN_MAX_SPANS;
  if (tiles_per_span < 1) tiles_per_span = 1;
  const uint32_t span = tiles_per_span * TILE;
  const uint32_t spans = (m + span - 1) / span;
  aux.ensure((size_t)SCAN_MAX_SPANS * 4);
  ARK_LAUNCH(scan_exclusive_kernel, dim3(spans), dim3(SCAN_THREADS), 0, stream, in, out, m, (uint32_t*)nullptr, span, aux.as<uint32_t>());
  ARK_CHECK_LAUNCH();
  ARK_LAUNCH(scan_add_spans_kernel, dim3(spans), dim3(SCAN_THREADS), 0, stream, out, m, total, span, (const uint32_t*)aux.as<uint32_t>(), spans);
  ARK_CHECK_LAUNCH();
}

// ---- K3: scatter (counting sort) ----------------------------------------------------------------------
static __global__ void __launch_bounds__(MSM_THREADS)
msm_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t entries,
                   const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                   uint32_t* __restrict__ sorted_keys, uint32_t* __restrict__ sorted_vals) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t key = (e < entries) ? keys[e] : MSM_INVALID;
  const bool valid = key != MSM_INVALID;
  const uint32_t slot = wave_agg_inc(cursor, valid ? key : 0u, valid);
  if (valid) {
    const uint32_t pos = offsets[key] + slot;
    sorted_keys[pos] = key;
    sorted_vals[pos] = vals[e];
  }
}

// ---- two-level counting sort (the default) -----------------------------------------------------------------
// The one-pass counting sort above pays one device-scope atomic per entry in the histogram, one returning atomic
// per entry in the scatter, and scatters 4-byte writes all over a 64-130 MB output: 4.6 ms per 2^20 proof (14%).
// Two levels keep every hot counter in LDS and every write run inside one workgroup:
//   level 1 (bins = key >> 8): each workgroup owns a slice of the scalars, counts its entries per bin in LDS
//     (sort_hi_hist), a scan over the bin-major [bin][workgroup] table assigns it a private range per bin, and
//     sort_hi_scatter re-derives the digits and writes (key, value) pairs into those ranges -- no global atomic.
//   level 2 (key & 255): the bin-grouped pairs are cut into tiles of SORT_TILE entries; a tile covers one or two
//     bins, counts its <= 256 keys per bin in LDS, adds those counts to the bucket histogram (sort_lo_hist: one
//     global atomic per (tile, key) instead of one per entry), and after the usual bucket scan reserves a range
//     per (tile, key) and writes its entries there (sort_lo_scatter): runs of ~16 entries per key.
// The order of entries inside a bucket is unspecified (as before); nothing downstream depends on it.
constexpr uint32_t SORT_LO_BITS = 8;
constexpr uint32_t SORT_LO = 1u << SORT_LO_BITS;
constexpr uint32_t SORT_MAX_BINS = 4096;       // LDS counters of level 1
constexpr uint32_t SORT_SPT = 4;               // scalars per thread in level 1
constexpr uint32_t SORT_HI_THREADS = 1024;     // level-1 workgroup: 4096 scalars, [bin][workgroup] table stays small
constexpr uint32_t SORT_EPT = 16;              // entries per thread in level 2
constexpr uint32_t SORT_TILE = MSM_THREADS * SORT_EPT;

// calls fn(w, key, val) for every non-zero signed digit of scalar i
template <class Fr, class Fn>
ARK_D void msm_for_each_digit(Fr k, int mont, uint32_t i, uint32_t n, uint32_t c, uint32_t windows, int precomp_flags,
                              uint32_t table_stride, Fn&& fn) {
  if (mont) k = Fr::from_mont(k);
  const uint32_t wstride = (uint32_t)precomp_flags >> 8;
  const uint32_t flip = ((precomp_flags & MSM_DIGITS_NEGATE_HIGH) && msm_negate_if_high(k)) ? 1u : 0u;
  const uint32_t B = 1u << (c - 1);
  const uint32_t full = 1u << c;
  uint32_t carry = 0;
  uint32_t q = 0, j = 0;                              // w = wstride * q + j: bucket set j, table row block q
  for (uint32_t w = 0; w < windows; w++) {
    const uint32_t bit = w * c;
    const uint32_t limb = bit >> 5, off = bit & 31;
    uint32_t d = 0;
    if (limb < (uint32_t)Fr::N) {
      uint64_t v = k.l[limb];
      if (limb + 1 < (uint32_t)Fr::N) v |= (uint64_t)k.l[limb + 1] << 32;
      d = (uint32_t)(v >> off) & (full - 1);
    }
    d += carry;
    uint32_t neg = 0;
    if (d > B) {
      d = full - d;
      neg = 1;
      carry = 1;
    } else {
      carry = 0;
    }
    if (d != 0) {
      const uint32_t key = j * B + d - 1;
      const uint32_t val = (q * table_stride + i) | ((neg ^ flip) << 31);
      fn(w, key, val);
    }
    if (++j == wstride) {
      j = 0;
      q++;
    }
  }
}

template <class Fr>
__global__ void __launch_bounds__(SORT_HI_THREADS)
sort_hi_hist_kernel(const Fr* __restrict__ scalars, uint32_t n, int mont, uint32_t c, uint32_t windows, int precomp,
                    uint32_t table_stride, uint32_t bins, uint32_t* __restrict__ hist /* [bin][gridDim.x] */) {
  __shared__ uint32_t lds[SORT_MAX_BINS];
  for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) lds[b] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (SORT_HI_THREADS * SORT_SPT);
  for (uint32_t j = 0; j < SORT_SPT; j++) {
    const uint32_t i = base + j * SORT_HI_THREADS + threadIdx.x;
    if (i < n) {
      msm_for_each_digit<Fr>(scalars[i], mont, i, n, c, windows, precomp, table_stride,
                             [&](uint32_t, uint32_t key, uint32_t) { atomicAdd(&lds[key >> SORT_LO_BITS], 1u); });
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) hist[(size_t)b * gridDim.x + blockIdx.x] = lds[b];
}

template <class Fr>
__global__ void __launch_bounds__(SORT_HI_THREADS)
sort_hi_scatter_kernel(const Fr* __restrict__ scalars, uint32_t n, int mont, uint32_t c, uint32_t windows,
                       int precomp, uint32_t table_stride, uint32_t bins,
                       const uint32_t* __restrict__ hist_scanned /* [bin][gridDim.x], exclusive */,
                       uint2* __restrict__ tmp /* (key, value) pairs */) {
  __shared__ uint32_t lds[SORT_MAX_BINS];
  for (uint32_t b = threadIdx.x; b < bins; b += blockDim.x) lds[b] = hist_scanned[(size_t)b * gridDim.x + blockIdx.x];
  __syncthreads();
  const uint32_t base = blockIdx.x * (SORT_HI_THREADS * SORT_SPT);
  for (uint32_t j = 0; j < SORT_SPT; j++) {
    const uint32_t i = base + j * SORT_HI_THREADS + threadIdx.x;
    if (i < n) {
      msm_for_each_digit<Fr>(scalars[i], mont, i, n, c, windows, precomp, table_stride,
                             [&](uint32_t, uint32_t key, uint32_t val) {
                               const uint32_t pos = atomicAdd(&lds[key >> SORT_LO_BITS], 1u);
                               tmp[pos] = make_uint2(key, val);
                             });
    }
  }
}

// level 2, shared by the histogram (SCATTER = false) and the scatter pass
template <bool SCATTER>
static __global__ void __launch_bounds__(MSM_THREADS)
sort_lo_kernel(const uint2* __restrict__ tmp /* (key, value) pairs grouped by bin */,
               const uint32_t* __restrict__ total_ptr, uint32_t* __restrict__ counts,
               const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
               uint32_t* __restrict__ sorted_keys, uint32_t* __restrict__ sorted_vals) {
  __shared__ uint32_t cnt[SORT_LO];
  __shared__ uint32_t base_s[SORT_LO];
  const uint32_t total = *total_ptr;
  const uint64_t t0 = (uint64_t)blockIdx.x * SORT_TILE;
  if (t0 >= total) return;                                   // whole workgroup
  const uint32_t first = (uint32_t)t0;
  const uint32_t last = (first + SORT_TILE < total) ? first + SORT_TILE : total;     // exclusive
  uint32_t key[SORT_EPT], val[SORT_EPT];
#pragma unroll
  for (uint32_t j = 0; j < SORT_EPT; j++) {
    const uint32_t e = first + j * MSM_THREADS + threadIdx.x;
    if (SCATTER) {
      const uint2 t = (e < last) ? tmp[e] : make_uint2(MSM_INVALID, 0u);
      key[j] = t.x;
      val[j] = t.y;
    } else {
      key[j] = (e < last) ? tmp[e].x : MSM_INVALID;
    }
  }
  const uint32_t h0 = tmp[first].x >> SORT_LO_BITS, h1 = tmp[last - 1].x >> SORT_LO_BITS;
  for (uint32_t hb = h0; hb <= h1; hb++) {                   // entries are grouped by bin: usually 1-2 rounds
    for (uint32_t t = threadIdx.x; t < SORT_LO; t += blockDim.x) cnt[t] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < SORT_EPT; j++) {
      if (key[j] != MSM_INVALID && (key[j] >> SORT_LO_BITS) == hb) atomicAdd(&cnt[key[j] & (SORT_LO - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < SORT_LO; t += blockDim.x) {
      const uint32_t k = (hb << SORT_LO_BITS) | t;
      const uint32_t n_k = cnt[t];
      if (SCATTER) {
        base_s[t] = n_k ? offsets[k] + atomicAdd(&cursor[k], n_k) : 0u;
      } else if (n_k) {
        atomicAdd(&counts[k], n_k);
      }
    }
    if (SCATTER) {
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < SORT_EPT; j++) {
        if (key[j] != MSM_INVALID && (key[j] >> SORT_LO_BITS) == hb) {
          const uint32_t pos = atomicAdd(&base_s[key[j] & (SORT_LO - 1)], 1u);
          sorted_keys[pos] = key[j];
          sorted_vals[pos] = val[j];
        }
      }
    }
    __syncthreads();
  }
}

// ---- K4: bucket accumulation -----------------------------------------------------------------------------
template <class F>
struct SegPartial {
  XYZZ<F> pt;
};

template <class F>
ARK_D void msm_flush_run(uint32_t key, const XYZZ<F>& acc, bool first_run, uint32_t run_start, uint32_t run_end,
                         uint32_t seg, const uint32_t* offsets, const uint32_t* counts, XYZZ<F>* buckets,
                         XYZZ<F>* head, uint32_t* head_key, XYZZ<F>* tail, uint32_t* tail_key) {
  const uint32_t o = offsets[key], cnt = counts[key];
  const bool complete = (run_start == o) && (run_end == o + cnt);
  if (complete) {
    buckets[key] = acc;
  } else if (first_run) {
    head[seg] = acc;
    head_key[seg] = key;
  } else {
    tail[seg] = acc;
    tail_key[seg] = key;
  }
}

}  // namespace ark355
#include "tails28_impl.cuh"     // (includes msm28_impl.cuh: the 28-bit accumulation kernels and their bucket slots)
namespace ark355 {

template <class F, bool NI>
__global__ void __launch_bounds__(MSM_THREADS)
msm_accumulate_kernel(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                      const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                      const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                      XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ head, uint32_t* __restrict__ head_key,
                      XYZZ<F>* __restrict__ tail, uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = *total_ptr;
  const uint32_t MSM_SEG = seg_len;
  const uint64_t start64 = (uint64_t)seg * seg_len;
  if (start64 >= total) return;
  const uint32_t start = (uint32_t)start64;
  const uint32_t end = (start + MSM_SEG < total) ? start + MSM_SEG : total;
  uint32_t cur_key = sorted_keys[start];
  uint32_t run_start = start;
  bool first_run = true;
  XYZZ<F> acc = XYZZ<F>::inf();
  // Software prefetch: the next base (a random 96/192-byte gather from the window tables) is in flight while the
  // current mixed addition runs.  It is held in explicit 16-byte registers: a struct copy here was lowered by
  // hipcc to scratch-to-scratch copies with a vmcnt(0) after every load (seen in the round-1 ISA).
  constexpr int Q = sizeof(Affine<F>) / 16;
  // G2 lives at the edge of the 512-register file: holding the prefetched point (48 more registers) costs more
  // in spills than the gather latency it hides (one ~2 us gather per ~60 us mixed addition)
  constexpr bool PREFETCH = ARK_G1_PREFETCH && sizeof(F) <= 64;
  uint4 nx[Q];
  uint32_t v_next = sorted_vals[start];
  if constexpr (PREFETCH) {
    const uint4* src = reinterpret_cast<const uint4*>(bases + (v_next & ARK_TBL_MASK));
#pragma unroll
    for (int k = 0; k < Q; k++) nx[k] = src[k];
  }
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = sorted_keys[e];
    uint32_t v = v_next;
    Affine<F> p;
    if constexpr (PREFETCH) {
      uint32_t* d = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        d[4 * k + 0] = nx[k].x;
        d[4 * k + 1] = nx[k].y;
        d[4 * k + 2] = nx[k].z;
        d[4 * k + 3] = nx[k].w;
      }
      const uint32_t en = (e + 1 < end) ? e + 1 : e;     // clamp: the last iteration re-reads its own entry
      v_next = sorted_vals[en];
      const uint4* src = reinterpret_cast<const uint4*>(bases + (v_next & ARK_TBL_MASK));
#pragma unroll
      for (int k = 0; k < Q; k++) nx[k] = src[k];
    } else {
      v = sorted_vals[e];
      p = bases[v & ARK_TBL_MASK];
    }
    if (key != cur_key) {
      msm_flush_run<F>(cur_key, acc, first_run, run_start, e, seg, offsets, counts, buckets, head, head_key, tail,
                       tail_key);
      cur_key = key;
      run_start = e;
      first_run = false;
      acc = XYZZ<F>::inf();
    }
    if (v >> 31) p.y = F::neg(p.y);
    if constexpr (NI) xyzz_madd_ni(acc, p);
    else xyzz_madd(acc, p);
  }
  msm_flush_run<F>(cur_key, acc, first_run, run_start, end, seg, offsets, counts, buckets, head, head_key, tail,
                   tail_key);
}

// ---- K4 for G2: lane-split accumulation -------------------------------------------------------------------------
// Two lanes per segment: even lanes carry the c0 components, odd lanes the c1 components of every Fq2 value
// (field.cuh Fp2L).  Same segment/run logic as msm_accumulate_kernel; loads and stores touch this lane's half
// of each Fq2 coordinate.
template <class P>
struct is_fp2 {
  static constexpr bool value = false;
};
template <class P>
struct is_fp2<Fp2<P>> {
  static constexpr bool value = true;
};

template <class P>
ARK_D void g2l_store(XYZZ<Fp2<P>>* dst, const XYZZ<Fp2L<P>>& v, uint32_t par) {
  Fp<P>* d = reinterpret_cast<Fp<P>*>(dst);
  d[0 + par] = v.x.c;
  d[2 + par] = v.y.c;
  d[4 + par] = v.zz.c;
  d[6 + par] = v.zzz.c;
}

#ifndef ARK_G2L_PREFETCH
#define ARK_G2L_PREFETCH 0
#endif
// Two waves per SIMD need <= 256 registers per lane.  With the register prefetch of the next base the kernel
// lands on 251-256, and that build ran at 11.2 ms on some MI355X boxes but 17-24 ms on others (same binary);
// without it there is head-room below the cliff.
template <class P>
__global__ void __launch_bounds__(MSM_THREADS, 2)
msm_accumulate_g2l_kernel(const Affine<Fp2<P>>* __restrict__ bases, const uint32_t* __restrict__ sorted_keys,
                          const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ total_ptr,
                          const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                          XYZZ<Fp2<P>>* __restrict__ buckets, XYZZ<Fp2<P>>* __restrict__ head,
                          uint32_t* __restrict__ head_key, XYZZ<Fp2<P>>* __restrict__ tail,
                          uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using FL = Fp2L<P>;
  using Fq = Fp<P>;
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
  XYZZ<FL> acc = XYZZ<FL>::inf();
  auto flush = [&](uint32_t key, uint32_t run_end) {
    const uint32_t o = offsets[key], cnt = counts[key];
    const bool complete = (run_start == o) && (run_end == o + cnt);
    if (complete) {
      g2l_store<P>(&buckets[key], acc, par);
    } else if (first_run) {
      g2l_store<P>(&head[seg], acc, par);
      if (par == 0) head_key[seg] = key;
    } else {
      g2l_store<P>(&tail[seg], acc, par);
      if (par == 0) tail_key[seg] = key;
    }
  };
  // this lane's halves of a base: x.c[par] and y.c[par], 3 (BLS) / 2 (BN) x 16 B each; optionally prefetched one
  // iteration ahead into registers
  constexpr int Q = sizeof(Fq) / 16;
  constexpr bool PF = ARK_G2L_PREFETCH != 0;
  uint4 nx[2 * Q];
  auto fetch = [&](uint32_t v) {
    const Fq* b = reinterpret_cast<const Fq*>(bases + (v & ARK_TBL_MASK));
    const uint4* sx = reinterpret_cast<const uint4*>(b + par);
    const uint4* sy = reinterpret_cast<const uint4*>(b + 2 + par);
#pragma unroll
    for (int k = 0; k < Q; k++) {
      nx[k] = sx[k];
      nx[Q + k] = sy[k];
    }
  };
  uint32_t v_next = sorted_vals[start];
  if constexpr (PF) fetch(v_next);
  for (uint32_t e = start; e < end; e++) {
    const uint32_t key = sorted_keys[e];
    uint32_t v = v_next;
    if constexpr (!PF) {
      v = sorted_vals[e];
      fetch(v);
    }
    Affine<FL> p;
    {
      uint32_t* dx = p.x.c.l;
      uint32_t* dy = p.y.c.l;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        dx[4 * k + 0] = nx[k].x;
        dx[4 * k + 1] = nx[k].y;
        dx[4 * k + 2] = nx[k].z;
        dx[4 * k + 3] = nx[k].w;
        dy[4 * k + 0] = nx[Q + k].x;
        dy[4 * k + 1] = nx[Q + k].y;
        dy[4 * k + 2] = nx[Q + k].z;
        dy[4 * k + 3] = nx[Q + k].w;
      }
    }
    if constexpr (PF) {
      const uint32_t en = (e + 1 < end) ? e + 1 : e;
      v_next = sorted_vals[en];
      fetch(v_next);
    }
    if (key != cur_key) {
      flush(cur_key, e);
      cur_key = key;
      run_start = e;
      first_run = false;
      acc = XYZZ<FL>::inf();
    }
    if (v >> 31) p.y = FL::neg(p.y);
    xyzz_madd(acc, p);
  }
  flush(cur_key, end);
}

// The latency-bound tail kernels (merge, bucket reduction) keep to 256 registers per lane: two waves per SIMD, so that a
// wave of theirs and a wave of an accumulation kernel of another proof in flight can share a SIMD.  (Left to itself the
// Fq2 group addition takes 456 registers -- one wave per SIMD -- and the CUs a tail kernel sits on stop accumulating.)
#ifndef MSM_TAIL_WAVES
#define MSM_TAIL_WAVES 2
#endif
// buckets whose entries straddle segment boundaries: add their partial runs.  Buckets spread over more than
// heavy_span segments (skewed scalars: boolean witnesses, the all-equal DummyCircuit, a short top window) are
// only recorded here and summed by a whole workgroup each in msm_merge_heavy_kernel.  heavy_span is MSM_HEAVY_SPAN, or
// twice the AVERAGE span when that is larger: in a 2^24-term MSM every bucket of a uniform input spans
// ~130 segments, and sending all 32 768 of them through the 128 workgroups of the heavy kernel took 46 ms where one lane
// per bucket takes 4 (profiles/r02_msm_microbench.txt).
#ifndef ARK_MSM_HEAVY_SPAN
#define ARK_MSM_HEAVY_SPAN 48   // tests shrink it so that tiny cases take the heavy path
#endif
constexpr uint32_t MSM_HEAVY_SPAN = ARK_MSM_HEAVY_SPAN;
#ifndef ARK_MSM_HEAVY_GRID
// Workgroups of the heavy-bucket merge (each loops over the heavy list).  128 is plenty -- heavy buckets are few by
// definition -- and 1024 mostly-empty workgroups queued behind the accumulation kernels of the other proofs in flight
// held the reduction stream for ~0.6 ms per MSM (round-2 timeline).
#define ARK_MSM_HEAVY_GRID 128u
#endif
template <class F>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_merge_kernel(uint32_t total_buckets, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                 XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ head, const uint32_t* __restrict__ head_key,
                 const XYZZ<F>* __restrict__ tail, const uint32_t* __restrict__ tail_key,
                 uint32_t* __restrict__ heavy_count, uint32_t* __restrict__ heavy_list, uint32_t seg_len,
                 uint32_t heavy_span) {
  const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= total_buckets) return;
  const uint32_t cnt = counts[key];
  if (cnt == 0) return;
  const uint32_t o = offsets[key];
  const uint32_t t0 = o / seg_len, t1 = (o + cnt - 1) / seg_len;
  if (t0 == t1) return;   // the single run was complete and already written
  if (t1 - t0 > heavy_span) {
    heavy_list[atomicAdd(heavy_count, 1u)] = key;
    return;
  }
  XYZZ<F> sum = XYZZ<F>::inf();
  for (uint32_t t = t0; t <= t1; t++) {
    if (head_key[t] == key) sum = xyzz_add(sum, head[t]);
    if (tail_key[t] == key) sum = xyzz_add(sum, tail[t]);
  }
  buckets[key] = sum;
}

// ---- wave-level reduction of XYZZ points with __shfl_xor ---------------------------------------------------
template <class F>
ARK_D XYZZ<F> xyzz_shfl_xor(const XYZZ<F>& p, int mask) {
  XYZZ<F> r;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&r);
  constexpr int WORDS = sizeof(XYZZ<F>) / 4;
#pragma unroll
  for (int i = 0; i < WORDS; i++) dst[i] = (uint32_t)__shfl_xor((int)src[i], mask, 64);
  return r;
}

template <class F>
ARK_D XYZZ<F> wave_reduce_sum(XYZZ<F> v) {
  for (int mask = 32; mask >= 1; mask >>= 1) {
    XYZZ<F> o = xyzz_shfl_xor(v, mask);
    v = xyzz_add(v, o);
  }
  return v;
}

// one workgroup per heavy bucket: lanes stride over the bucket's segments, wave butterfly, 4 waves through LDS
template <class F>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_merge_heavy_kernel(const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list,
                       const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                       XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ head,
                       const uint32_t* __restrict__ head_key, const XYZZ<F>* __restrict__ tail,
                       const uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  __shared__ uint32_t wave_out[(MSM_THREADS / 64) * (sizeof(XYZZ<F>) / 4)];
  constexpr int WORDS = sizeof(XYZZ<F>) / 4;
  const uint32_t nheavy = *heavy_count;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t key = heavy_list[h];
    const uint32_t o = offsets[key], cnt = counts[key];
    const uint32_t t0 = o / seg_len, t1 = (o + cnt - 1) / seg_len;
    XYZZ<F> sum = XYZZ<F>::inf();
    for (uint32_t t = t0 + threadIdx.x; t <= t1; t += blockDim.x) {
      if (head_key[t] == key) sum = xyzz_add(sum, head[t]);
      if (tail_key[t] == key) sum = xyzz_add(sum, tail[t]);
    }
    sum = wave_reduce_sum(sum);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&sum);
      for (int i = 0; i < WORDS; i++) wave_out[wave * WORDS + i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      XYZZ<F> tot = XYZZ<F>::inf();
      for (uint32_t v = 0; v < blockDim.x / 64; v++) {
        XYZZ<F> t;
        uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
        for (int i = 0; i < WORDS; i++) dst[i] = wave_out[v * WORDS + i];
        tot = xyzz_add(tot, t);
      }
      buckets[key] = tot;
    }
    __syncthreads();
  }
}

// ---- K5: bucket reduction: per window sum_{b} (b+1) * bucket[b] -----------------------------------------------
// grid.x = blocks per window, grid.y = windows.  Output: partials[window * gridDim.x + blockIdx.x].
template <class F>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_reduce_kernel(const XYZZ<F>* __restrict__ buckets, uint32_t buckets_per_window, XYZZ<F>* __restrict__ partials) {
  __shared__ uint32_t wave_out[(MSM_THREADS / 64) * (sizeof(XYZZ<F>) / 4)];
  const uint32_t w = blockIdx.y;
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;   // chunk index inside the window
  const uint32_t first = chunk * MSM_RED_K;
  XYZZ<F> contrib = XYZZ<F>::inf();
  if (first < buckets_per_window) {
    const uint32_t last = (first + MSM_RED_K < buckets_per_window) ? first + MSM_RED_K : buckets_per_window;
    const XYZZ<F>* wb = buckets + (uint64_t)w * buckets_per_window;
    XYZZ<F> running = XYZZ<F>::inf();
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t b = last; b-- > first;) {
      running = xyzz_add(running, wb[b]);
      acc = xyzz_add(acc, running);
    }
    // acc = sum (b - first + 1) * bucket[b];  add first * running
    if (first != 0 && !running.is_inf()) {
      uint32_t k = first;
      acc = xyzz_add(acc, xyzz_mul_scalar(running, &k, 1));
    }
    contrib = acc;
  }
  contrib = wave_reduce_sum(contrib);
  constexpr int WORDS = sizeof(XYZZ<F>) / 4;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&contrib);
    for (int i = 0; i < WORDS; i++) wave_out[wave * WORDS + i] = src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    XYZZ<F> sum = XYZZ<F>::inf();
    for (uint32_t v = 0; v < blockDim.x / 64; v++) {
      XYZZ<F> t;
      uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
      for (int i = 0; i < WORDS; i++) dst[i] = wave_out[v * WORDS + i];
      sum = xyzz_add(sum, t);
    }
    partials[w * gridDim.x + blockIdx.x] = sum;
  }
}

// ---- the same tails for G2 with LANE PAIRS (field.cuh Fp2L) ------------------------------------------------------------
// Over Fq2 the out-of-line group addition with inlined multiplications needs 456 registers, so the kernels above call a
// function per field multiplication when F = Fq2 -- 3.9 ms of bucket reduction and 1.2 ms of merge per 2^20-term MSM on a
// few dozen workgroups.  Here every bucket / chunk belongs to TWO lanes, each holding one component of every coordinate:
// G1-like registers per lane, multiplications inlined, and each Fq2 product is shared by the pair.  Control flow is
// uniform inside a pair (keys, counts and the pair-wide predicates of Fp2L).
#ifndef ARK_PLAIN_HOST
template <class P>
ARK_D XYZZ<Fp2L<P>> pair_load(const XYZZ<Fp2<P>>* p) {
  const Fp<P>* q = reinterpret_cast<const Fp<P>*>(p);
  const uint32_t par = threadIdx.x & 1u;
  return XYZZ<Fp2L<P>>{Fp2L<P>{q[0 + par]}, Fp2L<P>{q[2 + par]}, Fp2L<P>{q[4 + par]}, Fp2L<P>{q[6 + par]}};
}
template <class P>
ARK_D void pair_store(XYZZ<Fp2<P>>* p, const XYZZ<Fp2L<P>>& v) {
  Fp<P>* q = reinterpret_cast<Fp<P>*>(p);
  const uint32_t par = threadIdx.x & 1u;
  q[0 + par] = v.x.c;
  q[2 + par] = v.y.c;
  q[4 + par] = v.zz.c;
  q[6 + par] = v.zzz.c;
}
// butterfly over the 32 pairs of a wave (the masks keep the lane parity)
template <class P>
ARK_D XYZZ<Fp2L<P>> wave_reduce_sum_pairs(XYZZ<Fp2L<P>> v) {
  for (int mask = 32; mask >= 2; mask >>= 1) {
    XYZZ<Fp2L<P>> o = xyzz_shfl_xor(v, mask);
    v = xyzz_add(v, o);
  }
  return v;
}

template <class P>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_merge_pair_kernel(uint32_t total_buckets, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                      XYZZ<Fp2<P>>* __restrict__ buckets, const XYZZ<Fp2<P>>* __restrict__ head,
                      const uint32_t* __restrict__ head_key, const XYZZ<Fp2<P>>* __restrict__ tail,
                      const uint32_t* __restrict__ tail_key, uint32_t* __restrict__ heavy_count,
                      uint32_t* __restrict__ heavy_list, uint32_t seg_len, uint32_t heavy_span) {
  using L = Fp2L<P>;
  const uint32_t key = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  if (key >= total_buckets) return;
  const uint32_t cnt = counts[key];
  if (cnt == 0) return;
  const uint32_t o = offsets[key];
  const uint32_t t0 = o / seg_len, t1 = (o + cnt - 1) / seg_len;
  if (t0 == t1) return;   // the single run was complete and already written
  if (t1 - t0 > heavy_span) {
    if ((threadIdx.x & 1u) == 0) heavy_list[atomicAdd(heavy_count, 1u)] = key;
    return;
  }
  XYZZ<L> sum = XYZZ<L>::inf();
  for (uint32_t t = t0; t <= t1; t++) {
    if (head_key[t] == key) sum = xyzz_add(sum, pair_load<P>(&head[t]));
    if (tail_key[t] == key) sum = xyzz_add(sum, pair_load<P>(&tail[t]));
  }
  pair_store<P>(&buckets[key], sum);
}

// one workgroup per heavy bucket: the 128 pairs stride over the bucket's segments, pair butterfly, 4 waves through LDS
template <class P>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_merge_heavy_pair_kernel(const uint32_t* __restrict__ heavy_count, const uint32_t* __restrict__ heavy_list,
                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                            XYZZ<Fp2<P>>* __restrict__ buckets, const XYZZ<Fp2<P>>* __restrict__ head,
                            const uint32_t* __restrict__ head_key, const XYZZ<Fp2<P>>* __restrict__ tail,
                            const uint32_t* __restrict__ tail_key, uint32_t seg_len) {
  using L = Fp2L<P>;
  constexpr int WORDS = sizeof(XYZZ<Fp2<P>>) / 4, HALF = sizeof(Fp<P>) / 4;
  __shared__ uint32_t wave_out[(MSM_THREADS / 64) * WORDS];
  const uint32_t nheavy = *heavy_count;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t key = heavy_list[h];
    const uint32_t o = offsets[key], cnt = counts[key];
    const uint32_t t0 = o / seg_len, t1 = (o + cnt - 1) / seg_len;
    XYZZ<L> sum = XYZZ<L>::inf();
    for (uint32_t t = t0 + (threadIdx.x >> 1); t <= t1; t += blockDim.x / 2) {
      if (head_key[t] == key) sum = xyzz_add(sum, pair_load<P>(&head[t]));
      if (tail_key[t] == key) sum = xyzz_add(sum, pair_load<P>(&tail[t]));
    }
    sum = wave_reduce_sum_pairs<P>(sum);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 2) {
      const Fp<P>* src[4] = {&sum.x.c, &sum.y.c, &sum.zz.c, &sum.zzz.c};
      for (int k = 0; k < 4; k++)
        for (int i = 0; i < HALF; i++) wave_out[wave * WORDS + (2 * k + lane) * HALF + i] = src[k]->l[i];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      XYZZ<L> tot = XYZZ<L>::inf();
      for (uint32_t v = 0; v < blockDim.x / 64; v++)
        tot = xyzz_add(tot, pair_load<P>(reinterpret_cast<const XYZZ<Fp2<P>>*>(&wave_out[v * WORDS])));
      pair_store<P>(&buckets[key], tot);
    }
    __syncthreads();
  }
}

// grid.x = blocks per window (MSM_THREADS / 2 chunks each), grid.y = windows
template <class P>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_reduce_pair_kernel(const XYZZ<Fp2<P>>* __restrict__ buckets, uint32_t buckets_per_window,
                       XYZZ<Fp2<P>>* __restrict__ partials) {
  using L = Fp2L<P>;
  constexpr int WORDS = sizeof(XYZZ<Fp2<P>>) / 4, HALF = sizeof(Fp<P>) / 4;
  __shared__ uint32_t wave_out[(MSM_THREADS / 64) * WORDS];
  const uint32_t w = blockIdx.y;
  const uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  const uint32_t first = chunk * MSM_RED_K;
  XYZZ<L> contrib = XYZZ<L>::inf();
  if (first < buckets_per_window) {
    const uint32_t last = (first + MSM_RED_K < buckets_per_window) ? first + MSM_RED_K : buckets_per_window;
    const XYZZ<Fp2<P>>* wb = buckets + (uint64_t)w * buckets_per_window;
    XYZZ<L> running = XYZZ<L>::inf();
    XYZZ<L> acc = XYZZ<L>::inf();
    for (uint32_t b = last; b-- > first;) {
      running = xyzz_add(running, pair_load<P>(&wb[b]));
      acc = xyzz_add(acc, running);
    }
    if (first != 0 && !running.is_inf()) {
      uint32_t k = first;
      acc = xyzz_add(acc, xyzz_mul_scalar(running, &k, 1));
    }
    contrib = acc;
  }
  contrib = wave_reduce_sum_pairs<P>(contrib);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 2) {          // the first pair of the wave: component `lane` of each coordinate
    const Fp<P>* src[4] = {&contrib.x.c, &contrib.y.c, &contrib.zz.c, &contrib.zzz.c};
    for (int k = 0; k < 4; k++)
      for (int i = 0; i < HALF; i++) wave_out[wave * WORDS + (2 * k + lane) * HALF + i] = src[k]->l[i];
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    XYZZ<L> sum = XYZZ<L>::inf();
    for (uint32_t v = 0; v < blockDim.x / 64; v++)
      sum = xyzz_add(sum, pair_load<P>(reinterpret_cast<const XYZZ<Fp2<P>>*>(&wave_out[v * WORDS])));
    pair_store<P>(&partials[w * gridDim.x + blockIdx.x], sum);
  }
}

// one wave = 32 pairs; window tables only (ONE bucket set: the pairs share its partials)
template <class P>
__global__ void __launch_bounds__(64)
msm_combine_pair_kernel(const XYZZ<Fp2<P>>* __restrict__ partials, uint32_t count, XYZZ<Fp2<P>>* __restrict__ out,
                        int accumulate) {
  using L = Fp2L<P>;
  const uint32_t pr = threadIdx.x >> 1;
  XYZZ<L> v = XYZZ<L>::inf();
  for (uint32_t i = pr; i < count; i += 32) v = xyzz_add(v, pair_load<P>(&partials[i]));
  v = wave_reduce_sum_pairs<P>(v);
  if (threadIdx.x < 2) {
    if (accumulate) v = xyzz_add(v, pair_load<P>(out));
    pair_store<P>(out, v);
  }
}
#endif  // ARK_PLAIN_HOST

// ---- two-level bucket reduction for large bucket sets (window sizes c >= 18 over window tables) ---------------------
// sum_b (b+1) B_b with b = K j + i:  sum_j [ W_j + K j T_j ],  T_j = sum_i B_{Kj+i},  W_j = sum_i (i+1) B_{Kj+i}.
// Level 1 gives every lane K consecutive buckets (two additions per bucket, no scalar multiplication at all); level 2
// is the weighted sum of the T_j -- the same shape as msm_reduce_kernel, 1/K of its size -- plus the plain sum of the
// W_j.  The one-level kernel pays a double-and-add by the chunk's first index (~28 group operations per 4 buckets); at
// 2^19 buckets that alone was 40 % of the accumulation work (round-1 measurement: c = 20 saved 5 % of the accumulation
// and lost 14 ms in the reduction), which is what kept the window size at 16.
constexpr uint32_t MSM_RED_L1 = 16;       // buckets per level-1 lane (power of two: K T is log2 K doublings)
constexpr uint32_t MSM_RED_L1_LOG = 4;
template <class F>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_reduce_l1_kernel(const XYZZ<F>* __restrict__ buckets, uint32_t nbuckets, XYZZ<F>* __restrict__ T,
                     XYZZ<F>* __restrict__ W) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t first = (uint64_t)j * MSM_RED_L1;
  if (first >= nbuckets) return;
  const uint32_t last = (first + MSM_RED_L1 < nbuckets) ? (uint32_t)first + MSM_RED_L1 : nbuckets;
  XYZZ<F> running = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
  for (uint32_t b = last; b-- > (uint32_t)first;) {
    running = xyzz_add(running, buckets[b]);
    acc = xyzz_add(acc, running);
  }
  T[j] = running;
  W[j] = acc;
}

// Output: partials[blockIdx.x] = sum over this workgroup's lanes of  sum_j W_j + K * sum_j j T_j
template <class F>
__global__ void __launch_bounds__(MSM_THREADS, MSM_TAIL_WAVES)
msm_reduce_l2_kernel(const XYZZ<F>* __restrict__ T, const XYZZ<F>* __restrict__ W, uint32_t items,
                     XYZZ<F>* __restrict__ partials) {
  __shared__ uint32_t wave_out[(MSM_THREADS / 64) * (sizeof(XYZZ<F>) / 4)];
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t first = (uint64_t)chunk * MSM_RED_K;
  XYZZ<F> contrib = XYZZ<F>::inf();
  if (first < items) {
    const uint32_t last = (first + MSM_RED_K < items) ? (uint32_t)first + MSM_RED_K : items;
    XYZZ<F> running = XYZZ<F>::inf(), acc = XYZZ<F>::inf(), wsum = XYZZ<F>::inf();
    for (uint32_t j = last; j-- > (uint32_t)first;) {
      acc = xyzz_add(acc, running);             // acc = sum (j - first) T_j
      running = xyzz_add(running, T[j]);
      wsum = xyzz_add(wsum, W[j]);
    }
    if (first != 0 && !running.is_inf()) {
      uint32_t k = (uint32_t)first;
      acc = xyzz_add(acc, xyzz_mul_scalar(running, &k, 1));
    }
    if (!acc.is_inf())
      for (uint32_t d = 0; d < MSM_RED_L1_LOG; d++) acc = xyzz_dbl(acc);
    contrib = xyzz_add(wsum, acc);
  }
  contrib = wave_reduce_sum(contrib);
  constexpr int WORDS = sizeof(XYZZ<F>) / 4;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&contrib);
    for (int i = 0; i < WORDS; i++) wave_out[wave * WORDS + i] = src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    XYZZ<F> sum = XYZZ<F>::inf();
    for (uint32_t v = 0; v < blockDim.x / 64; v++) {
      XYZZ<F> t;
      uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
      for (int i = 0; i < WORDS; i++) dst[i] = wave_out[v * WORDS + i];
      sum = xyzz_add(sum, t);
    }
    partials[blockIdx.x] = sum;
  }
}

// one wave.  Several bucket sets (no window tables): lane w -> 2^(c*w) * (sum of window w's partials), butterfly
// over lanes.  One bucket set (window tables): the lanes share the partials of the single set.  Lane 0
// writes / accumulates into *out.
template <class F>
__global__ void __launch_bounds__(64)
msm_combine_kernel(const XYZZ<F>* __restrict__ partials, uint32_t per_window, uint32_t windows, uint32_t c,
                   XYZZ<F>* __restrict__ out, int accumulate) {
  const uint32_t w = threadIdx.x;
  XYZZ<F> v = XYZZ<F>::inf();
  if (windows == 1) {
    for (uint32_t i = w; i < per_window; i += 64) v = xyzz_add(v, partials[i]);
  } else if (w < windows) {
    for (uint32_t i = 0; i < per_window; i++) v = xyzz_add(v, partials[w * per_window + i]);
    if (!v.is_inf()) {
      const uint32_t dbl = c * w;
      for (uint32_t i = 0; i < dbl; i++) v = xyzz_dbl(v);
    }
  }
  v = wave_reduce_sum(v);
  if (threadIdx.x == 0) {
    if (accumulate) v = xyzz_add(v, *out);
    *out = v;
  }
}

// XYZZ -> affine for `count` points (one lane each)
template <class F>
__global__ void __launch_bounds__(64)
xyzz_to_affine_kernel(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = xyzz_to_affine(in[i]);
}

// sum of `count` XYZZ points -> out[0] (single wave; the cross-GPU combine of SURVEY 8e)
template <class F>
__global__ void __launch_bounds__(64)
xyzz_sum_kernel(const XYZZ<F>* __restrict__ in, uint32_t count, XYZZ<F>* __restrict__ out) {
  XYZZ<F> v = XYZZ<F>::inf();
  for (uint32_t i = threadIdx.x; i < count; i += 64) v = xyzz_add(v, in[i]);
  v = wave_reduce_sum(v);
  if (threadIdx.x == 0) *out = v;
}

// ---- per-window base tables (built once per key) --------------------------------------------------------------------
// out[i] = 2^c * in[i]  (affine in, XYZZ out)
template <class F>
__global__ void __launch_bounds__(MSM_THREADS)
precomp_shift_kernel(const Affine<F>* __restrict__ in, XYZZ<F>* __restrict__ out, uint32_t n, uint32_t c) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> p = xyzz_dbl_affine_t<true>(in[i]);
  for (uint32_t k = 1; k < c; k++) p = xyzz_dbl(p);
  out[i] = p;
}

// XYZZ -> affine for n points, PRE_K consecutive points per lane sharing ONE field inversion (Montgomery's trick)
constexpr uint32_t PRE_K = 16;
template <class F>
__global__ void __launch_bounds__(MSM_THREADS)
batch_to_affine_kernel(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, uint32_t n) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t first64 = (uint64_t)t * PRE_K;
  if (first64 >= n) return;
  const uint32_t first = (uint32_t)first64;
  const uint32_t cnt = (first + PRE_K <= n) ? PRE_K : (n - first);
  F pref[PRE_K];
  F acc = F::one();
  for (uint32_t j = 0; j < cnt; j++) {
    const XYZZ<F> p = in[first + j];
    if (!p.is_inf()) acc = F::mul_ni(acc, p.zzz);
    pref[j] = acc;
  }
  F inv = F::inv(acc);
  for (uint32_t j = cnt; j-- > 0;) {
    const XYZZ<F> p = in[first + j];
    if (p.is_inf()) {
      out[first + j] = Affine<F>::inf();
      continue;
    }
    const F prev = (j > 0) ? pref[j - 1] : F::one();
    const F i3 = F::mul_ni(inv, prev);      // 1 / zzz_j
    inv = F::mul_ni(inv, p.zzz);
    const F iz = F::mul_ni(p.zz, i3);
    const F i2 = F::sqr_ni(iz);
    out[first + j] = Affine<F>{F::mul_ni(p.x, i2), F::mul_ni(p.y, i3)};
  }
}

// Window tables of a base vector: T[w*n + i] = 2^(c*w) * P_i, affine, for the plan the MSM of this length uses.
struct PrecompTable {
  MsmPlan plan;
  DevBuf table;          // windows * n rows: Affine<F>, or Affine28 rows when limb28 is set
  uint64_t n = 0;
  bool limb28 = false;   // rows are stored in the radix-2^28 form of field28.cuh (msm28_impl.cuh)
  bool packed = false;   // ... bit-packed (Affine28: 96 / 64 B per G1 row) instead of one word per limb (Affine28U: 128 / 80 B)
  int fmt() const { return limb28 ? (packed ? 2 : 1) : 0; }       // what msm_accumulate_phase is told
};

// ---- HBM footprint of window tables -------------------------------------------------------------------------------------
// Full tables (every window) are W x the key: 16 x 128-byte rows per G1 base at c = 16, i.e. 15 GB for a 2^20-constraint
// BLS12-381 key, 60 GB at 2^22, ~120 GB at 2^23 and more than one MI355X holds at 2^24.  table_row_bytes / the planner
// below pick the smallest window stride (MsmPlan::wstride) whose tables fit the budget, so that such a key still loads
// -- with every second (third, ...) window's table and that many bucket sets -- instead of failing in hipMalloc.
template <class F>
static inline size_t table_row_bytes(bool limb28, bool packed = false) {
  if (!limb28) return sizeof(Affine<F>);
  if constexpr (is_fp2<F>::value) {
    using P = typename F::Base::Params;
    return packed ? sizeof(Affine28G2<P, true>) : sizeof(Affine28G2<P, false>);
  } else {
    using P = typename F::Params;
    return packed ? sizeof(Affine28<P>) : sizeof(Affine28U<P>);
  }
}
// Row format of the 28-bit tables of one key (policy PACK_ROWS, read when a key / base set is loaded): 1 = packed, 0 = one word
// per limb, -1 (default) = packed when a packed G1 row is a whole number of 64-byte sectors (BN254: 64 B -- faster and
// smaller), else unpacked UNLESS the key's tables would not fit HBM at stride 1 that way (table_stride_plan packs them
// before it gives up windows).  One decision per key: all five tables use the same format.
template <class Fq>
static inline bool table_pack_default(const TunePolicy& pol) {
  if (pol.pack_rows >= 0) return pol.pack_rows != 0;
  return sizeof(Affine28<typename Fq::Params>) % 64 == 0;
}

struct TableNeed {          // one base vector of a key
  uint64_t n;               // rows per window block
  uint64_t plan_n;          // length the window size is planned for (shards: the largest shard)
  bool g2;
  int pref_c = 0;           // window size asked for this vector alone (policy MSM_C_H for h_query); 0: policy MSM_C
};

// HBM budget for the resident tables of ONE key / base set: policy HBM_BUDGET_MB (tests, A/B) or 80 % of the device
// minus a reserve for the proving contexts' scratch (`scratch_bytes`, the caller's estimate); `use_free`: also stay
// inside what is free right now (whole keys; the shards of one key plan from the device size alone so that every rank
// reaches the same stride -- the bucket-level exchange adds bucket arrays of different ranks).
static inline size_t table_budget_bytes(const TunePolicy& pol, size_t scratch_bytes, bool use_free) {
  if (pol.hbm_budget_mb > 0) return (size_t)pol.hbm_budget_mb << 20;
#if defined(ARK_EMUL)
  (void)scratch_bytes;
  (void)use_free;
  return ~(size_t)0 >> 1;
#else
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) return ~(size_t)0 >> 1;
  size_t budget = total_b / 5 * 4;
  if (use_free && free_b < budget) budget = free_b;
  return budget > scratch_bytes ? budget - scratch_bytes : 0;
#endif
}

// smallest stride whose tables (all vectors resident + the two-window staging area of the one being built) fit `budget`;
// 0 when not even the bare base vectors do
// *packed (out): the row format of the key's 28-bit tables (table_pack_default, or packed when that is what makes stride 1 fit)
template <class Fq, class Fq2, class Fr>
static inline uint32_t table_stride_plan(const TunePolicy& pol, const TableNeed* need, int count, size_t budget, std::string* why,
                                         bool* packed_out = nullptr) {
  bool pk = table_pack_default<Fq>(pol);
  if (packed_out) *packed_out = pk;
  if (pol.table_stride >= 1 && pol.table_stride <= 64) return (uint32_t)pol.table_stride;       // tests / A/B: force a stride
  uint32_t max_windows = 1;
  for (uint32_t s = 1;; s++) {
    if (s == 2 && !pk && pol.pack_rows < 0) {       // stride 1 did not fit with unpacked rows: pack them and try again
      pk = true;
      if (packed_out) *packed_out = true;
      s = 1;
    }
    size_t resident = 0, stage = 0;
    for (int i = 0; i < count; i++) {
      const int pc = need[i].pref_c ? need[i].pref_c : pol.msm_c;
      const MsmPlan p = msm_plan(need[i].n, Fr::Params::BITS, true,
                                 need[i].plan_n ? (int)msm_plan(need[i].plan_n, Fr::Params::BITS, true, 0, 0, pc).c : 0, s, pc);
      if (p.windows > max_windows) max_windows = p.windows;
      const size_t row = need[i].g2 ? table_row_bytes<Fq2>(true, pk) : table_row_bytes<Fq>(true, pk);
      const size_t aff = need[i].g2 ? sizeof(Affine<Fq2>) : sizeof(Affine<Fq>);
      const size_t xyzz = need[i].g2 ? sizeof(XYZZ<Fq2>) : sizeof(XYZZ<Fq>);
      resident += (size_t)p.table_windows * need[i].n * row;
      const size_t st = need[i].n * (2 * aff + xyzz);
      if (st > stage) stage = st;
    }
    if (resident + stage <= budget) return s;
    if (s >= max_windows) {
      if (why)
        *why = "window tables do not fit HBM: " + std::to_string((resident + stage) >> 20) + " MiB needed even with one table window, budget " +
               std::to_string(budget >> 20) + " MiB";
      return 0;
    }
  }
}

// plan_n: the length the window size is chosen for (0 = n).  The shards of one key pass the LARGEST shard length so
// that every rank uses the same window size -- the bucket-level exchange adds bucket arrays of different ranks.
// wstride: MsmPlan::wstride (from table_stride_plan).
// pref_c: window size asked for this table alone (0: policy MSM_C).
template <class F, class Fr>
// packed: row format of a 28-bit table (-1: table_pack_default; a key passes what table_stride_plan decided for all its tables)
static void precomp_build(const TunePolicy& pol, PrecompTable& t, const void* d_bases, uint64_t n, hipStream_t stream,
                          uint64_t plan_n = 0, uint32_t wstride = 1, int pref_c = 0, int packed = -1) {
  t.n = n;
  const int pc = pref_c ? pref_c : pol.msm_c;
  t.plan = msm_plan(n, Fr::Params::BITS, /*precomp=*/true,
                    plan_n ? (int)msm_plan(plan_n, Fr::Params::BITS, /*precomp=*/true, 0, 0, pc).c : 0, wstride, pc);
  const MsmPlan& p = t.plan;
  const uint32_t TW = p.table_windows;
  ARK_REQUIRE((uint64_t)TW * n < (1ull << 31), ARK355_EINVAL, "window table too large for 31-bit indices");
  if (pol.trace_host)
    fprintf(stderr, "[ark355] window table: %llu bases, c = %u, %u windows (%u table blocks, %u bucket sets)%s\n",
            (unsigned long long)n, p.c, p.windows, TW, p.key_windows, p.negate_high ? ", scalars above (r - 1) / 2 negated" : "");
  const uint32_t grid = (uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS);
  const uint32_t gridb = (uint32_t)(((n + PRE_K - 1) / PRE_K + MSM_THREADS - 1) / MSM_THREADS);
  const uint32_t shift = p.c * p.wstride;              // consecutive table blocks differ by 2^(c * wstride)
  // radix-2^28 rows (msm28_impl.cuh): the canonical form of a block only lives in a two-block staging area while the
  // next block is derived from it; each finished block is re-encoded straight into the final table.  (The first version
  // built the whole canonical table first: 75 % more HBM at the peak than the key keeps -- what a 2^23-constraint key
  // cannot spare.)
  bool pack;
  if constexpr (is_fp2<F>::value) pack = packed >= 0 ? packed != 0 : table_pack_default<typename F::Base>(pol);
  else pack = packed >= 0 ? packed != 0 : table_pack_default<F>(pol);
  const size_t row = table_row_bytes<F>(true, pack);
  t.table.alloc((size_t)TW * (n ? n : 1) * row);
  t.limb28 = true;
  t.packed = pack;
  if (n == 0) return;
  DevBuf stage[2] = {DevBuf(n * sizeof(Affine<F>)), DevBuf(n * sizeof(Affine<F>))};
  DevBuf tmp(n * sizeof(XYZZ<F>));
  const dim3 grid28((uint32_t)((n + 255) / 256));
  auto encode = [&](const DevBuf& src, uint32_t w) {
    uint8_t* dst = t.table.as<uint8_t>() + (size_t)w * n * row;
    if constexpr (is_fp2<F>::value) {
      using P = typename F::Base::Params;
      if (pack)
        ARK_LAUNCH((table_to28_g2_kernel<P, true>), grid28, dim3(256), 0, stream, (const Affine<F>*)src.as<Affine<F>>(),
                   reinterpret_cast<Affine28G2<P, true>*>(dst), n);
      else
        ARK_LAUNCH((table_to28_g2_kernel<P, false>), grid28, dim3(256), 0, stream, (const Affine<F>*)src.as<Affine<F>>(),
                   reinterpret_cast<Affine28G2<P, false>*>(dst), n);
    } else {
      using P = typename F::Params;
      if (pack)
        ARK_LAUNCH((table_to28_kernel<P, true>), grid28, dim3(256), 0, stream, (const Affine<F>*)src.as<Affine<F>>(),
                   reinterpret_cast<Affine28<P>*>(dst), n);
      else
        ARK_LAUNCH((table_to28_kernel<P, false>), grid28, dim3(256), 0, stream, (const Affine<F>*)src.as<Affine<F>>(),
                   reinterpret_cast<Affine28U<P>*>(dst), n);
    }
    ARK_CHECK_LAUNCH();
  };
  ARK_CHECK_HIP(hipMemcpyAsync(stage[0].p, d_bases, n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, stream));
  encode(stage[0], 0);
  for (uint32_t w = 1; w < TW; w++) {
    DevBuf& prev = stage[(w - 1) & 1];
    DevBuf& cur = stage[w & 1];
    ARK_LAUNCH((precomp_shift_kernel<F>), dim3(grid), dim3(MSM_THREADS), 0, stream, (const Affine<F>*)prev.as<Affine<F>>(),
               tmp.as<XYZZ<F>>(), (uint32_t)n, shift);
    ARK_CHECK_LAUNCH();
    ARK_LAUNCH((batch_to_affine_kernel<F>), dim3(gridb), dim3(MSM_THREADS), 0, stream, (const XYZZ<F>*)tmp.as<XYZZ<F>>(),
               cur.as<Affine<F>>(), (uint32_t)n);
    ARK_CHECK_LAUNCH();
    encode(cur, w);
  }
  ARK_CHECK_HIP(hipStreamSynchronize(stream));     // staging buffers are freed on return
}

// ---- one fill dispatch per proof ----------------------------------------------------------------------------------------
// A proof used to queue 31 hipMemsetAsync calls (sort counters, bucket sets, head / tail keys, heavy-bucket counters): 31
// dispatches of ~5 us of work each.  Kernel time was never the issue -- the DISPATCHES are: on some boxes of the pool an
// in-order queue pays 50-90 us per dispatch (ark355_diag_dispatch), i.e. 2-3 ms per proof for clearing 50 MB.  A FillBatch
// collects the ranges while the proof's scratch is planned and clears them with ONE kernel.
struct FillRange {
  void* p;
  uint32_t n16;       // whole 16-byte units
  uint32_t rem;       // 0..3 trailing 32-bit words
  uint32_t value;     // the byte value replicated into a word
  uint32_t pad;
};
constexpr uint32_t FILL_MAX = 28;
struct FillList {
  FillRange r[FILL_MAX];
  uint32_t count;
};
static __global__ void __launch_bounds__(256) multi_fill_kernel(FillList L) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint32_t k = 0; k < L.count; k++) {
    const FillRange r = L.r[k];
    const uint4 v = make_uint4(r.value, r.value, r.value, r.value);
    uint4* d = reinterpret_cast<uint4*>(r.p);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n16; i += stride) d[i] = v;
    if (blockIdx.x == 0 && threadIdx.x < r.rem) reinterpret_cast<uint32_t*>(r.p)[4ull * r.n16 + threadIdx.x] = r.value;
  }
}
struct FillBatch {
  FillList list;
  hipStream_t stream;
  explicit FillBatch(hipStream_t st) : stream(st) { list.count = 0; }
  // bytes must be a multiple of 4 and p 16-byte aligned (hipMalloc'ed scratch); anything else is cleared the old way
  void add(void* p, size_t bytes, uint8_t byte_value) {
    if (bytes == 0) return;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 15) || (bytes >> 4) > 0xFFFFFFFFull) {
      ARK_CHECK_HIP(hipMemsetAsync(p, byte_value, bytes, stream));
      return;
    }
    if (list.count == FILL_MAX) flush();
    FillRange& r = list.r[list.count++];
    r.p = p;
    r.n16 = (uint32_t)(bytes >> 4);
    r.rem = (uint32_t)((bytes & 15) >> 2);
    r.value = 0x01010101u * byte_value;
    r.pad = 0;
  }
  void flush() {
    if (list.count == 0) return;
    uint64_t units = 0;
    for (uint32_t k = 0; k < list.count; k++) units += list.r[k].n16;
#if defined(ARK_EMUL)
    const uint32_t grid = 1;               // the emulator runs lanes one after another
#else
    uint64_t g = (units + 256ull * 8 - 1) / (256ull * 8);
    const uint32_t grid = (uint32_t)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
#endif
    ARK_LAUNCH(multi_fill_kernel, dim3(grid), dim3(256), 0, stream, list);
    ARK_CHECK_LAUNCH();
    list.count = 0;
  }
};
// clear through the batch when there is one, directly otherwise
static inline void fill_bytes(FillBatch* fb, void* p, uint8_t v, size_t bytes, hipStream_t stream) {
  if (fb) fb->add(p, bytes, v);
  else ARK_CHECK_HIP(hipMemsetAsync(p, v, bytes, stream));
}

// ---- host driver ---------------------------------------------------------------------------------------------
// Scratch for one MSM "sort" (shared by several accumulations over the same scalars).
struct MsmSort {
  MsmPlan plan;
  DevBuf keys, vals, counts, offsets, cursor, sorted_keys, sorted_vals, total, hist, hist_scanned, scan_aux;
};

// Plan one sort (window size, buffer sizes) and clear its counters -- through `fb` when the caller batches the fills of
// a whole proof (prove_run), on `stream` otherwise.  msm_sort_run queues the kernels.
template <class Fr>
static void msm_sort_plan(ark355_ctx* ctx, MsmSort& s, uint64_t n, hipStream_t stream, const PrecompTable* tab = nullptr,
                          FillBatch* fb = nullptr) {
  (void)ctx;
  ARK_REQUIRE(n < (1ull << 31), ARK355_EINVAL, "MSM size must be < 2^31");
  const bool precomp = tab != nullptr;
  if (tab) ARK_REQUIRE(n <= tab->n, ARK355_EINVAL, "more scalars than table rows");
  // with window tables the window size is the one the tables were built for
  s.plan = msm_plan(n, Fr::Params::BITS, precomp, tab ? (int)tab->plan.c : 0, tab ? tab->plan.wstride : 0);
  const MsmPlan& p = s.plan;
  const uint64_t entries = (uint64_t)p.windows * n;
  ARK_REQUIRE(entries < (1ull << 31), ARK355_EINVAL, "MSM entry count must be < 2^31");
  // two-level sort: level 1 writes (key, value) PAIRS into `keys` (one 8-byte store per entry instead of two 4-byte stores into
  // two arrays: round 6); the one-pass fallback keeps separate arrays
  const bool two_level = (p.total_buckets + SORT_LO - 1) / SORT_LO <= SORT_MAX_BINS;
  s.keys.ensure(entries * (two_level ? 8 : 4));
  if (!two_level) s.vals.ensure(entries * 4);
  s.sorted_keys.ensure(entries * 4 + 16);
  s.sorted_vals.ensure(entries * 4 + 16);
  s.counts.ensure((size_t)p.total_buckets * 4);
  s.offsets.ensure((size_t)p.total_buckets * 4);
  s.cursor.ensure((size_t)p.total_buckets * 4);
  s.total.ensure(16);
  if (n == 0) {
    fill_bytes(fb, s.total.p, 0, 4, stream);
    fill_bytes(fb, s.counts.p, 0, (size_t)p.total_buckets * 4, stream);
    fill_bytes(fb, s.offsets.p, 0, (size_t)p.total_buckets * 4, stream);
    return;
  }
  fill_bytes(fb, s.counts.p, 0, (size_t)p.total_buckets * 4, stream);
  fill_bytes(fb, s.cursor.p, 0, (size_t)p.total_buckets * 4, stream);
}

template <class Fr>
static void msm_sort_run(ark355_ctx* ctx, MsmSort& s, const void* d_scalars, uint64_t n, int mont, hipStream_t stream,
                         const PrecompTable* tab = nullptr) {
  const MsmPlan& p = s.plan;
  ARK_REQUIRE(p.n == n, ARK355_EINVAL, "sort was planned for another length");
  if (n == 0) return;
  const uint32_t stride = tab ? (uint32_t)tab->n : 0;
  const uint64_t entries = (uint64_t)p.windows * n;
  const uint32_t bins = (p.total_buckets + SORT_LO - 1) / SORT_LO;
  // the one-pass counting sort of round 1 when level 1 would not fit LDS
  if (bins > SORT_MAX_BINS) {
    const uint32_t grid_n = (uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS);
    ARK_LAUNCH((msm_digits_kernel<Fr>), dim3(grid_n), dim3(MSM_THREADS), 0, stream, (const Fr*)d_scalars, (uint32_t)n,
               mont, p.c, p.windows, msm_digit_flags(p), stride, s.keys.as<uint32_t>(), s.vals.as<uint32_t>(),
               s.counts.as<uint32_t>());
    ARK_CHECK_LAUNCH();
    scan_exclusive(stream, s.counts.as<uint32_t>(), s.offsets.as<uint32_t>(), p.total_buckets, s.total.as<uint32_t>(), s.scan_aux);
    const uint32_t grid_e = (uint32_t)((entries + MSM_THREADS - 1) / MSM_THREADS);
    ARK_LAUNCH(msm_scatter_kernel, dim3(grid_e), dim3(MSM_THREADS), 0, stream, s.keys.as<uint32_t>(),
               s.vals.as<uint32_t>(), entries, s.offsets.as<uint32_t>(), s.cursor.as<uint32_t>(),
               s.sorted_keys.as<uint32_t>(), s.sorted_vals.as<uint32_t>());
    ARK_CHECK_LAUNCH();
    return;
  }
  // level 1: group by bin (key >> 8) into keys/vals
  const uint32_t per_wg = SORT_HI_THREADS * SORT_SPT;
  const uint32_t grid1 = (uint32_t)((n + per_wg - 1) / per_wg);
  const size_t hist_elems = (size_t)bins * grid1;
  s.hist.ensure(hist_elems * 4);
  s.hist_scanned.ensure(hist_elems * 4);
  ARK_LAUNCH((sort_hi_hist_kernel<Fr>), dim3(grid1), dim3(SORT_HI_THREADS), 0, stream, (const Fr*)d_scalars, (uint32_t)n,
             mont, p.c, p.windows, msm_digit_flags(p), stride, bins, s.hist.as<uint32_t>());
  ARK_CHECK_LAUNCH();
  ARK_REQUIRE(hist_elems < (1ull << 31), ARK355_EINVAL, "sort histogram too large");
  scan_exclusive(stream, s.hist.as<uint32_t>(), s.hist_scanned.as<uint32_t>(), (uint32_t)hist_elems, s.total.as<uint32_t>(), s.scan_aux);
  ARK_LAUNCH((sort_hi_scatter_kernel<Fr>), dim3(grid1), dim3(SORT_HI_THREADS), 0, stream, (const Fr*)d_scalars,
             (uint32_t)n, mont, p.c, p.windows, msm_digit_flags(p), stride, bins, s.hist_scanned.as<uint32_t>(),
             s.keys.as<uint2>());
  ARK_CHECK_LAUNCH();
  // level 2: bucket histogram, bucket offsets, final placement
  const uint32_t grid2 = (uint32_t)((entries + SORT_TILE - 1) / SORT_TILE);
  ARK_LAUNCH((sort_lo_kernel<false>), dim3(grid2), dim3(MSM_THREADS), 0, stream, (const uint2*)s.keys.as<uint2>(), s.total.as<uint32_t>(), s.counts.as<uint32_t>(), s.offsets.as<uint32_t>(),
             s.cursor.as<uint32_t>(), s.sorted_keys.as<uint32_t>(), s.sorted_vals.as<uint32_t>());
  ARK_CHECK_LAUNCH();
  scan_exclusive(stream, s.counts.as<uint32_t>(), s.offsets.as<uint32_t>(), p.total_buckets, s.total.as<uint32_t>(), s.scan_aux);
  ARK_LAUNCH((sort_lo_kernel<true>), dim3(grid2), dim3(MSM_THREADS), 0, stream, (const uint2*)s.keys.as<uint2>(), s.total.as<uint32_t>(), s.counts.as<uint32_t>(), s.offsets.as<uint32_t>(),
             s.cursor.as<uint32_t>(), s.sorted_keys.as<uint32_t>(), s.sorted_vals.as<uint32_t>());
  ARK_CHECK_LAUNCH();
}

template <class Fr>
static void msm_sort(ark355_ctx* ctx, MsmSort& s, const void* d_scalars, uint64_t n, int mont, hipStream_t stream,
                     const PrecompTable* tab = nullptr) {
  msm_sort_plan<Fr>(ctx, s, n, stream, tab);
  msm_sort_run<Fr>(ctx, s, d_scalars, n, mont, stream, tab);
}

// Scratch for the bucket phase of one group type.
struct MsmBuckets {
  DevBuf buckets, head, tail, head_key, tail_key, partials, heavy_count, heavy_list, lvl_t, lvl_w;
  DevBuf rc;                            // row / column sums of the bucket matrix (tails28_impl.cuh, stage A)
  uint32_t seg_len = 32, segs = 0;      // of the last accumulation over this bucket set (G1 and G2 differ)
  int fmt = 0;                          // slot format of buckets / head / tail: 0 canonical XYZZ<F>, else Slot28 / Slot28G2
  bool prepared = false;                // msm_prepare_phase ran for the coming accumulation
  bool heavy_cleared = false;           // msm_prepare_phase already cleared heavy_count for the coming merge
};

// The field a group's coordinates live in, seen from the 28-bit kernels: base-field parameters + "on lane pairs"
template <class F>
struct Tail28Of {
  using P = typename F::Params;
  static constexpr bool G2 = false;
};
template <class P_>
struct Tail28Of<Fp2<P_>> {
  using P = P_;
  static constexpr bool G2 = true;
};
template <class F>
static inline size_t msm_slot_bytes(int fmt) {
  return fmt ? sizeof(typename Tail28<typename Tail28Of<F>::P, Tail28Of<F>::G2>::Slot) : sizeof(XYZZ<F>);
}
// What an MSM leaves on the device for the host: ONE XYZZ sum from the 32-bit tails (one-shot MSMs over caller's bases), c
// partial sums per bucket set from the 28-bit tails (resident tables); msm_parts_finish turns either into the sum.
static inline uint32_t msm_parts_count(const MsmPlan& p, int fmt) { return fmt ? p.key_windows * p.c : 1u; }
template <class F>
static XYZZ<F> msm_parts_finish(const XYZZ<F>* parts, const MsmPlan& p, int fmt) {
  return fmt ? msm_finish_host<F>(parts, p.key_windows, p.c) : parts[0];
}

// Phase 1 of the bucket method over an existing sort: bucket accumulation (the chip-filling kernel).
// Size and clear the bucket set of one MSM over an existing sort.  The prover issues this on its (high-priority) sort
// stream right behind the sort, so that the accumulation stream carries nothing but accumulation kernels: three small
// fill kernels in front of every accumulation launch sat behind the other proofs' workgroups and opened a gap between
// consecutive accumulations (31 fills per proof, 2.5 ms of stream time with four proofs in flight).
template <class F>
static void msm_prepare_phase(const TunePolicy& pol, const MsmSort& s, MsmBuckets& b, hipStream_t stream, int fmt = 0,
                              FillBatch* fb = nullptr) {
  const MsmPlan& p = s.plan;
  b.prepared = true;
  b.heavy_cleared = false;
  b.fmt = fmt;
  if (p.n == 0) return;
  const size_t slot = msm_slot_bytes<F>(fmt);
  const uint64_t entries = (uint64_t)p.windows * p.n;
  b.seg_len = msm_seg_len(entries, is_fp2<F>::value, pol.msm_seg);
  b.segs = (uint32_t)((entries + b.seg_len - 1) / b.seg_len);
  const uint32_t segs = b.segs;
  b.buckets.ensure((size_t)p.total_buckets * slot);
  b.head.ensure((size_t)segs * slot);
  b.tail.ensure((size_t)segs * slot);
  b.head_key.ensure((size_t)segs * 4);
  b.tail_key.ensure((size_t)segs * 4);
  fill_bytes(fb, b.buckets.p, 0, (size_t)p.total_buckets * slot, stream);
  fill_bytes(fb, b.head_key.p, 0xFF, (size_t)segs * 4, stream);
  fill_bytes(fb, b.tail_key.p, 0xFF, (size_t)segs * 4, stream);
  // the heavy-bucket counter of the merge that follows the accumulation (msm_reduce_phase clears it itself otherwise)
  b.heavy_count.ensure(16);
  fill_bytes(fb, b.heavy_count.p, 0, 4, stream);
  b.heavy_cleared = true;
}

// destination arrays of the 28-bit accumulation kernels: which = 0 buckets, 1 head, 2 tail
template <class P, int COORDS>
static Msm28Slot<P, COORDS>* msm_slots28(MsmBuckets& b, uint32_t total_buckets, int which) {
  (void)total_buckets;
  return (which == 0 ? b.buckets : (which == 1 ? b.head : b.tail)).as<Msm28Slot<P, COORDS>>();
}

template <class F>
static void msm_accumulate_phase(ark355_ctx* ctx, const MsmSort& s, MsmBuckets& b, const Affine<F>* d_bases,
                                 hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                                 int fmt = 0) {
  // fmt: PrecompTable::fmt() of the table d_bases points to -- 0 canonical Affine<F> rows, 1 unpacked, 2 packed 28-bit rows
  const MsmPlan& p = s.plan;
  if (!b.prepared) msm_prepare_phase<F>(ctx->policy, s, b, stream, fmt);      // stand-alone MSMs: same stream
  ARK_REQUIRE(b.fmt == fmt, ARK355_EINVAL, "bucket set was prepared for another slot format");
  b.prepared = false;
  if (p.n == 0) return;
  const uint32_t segs = b.segs;
  const uint32_t grid_s = (segs + MSM_THREADS - 1) / MSM_THREADS;
  // Workgroup size of the two kernels without LDS (unpacked 28-bit rows).  A 256-lane workgroup needs a free register slot on
  // four SIMDs at once and gives its slots back only when its slowest wave is done; with 64 lanes every SIMD refills by itself.
  // Measured, same box, interleaved (runs T, V): a LONE one-stream 2^20 proof 24.4-24.6 -> 23.8 ms device-resident with 64; four
  // proofs in flight 21.8-22.0 ms either way; the five-stream pipeline of a sharded proof's rank 15.9-16.0 -> 16.4-16.5 ms, i.e.
  // worse (its feeder kernels then compete with four times as many workgroups for the dispatcher).  So the CALL says what it
  // wants (ark355_ctx::acc_threads_hint: prove_run asks for 64 for a proof alone on one stream) and policy ACC_THREADS = 64 / 128 /
  // 256 overrides it (0, the default: the hint, else 256).
  const int acc_want = ctx->policy.acc_threads ? ctx->policy.acc_threads : ctx->acc_threads_hint;
  const uint32_t acc_t = (acc_want == 64 || acc_want == 128) ? (uint32_t)acc_want : MSM_THREADS;
  if (ev0) ARK_CHECK_HIP(hipEventRecord(ev0, stream));
  if constexpr (is_fp2<F>::value) {
    // G2: lane-split kernels (two lanes per segment; the whole-element kernels of round 1 lost to them by 2x and are gone)
    using P = typename F::Base::Params;
    const uint32_t grid_l = (2 * segs + MSM_THREADS - 1) / MSM_THREADS;
    if (fmt == 2) {
      ARK_LAUNCH((msm_accumulate_g2l28p_kernel<P>), dim3(grid_l), dim3(MSM_THREADS), 0, stream,
                 reinterpret_cast<const Affine28G2<P, true>*>(d_bases), s.sorted_keys.as<uint32_t>(),
                 s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(), s.offsets.as<uint32_t>(),
                 s.counts.as<uint32_t>(), msm_slots28<P, 8>(b, p.total_buckets, 0), msm_slots28<P, 8>(b, p.total_buckets, 1),
                 b.head_key.as<uint32_t>(), msm_slots28<P, 8>(b, p.total_buckets, 2), b.tail_key.as<uint32_t>(), b.seg_len);
    } else if (fmt == 1) {
      constexpr bool ZL = g2l28_zz_in_lds<P>();
      ARK_LAUNCH((msm_accumulate_g2l28_kernel<P, ZL>), dim3((2 * segs + acc_t - 1) / acc_t), dim3(acc_t),
                 ZL ? ZzLds<P>::bytes(acc_t) : (size_t)0, stream,
                 reinterpret_cast<const Affine28G2<P, false>*>(d_bases), s.sorted_keys.as<uint32_t>(),
                 s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(), s.offsets.as<uint32_t>(),
                 s.counts.as<uint32_t>(), msm_slots28<P, 8>(b, p.total_buckets, 0), msm_slots28<P, 8>(b, p.total_buckets, 1),
                 b.head_key.as<uint32_t>(), msm_slots28<P, 8>(b, p.total_buckets, 2), b.tail_key.as<uint32_t>(), b.seg_len);
    } else {
      ARK_LAUNCH((msm_accumulate_g2l_kernel<P>), dim3(grid_l), dim3(MSM_THREADS), 0, stream, d_bases,
                 s.sorted_keys.as<uint32_t>(), s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(),
                 s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(), b.head.as<XYZZ<F>>(),
                 b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(), b.seg_len);
    }
  } else if (fmt == 2) {
    using P = typename F::Params;
    ARK_LAUNCH((msm_accumulate28p_kernel<P>), dim3(grid_s), dim3(MSM_THREADS), 0, stream,
               reinterpret_cast<const Affine28<P>*>(d_bases), s.sorted_keys.as<uint32_t>(),
               s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(), s.offsets.as<uint32_t>(),
               s.counts.as<uint32_t>(), msm_slots28<P, 4>(b, p.total_buckets, 0), msm_slots28<P, 4>(b, p.total_buckets, 1),
               b.head_key.as<uint32_t>(), msm_slots28<P, 4>(b, p.total_buckets, 2), b.tail_key.as<uint32_t>(), b.seg_len);
  } else if (fmt == 1) {
    using P = typename F::Params;
    ARK_LAUNCH((msm_accumulate28_kernel<P>), dim3((segs + acc_t - 1) / acc_t), dim3(acc_t), 0, stream,
               reinterpret_cast<const Affine28U<P>*>(d_bases), s.sorted_keys.as<uint32_t>(),
               s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(), s.offsets.as<uint32_t>(),
               s.counts.as<uint32_t>(), msm_slots28<P, 4>(b, p.total_buckets, 0), msm_slots28<P, 4>(b, p.total_buckets, 1),
               b.head_key.as<uint32_t>(), msm_slots28<P, 4>(b, p.total_buckets, 2), b.tail_key.as<uint32_t>(), b.seg_len);
  } else {
    ARK_LAUNCH((msm_accumulate_kernel<F, false>), dim3(grid_s), dim3(MSM_THREADS), 0, stream, d_bases,
               s.sorted_keys.as<uint32_t>(), s.sorted_vals.as<uint32_t>(), s.total.as<uint32_t>(),
               s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(), b.head.as<XYZZ<F>>(),
               b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(), b.seg_len);
  }
  ARK_CHECK_LAUNCH();
  if (ev1) ARK_CHECK_HIP(hipEventRecord(ev1, stream));
}

// Phase 2 of `count` (<= TAIL28_MAX) MSMs over 28-bit slots whose accumulations have been queued on `stream`, as ONE launch
// per step (tails28_impl.cuh: merge, heavy merge, row / column sums, bit sums; blockIdx.z = MSM) -- a one-stream proof runs
// the tails of its four G1 MSMs side by side.  outs[i] receives msm_parts_count() partial sums (msm_parts_finish on the host).
// Returns false -- nothing queued -- when the MSMs cannot share launches (different bucket layouts, an empty one, 32-bit
// slots): the caller then runs msm_reduce_phase per MSM.
// after_merge (optional, count == 1): called with the complete local bucket array (slots) between the merge and the
// reduction -- the bucket-level cross-GPU exchange of comm_impl.cuh.
template <class F, class Hook = std::nullptr_t>
static bool msm_tails28_launch(ark355_ctx* ctx, int count, const MsmSort* const* sorts, MsmBuckets* const* bks,
                               XYZZ<F>* const* outs, hipStream_t stream, Hook after_merge = nullptr) {
  (void)ctx;
  using P = typename Tail28Of<F>::P;
  constexpr bool G2 = Tail28Of<F>::G2;
  constexpr bool HOOK = !std::is_same<Hook, std::nullptr_t>::value;
  using T = Tail28<P, G2>;
  using Slot = typename T::Slot;
  if (count < 1 || count > TAIL28_MAX) return false;
  const MsmPlan& p0 = sorts[0]->plan;
  // (an EMPTY shard still takes part in the bucket exchange: all-infinity bucket array, no merge)
  const bool empty_with_hook = HOOK && count == 1 && p0.n == 0;
  for (int i = 0; i < count; i++) {
    const MsmPlan& p = sorts[i]->plan;
    if ((p.n == 0 && !empty_with_hook) || !bks[i]->fmt) return false;
    if (p.total_buckets != p0.total_buckets || p.buckets_per_window != p0.buckets_per_window || p.key_windows != p0.key_windows ||
        p.c != p0.c)
      return false;
  }
  ARK_REQUIRE(p0.c >= 3 && p0.buckets_per_window == (1u << (p0.c - 1)), ARK355_EINVAL, "bucket layout the 28-bit tails do not know");
  uint32_t lb, hb;
  tails28_split(p0.c, &lb, &hb);
  const uint32_t L = 1u << lb, H = 1u << hb;
  TailBatch28 tb;
  memset(&tb, 0, sizeof(tb));
  uint32_t max_heavy_all = 1;
  for (int i = 0; i < count; i++) {
    const MsmSort& s = *sorts[i];
    MsmBuckets& b = *bks[i];
    const uint32_t segs = b.segs;
    // at most entries / (MSM_HEAVY_SPAN * segment length) buckets can be heavy; "heavy" is relative: twice the average span
    // of a bucket once that exceeds the fixed threshold
    const uint32_t max_heavy = segs / MSM_HEAVY_SPAN + 1;
    const uint32_t avg_span = (uint32_t)(((uint64_t)segs + p0.total_buckets - 1) / p0.total_buckets);
    const uint32_t heavy_span = (2 * avg_span > MSM_HEAVY_SPAN) ? 2 * avg_span : MSM_HEAVY_SPAN;
    if (max_heavy > max_heavy_all) max_heavy_all = max_heavy;
    b.heavy_count.ensure(16);
    b.heavy_list.ensure((size_t)max_heavy * 4);
    b.rc.ensure((size_t)p0.key_windows * (H + L) * sizeof(Slot));
    if (!b.heavy_cleared) ARK_CHECK_HIP(hipMemsetAsync(b.heavy_count.p, 0, 4, stream));
    b.heavy_cleared = false;
    TailJob28& J = tb.j[i];
    J.offsets = s.offsets.as<uint32_t>();
    J.counts = s.counts.as<uint32_t>();
    J.buckets = b.buckets.p;
    J.head = b.head.p;
    J.head_key = b.head_key.as<uint32_t>();
    J.tail = b.tail.p;
    J.tail_key = b.tail_key.as<uint32_t>();
    J.heavy_count = b.heavy_count.as<uint32_t>();
    J.heavy_list = b.heavy_list.as<uint32_t>();
    J.rc = b.rc.p;
    J.out = outs[i];
    J.seg_len = b.seg_len;
    J.heavy_span = heavy_span;
  }
  if (empty_with_hook) {
    bks[0]->buckets.ensure((size_t)p0.total_buckets * sizeof(Slot));
    tb.j[0].buckets = bks[0]->buckets.p;
    ARK_CHECK_HIP(hipMemsetAsync(bks[0]->buckets.p, 0, (size_t)p0.total_buckets * sizeof(Slot), stream));
  } else {
    const uint32_t grid_b = (uint32_t)(((uint64_t)p0.total_buckets * T::LPI + MSM_THREADS - 1) / MSM_THREADS);
    ARK_LAUNCH((msm_merge28_kernel<P, G2>), dim3(grid_b, 1, (uint32_t)count), dim3(MSM_THREADS), 0, stream, tb, p0.total_buckets);
    ARK_CHECK_LAUNCH();
    const uint32_t grid_h = max_heavy_all < ARK_MSM_HEAVY_GRID ? max_heavy_all : ARK_MSM_HEAVY_GRID;
    ARK_LAUNCH((msm_merge_heavy28_kernel<P, G2>), dim3(grid_h, 1, (uint32_t)count), dim3(MSM_THREADS), 0, stream, tb);
    ARK_CHECK_LAUNCH();
  }
  if constexpr (HOOK) {
    ARK_REQUIRE(count == 1, ARK355_EINVAL, "the bucket exchange runs per MSM");
    after_merge(bks[0]->buckets.p, p0.total_buckets, stream);
  }
  ARK_LAUNCH((msm_selsum28_kernel<P, G2, 0>), dim3(H + L, p0.key_windows, (uint32_t)count), dim3(64), 0, stream, tb, lb, hb);
  ARK_CHECK_LAUNCH();
  ARK_LAUNCH((msm_selsum28_kernel<P, G2, 1>), dim3(p0.c, p0.key_windows, (uint32_t)count), dim3(64), 0, stream, tb, lb, hb);
  ARK_CHECK_LAUNCH();
  return true;
}

// G2 tails run on lane pairs (round 2; the one-lane-per-bucket kernels with out-of-line Fq2 arithmetic they replaced took
// 2.5x as long and are gone -- profiles/r02_g2_pair_tails_ab.txt)
static inline bool msm_g2_pair_tails(const ark355_ctx*) { return true; }

// Phase 2: straddling-run merge, weighted bucket reduction, window combination; writes/accumulates the XYZZ
// result into d_out.  Only a handful of workgroups and latency-bound, so the prover runs it on its own stream
// underneath the next MSM's accumulation.
// after_merge (optional): called with the complete local bucket array between the merge and the weighted reduction
// (the bucket-level cross-GPU exchange of comm_impl.cuh).
template <class F, class Hook = std::nullptr_t>
static void msm_reduce_phase(ark355_ctx* ctx, const MsmSort& s, MsmBuckets& b, XYZZ<F>* d_out, int accumulate,
                             hipStream_t stream, Hook after_merge = nullptr) {
  const MsmPlan& p = s.plan;
  constexpr bool HOOK = !std::is_same<Hook, std::nullptr_t>::value;
  if (b.fmt) {
    // buckets in 28-bit slots (every MSM over resident window tables): the round-6 tails; d_out receives msm_parts_count()
    // partial sums
    ARK_REQUIRE(!accumulate, ARK355_EINVAL, "the 28-bit tails do not accumulate into a previous result");
    if (p.n == 0 && !HOOK) {
      ARK_CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)msm_parts_count(p, b.fmt) * sizeof(XYZZ<F>), stream));
      return;
    }
    const MsmSort* sorts[1] = {&s};
    MsmBuckets* bks[1] = {&b};
    XYZZ<F>* outs[1] = {d_out};
    const bool ok = msm_tails28_launch<F, Hook>(ctx, 1, sorts, bks, outs, stream, after_merge);
    ARK_REQUIRE(ok, ARK355_EINVAL, "28-bit tails refused a bucket set");
    return;
  }
  if (p.n == 0) {
    if constexpr (HOOK) {
      // an empty shard still takes part in the exchange: all-infinity bucket array
      b.buckets.ensure((size_t)p.total_buckets * sizeof(XYZZ<F>));
      ARK_CHECK_HIP(hipMemsetAsync(b.buckets.p, 0, (size_t)p.total_buckets * sizeof(XYZZ<F>), stream));
    } else {
      if (!accumulate) ARK_CHECK_HIP(hipMemsetAsync(d_out, 0, sizeof(XYZZ<F>), stream));
      return;
    }
  } else {
    const uint32_t segs = b.segs;
    const uint32_t grid_b = (p.total_buckets + MSM_THREADS - 1) / MSM_THREADS;
    // at most entries / (MSM_HEAVY_SPAN * segment length) buckets can be heavy
    const uint32_t max_heavy = segs / MSM_HEAVY_SPAN + 1;
    // "heavy" is relative: twice the average span of a bucket once that exceeds the fixed threshold
    const uint32_t avg_span = (uint32_t)(((uint64_t)segs + p.total_buckets - 1) / p.total_buckets);
    const uint32_t heavy_span = (2 * avg_span > MSM_HEAVY_SPAN) ? 2 * avg_span : MSM_HEAVY_SPAN;
    b.heavy_count.ensure(16);
    b.heavy_list.ensure((size_t)max_heavy * 4);
    if (!b.heavy_cleared) ARK_CHECK_HIP(hipMemsetAsync(b.heavy_count.p, 0, 4, stream));
    b.heavy_cleared = false;
    if constexpr (is_fp2<F>::value) {
      if (msm_g2_pair_tails(ctx)) {
        using P = typename F::Base::Params;
        ARK_LAUNCH((msm_merge_pair_kernel<P>), dim3(2 * grid_b), dim3(MSM_THREADS), 0, stream, p.total_buckets,
                   s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(), b.head.as<XYZZ<F>>(),
                   b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(),
                   b.heavy_count.as<uint32_t>(), b.heavy_list.as<uint32_t>(), b.seg_len, heavy_span);
      } else {
        ARK_LAUNCH((msm_merge_kernel<F>), dim3(grid_b), dim3(MSM_THREADS), 0, stream, p.total_buckets,
                   s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(), b.head.as<XYZZ<F>>(),
                   b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(),
                   b.heavy_count.as<uint32_t>(), b.heavy_list.as<uint32_t>(), b.seg_len, heavy_span);
      }
    } else {
      ARK_LAUNCH((msm_merge_kernel<F>), dim3(grid_b), dim3(MSM_THREADS), 0, stream, p.total_buckets,
                 s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(), b.head.as<XYZZ<F>>(),
                 b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(),
                 b.heavy_count.as<uint32_t>(), b.heavy_list.as<uint32_t>(), b.seg_len, heavy_span);
    }
    ARK_CHECK_LAUNCH();
    const uint32_t grid_h = max_heavy < ARK_MSM_HEAVY_GRID ? max_heavy : ARK_MSM_HEAVY_GRID;
    bool heavy_done = false;
    if constexpr (is_fp2<F>::value) {
      if (msm_g2_pair_tails(ctx)) {
        using P = typename F::Base::Params;
        ARK_LAUNCH((msm_merge_heavy_pair_kernel<P>), dim3(grid_h), dim3(MSM_THREADS), 0, stream, b.heavy_count.as<uint32_t>(),
                   b.heavy_list.as<uint32_t>(), s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(),
                   b.head.as<XYZZ<F>>(), b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(), b.seg_len);
        heavy_done = true;
      }
    }
    if (!heavy_done)
      ARK_LAUNCH((msm_merge_heavy_kernel<F>), dim3(grid_h), dim3(MSM_THREADS), 0, stream, b.heavy_count.as<uint32_t>(),
                 b.heavy_list.as<uint32_t>(), s.offsets.as<uint32_t>(), s.counts.as<uint32_t>(), b.buckets.as<XYZZ<F>>(),
                 b.head.as<XYZZ<F>>(), b.head_key.as<uint32_t>(), b.tail.as<XYZZ<F>>(), b.tail_key.as<uint32_t>(), b.seg_len);
    ARK_CHECK_LAUNCH();
  }
  if constexpr (HOOK) after_merge(b.buckets.p, p.total_buckets, stream);

#ifndef ARK_MSM_TWO_LEVEL_MIN
#define ARK_MSM_TWO_LEVEL_MIN (1u << 17)        // bucket count from which the two-level reduction is used (tests: small)
#endif
  const uint32_t two_level_min = ctx->policy.msm_two_level_min >= 0 ? (uint32_t)ctx->policy.msm_two_level_min
                                                                     : (uint32_t)ARK_MSM_TWO_LEVEL_MIN;       // A/B knob
  if (p.key_windows == 1 && p.total_buckets >= two_level_min) {
    const uint32_t items = (p.total_buckets + MSM_RED_L1 - 1) / MSM_RED_L1;
    b.lvl_t.ensure((size_t)items * sizeof(XYZZ<F>));
    b.lvl_w.ensure((size_t)items * sizeof(XYZZ<F>));
    ARK_LAUNCH((msm_reduce_l1_kernel<F>), dim3((items + MSM_THREADS - 1) / MSM_THREADS), dim3(MSM_THREADS), 0, stream,
               (const XYZZ<F>*)b.buckets.as<XYZZ<F>>(), p.total_buckets, b.lvl_t.as<XYZZ<F>>(), b.lvl_w.as<XYZZ<F>>());
    ARK_CHECK_LAUNCH();
    const uint32_t chunks2 = (items + MSM_RED_K - 1) / MSM_RED_K;
    const uint32_t blocks2 = (chunks2 + MSM_THREADS - 1) / MSM_THREADS;
    b.partials.ensure((size_t)blocks2 * sizeof(XYZZ<F>));
    ARK_LAUNCH((msm_reduce_l2_kernel<F>), dim3(blocks2), dim3(MSM_THREADS), 0, stream, (const XYZZ<F>*)b.lvl_t.as<XYZZ<F>>(),
               (const XYZZ<F>*)b.lvl_w.as<XYZZ<F>>(), items, b.partials.as<XYZZ<F>>());
    ARK_CHECK_LAUNCH();
    ARK_LAUNCH((msm_combine_kernel<F>), dim3(1), dim3(64), 0, stream, b.partials.as<XYZZ<F>>(), blocks2, 1u, p.c, d_out,
               accumulate);
    ARK_CHECK_LAUNCH();
    return;
  }
  const uint32_t chunks = (p.buckets_per_window + MSM_RED_K - 1) / MSM_RED_K;
  if constexpr (is_fp2<F>::value) {
    if (msm_g2_pair_tails(ctx)) {
      using P = typename F::Base::Params;
      const uint32_t blocks_pw = (chunks + MSM_THREADS / 2 - 1) / (MSM_THREADS / 2);       // a lane pair per chunk
      b.partials.ensure((size_t)blocks_pw * p.key_windows * sizeof(XYZZ<F>));
      ARK_LAUNCH((msm_reduce_pair_kernel<P>), dim3(blocks_pw, p.key_windows), dim3(MSM_THREADS), 0, stream,
                 (const XYZZ<F>*)b.buckets.as<XYZZ<F>>(), p.buckets_per_window, b.partials.as<XYZZ<F>>());
      ARK_CHECK_LAUNCH();
      if (p.key_windows == 1) {
        ARK_LAUNCH((msm_combine_pair_kernel<P>), dim3(1), dim3(64), 0, stream, (const XYZZ<F>*)b.partials.as<XYZZ<F>>(),
                   blocks_pw, d_out, accumulate);
      } else {
        ARK_REQUIRE(p.key_windows <= 64, ARK355_EINVAL, "window count exceeds one wave");
        ARK_LAUNCH((msm_combine_kernel<F>), dim3(1), dim3(64), 0, stream, b.partials.as<XYZZ<F>>(), blocks_pw,
                   p.key_windows, p.c, d_out, accumulate);
      }
      ARK_CHECK_LAUNCH();
      return;
    }
  }
  const uint32_t blocks_per_window = (chunks + MSM_THREADS - 1) / MSM_THREADS;
  b.partials.ensure((size_t)blocks_per_window * p.key_windows * sizeof(XYZZ<F>));
  ARK_LAUNCH((msm_reduce_kernel<F>), dim3(blocks_per_window, p.key_windows), dim3(MSM_THREADS), 0, stream,
             b.buckets.as<XYZZ<F>>(), p.buckets_per_window, b.partials.as<XYZZ<F>>());
  ARK_CHECK_LAUNCH();
  ARK_REQUIRE(p.key_windows <= 64, ARK355_EINVAL, "window count exceeds one wave");
  ARK_LAUNCH((msm_combine_kernel<F>), dim3(1), dim3(64), 0, stream, b.partials.as<XYZZ<F>>(), blocks_per_window,
             p.key_windows, p.c, d_out, accumulate);
  ARK_CHECK_LAUNCH();
}

template <class F>
static bool msm_reduce_phase_batch(ark355_ctx* ctx, int count, const MsmSort* const* sorts, MsmBuckets* const* bks,
                                   XYZZ<F>* const* outs, hipStream_t stream) {
  if (count < 2) return false;
  return msm_tails28_launch<F>(ctx, count, sorts, bks, outs, stream);
}

// Both phases on one stream.
template <class F>
static void msm_buckets(ark355_ctx* ctx, const MsmSort& s, MsmBuckets& b, const Affine<F>* d_bases, XYZZ<F>* d_out,
                        int accumulate, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                        int fmt = 0) {
  msm_accumulate_phase<F>(ctx, s, b, d_bases, stream, ev0, ev1, fmt);
  msm_reduce_phase<F>(ctx, s, b, d_out, accumulate, stream);
}

}  // namespace ark355
