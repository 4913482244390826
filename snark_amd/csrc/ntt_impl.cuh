// Radix-2 NTT over Fr for gfx950: Stockham auto-sort passes of radix 2^r (r <= 9).  A workgroup owns a tile of
// R x P elements (R = 2^r, up to 2048 elements); every lane keeps EIGHT of them in registers and runs up to three
// butterfly stages there (a radix-8 group) before the tile is re-dealt through LDS for the next group -- one LDS round
// trip per three stages instead of one per stage.  The round trips move 16-byte halves (limbs 0-3, then 4-7), so a
// 2048-element tile needs 32 KiB of LDS, four workgroups share a CU, and every LDS access is a conflict-free b128.
//
// Replaces ark-poly `Radix2EvaluationDomain::{fft,ifft}_in_place` and the coset variants (un-vendored crate
// ark-poly/src/domain/radix2/fft.rs) for the 7 transforms of the Groth16 witness map (SURVEY.md 3.1, Appendix A
// steps 2-5).  Natural order in, natural order out:  X[k] = sum_j x[j] w^(jk).
//
// One pass with current stride s (sequence length n = N/s, m = n/R):
//   y[q + s(Rp + k)] = w_N^(s p k) * sum_t x[q + s p + (N/R) t] * w_R^(t k),   q < s, p < m, k < R
// The kernel is bound by the integer multiplier, not by HBM (an Fr multiplication is 136 v_mad/v_mul_u32 against 64 B
// moved per element and pass), so the design removes multiplications wherever the structure allows:
//   * butterflies in the LAST group of a pass use compile-time twiddle indices; the trivial ones (w^0) are skipped;
//   * the twiddle between passes comes from a direct table whenever it has <= 2^16 entries (every pass but the first:
//     one multiplication instead of the two of the hi/lo composition w^e = hi[e >> LO] * lo[e & mask]);
//   * inverse NTT -> coset NTT is ONE kernel at the seam: the last pass of the inverse transform leaves its tile in
//     LDS, scales it by g^j / N (one composed factor instead of 1/N and g^j separately) and runs the first pass of
//     the coset transform on it -- the tile of a last pass and of a first pass of the same radix are the same index set;
// HBM traffic: one read + one write of the vector per pass (64 B per element per pass).
#pragma once
#include "common.h"

namespace ark355 {

#ifndef ARK_NTT_EMAX_LOG
#define ARK_NTT_EMAX_LOG 11   // elements per workgroup tile (2^11 x 16 B = 32 KiB exchange buffer)
#endif
#ifndef ARK_NTT_RMAX_LOG
#define ARK_NTT_RMAX_LOG 9    // largest radix 2^r of one pass (tests shrink it to force many passes)
#endif
constexpr uint32_t NTT_EMAX_LOG = ARK_NTT_EMAX_LOG;
constexpr uint32_t NTT_RMAX_LOG = ARK_NTT_RMAX_LOG;
constexpr uint32_t NTT_THREADS = 256;
#ifndef ARK_NTT_WAVES
#define ARK_NTT_WAVES 3       // waves per SIMD the pass kernels' register budget is sized for (3: 168 VGPRs, 2: 256)
#endif
// Direct inter-pass twiddle tables up to 2^23 entries (256 MiB each; MI355X has the HBM for it).  The kernel is bound by
// the integer multiplier, so a 32-byte table read per element (+50 % of that pass's traffic) is cheaper than the second
// multiplication of the hi/lo composition: the FIRST pass of a 2^21-point transform has a table of 2^21 entries.
constexpr uint32_t NTT_DIRECT_MAX_LOG = 23;
static_assert(NTT_RMAX_LOG >= 1 && NTT_RMAX_LOG <= 9, "pass radix 2^1 .. 2^9");
static_assert((1u << NTT_EMAX_LOG) == 8 * NTT_THREADS, "eight elements per lane");

struct NttTables {
  uint32_t log_n = 0, lo_bits = 0;
  DevBuf w_lo, w_hi, wi_lo, wi_hi;     // w^e, w^-e
  DevBuf g_lo, g_hi, gi_lo, gi_hi;     // g^j ; g^-j / N
  DevBuf seam;                         // g^j / N, j < N (the seam of the fused inverse -> coset kernel)
  DevBuf n_inv;                        // 1/N
  DevBuf gz_hi, zconst;                // the quotient step of the witness map (ntt_quotient_tables): gi_hi scaled by 1 / (g^N - 1);
                                       // zconst[0] = 1 / (g^N - 1), zconst[1] = that / N
  DevBuf tw_r[2][10];                  // [inverse][r]: w_R^e, e < R/2   (in-tile butterflies)
  std::map<uint32_t, DevBuf> direct;   // (inverse << 16 | s_log << 8 | r) -> w^(s p k), index (p << r) | k
  std::mutex mu;                       // the lazily built members (tw_r, direct, seam): contexts of one device share a set
};

template <class Fr>
ARK_D Fr pow_lookup(const Fr* lo, const Fr* hi, uint32_t lo_bits, uint64_t e) {
  const uint64_t h = e >> lo_bits;
  const uint32_t l = (uint32_t)(e & ((1ull << lo_bits) - 1));
  // lo[0] == 1 always; hi[0] may carry a folded constant (1/N), so only the l == 0 shortcut is valid
  if (l == 0) return hi[h];
  return Fr::mul(hi[h], lo[l]);
}

ARK_D uint32_t bitrev_bits(uint32_t v, uint32_t bits) {
#if defined(ARK_EMUL)
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) {
    r = (r << 1) | (v & 1);
    v >>= 1;
  }
  return r;
#else
  return bits ? (__brev(v) >> (32u - bits)) : 0u;
#endif
}

// ---- the tile: eight elements per lane, radix-8 groups, LDS re-deals ---------------------------------------------------
template <class Fr>
struct NttLane {
  static_assert(Fr::N == 8, "Fr is eight 32-bit limbs for both curves");
  Fr x[8];
};

// Re-deal the tile: lane slot i leaves position from[i] and takes the element at position to[i].  Two 16-byte halves.
template <class Fr>
ARK_D void ntt_exchange(NttLane<Fr>& v, uint4* xbuf, const uint32_t (&from)[8], const uint32_t (&to)[8], bool active) {
#pragma unroll
  for (int half = 0; half < 2; half++) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        xbuf[from[i]] = make_uint4(v.x[i].l[4 * half + 0], v.x[i].l[4 * half + 1], v.x[i].l[4 * half + 2], v.x[i].l[4 * half + 3]);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint4 t = xbuf[to[i]];
        v.x[i].l[4 * half + 0] = t.x;
        v.x[i].l[4 * half + 1] = t.y;
        v.x[i].l[4 * half + 2] = t.z;
        v.x[i].l[4 * half + 3] = t.w;
      }
    }
    __syncthreads();
  }
}

template <class Fr>
ARK_D Fr ntt_tw_load(const uint4* tw, uint32_t e) {
  const uint4 a = tw[2 * e], b = tw[2 * e + 1];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}

// Tile geometry of one pass, shared by every lane.
struct NttGeo {
  uint32_t p_log;     // log2 P (columns of the tile)
  uint32_t nt;        // active lanes = R P / 8
};

// positions (t P + c) of this lane's eight slots in the group that starts at stage ST with G stages:
// sub-transform f = (blk u + kk_low) P + c, f = tid + h NT for h < 8 / 2^G; slot h 2^G + j holds t = blk u 2^G + kk_low + j u
template <int R_LOG, int ST, int G>
ARK_D void ntt_group_pos(const NttGeo& geo, uint32_t tid, uint32_t (&pos)[8], uint32_t (&kk)[8]) {
  constexpr uint32_t SUB = 1u << G, H = 8u >> G;
  constexpr uint32_t U_LOG = R_LOG - ST - G, U = 1u << U_LOG;
#pragma unroll
  for (uint32_t h = 0; h < H; h++) {
    const uint32_t f = tid + h * geo.nt;
    const uint32_t c = f & ((1u << geo.p_log) - 1u), sf = f >> geo.p_log;
    const uint32_t kk_low = sf & (U - 1u), blk = sf >> U_LOG;
#pragma unroll
    for (uint32_t j = 0; j < SUB; j++) {
      const uint32_t t = (blk << (U_LOG + G)) + kk_low + (j << U_LOG);
      pos[h * SUB + j] = (t << geo.p_log) + c;
      kk[h * SUB + j] = kk_low;
    }
  }
}

// G radix-2 DIF stages (ST .. ST+G-1) on the lane's registers
template <class Fr, int R_LOG, int ST, int G>
ARK_D void ntt_group_butterflies(NttLane<Fr>& v, const uint32_t (&kk)[8], const uint4* tw) {
  constexpr uint32_t SUB = 1u << G, H = 8u >> G;
  constexpr uint32_t U_LOG = R_LOG - ST - G;
#pragma unroll
  for (int a = 0; a < G; a++) {
    const uint32_t d = SUB >> (a + 1);
#pragma unroll
    for (uint32_t h = 0; h < H; h++) {
#pragma unroll
      for (uint32_t j = 0; j < SUB; j++) {
        if (j & d) continue;
        const uint32_t lo = h * SUB + j, hi = lo + d;
        const Fr u = v.x[lo], w = v.x[hi];
        v.x[lo] = Fr::add(u, w);
        Fr dif = Fr::sub(u, w);
        // twiddle exponent (t mod half') << stage; with U == 1 (last group of the pass) it is a compile-time constant
        // and the trivial ones vanish
        if (U_LOG == 0) {
          const uint32_t e = (j & (d - 1u)) << (ST + a);
          if (e != 0) dif = Fr::mul(dif, ntt_tw_load<Fr>(tw, e));
        } else {
          const uint32_t e = (kk[lo] + ((j & (d - 1u)) << U_LOG)) << (ST + a);
          dif = Fr::mul(dif, ntt_tw_load<Fr>(tw, e));
        }
        v.x[hi] = dif;
      }
    }
  }
}

// groups of a radix-2^R_LOG pass: full groups of three stages, then the remainder
template <int R_LOG>
struct NttGroups {
  static constexpr int FULL = R_LOG / 3, REM = R_LOG % 3;
  static constexpr int COUNT = FULL + (REM ? 1 : 0);
};

// The in-tile R-point transforms.  On entry the lane holds the slots of group 0 (positions pos); on exit it holds the
// slots of the LAST group (positions returned in pos); output k of a column sits at position bitrev_r(k).
template <class Fr, int R_LOG>
ARK_D void ntt_tile_dft(NttLane<Fr>& v, const NttGeo& geo, uint32_t tid, bool active, uint4* xbuf, const uint4* tw,
                        uint32_t (&pos)[8]) {
  uint32_t kk[8], npos[8];
  constexpr int G0 = R_LOG >= 3 ? 3 : R_LOG;
  ntt_group_pos<R_LOG, 0, G0>(geo, tid, pos, kk);
  if (active) ntt_group_butterflies<Fr, R_LOG, 0, G0>(v, kk, tw);
  if constexpr (R_LOG > 3) {
    constexpr int G1 = R_LOG >= 6 ? 3 : R_LOG - 3;
    ntt_group_pos<R_LOG, 3, G1>(geo, tid, npos, kk);
    ntt_exchange<Fr>(v, xbuf, pos, npos, active);
#pragma unroll
    for (int i = 0; i < 8; i++) pos[i] = npos[i];
    if (active) ntt_group_butterflies<Fr, R_LOG, 3, G1>(v, kk, tw);
  }
  if constexpr (R_LOG > 6) {
    constexpr int G2 = R_LOG - 6;
    ntt_group_pos<R_LOG, 6, G2>(geo, tid, npos, kk);
    ntt_exchange<Fr>(v, xbuf, pos, npos, active);
#pragma unroll
    for (int i = 0; i < 8; i++) pos[i] = npos[i];
    if (active) ntt_group_butterflies<Fr, R_LOG, 6, G2>(v, kk, tw);
  }
}

// first group's positions only (for the loads)
template <int R_LOG>
ARK_D void ntt_first_pos(const NttGeo& geo, uint32_t tid, uint32_t (&pos)[8]) {
  uint32_t kk[8];
  constexpr int G0 = R_LOG >= 3 ? 3 : R_LOG;
  ntt_group_pos<R_LOG, 0, G0>(geo, tid, pos, kk);
}

struct NttPassArgs {
  const void* in;        // Fr*
  void* out;
  uint32_t log_n, s_log, p_log;
  const void* tw;        // w_R^e, e < R/2 (Fr, global)
  const void* direct;    // optional direct inter-pass table
  const void* w_lo;      // hi/lo tables of w (composition when no direct table)
  const void* w_hi;
  uint32_t lo_bits;
  // several vectors of one shape in ONE launch (the witness map's a, b, c): vector y = blockIdx.y reads in + y * batch_stride
  // and writes out + y * batch_stride (elements); 0 for a single vector
  uint64_t batch_stride;
};

template <class Fr>
ARK_D void ntt_stage_twiddles(uint4* tw_lds, const Fr* tw_glob, uint32_t half_r, uint32_t tid) {
  const uint4* src = reinterpret_cast<const uint4*>(tw_glob);
  for (uint32_t i = tid; i < 2 * half_r; i += NTT_THREADS) tw_lds[i] = src[i];
}

// load this lane's eight inputs of a pass whose tile starts at column pq0
template <class Fr>
ARK_D void ntt_load_inputs(NttLane<Fr>& v, const NttPassArgs& a, const uint32_t (&pos)[8], uint64_t pq0, uint32_t cols_log,
                           bool active) {
  if (!active) return;
  const Fr* in = reinterpret_cast<const Fr*>(a.in) + (uint64_t)blockIdx.y * a.batch_stride;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t c = pos[i] & ((1u << a.p_log) - 1u), t = pos[i] >> a.p_log;
    v.x[i] = in[pq0 + c + ((uint64_t)t << cols_log)];
  }
}

// Output mapping of a pass: slot i is output index idx = tid + i NT of the tile, (k, c) chosen so that the global
// stores of consecutive lanes are contiguous; the value sits at position bitrev_r(k) P + c after the stages.
template <int R_LOG>
ARK_D void ntt_out_pos(const NttGeo& geo, uint32_t tid, uint32_t s_log, uint32_t (&pos)[8], uint32_t (&kout)[8],
                       uint32_t (&cout)[8]) {
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) {
    const uint32_t idx = tid + i * geo.nt;
    uint32_t k, c;
    if (s_log < geo.p_log) {
      k = idx & ((1u << R_LOG) - 1u);
      c = idx >> R_LOG;
    } else {
      c = idx & ((1u << geo.p_log) - 1u);
      k = idx >> geo.p_log;
    }
    kout[i] = k;
    cout[i] = c;
    pos[i] = (bitrev_bits(k, R_LOG) << geo.p_log) + c;
  }
}

// inter-pass twiddle w^(s p k) from the direct table (when the pass has one) and store.
template <class Fr, int R_LOG>
ARK_D void ntt_store_outputs(NttLane<Fr>& v, const NttPassArgs& a, const uint32_t (&kout)[8], const uint32_t (&cout)[8],
                             uint64_t pq0, bool active) {
  if (!active) return;
  Fr* out = reinterpret_cast<Fr*>(a.out) + (uint64_t)blockIdx.y * a.batch_stride;
  const uint64_t s_mask = (1ull << a.s_log) - 1ull;
  if (a.direct) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t p = (pq0 + cout[i]) >> a.s_log;
      v.x[i] = Fr::mul(v.x[i], reinterpret_cast<const Fr*>(a.direct)[(p << R_LOG) | kout[i]]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t pq = pq0 + cout[i];
    const uint64_t q = pq & s_mask, p = pq >> a.s_log;
    out[q + ((p * (1u << R_LOG) + kout[i]) << a.s_log)] = v.x[i];
  }
}

// One pass.  grid = N / (R P) tiles, NTT_THREADS lanes, dynamic LDS = exchange buffer (R P x 16 B) + twiddles (R/2 x 32 B)
template <class Fr, int R_LOG>
__global__ void __launch_bounds__(NTT_THREADS, ARK_NTT_WAVES)
ntt_pass_kernel(const NttPassArgs a) {
  ARK_DYN_SMEM(uint4, lds);
  const uint32_t tid = threadIdx.x;
  NttGeo geo;
  geo.p_log = a.p_log;
  geo.nt = (1u << (R_LOG + a.p_log)) >> 3;
  const bool active = tid < geo.nt;
  uint4* xbuf = lds;
  uint4* tw = lds + ((size_t)1 << (R_LOG + a.p_log));
  ntt_stage_twiddles<Fr>(tw, reinterpret_cast<const Fr*>(a.tw), (1u << R_LOG) >> 1, tid);
  const uint32_t cols_log = a.log_n - R_LOG;
  const uint64_t pq0 = (uint64_t)blockIdx.x << a.p_log;
  NttLane<Fr> v;
  uint32_t pos[8];
  ntt_first_pos<R_LOG>(geo, tid, pos);
  ntt_load_inputs<Fr>(v, a, pos, pq0, cols_log, active);
  __syncthreads();                                   // twiddles staged
  ntt_tile_dft<Fr, R_LOG>(v, geo, tid, active, xbuf, tw, pos);
  uint32_t opos[8], kout[8], cout[8];
  ntt_out_pos<R_LOG>(geo, tid, a.s_log, opos, kout, cout);
  ntt_exchange<Fr>(v, xbuf, pos, opos, active);
  ntt_store_outputs<Fr, R_LOG>(v, a, kout, cout, pq0, active);
}

// The seam of inverse NTT -> coset NTT in one kernel: LAST pass of the first transform (stride N/R), the factor
// seam[o] = g^o / N on its outputs (a direct table of N entries), FIRST pass of the second transform (stride 1) on the
// same tile.
//   a  : arguments of the last pass (its output scaling fields are ignored)
//   b  : arguments of the first pass (its `in` is ignored; tw / direct / w tables / out belong to the second transform)
template <class Fr, int R_LOG>
__global__ void __launch_bounds__(NTT_THREADS, ARK_NTT_WAVES)       // 56 KiB of LDS per workgroup: three per CU
ntt_seam_kernel(const NttPassArgs a, const NttPassArgs b, const Fr* __restrict__ seam) {
  ARK_DYN_SMEM(uint4, lds);
  const uint32_t tid = threadIdx.x;
  NttGeo geo;
  geo.p_log = a.p_log;
  geo.nt = (1u << (R_LOG + a.p_log)) >> 3;
  const bool active = tid < geo.nt;
  uint4* xbuf = lds;
  uint4* tw = lds + ((size_t)1 << (R_LOG + a.p_log));
  uint4* tw2 = tw + (1u << R_LOG);                   // second transform's twiddles (R/2 x 2 uint4 each)
  ntt_stage_twiddles<Fr>(tw, reinterpret_cast<const Fr*>(a.tw), (1u << R_LOG) >> 1, tid);
  ntt_stage_twiddles<Fr>(tw2, reinterpret_cast<const Fr*>(b.tw), (1u << R_LOG) >> 1, tid);
  const uint32_t cols_log = a.log_n - R_LOG;          // == a.s_log: last pass
  const uint64_t pq0 = (uint64_t)blockIdx.x << a.p_log;
  NttLane<Fr> v;
  uint32_t pos[8];
  ntt_first_pos<R_LOG>(geo, tid, pos);
  ntt_load_inputs<Fr>(v, a, pos, pq0, cols_log, active);
  __syncthreads();
  ntt_tile_dft<Fr, R_LOG>(v, geo, tid, active, xbuf, tw, pos);
  // outputs of the last pass: p = 0 (no inter-pass twiddle), o = pq + k (N/R).  Re-deal straight into the slots of the
  // next transform's first group: its input (t, c) is the value with k = t, i.e. the one at position bitrev_r(t) P + c.
  uint32_t npos[8], from[8];
  ntt_first_pos<R_LOG>(geo, tid, npos);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t c = npos[i] & ((1u << geo.p_log) - 1u), t = npos[i] >> geo.p_log;
    from[i] = (bitrev_bits(t, R_LOG) << geo.p_log) + c;
  }
  ntt_exchange<Fr>(v, xbuf, pos, from, active);
  if (active) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t c = npos[i] & ((1u << geo.p_log) - 1u), t = npos[i] >> geo.p_log;
      v.x[i] = Fr::mul(v.x[i], seam[pq0 + c + ((uint64_t)t << cols_log)]);
    }
  }
  ntt_tile_dft<Fr, R_LOG>(v, geo, tid, active, xbuf, tw2, npos);
  uint32_t opos[8], kout[8], cout[8];
  ntt_out_pos<R_LOG>(geo, tid, b.s_log, opos, kout, cout);
  ntt_exchange<Fr>(v, xbuf, npos, opos, active);
  ntt_store_outputs<Fr, R_LOG>(v, b, kout, cout, pq0, active);
}

// ---- host side -----------------------------------------------------------------------------------
template <class Fr>
static Fr fr_pow2k(Fr x, uint32_t k) {          // x^(2^k)
  for (uint32_t i = 0; i < k; i++) x = Fr::sqr(x);
  return x;
}
template <class Fr>
static Fr fr_pow_u64(Fr x, uint64_t e) {
  Fr r = Fr::one();
  while (e) {
    if (e & 1) r = Fr::mul(r, x);
    x = Fr::sqr(x);
    e >>= 1;
  }
  return r;
}
template <class Fr>
static Fr fr_from_params(uint32_t (*f)(int)) {
  Fr r;
  for (int i = 0; i < Fr::N; i++) r.l[i] = f(i);
  return r;
}

template <class Fr>
static void upload_powers(DevBuf& dst, Fr base, uint64_t count, Fr scale_all) {
  std::vector<Fr> h(count);
  Fr cur = scale_all;
  for (uint64_t i = 0; i < count; i++) {
    h[i] = cur;
    cur = Fr::mul(cur, base);
  }
  dst.alloc(count * sizeof(Fr));
  ARK_CHECK_HIP(hipMemcpy(dst.p, h.data(), count * sizeof(Fr), hipMemcpyHostToDevice));
}

template <class Fr>
static Fr ntt_root(uint32_t log_n, bool inverse) {
  using P = typename Fr::Params;
  const Fr root = fr_from_params<Fr>(inverse ? &P::root_inv : &P::root);
  return fr_pow2k(root, P::TWO_ADICITY - log_n);
}

template <class Fr>
static NttTables* build_ntt_tables(uint32_t log_n) {
  using P = typename Fr::Params;
  ARK_REQUIRE(log_n <= (uint32_t)P::TWO_ADICITY, ARK355_E_POLY_DEGREE_TOO_LARGE,
              "domain size exceeds the two-adicity of Fr");
  auto* t = new NttTables();
  t->log_n = log_n;
  const uint32_t lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
  t->lo_bits = lo_bits;
  const Fr w = ntt_root<Fr>(log_n, false), wi = ntt_root<Fr>(log_n, true);
  const Fr g = fr_from_params<Fr>(&P::gen), gi = fr_from_params<Fr>(&P::gen_inv);
  // 1/N = (1/2)^log_n ; 1/2 = (p+1)/2
  Fr two = Fr::add(Fr::one(), Fr::one());
  Fr half = Fr::inv(two);
  Fr n_inv = fr_pow_u64(half, log_n);
  const uint64_t nlo = 1ull << lo_bits, nhi = 1ull << hi_bits;
  upload_powers(t->w_lo, w, nlo, Fr::one());
  upload_powers(t->w_hi, fr_pow2k(w, lo_bits), nhi, Fr::one());
  upload_powers(t->wi_lo, wi, nlo, Fr::one());
  upload_powers(t->wi_hi, fr_pow2k(wi, lo_bits), nhi, Fr::one());
  upload_powers(t->g_lo, g, nlo, Fr::one());
  upload_powers(t->g_hi, fr_pow2k(g, lo_bits), nhi, Fr::one());
  upload_powers(t->gi_lo, gi, nlo, Fr::one());
  upload_powers(t->gi_hi, fr_pow2k(gi, lo_bits), nhi, n_inv);   // 1/N folded into the high table
  t->n_inv.alloc(sizeof(Fr));
  ARK_CHECK_HIP(hipMemcpy(t->n_inv.p, &n_inv, sizeof(Fr), hipMemcpyHostToDevice));
  return t;
}

// Tables are read-only once built and identical for every context of a device: the proofs in flight (one context
// each) share one set -- 200 MiB at N = 2^21 -- through a per-process registry of weak references; a context keeps the
// sets it used alive.
template <class Curve>
static NttTables* get_ntt_tables(ark355_ctx* ctx, uint32_t log_n) {
  const uint32_t key = ((uint32_t)Curve::ID << 8) | log_n;
  auto it = ctx->ntt_tables.find(key);
  if (it != ctx->ntt_tables.end()) return it->second.get();
  static std::mutex reg_mu;
  static std::map<uint64_t, std::weak_ptr<NttTables>> registry;
  const uint64_t gkey = ((uint64_t)(uint32_t)ctx->device << 32) | key;
  std::shared_ptr<NttTables> t;
  {
    std::lock_guard<std::mutex> lk(reg_mu);
    t = registry[gkey].lock();
    if (!t) {
      t.reset(build_ntt_tables<typename Curve::Fr>(log_n));
      registry[gkey] = t;
    }
  }
  ctx->ntt_tables[key] = t;
  return t.get();
}

// w_R^e, e < R/2, R = 2^r (at least one entry so that the pointer is valid for r = 1: w^0)
template <class Fr>
static const Fr* ntt_tw_table(NttTables* t, uint32_t r, bool inverse) {
  std::lock_guard<std::mutex> lk(t->mu);
  DevBuf& b = t->tw_r[inverse ? 1 : 0][r];
  if (!b.p) {
    const Fr wr = fr_pow2k(ntt_root<Fr>(t->log_n, inverse), t->log_n - r);
    const uint64_t cnt = (1ull << r) >> 1;
    upload_powers(b, wr, cnt ? cnt : 1, Fr::one());
  }
  return b.as<Fr>();
}

// g^j / N for j < N
template <class Fr>
static const Fr* ntt_seam_table(NttTables* t) {
  std::lock_guard<std::mutex> lk(t->mu);
  if (!t->seam.p) {
    using P = typename Fr::Params;
    Fr n_inv;
    ARK_CHECK_HIP(hipMemcpy(&n_inv, t->n_inv.p, sizeof(Fr), hipMemcpyDeviceToHost));
    upload_powers(t->seam, fr_from_params<Fr>(&P::gen), 1ull << t->log_n, n_inv);
  }
  return t->seam.as<Fr>();
}

// The constants of the witness map's last step, h_k = (rho_k - c_k) / (g^N - 1) with rho = the inverse coset transform of
// a' b' (witness_impl.cuh): the high table of g^-k / N with 1 / (g^N - 1) folded in (the low table stays gi_lo), and the
// two constants a coefficient vector of c is scaled with (c as coefficients: zconst[0]; c as N c_k, an inverse transform
// without its 1/N: zconst[1]).
template <class Fr>
static void ntt_quotient_tables(NttTables* t, const Fr** gz_hi, const Fr** zconst) {
  std::lock_guard<std::mutex> lk(t->mu);
  if (!t->gz_hi.p) {
    using P = typename Fr::Params;
    Fr n_inv;
    ARK_CHECK_HIP(hipMemcpy(&n_inv, t->n_inv.p, sizeof(Fr), hipMemcpyDeviceToHost));
    const Fr zinv = Fr::inv(Fr::sub(fr_pow2k(fr_from_params<Fr>(&P::gen), t->log_n), Fr::one()));
    const Fr zc[2] = {zinv, Fr::mul(zinv, n_inv)};
    t->zconst.alloc(sizeof(zc));
    ARK_CHECK_HIP(hipMemcpy(t->zconst.p, zc, sizeof(zc), hipMemcpyHostToDevice));
    const uint32_t hi_bits = t->log_n - t->lo_bits;
    upload_powers(t->gz_hi, fr_pow2k(fr_from_params<Fr>(&P::gen_inv), t->lo_bits), 1ull << hi_bits, zc[1]);
  }
  *gz_hi = t->gz_hi.as<Fr>();
  *zconst = t->zconst.as<Fr>();
}

// w^(s p k) for p < N/(s R), k < R at index (p << r) | k; nullptr when the table would exceed 2^NTT_DIRECT_MAX_LOG entries
template <class Fr>
static const Fr* ntt_direct_table(const TunePolicy& pol, NttTables* t, uint32_t s_log, uint32_t r, bool inverse) {
  const uint32_t ent_log = t->log_n - s_log;
  const uint32_t max_log = pol.ntt_direct_max >= 0 ? (uint32_t)pol.ntt_direct_max : NTT_DIRECT_MAX_LOG;   // tests: force the fallback
  if (ent_log > max_log || ent_log == r) return nullptr;                  // too large / last pass (p == 0 only)
  std::lock_guard<std::mutex> lk(t->mu);
  const uint32_t key = ((inverse ? 1u : 0u) << 16) | (s_log << 8) | r;
  auto it = t->direct.find(key);
  if (it == t->direct.end()) {
    const Fr ws = fr_pow2k(ntt_root<Fr>(t->log_n, inverse), s_log);        // w^s
    const uint64_t m = 1ull << (ent_log - r), R = 1ull << r;
    std::vector<Fr> h(m * R);
    Fr wp = Fr::one();                                                    // (w^s)^p
    for (uint64_t p = 0; p < m; p++) {
      Fr cur = Fr::one();
      for (uint64_t k = 0; k < R; k++) {
        h[(p << r) | k] = cur;
        cur = Fr::mul(cur, wp);
      }
      wp = Fr::mul(wp, ws);
    }
    DevBuf b(h.size() * sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpy(b.p, h.data(), h.size() * sizeof(Fr), hipMemcpyHostToDevice));
    it = t->direct.emplace(key, std::move(b)).first;
  }
  return it->second.as<Fr>();
}

// Multiplications per eight elements of a radix-2^r pass: 12 per full three-stage group that is not the last one; the
// last group has compile-time twiddle indices and skips the trivial ones (5 for three stages, 2 for two, 0 for one);
// 8 more for the twiddle between passes.
static inline uint32_t ntt_pass_muls(uint32_t r, bool last) {
  const uint32_t full = r / 3, rem = r % 3;
  const uint32_t groups = full + (rem ? 1 : 0);
  const uint32_t last_g = rem ? rem : 3;
  const uint32_t tail = last_g == 3 ? 5 : (last_g == 2 ? 2 : 0);
  return 12 * (groups - 1) + tail + (last ? 0 : 8);
}

// Radices of the passes: as few passes as the largest radix allows; among those, the first and the last pass share a
// radix whenever possible (the seam kernel needs that) and the split with the fewest multiplications wins -- 2^21 points
// run as 2^6 x 2^9 x 2^6 (79 multiplications per eight elements and transform) rather than 2^7 x 2^7 x 2^7 (88).
static inline std::vector<uint32_t> ntt_radices(const TunePolicy& pol, uint32_t log_n) {
  uint32_t rmax = NTT_RMAX_LOG;
  if (pol.ntt_rmax >= 1 && pol.ntt_rmax <= (int)NTT_RMAX_LOG) rmax = (uint32_t)pol.ntt_rmax;   // tests: many passes on small vectors
  const uint32_t npass = (log_n + rmax - 1) / rmax;
  if (npass <= 1) return {log_n};
  std::vector<uint32_t> best;
  uint32_t best_cost = ~0u;
  // ends of radix a, the middle split evenly
  for (uint32_t a = 1; a <= rmax; a++) {
    if (2 * a > log_n) break;
    const uint32_t mid = log_n - 2 * a, nmid = npass - 2;
    std::vector<uint32_t> r;
    r.push_back(a);
    if (nmid == 0) {
      if (mid != 0) continue;
    } else {
      if (mid < nmid || mid > nmid * rmax) continue;
      for (uint32_t i = 0; i < nmid; i++) r.push_back(mid / nmid + (i < mid % nmid ? 1 : 0));
    }
    r.push_back(a);
    uint32_t cost = 0;
    for (size_t i = 0; i < r.size(); i++) cost += ntt_pass_muls(r[i], i + 1 == r.size());
    if (cost < best_cost) {
      best_cost = cost;
      best = r;
    }
  }
  if (!best.empty()) return best;
  // no symmetric split (two passes over an odd log_n): balanced
  std::vector<uint32_t> r(npass, log_n / npass);
  for (uint32_t i = 0; i < log_n % npass; i++) r[i] += 1;
  return r;
}

template <class Fr, int R_LOG>
static void ntt_launch_pass(const NttPassArgs& a, hipStream_t stream, uint32_t batch = 1) {
  const uint32_t grid = 1u << (a.log_n - R_LOG - a.p_log);
  const size_t smem = ((size_t)16 << (R_LOG + a.p_log)) + (size_t)32 * ((1u << R_LOG) >> 1) + 32;
  ARK_LAUNCH((ntt_pass_kernel<Fr, R_LOG>), dim3(grid, batch), dim3(NTT_THREADS), smem, stream, a);
  ARK_CHECK_LAUNCH();
}
template <class Fr, int R_LOG>
static void ntt_launch_seam(const NttPassArgs& a, const NttPassArgs& b, const Fr* seam, hipStream_t stream, uint32_t batch = 1) {
  const uint32_t grid = 1u << (a.log_n - R_LOG - a.p_log);
  const size_t smem = ((size_t)16 << (R_LOG + a.p_log)) + (size_t)64 * ((1u << R_LOG) >> 1) + 64;
  ARK_LAUNCH((ntt_seam_kernel<Fr, R_LOG>), dim3(grid, batch), dim3(NTT_THREADS), smem, stream, a, b, seam);
  ARK_CHECK_LAUNCH();
}

#define ARK_NTT_DISPATCH(r, CALL)                                                          \
  switch (r) {                                                                             \
    case 1: { constexpr int RL = 1; CALL; } break;                                         \
    case 2: { constexpr int RL = 2; CALL; } break;                                         \
    case 3: { constexpr int RL = 3; CALL; } break;                                         \
    case 4: { constexpr int RL = 4; CALL; } break;                                         \
    case 5: { constexpr int RL = 5; CALL; } break;                                         \
    case 6: { constexpr int RL = 6; CALL; } break;                                         \
    case 7: { constexpr int RL = 7; CALL; } break;                                         \
    case 8: { constexpr int RL = 8; CALL; } break;                                         \
    case 9: { constexpr int RL = 9; CALL; } break;                                         \
    default: throw HipError{ARK355_EINVAL, "NTT pass radix out of range"};                 \
  }

// tile columns of a pass: 2048 elements per tile when the vector has that many, at least 8 (one per lane slot)
static inline uint32_t ntt_p_log(uint32_t log_n, uint32_t r) {
  const uint32_t cols_log = log_n - r;
  uint32_t p_log = NTT_EMAX_LOG > r ? NTT_EMAX_LOG - r : 0;
  if (p_log > cols_log) p_log = cols_log;
  return p_log;
}

// vectors shorter than 8 elements: one lane, schoolbook (tests and degenerate circuits only)
template <class Fr>
__global__ void __launch_bounds__(64)
ntt_tiny_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, uint32_t log_n, const Fr* __restrict__ w_lo,
                const Fr* __restrict__ w_hi, uint32_t lo_bits, const Fr* __restrict__ in_lo, const Fr* __restrict__ in_hi,
                const Fr* __restrict__ out_lo, const Fr* __restrict__ out_hi, const Fr* __restrict__ out_const) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t n = 1u << log_n;
  Fr x[4], y[4];
  for (uint32_t j = 0; j < n; j++) {
    x[j] = in[j];
    if (in_lo) x[j] = Fr::mul(x[j], pow_lookup<Fr>(in_lo, in_hi, lo_bits, j));
  }
  for (uint32_t k = 0; k < n; k++) {
    Fr acc = Fr::zero();
    for (uint32_t j = 0; j < n; j++) acc = Fr::add(acc, Fr::mul(x[j], pow_lookup<Fr>(w_lo, w_hi, lo_bits, (uint64_t)((j * k) & (n - 1)))));
    if (out_lo) acc = Fr::mul(acc, pow_lookup<Fr>(out_lo, out_hi, lo_bits, k));
    if (out_const) acc = Fr::mul(acc, *out_const);
    y[k] = acc;
  }
  for (uint32_t k = 0; k < n; k++) out[k] = y[k];
}

// x[i] *= hi[i >> lo_bits] * lo[i & mask]   (or *= *cst): the coset shift g^j of a forward coset transform and the
// 1/N (g^-j / N) of an inverse one.  Elementwise and HBM-cheap; the passes themselves stay free of optional paths.
template <class Fr>
__global__ void __launch_bounds__(256)
ntt_scale_kernel(Fr* __restrict__ x, uint64_t n, const Fr* __restrict__ lo, const Fr* __restrict__ hi, uint32_t lo_bits,
                 const Fr* __restrict__ cst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr v = x[i];
  if (lo) v = Fr::mul(v, pow_lookup<Fr>(lo, hi, lo_bits, i));
  else v = Fr::mul(v, *cst);
  x[i] = v;
}

// Inter-pass twiddle of a pass that is too large for a direct table (domains beyond 2^23 points): y[o] *= w^(s p k)
// with o = q + s (R p + k), composed from the hi/lo tables.  Elementwise, after the pass.
template <class Fr>
__global__ void __launch_bounds__(256)
ntt_twiddle_kernel(Fr* __restrict__ y, uint64_t n, uint32_t s_log, uint32_t r, const Fr* __restrict__ w_lo,
                   const Fr* __restrict__ w_hi, uint32_t lo_bits) {
  const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  const uint64_t j = o >> s_log, k = j & ((1ull << r) - 1ull), p = j >> r;
  const uint64_t e = (p * k) << s_log;
  if (e != 0) y[o] = Fr::mul(y[o], pow_lookup<Fr>(w_lo, w_hi, lo_bits, e));
}

template <class Fr>
static void ntt_twiddle_fallback(const NttPassArgs& a, uint32_t r, hipStream_t stream, uint32_t batch = 1) {
  if (a.direct || a.s_log + r == a.log_n) return;
  const uint64_t n = 1ull << a.log_n;
  for (uint32_t y = 0; y < batch; y++) {
    ARK_LAUNCH((ntt_twiddle_kernel<Fr>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream,
               reinterpret_cast<Fr*>(a.out) + (uint64_t)y * a.batch_stride, n, a.s_log, r, reinterpret_cast<const Fr*>(a.w_lo),
               reinterpret_cast<const Fr*>(a.w_hi), a.lo_bits);
    ARK_CHECK_LAUNCH();
  }
}

// NTT of 2^log_n elements.  `data` holds the input; `scratch` is a same-size buffer.  Returns the buffer that holds
// the result (passes ping-pong), so callers can avoid a final copy.  inverse: 1/N (or g^-k / N with coset) is applied
// after the last pass; coset forward multiplies the input by g^j before the first pass (in place).
template <class Curve>
static void* ntt_run(ark355_ctx* ctx, void* data, void* scratch, uint32_t log_n, bool inverse, bool coset,
                     hipStream_t stream) {
  using Fr = typename Curve::Fr;
  if (log_n == 0) return data;
  NttTables* t = get_ntt_tables<Curve>(ctx, log_n);
  const Fr* w_lo = (inverse ? t->wi_lo : t->w_lo).template as<Fr>();
  const Fr* w_hi = (inverse ? t->wi_hi : t->w_hi).template as<Fr>();
  const Fr* in_lo = nullptr; const Fr* in_hi = nullptr;
  const Fr* out_lo = nullptr; const Fr* out_hi = nullptr; const Fr* out_const = nullptr;
  if (coset && !inverse) { in_lo = t->g_lo.as<Fr>(); in_hi = t->g_hi.as<Fr>(); }
  if (inverse) {
    if (coset) { out_lo = t->gi_lo.as<Fr>(); out_hi = t->gi_hi.as<Fr>(); }
    else out_const = t->n_inv.as<Fr>();
  }
  if (log_n < 3) {
    ARK_LAUNCH((ntt_tiny_kernel<Fr>), dim3(1), dim3(64), 0, stream, (const Fr*)data, (Fr*)scratch, log_n, w_lo, w_hi,
               t->lo_bits, in_lo, in_hi, out_lo, out_hi, out_const);
    ARK_CHECK_LAUNCH();
    return scratch;
  }
  const uint64_t n = 1ull << log_n;
  const dim3 sgrid((uint32_t)((n + 255) / 256));
  if (in_lo) {
    ARK_LAUNCH((ntt_scale_kernel<Fr>), sgrid, dim3(256), 0, stream, (Fr*)data, n, in_lo, in_hi, t->lo_bits, (const Fr*)nullptr);
    ARK_CHECK_LAUNCH();
  }
  const std::vector<uint32_t> radices = ntt_radices(ctx->policy, log_n);
  uint32_t s_log = 0;
  Fr* src = (Fr*)data;
  Fr* dst = (Fr*)scratch;
  for (size_t pass = 0; pass < radices.size(); pass++) {
    const uint32_t r = radices[pass];
    NttPassArgs a{};
    a.in = src;
    a.out = dst;
    a.log_n = log_n;
    a.s_log = s_log;
    a.p_log = ntt_p_log(log_n, r);
    a.tw = ntt_tw_table<Fr>(t, r, inverse);
    a.direct = ntt_direct_table<Fr>(ctx->policy, t, s_log, r, inverse);
    a.w_lo = w_lo;
    a.w_hi = w_hi;
    a.lo_bits = t->lo_bits;
    ARK_NTT_DISPATCH(r, (ntt_launch_pass<Fr, RL>(a, stream)));
    ntt_twiddle_fallback<Fr>(a, r, stream);
    s_log += r;
    Fr* tmp = src; src = dst; dst = tmp;
  }
  if (out_lo || out_const) {
    ARK_LAUNCH((ntt_scale_kernel<Fr>), sgrid, dim3(256), 0, stream, src, n, out_lo, out_hi, t->lo_bits, out_const);
    ARK_CHECK_LAUNCH();
  }
  return src;
}

// The passes of one transform and nothing else (no coset shift, no 1/N): `batch` vectors of 2^log_n elements, vector y at
// data + y * batch_stride with its ping-pong partner at scratch + y * batch_stride, one launch per pass.  Returns where
// vector 0 ended up.  The distributed witness map (witness_dist_impl.cuh) runs its local N/G-point transforms with this
// and folds every scaling into its exchange kernels.  log_n >= 3.
template <class Curve>
static void* ntt_passes(ark355_ctx* ctx, void* data, void* scratch, uint32_t log_n, bool inverse, hipStream_t stream,
                        uint32_t batch = 1, uint64_t batch_stride = 0) {
  using Fr = typename Curve::Fr;
  ARK_REQUIRE(log_n >= 3, ARK355_EINVAL, "ntt_passes: at least eight points");
  NttTables* t = get_ntt_tables<Curve>(ctx, log_n);
  const std::vector<uint32_t> radices = ntt_radices(ctx->policy, log_n);
  uint32_t s_log = 0;
  Fr* src = (Fr*)data;
  Fr* dst = (Fr*)scratch;
  for (size_t pass = 0; pass < radices.size(); pass++) {
    const uint32_t r = radices[pass];
    NttPassArgs a{};
    a.in = src;
    a.out = dst;
    a.log_n = log_n;
    a.s_log = s_log;
    a.p_log = ntt_p_log(log_n, r);
    a.tw = ntt_tw_table<Fr>(t, r, inverse);
    a.direct = ntt_direct_table<Fr>(ctx->policy, t, s_log, r, inverse);
    a.w_lo = (inverse ? t->wi_lo : t->w_lo).template as<Fr>();
    a.w_hi = (inverse ? t->wi_hi : t->w_hi).template as<Fr>();
    a.lo_bits = t->lo_bits;
    a.batch_stride = batch > 1 ? batch_stride : 0;
    ARK_NTT_DISPATCH(r, (ntt_launch_pass<Fr, RL>(a, stream, batch)));
    ntt_twiddle_fallback<Fr>(a, r, stream, batch);
    s_log += r;
    Fr* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

// inverse NTT followed by coset NTT of the same vector (witness map: evaluations on H -> evaluations on g H), with the
// seam fused when the first and the last pass share a radix.  Same contract as ntt_run.
// batch > 1: `batch` vectors of the same length, vector y at data + y * batch_stride with its ping-pong partner at
// scratch + y * batch_stride (elements), every pass ONE launch over all of them (the witness map's a, b, c: 15 dispatches
// become 5, and a pass fills the chip three times as long); the result of vector y sits at (return value) + y * batch_stride.
// tail > 0: the LAST `tail` vectors of the batch stop after the inverse transform (the witness map's c, whose coset
// evaluations nobody needs: witness_impl.cuh).  *tail_res = where the first of them ended up (data or scratch side, same
// stride), *tail_scaled = whether its 1/N has been applied (the fused path leaves N x the coefficients: the factor is folded
// into the caller's next step instead of costing a pass of its own).
template <class Curve>
static void* ntt_inverse_then_coset(ark355_ctx* ctx, void* data, void* scratch, uint32_t log_n, hipStream_t stream,
                                    uint32_t batch = 1, uint64_t batch_stride = 0, uint32_t tail = 0, void** tail_res = nullptr,
                                    bool* tail_scaled = nullptr) {
  using Fr = typename Curve::Fr;
  ARK_REQUIRE(tail < batch && (tail == 0 || (tail_res && tail_scaled)), ARK355_EINVAL, "ntt_inverse_then_coset: bad tail");
  const uint32_t full = batch - tail;
  const std::vector<uint32_t> radices = log_n >= 3 ? ntt_radices(ctx->policy, log_n) : std::vector<uint32_t>();
  // the seam kernel exists for first/last radices 2^5 .. 2^9 (every domain of 2^10 points or more with matching ends);
  // tests lower the bound through policy NTT_RMAX and then use radices < 6: those take the unfused path
  const bool fuse = log_n >= 3 && radices.front() == radices.back() && radices.front() >= 5 && !ctx->policy.ntt_nofuse;
  if (!fuse) {
    void* first = nullptr;
    for (uint32_t y = 0; y < batch; y++) {       // (every vector takes the same ping-pong path)
      void* cur = (Fr*)data + (uint64_t)y * batch_stride;
      void* oth = (Fr*)scratch + (uint64_t)y * batch_stride;
      void* res = ntt_run<Curve>(ctx, cur, oth, log_n, /*inverse=*/true, /*coset=*/false, stream);
      if (y >= full) {
        if (y == full) {
          *tail_res = res;
          *tail_scaled = true;
        }
        continue;
      }
      if (res != cur) { oth = cur; cur = res; }
      res = ntt_run<Curve>(ctx, cur, oth, log_n, /*inverse=*/false, /*coset=*/true, stream);
      if (y == 0) first = res;
    }
    return first;
  }
  NttTables* t = get_ntt_tables<Curve>(ctx, log_n);
  const size_t np = radices.size();
  Fr* src = (Fr*)data;
  Fr* dst = (Fr*)scratch;
  auto make = [&](bool inverse, size_t pass, uint32_t s_log) {
    const uint32_t r = radices[pass];
    NttPassArgs a{};
    a.log_n = log_n;
    a.s_log = s_log;
    a.p_log = ntt_p_log(log_n, r);
    a.tw = ntt_tw_table<Fr>(t, r, inverse);
    a.direct = ntt_direct_table<Fr>(ctx->policy, t, s_log, r, inverse);
    a.w_lo = (inverse ? t->wi_lo : t->w_lo).template as<Fr>();
    a.w_hi = (inverse ? t->wi_hi : t->w_hi).template as<Fr>();
    a.lo_bits = t->lo_bits;
    a.batch_stride = batch > 1 ? batch_stride : 0;
    return a;
  };
  // inverse transform, all passes but the last
  uint32_t s_log = 0;
  for (size_t pass = 0; pass + 1 < np; pass++) {
    NttPassArgs a = make(true, pass, s_log);
    a.in = src;
    a.out = dst;
    ARK_NTT_DISPATCH(radices[pass], (ntt_launch_pass<Fr, RL>(a, stream, batch)));
    ntt_twiddle_fallback<Fr>(a, radices[pass], stream, batch);
    s_log += radices[pass];
    Fr* tmp = src; src = dst; dst = tmp;
  }
  // the seam: last inverse pass + g^j / N + first coset pass
  {
    NttPassArgs a = make(true, np - 1, s_log);
    NttPassArgs b = make(false, 0, 0);
    a.in = src;
    b.out = dst;
    const Fr* seam = ntt_seam_table<Fr>(t);
    switch (radices[0]) {      // instantiated for the radices a fused transform can start with (log_n >= NTT_SEAM_MIN_LOG)
      case 5: ntt_launch_seam<Fr, 5>(a, b, seam, stream, full); break;
      case 6: ntt_launch_seam<Fr, 6>(a, b, seam, stream, full); break;
      case 7: ntt_launch_seam<Fr, 7>(a, b, seam, stream, full); break;
      case 8: ntt_launch_seam<Fr, 8>(a, b, seam, stream, full); break;
      case 9: ntt_launch_seam<Fr, 9>(a, b, seam, stream, full); break;
      default: throw HipError{ARK355_EINVAL, "no seam kernel for this radix"};
    }
    ntt_twiddle_fallback<Fr>(b, radices[0], stream, full);
    if (tail) {
      // the vectors that stop here: the last inverse pass as a plain pass (no inter-pass twiddle: p == 0), unscaled
      NttPassArgs c = make(true, np - 1, s_log);
      c.in = src + (uint64_t)full * batch_stride;
      c.out = dst + (uint64_t)full * batch_stride;
      ARK_NTT_DISPATCH(radices[np - 1], (ntt_launch_pass<Fr, RL>(c, stream, tail)));
      *tail_res = c.out;
      *tail_scaled = false;
    }
    Fr* tmp = src; src = dst; dst = tmp;
  }
  // coset transform, remaining passes
  s_log = radices[0];
  for (size_t pass = 1; pass < np; pass++) {
    NttPassArgs a = make(false, pass, s_log);
    a.in = src;
    a.out = dst;
    ARK_NTT_DISPATCH(radices[pass], (ntt_launch_pass<Fr, RL>(a, stream, full)));
    ntt_twiddle_fallback<Fr>(a, radices[pass], stream, full);
    s_log += radices[pass];
    Fr* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

}  // namespace ark355
