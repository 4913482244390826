// Radix-2 NTT over Fr for gfx950: Stockham auto-sort passes of radix 2^r (r <= 8), each pass staging
// a tile of R x P elements through LDS (limb-major, so butterfly reads/writes are bank-conflict free)
// and running its r radix-2 DIF butterfly stages there.
//
// Replaces ark-poly `Radix2EvaluationDomain::{fft,ifft}_in_place` and the coset variants
// (un-vendored crate ark-poly/src/domain/radix2/fft.rs) for the 7 transforms of the Groth16
// witness map.  Natural order in, natural order out:
// X[k] = sum_j x[j] w^(jk).
//
// One pass with current stride s (sequence length n = N/s, m = n/R):
//   y[q + s(Rp + k)] = w_N^(s p k) * sum_t x[q + s p + (N/R) t] * w_R^(t k),   q < s, p < m, k < R
// Reads are contiguous in pq = q + s p for every t; writes are contiguous in q (or in k when s < P).
// HBM traffic: one read + one write of the vector per pass (64 B per element per pass), twiddles come
// from two small L2-resident tables (w^e = hi[e >> LO] * lo[e & mask]).
#pragma once
#include "common.h"

namespace ark355 {

#ifndef ARK_NTT_EMAX_LOG
#define ARK_NTT_EMAX_LOG 10   // elements staged per workgroup (32 KiB of LDS)
#endif
#ifndef ARK_NTT_RMAX_LOG
#define ARK_NTT_RMAX_LOG 8    // largest radix 2^r of one pass (tests shrink it to force many passes)
#endif
constexpr uint32_t NTT_EMAX_LOG = ARK_NTT_EMAX_LOG;
constexpr uint32_t NTT_RMAX_LOG = ARK_NTT_RMAX_LOG;
constexpr uint32_t NTT_THREADS = 256;

struct NttTables {
  uint32_t log_n = 0, lo_bits = 0;
  DevBuf w_lo, w_hi, wi_lo, wi_hi;     // w^e, w^-e
  DevBuf g_lo, g_hi, gi_lo, gi_hi;     // g^j ; g^-j / N
  DevBuf n_inv;                        // 1/N
};

template <class Fr>
ARK_D Fr fr_load_soa(const uint32_t* base, uint32_t stride, uint32_t pos) {
  Fr x;
#pragma unroll
  for (int l = 0; l < Fr::N; l++) x.l[l] = base[l * stride + pos];
  return x;
}
template <class Fr>
ARK_D void fr_store_soa(uint32_t* base, uint32_t stride, uint32_t pos, const Fr& x) {
#pragma unroll
  for (int l = 0; l < Fr::N; l++) base[l * stride + pos] = x.l[l];
}

template <class Fr>
ARK_D Fr pow_lookup(const Fr* lo, const Fr* hi, uint32_t lo_bits, uint64_t e) {
  const uint64_t h = e >> lo_bits;
  const uint32_t l = (uint32_t)(e & ((1ull << lo_bits) - 1));
  // lo[0] == 1 always; hi[0] may carry a folded constant (1/N), so only the l == 0 shortcut is valid
  if (l == 0) return hi[h];
  return Fr::mul(hi[h], lo[l]);
}

ARK_D uint32_t bitrev_bits(uint32_t v, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) {
    r = (r << 1) | (v & 1);
    v >>= 1;
  }
  return r;
}

template <class Fr>
__global__ void __launch_bounds__(NTT_THREADS)
ntt_pass_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, uint32_t log_n, uint32_t s_log, uint32_t r,
                uint32_t p_log, const Fr* __restrict__ w_lo, const Fr* __restrict__ w_hi, uint32_t lo_bits,
                const Fr* __restrict__ in_lo, const Fr* __restrict__ in_hi, const Fr* __restrict__ out_lo,
                const Fr* __restrict__ out_hi, const Fr* __restrict__ out_const) {
  const uint32_t R = 1u << r, P = 1u << p_log, E = R << p_log;
  const uint32_t cols_log = log_n - r;
  const uint32_t tid = threadIdx.x, nth = blockDim.x;
  ARK_DYN_SMEM(uint32_t, lds);
  uint32_t* tw = lds + Fr::N * E;
  const uint32_t TW = R >> 1;

  for (uint32_t e = tid; e < TW; e += nth) {
    Fr t = pow_lookup<Fr>(w_lo, w_hi, lo_bits, (uint64_t)e << cols_log);
    fr_store_soa<Fr>(tw, TW ? TW : 1, e, t);
  }
  for (uint32_t idx = tid; idx < E; idx += nth) {
    const uint32_t t = idx >> p_log, c = idx & (P - 1);
    const uint64_t pq = (uint64_t)blockIdx.x * P + c;
    const uint64_t g = pq + ((uint64_t)t << cols_log);
    Fr x = in[g];
    if (in_lo) x = Fr::mul(x, pow_lookup<Fr>(in_lo, in_hi, lo_bits, g));
    fr_store_soa<Fr>(lds, E, idx, x);
  }
  __syncthreads();

  for (uint32_t st = 0; st < r; st++) {
    const uint32_t half_log = r - 1 - st, half = 1u << half_log;
    for (uint32_t bidx = tid; bidx < (E >> 1); bidx += nth) {
      const uint32_t c = bidx & (P - 1), j = bidx >> p_log;
      const uint32_t k = j & (half - 1), b = j >> half_log;
      const uint32_t i0 = (b << (half_log + 1)) + k, i1 = i0 + half;
      const uint32_t a0 = (i0 << p_log) + c, a1 = (i1 << p_log) + c;
      Fr u = fr_load_soa<Fr>(lds, E, a0);
      Fr v = fr_load_soa<Fr>(lds, E, a1);
      Fr sum = Fr::add(u, v);
      Fr dif = Fr::sub(u, v);
      if (k != 0) dif = Fr::mul(dif, fr_load_soa<Fr>(tw, TW, k << st));
      fr_store_soa<Fr>(lds, E, a0, sum);
      fr_store_soa<Fr>(lds, E, a1, dif);
    }
    __syncthreads();
  }

  const uint32_t s_mask = (1u << s_log) - 1;   // s <= N/2 < 2^32 whenever this mask is used with s_log < 32
  for (uint32_t idx = tid; idx < E; idx += nth) {
    uint32_t k, c;
    if (s_log < p_log) {
      k = idx & (R - 1);
      c = idx >> r;
    } else {
      c = idx & (P - 1);
      k = idx >> p_log;
    }
    const uint32_t pos = bitrev_bits(k, r);
    Fr x = fr_load_soa<Fr>(lds, E, (pos << p_log) + c);
    const uint64_t pq = (uint64_t)blockIdx.x * P + c;
    const uint64_t q = pq & s_mask, p = pq >> s_log;
    const uint64_t e = (p * k) << s_log;
    if (e != 0) x = Fr::mul(x, pow_lookup<Fr>(w_lo, w_hi, lo_bits, e));
    const uint64_t o = q + ((p * R + k) << s_log);
    if (out_lo) x = Fr::mul(x, pow_lookup<Fr>(out_lo, out_hi, lo_bits, o));
    if (out_const) x = Fr::mul(x, *out_const);
    out[o] = x;
  }
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
static NttTables* build_ntt_tables(uint32_t log_n) {
  using P = typename Fr::Params;
  ARK_REQUIRE(log_n <= (uint32_t)P::TWO_ADICITY, ARK355_E_POLY_DEGREE_TOO_LARGE,
              "domain size exceeds the two-adicity of Fr");
  auto* t = new NttTables();
  t->log_n = log_n;
  const uint32_t lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
  t->lo_bits = lo_bits;
  const Fr root = fr_from_params<Fr>(&P::root), root_inv = fr_from_params<Fr>(&P::root_inv);
  const Fr w = fr_pow2k(root, P::TWO_ADICITY - log_n), wi = fr_pow2k(root_inv, P::TWO_ADICITY - log_n);
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

template <class Curve>
static NttTables* get_ntt_tables(ark355_ctx* ctx, uint32_t log_n) {
  const uint32_t key = ((uint32_t)Curve::ID << 8) | log_n;
  auto it = ctx->ntt_tables.find(key);
  if (it != ctx->ntt_tables.end()) return it->second;
  NttTables* t = build_ntt_tables<typename Curve::Fr>(log_n);
  ctx->ntt_tables[key] = t;
  return t;
}

// NTT of 2^log_n elements.  `data` holds the input; `scratch` is a same-size buffer.  Returns the
// buffer that holds the result (passes ping-pong), so callers can avoid a final copy.
// mode bits: inverse, coset.  GH multiplies by 1/N or g^-K/N in the last pass; coset-forward multiplies the
// input by g^j in the first pass.
template <class Curve>
static void* ntt_run(ark355_ctx* ctx, void* data, void* scratch, uint32_t log_n, bool inverse, bool coset,
                     hipStream_t stream) {
  using Fr = typename Curve::Fr;
  if (log_n == 0) return data;
  NttTables* t = get_ntt_tables<Curve>(ctx, log_n);
  const Fr* w_lo = (inverse ? t->wi_lo : t->w_lo).template as<Fr>();
  const Fr* w_hi = (inverse ? t->wi_hi : t->w_hi).template as<Fr>();
  const uint32_t npass = (log_n + NTT_RMAX_LOG - 1) / NTT_RMAX_LOG;
  uint32_t s_log = 0;
  Fr* src = (Fr*)data;
  Fr* dst = (Fr*)scratch;
  uint32_t remaining = log_n;
  for (uint32_t pass = 0; pass < npass; pass++) {
    const uint32_t r = (remaining + (npass - pass) - 1) / (npass - pass);
    remaining -= r;
    const uint32_t cols_log = log_n - r;
    uint32_t p_log = NTT_EMAX_LOG - r;
    if (p_log > cols_log) p_log = cols_log;
    const uint32_t grid = 1u << (cols_log - p_log);
    const bool first = pass == 0, last = pass + 1 == npass;
    const Fr* in_lo = nullptr; const Fr* in_hi = nullptr;
    const Fr* out_lo = nullptr; const Fr* out_hi = nullptr; const Fr* out_const = nullptr;
    if (first && coset && !inverse) { in_lo = t->g_lo.as<Fr>(); in_hi = t->g_hi.as<Fr>(); }
    if (last && inverse) {
      if (coset) { out_lo = t->gi_lo.as<Fr>(); out_hi = t->gi_hi.as<Fr>(); }
      else out_const = t->n_inv.as<Fr>();
    }
    const size_t smem = (size_t)Fr::N * 4 * ((1u << (r + p_log)) + (1u << r) / 2 + 1);
    ARK_LAUNCH((ntt_pass_kernel<Fr>), dim3(grid), dim3(NTT_THREADS), smem, stream, src, dst, log_n, s_log, r,
               p_log, w_lo, w_hi, t->lo_bits, in_lo, in_hi, out_lo, out_hi, out_const);
    ARK_CHECK_LAUNCH();
    s_log += r;
    Fr* tmp = src; src = dst; dst = tmp;
  }
  return src;
}

}  // namespace ark355
