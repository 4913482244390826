// Per-curve implementation of the C ABI (explicitly instantiated in ark355_bls.hip / ark355_bn.hip).
#pragma once
#include <functional>
#include <thread>
#include "common.h"
#include "groth16_impl.cuh"
#include "wire_impl.cuh"
#include "pairing_host.hpp"

namespace ark355 {

struct BasesDev {
  int curve = 0, group = 1;
  uint64_t n = 0;
  PrecompTable tab;      // per-window tables of the resident bases
};

// ---- fixed-base multiplication (setup: the generator's five query vectors are k_i * G) -----------------------------
// Windowed: T[w][d-1] = d * 2^(8w) * base for d = 1..255, w < 32 (8 160 affine points, built once per call on the
// device), then every scalar costs 32 mixed additions instead of the ~383 group operations of double-and-add, and the
// results are normalised 16 at a time with one inversion (batch_to_affine_kernel).  The first version (one lane per
// scalar, double-and-add, an inversion each) spilled 2.9 KB per lane and took most of the 9 s of bench.py's preparation.
constexpr uint32_t FB_WBITS = 8, FB_WINDOWS = 32, FB_ROW = (1u << FB_WBITS) - 1u;

// row w of the table in XYZZ form: lane d-1 computes d * (2^(8w) base) by double-and-add over the 8-bit d
template <class F>
__global__ void __launch_bounds__(256)
fixed_base_table_kernel(const Affine<F>* __restrict__ base, XYZZ<F>* __restrict__ table) {
  const uint32_t w = blockIdx.x, d = threadIdx.x + 1;
  if (d > FB_ROW) return;
  XYZZ<F> p = XYZZ<F>::from_affine(*base);
  for (uint32_t i = 0; i < w * FB_WBITS; i++) p = xyzz_dbl(p);
  uint32_t k = d;
  table[w * FB_ROW + (d - 1)] = xyzz_mul_scalar(p, &k, 1);
}

// a handful of scalars (the alpha/beta/gamma/delta points of a key, closed-form checks): one lane each, double-and-add --
// building the 8 160-point table first would cost more than it saves
template <class F, class Fr>
__global__ void __launch_bounds__(128)
fixed_base_small_kernel(const Affine<F>* __restrict__ base, const Fr* __restrict__ scalars, uint64_t n,
                        XYZZ<F>* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr k = scalars[i];
  out[i] = xyzz_mul_scalar(XYZZ<F>::from_affine(*base), k.l, Fr::N);
}

template <class F, class Fr>
__global__ void __launch_bounds__(128)
fixed_base_mul_kernel(const Affine<F>* __restrict__ table, const Fr* __restrict__ scalars, uint64_t n,
                      XYZZ<F>* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr k = scalars[i];
  XYZZ<F> acc = XYZZ<F>::inf();
  for (uint32_t w = 0; w < FB_WINDOWS; w++) {
    const uint32_t d = (k.l[w >> 2] >> ((w & 3u) * 8u)) & 0xFFu;
    if (d) xyzz_madd_ni(acc, table[w * FB_ROW + (d - 1)]);
  }
  out[i] = acc;
}

struct GenericScratch {
  MsmSort sort;
  MsmBuckets bk;
  DevBuf a, b, c;
};

template <class Curve>
struct Api {
  using Fr = typename Curve::Fr;
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using W = Wire<Curve>;

  static void sizes(uint32_t what[4]) {
    what[0] = sizeof(Fr);
    what[1] = sizeof(Fq);
    what[2] = sizeof(Affine<Fq>);
    what[3] = sizeof(Affine<Fq2>);
  }

  static PkDev* pk_load(const TunePolicy& pol, const ark355_pk_desc* d, hipStream_t st, uint32_t shard_index = 0,
                        uint32_t shard_count = 1) {
    return pk_upload<Curve>(pol, d, st, shard_index, shard_count);
  }

  static R1csDev* r1cs_load(uint64_t n, uint64_t ell, uint64_t w, const uint64_t* const rp[3],
                            const uint32_t* const col[3], const uint8_t* const coeff[3]) {
    return r1cs_upload<Curve>(n, ell, w, rp, col, coeff);
  }

  static void prove(ark355_ctx* ctx, ProverScratch& sc, const PkDev& pk, const R1csDev& r1, const void* z,
                    bool on_dev, const uint8_t* r, const uint8_t* s, ark355_proof_raw* out, uint8_t* partials = nullptr,
                    CommDev* cm = nullptr, int shard_mode = 0) {
    prove_run<Curve>(ctx, sc, pk, r1, z, on_dev, r, s, out, partials, cm, shard_mode);
  }
  static void combine(const uint8_t* partials, uint64_t count, const uint8_t* r, const uint8_t* s, ark355_proof_raw* out) {
    combine_partials_host<Curve>(partials, count, r, s, out);
  }
  static size_t partial_size() { return 4 * sizeof(XYZZ<Fq>) + sizeof(XYZZ<Fq2>); }

  static void witness_map(ark355_ctx* ctx, ProverScratch& sc, const R1csDev& r1, const uint8_t* z, uint8_t* h_out) {
    hipStream_t st = ctx->stream;
    sc.zx.ensure((r1.m + 4) * sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpyAsync(sc.zx.p, z, r1.m * sizeof(Fr), hipMemcpyHostToDevice, st));
    void* d_h = witness_map_run<Curve>(ctx, r1, sc.zx.p, sc.ws, st);
    ARK_CHECK_HIP(hipMemcpyAsync(h_out, d_h, r1.N * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  static void witness_map_dist_sim(ark355_ctx* ctx, ProverScratch& sc, const R1csDev& r1, const uint8_t* z, uint32_t world,
                                   uint8_t* h_out) {
    hipStream_t st = ctx->stream;
    sc.zx.ensure((r1.m + 4) * sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpyAsync(sc.zx.p, z, r1.m * sizeof(Fr), hipMemcpyHostToDevice, st));
    DevBuf h(r1.N * sizeof(Fr));
    ark355::witness_map_dist_sim<Curve>(ctx, r1, sc.zx.p, world, h.p, st);
    ARK_CHECK_HIP(hipMemcpyAsync(h_out, h.p, r1.N * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  static void mat_vec(ark355_ctx* ctx, ProverScratch& sc, const R1csDev& r1, const uint8_t* z, uint8_t* az,
                      uint8_t* bz, uint8_t* cz, int64_t* first_bad) {
    hipStream_t st = ctx->stream;
    sc.zx.ensure((r1.m + 4) * sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpyAsync(sc.zx.p, z, r1.m * sizeof(Fr), hipMemcpyHostToDevice, st));
    spmv_run<Curve>(r1, sc.zx.p, sc.ws, st);
    if (az) ARK_CHECK_HIP(hipMemcpyAsync(az, sc.ws.buf[0].p, r1.n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    if (bz) ARK_CHECK_HIP(hipMemcpyAsync(bz, sc.ws.buf[2].p, r1.n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    if (cz) ARK_CHECK_HIP(hipMemcpyAsync(cz, sc.ws.buf[4].p, r1.n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    if (first_bad) {
      sc.ws.first_bad.ensure(8);
      ARK_CHECK_HIP(hipMemsetAsync(sc.ws.first_bad.p, 0xFF, 8, st));
      if (r1.n) {
        const uint32_t grid = (uint32_t)((r1.n + 255) / 256);
        ARK_LAUNCH((r1cs_check_kernel<Fr>), dim3(grid), dim3(256), 0, st, sc.ws.buf[0].as<Fr>(),
                   sc.ws.buf[2].as<Fr>(), sc.ws.buf[4].as<Fr>(), r1.n, sc.ws.first_bad.as<unsigned long long>());
        ARK_CHECK_LAUNCH();
      }
      unsigned long long fb = 0;
      ARK_CHECK_HIP(hipMemcpyAsync(&fb, sc.ws.first_bad.p, 8, hipMemcpyDeviceToHost, st));
      ARK_CHECK_HIP(hipStreamSynchronize(st));
      *first_bad = (fb == ~0ull) ? -1 : (int64_t)fb;
    }
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  static void ntt_host(ark355_ctx* ctx, GenericScratch& g, uint8_t* data, uint32_t log_n, bool inverse, bool coset) {
    hipStream_t st = ctx->stream;
    const size_t bytes = sizeof(Fr) << log_n;
    g.a.ensure(bytes);
    g.b.ensure(bytes);
    ARK_CHECK_HIP(hipMemcpyAsync(g.a.p, data, bytes, hipMemcpyHostToDevice, st));
    void* res = ntt_run<Curve>(ctx, g.a.p, g.b.p, log_n, inverse, coset, st);
    ARK_CHECK_HIP(hipMemcpyAsync(data, res, bytes, hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  static void ntt_dev(ark355_ctx* ctx, void* d_data, void* d_scratch, uint32_t log_n, bool inverse, bool coset,
                      hipStream_t st) {
    const size_t bytes = sizeof(Fr) << log_n;
    void* res = ntt_run<Curve>(ctx, d_data, d_scratch, log_n, inverse, coset, st);
    if (res != d_data) ARK_CHECK_HIP(hipMemcpyAsync(d_data, res, bytes, hipMemcpyDeviceToDevice, st));
  }

  template <class F>
  static void msm_generic(ark355_ctx* ctx, GenericScratch& g, const Affine<F>* d_bases, const void* d_scalars,
                          uint64_t n, int mont, uint8_t* out, bool want_affine, const PrecompTable* tab = nullptr) {
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    ARK_CHECK_HIP(hipEventCreate(&e0));
    ARK_CHECK_HIP(hipEventCreate(&e1));
    try {
      const int fmt = tab != nullptr ? tab->fmt() : 0;
      msm_sort<Fr>(ctx, g.sort, d_scalars, n, mont, st, tab);
      // what the tails leave on the device: one sum (32-bit tails) or c partial sums per bucket set (28-bit tails)
      const uint32_t parts = msm_parts_count(g.sort.plan, fmt);
      g.c.ensure((size_t)parts * sizeof(XYZZ<F>));
      XYZZ<F>* d_res = g.c.as<XYZZ<F>>();
      msm_buckets<F>(ctx, g.sort, g.bk, d_bases, d_res, 0, st, n ? e0 : nullptr, n ? e1 : nullptr, fmt);
      // The last 2c group operations of the bucket reduction (Horner over the bit sums) and the one inversion of the
      // normalisation run in the library's host-compiled field code: ~1 us per group operation against ~12 us for a device
      // lane, tens of microseconds for the inversion against ~1 ms.
      std::vector<XYZZ<F>> h_parts(parts);
      ARK_CHECK_HIP(hipMemcpyAsync(h_parts.data(), d_res, (size_t)parts * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
      ARK_CHECK_HIP(hipStreamSynchronize(st));
      const XYZZ<F> h_res = msm_parts_finish<F>(h_parts.data(), g.sort.plan, fmt);
      if (want_affine) {
        const Affine<F> a = xyzz_to_affine(h_res);
        memcpy(out, &a, sizeof(a));
      } else {
        memcpy(out, &h_res, sizeof(h_res));
      }
      float ms = 0;
      if (n) (void)hipEventElapsedTime(&ms, e0, e1);
      ctx->acc_ms = ms;
      ctx->acc_launches = n ? 1 : 0;
      ctx->acc_points = (uint64_t)g.sort.plan.windows * n;
    } catch (...) {
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      throw;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }

  static void msm_host(ark355_ctx* ctx, GenericScratch& g, int group, const uint8_t* bases, const uint8_t* scalars,
                       uint64_t n, uint8_t* out) {
    hipStream_t st = ctx->stream;
    const size_t psz = group == 1 ? sizeof(Affine<Fq>) : sizeof(Affine<Fq2>);
    g.a.ensure(n * psz);
    g.b.ensure(n * sizeof(Fr));
    if (n) {
      ARK_CHECK_HIP(hipMemcpyAsync(g.a.p, bases, n * psz, hipMemcpyHostToDevice, st));
      ARK_CHECK_HIP(hipMemcpyAsync(g.b.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, st));
    }
    if (group == 1) msm_generic<Fq>(ctx, g, g.a.as<Affine<Fq>>(), g.b.p, n, 0, out, true);
    else msm_generic<Fq2>(ctx, g, g.a.as<Affine<Fq2>>(), g.b.p, n, 0, out, true);
  }

  static BasesDev* bases_load(const TunePolicy& pol, int group, const uint8_t* bases, uint64_t n, hipStream_t st) {
    auto* b = new BasesDev();
    try {
      b->curve = Curve::ID;
      b->group = group;
      b->n = n;
      const size_t psz = group == 1 ? sizeof(Affine<Fq>) : sizeof(Affine<Fq2>);
      DevBuf stage((n ? n : 1) * psz);
      if (n) ARK_CHECK_HIP(hipMemcpy(stage.p, bases, n * psz, hipMemcpyHostToDevice));
      const TableNeed need{n, 0, group == 2};
      std::string why;
      bool packed = false;
      const uint32_t ws = table_stride_plan<Fq, Fq2, Fr>(pol, &need, 1, table_budget_bytes(pol, (size_t)2 * 16 * 17 * n, true), &why, &packed);
      if (ws == 0) throw HipError{ARK355_ENOMEM, "base set: " + why};
      if (group == 1) precomp_build<Fq, Fr>(pol, b->tab, stage.p, n, st, 0, ws, 0, packed ? 1 : 0);
      else precomp_build<Fq2, Fr>(pol, b->tab, stage.p, n, st, 0, ws, 0, packed ? 1 : 0);
    } catch (...) {
      delete b;
      throw;
    }
    return b;
  }

  static void msm_dev(ark355_ctx* ctx, GenericScratch& g, const BasesDev& b, const void* d_scalars, uint64_t n, int mont,
                      uint8_t* out, bool want_affine) {
    ARK_REQUIRE(n <= b.n, ARK355_EINVAL, "more scalars than bases");
    if (b.group == 1) msm_generic<Fq>(ctx, g, b.tab.table.as<Affine<Fq>>(), d_scalars, n, mont, out, want_affine, &b.tab);
    else msm_generic<Fq2>(ctx, g, b.tab.table.as<Affine<Fq2>>(), d_scalars, n, mont, out, want_affine, &b.tab);
  }

  template <class F>
  static void xyzz_sum_t(ark355_ctx* ctx, GenericScratch& g, const uint8_t* partials, uint64_t count, uint8_t* out) {
    hipStream_t st = ctx->stream;
    g.a.ensure(count * sizeof(XYZZ<F>) + sizeof(XYZZ<F>) + sizeof(Affine<F>));
    XYZZ<F>* d_in = g.a.as<XYZZ<F>>();
    XYZZ<F>* d_sum = d_in + count;
    if (count) ARK_CHECK_HIP(hipMemcpyAsync(d_in, partials, count * sizeof(XYZZ<F>), hipMemcpyHostToDevice, st));
    ARK_LAUNCH((xyzz_sum_kernel<F>), dim3(1), dim3(64), 0, st, (const XYZZ<F>*)d_in, (uint32_t)count, d_sum);
    ARK_CHECK_LAUNCH();
    XYZZ<F> h_sum;
    ARK_CHECK_HIP(hipMemcpyAsync(&h_sum, d_sum, sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
    const Affine<F> a = xyzz_to_affine(h_sum);      // host-side normalisation (see msm_generic)
    memcpy(out, &a, sizeof(a));
  }

  static void xyzz_sum(ark355_ctx* ctx, GenericScratch& g, int group, const uint8_t* partials, uint64_t count,
                       uint8_t* out) {
    if (group == 1) xyzz_sum_t<Fq>(ctx, g, partials, count, out);
    else xyzz_sum_t<Fq2>(ctx, g, partials, count, out);
  }

  template <class F>
  static void fixed_base_t(ark355_ctx* ctx, GenericScratch& g, const uint8_t* base, const uint8_t* scalars, uint64_t n,
                           uint8_t* out) {
    static_assert(Fr::N * 4 == FB_WINDOWS, "one 8-bit window per scalar byte");
    hipStream_t st = ctx->stream;
    const uint32_t rows = FB_WINDOWS * FB_ROW;
    DevBuf d_base(sizeof(Affine<F>)), d_tx((size_t)rows * sizeof(XYZZ<F>)), d_ta((size_t)rows * sizeof(Affine<F>));
    g.b.ensure(n * sizeof(Fr));
    g.a.ensure((n ? n : 1) * sizeof(XYZZ<F>));
    g.c.ensure((n ? n : 1) * sizeof(Affine<F>));
    ARK_CHECK_HIP(hipMemcpyAsync(d_base.p, base, sizeof(Affine<F>), hipMemcpyHostToDevice, st));
    if (n) {
      ARK_REQUIRE(n < (1ull << 32), ARK355_EINVAL, "too many scalars");
      ARK_CHECK_HIP(hipMemcpyAsync(g.b.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, st));
      if (n < 512) {
        ARK_LAUNCH((fixed_base_small_kernel<F, Fr>), dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, st,
                   (const Affine<F>*)d_base.as<Affine<F>>(), g.b.as<Fr>(), n, g.a.as<XYZZ<F>>());
        ARK_CHECK_LAUNCH();
      } else {
        ARK_LAUNCH((fixed_base_table_kernel<F>), dim3(FB_WINDOWS), dim3(256), 0, st, (const Affine<F>*)d_base.as<Affine<F>>(),
                   d_tx.as<XYZZ<F>>());
        ARK_CHECK_LAUNCH();
        ARK_LAUNCH((batch_to_affine_kernel<F>), dim3(((rows + PRE_K - 1) / PRE_K + MSM_THREADS - 1) / MSM_THREADS),
                   dim3(MSM_THREADS), 0, st, (const XYZZ<F>*)d_tx.as<XYZZ<F>>(), d_ta.as<Affine<F>>(), rows);
        ARK_CHECK_LAUNCH();
        ARK_LAUNCH((fixed_base_mul_kernel<F, Fr>), dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, st,
                   (const Affine<F>*)d_ta.as<Affine<F>>(), g.b.as<Fr>(), n, g.a.as<XYZZ<F>>());
        ARK_CHECK_LAUNCH();
      }
      ARK_LAUNCH((batch_to_affine_kernel<F>), dim3((uint32_t)(((n + PRE_K - 1) / PRE_K + MSM_THREADS - 1) / MSM_THREADS)),
                 dim3(MSM_THREADS), 0, st, (const XYZZ<F>*)g.a.as<XYZZ<F>>(), g.c.as<Affine<F>>(), (uint32_t)n);
      ARK_CHECK_LAUNCH();
      ARK_CHECK_HIP(hipMemcpyAsync(out, g.c.p, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, st));
    }
    ARK_CHECK_HIP(hipStreamSynchronize(st));       // the table buffers are freed on return
  }

  static void fixed_base(ark355_ctx* ctx, GenericScratch& g, int group, const uint8_t* base, const uint8_t* scalars,
                         uint64_t n, uint8_t* out) {
    if (group == 1) fixed_base_t<Fq>(ctx, g, base, scalars, n, out);
    else fixed_base_t<Fq2>(ctx, g, base, scalars, n, out);
  }

  // ---- Groth16 generator scalars on the host (the library's own field code, std::thread) ------------------------------
  // Upstream `generate_parameters_with_qap` / `R1CSToQAP::instance_map_with_evaluation` (SURVEY.md Appendix A "Setup"):
  //   L_k(tau) = Z(tau)/N * w^k / (tau - w^k);  u_j = sum_k L_k A[k][j] (+ L_{n+j} for j < ell), v_j, w_j likewise;
  //   l_j = (beta u_j + alpha v_j + w_j) / delta (witness columns), gamma_abc_j = (...) / gamma (instance columns),
  //   h_i = Z(tau) tau^i / delta.  All outputs canonical 32-byte little-endian; the fixed-base multiplications that turn
  // them into the key's query vectors are ark355_fixed_base_mul's job.
  static void setup_scalars(uint64_t n, uint64_t ell, uint64_t w, const uint64_t* const rp[3], const uint32_t* const col[3],
                            const uint8_t* const coeff[3], const uint8_t* trapdoor, uint8_t* out_u, uint8_t* out_v,
                            uint8_t* out_w, uint8_t* out_l, uint8_t* out_gabc, uint8_t* out_h) {
    using P = typename Fr::Params;
    ARK_REQUIRE(ell >= 1, ARK355_EINVAL, "num_instance must include the constant One");
    const uint64_t m = ell + w;
    // the three CSR matrices come from the caller: monotone row pointers, columns inside [0, ell + w)
    for (int k = 0; k < 3; k++) {
      ARK_REQUIRE(rp[k][0] == 0, ARK355_EINVAL, "CSR row_ptr must start at 0");
      for (uint64_t i = 0; i < n; i++) ARK_REQUIRE(rp[k][i] <= rp[k][i + 1], ARK355_EINVAL, "CSR row_ptr is not monotone");
      const uint64_t nnz = rp[k][n];
      for (uint64_t t = 0; t < nnz; t++) ARK_REQUIRE(col[k][t] < m, ARK355_EINVAL, "CSR column index out of range");
    }
    uint32_t lg = 0;
    while ((1ull << lg) < n + ell) lg++;
    ARK_REQUIRE(lg <= (uint32_t)P::TWO_ADICITY, ARK355_E_POLY_DEGREE_TOO_LARGE, "n + ell exceeds the largest radix-2 domain of Fr");
    const uint64_t N = 1ull << lg;
    Fr td[5];
    for (int i = 0; i < 5; i++) {
      Fr c;
      memcpy(c.l, trapdoor + 32 * i, sizeof(Fr));
      td[i] = Fr::to_mont(c);
    }
    const Fr tau = td[0], alpha = td[1], beta = td[2], gamma = td[3], delta = td[4];
    const Fr omega = ntt_root<Fr>(lg, false);
    const Fr zt = Fr::sub(fr_pow_u64(tau, N), Fr::one());
    // tau inside the evaluation domain (Z(tau) = 0; probability N / r for an honest trapdoor, but a legal input): the
    // Lagrange coefficients degenerate to an indicator, L_k(tau) = [w^k == tau], as `evaluate_all_lagrange_coefficients`
    // upstream returns them; h_i = Z(tau) tau^i / delta = 0.
    const bool tau_in_domain = zt.is_zero();
    Fr nn = Fr::zero();
    nn.l[0] = (uint32_t)N;
    nn.l[1] = (uint32_t)(N >> 32);
    const Fr cN = Fr::mul(zt, Fr::inv(Fr::to_mont(nn)));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    auto parallel = [&](uint64_t count, const std::function<void(uint64_t, uint64_t)>& fn) {
      const uint64_t chunk = (count + nt - 1) / nt;
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; t++) {
        const uint64_t a = t * chunk, b = std::min<uint64_t>(count, a + chunk);
        if (a < b) th.emplace_back(fn, a, b);
      }
      for (auto& x : th) x.join();
    };
    std::vector<Fr> L(N), wk(N);
    if (tau_in_domain) {
      parallel(N, [&](uint64_t a, uint64_t b) {
        Fr p = fr_pow_u64(omega, a);
        for (uint64_t k = a; k < b; k++) {
          L[k] = (p == tau) ? Fr::one() : Fr::zero();
          p = Fr::mul(p, omega);
        }
      });
    } else
    parallel(N, [&](uint64_t a, uint64_t b) {
      Fr p = fr_pow_u64(omega, a), acc = Fr::one();
      for (uint64_t k = a; k < b; k++) {            // prefix products of the denominators of this chunk
        wk[k] = p;
        acc = Fr::mul(acc, Fr::sub(tau, p));
        L[k] = acc;
        p = Fr::mul(p, omega);
      }
      Fr inv = Fr::inv(acc);
      for (uint64_t k = b; k-- > a;) {
        const Fr iv = k > a ? Fr::mul(inv, L[k - 1]) : inv;
        inv = Fr::mul(inv, Fr::sub(tau, wk[k]));
        L[k] = Fr::mul(Fr::mul(iv, wk[k]), cN);
      }
    });
    std::vector<Fr> uvw[3];
    for (auto& x : uvw) x.assign(m, Fr::zero());
    for (uint64_t i = 0; i < ell; i++) uvw[0][i] = L[n + i];
    {
      const Fr one = Fr::one();
      std::vector<std::thread> th;
      for (int k = 0; k < 3; k++)
        th.emplace_back([&, k] {
          for (uint64_t i = 0; i < n; i++)
            for (uint64_t t = rp[k][i]; t < rp[k][i + 1]; t++) {
              Fr c;
              memcpy(c.l, coeff[k] + t * sizeof(Fr), sizeof(Fr));
              Fr& dst = uvw[k][col[k][t]];
              dst = Fr::add(dst, c == one ? L[i] : Fr::mul(L[i], c));
            }
        });
      for (auto& x : th) x.join();
    }
    const Fr gi = Fr::inv(gamma), di = Fr::inv(delta);
    auto put = [](uint8_t* dst, uint64_t i, const Fr& mont) {
      const Fr c = Fr::from_mont(mont);
      memcpy(dst + 32 * i, c.l, sizeof(Fr));
    };
    parallel(m, [&](uint64_t a, uint64_t b) {
      for (uint64_t i = a; i < b; i++) {
        const Fr abc = Fr::add(Fr::add(Fr::mul(beta, uvw[0][i]), Fr::mul(alpha, uvw[1][i])), uvw[2][i]);
        if (i < ell) put(out_gabc, i, Fr::mul(abc, gi));
        else put(out_l, i - ell, Fr::mul(abc, di));
        put(out_u, i, uvw[0][i]);
        put(out_v, i, uvw[1][i]);
        put(out_w, i, uvw[2][i]);
      }
    });
    const Fr h0 = Fr::mul(zt, di);
    parallel(N - 1, [&](uint64_t a, uint64_t b) {
      Fr p = Fr::mul(fr_pow_u64(tau, a), h0);
      for (uint64_t i = a; i < b; i++) {
        put(out_h, i, p);
        p = Fr::mul(p, tau);
      }
    });
  }

  // ---- batch verification (pairing_host.hpp) --------------------------------------------------------------------------
  // sum_j rho_j [ e(A_j, B_j) = e(alpha, beta) e(acc_j, gamma) e(C_j, delta) ]  <=>
  //   prod_j e(rho_j A_j, B_j) * e(-(sum rho_j) alpha, beta) * e(-sum_i (sum_j rho_j x_ji) gamma_abc_i, gamma)
  //                            * e(-sum_j rho_j C_j, delta) = 1
  // (k + 3 Miller loops and one final exponentiation for k proofs).  The two multi-scalar sums run on the device.
  static bool verify_batch(ark355_ctx* ctx, GenericScratch& g, const ark355_vk_desc* vk, const ark355_proof_raw* proofs,
                           const uint8_t* inputs, const uint8_t* rho, uint64_t count) {
    using PH = PairingHost<Curve>;
    const uint64_t ell = vk->num_instance;
    ARK_REQUIRE(ell >= 1 && count >= 1, ARK355_EINVAL, "empty batch or key");
    ARK_REQUIRE(rho || count == 1, ARK355_EINVAL, "a batch needs one random coefficient per proof");
    auto g1_of = [](const uint8_t* p) {
      Affine<Fq> a;
      memcpy(&a, p, sizeof(a));
      return a;
    };
    auto g2_of = [](const uint8_t* p) {
      Affine<Fq2> a;
      memcpy(&a, p, sizeof(a));
      return a;
    };
    // The raw entry point takes Montgomery images, not validated encodings: a point off the curve must not reach the
    // affine Miller loop (its line functions never use the curve constant, (0, y) / y = 0 cases would divide by zero
    // silently).  Three curve equations per proof on the host; subgroup membership is ark355_proof_from_bytes' job
    // (ARK355_VALIDATE_FULL), as upstream splits it between deserialization and verification.
    for (uint64_t j = 0; j < count; j++) {
      const Affine<Fq> a = g1_of(proofs[j].a), c = g1_of(proofs[j].c);
      const Affine<Fq2> b = g2_of(proofs[j].b);
      if (!a.is_inf() && !(Fq::sqr_ni(a.y) == W::curve_rhs(a.x))) return false;
      if (!c.is_inf() && !(Fq::sqr_ni(c.y) == W::curve_rhs(c.x))) return false;
      if (!b.is_inf() && !(Fq2::sqr_ni(b.y) == W::curve_rhs(b.x))) return false;
    }
    std::vector<Fr> r(count), coef(ell, Fr::zero());
    for (uint64_t j = 0; j < count; j++) {
      if (rho) {
        Fr c;
        memcpy(c.l, rho + 32 * j, sizeof(Fr));
        r[j] = Fr::to_mont(c);
        ARK_REQUIRE(!r[j].is_zero(), ARK355_EINVAL, "zero random coefficient");
      } else {
        r[j] = Fr::one();
      }
      coef[0] = Fr::add(coef[0], r[j]);
      for (uint64_t i = 1; i < ell; i++) {
        Fr x;
        memcpy(x.l, inputs + ((size_t)j * (ell - 1) + (i - 1)) * sizeof(Fr), sizeof(Fr));
        coef[i] = Fr::add(coef[i], Fr::mul(r[j], x));
      }
    }
    // device MSMs over canonical scalars
    std::vector<uint8_t> sc(std::max<uint64_t>(ell, count) * sizeof(Fr)), cpts(count * sizeof(Affine<Fq>));
    Affine<Fq> acc, csum;
    for (uint64_t i = 0; i < ell; i++) {
      const Fr c = Fr::from_mont(coef[i]);
      memcpy(sc.data() + i * sizeof(Fr), c.l, sizeof(Fr));
    }
    msm_host(ctx, g, 1, vk->gamma_abc_g1, sc.data(), ell, reinterpret_cast<uint8_t*>(&acc));
    for (uint64_t j = 0; j < count; j++) {
      const Fr c = Fr::from_mont(r[j]);
      memcpy(sc.data() + j * sizeof(Fr), c.l, sizeof(Fr));
      memcpy(cpts.data() + j * sizeof(Affine<Fq>), proofs[j].c, sizeof(Affine<Fq>));
    }
    msm_host(ctx, g, 1, cpts.data(), sc.data(), count, reinterpret_cast<uint8_t*>(&csum));
    std::vector<Affine<Fq>> Ps(count + 3);
    std::vector<Affine<Fq2>> Qs(count + 3);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 16) nt = 16;
    {
      // rho_j A_j on host threads (one 255-bit scalar multiplication each)
      std::vector<std::thread> th;
      const unsigned k = (unsigned)std::min<uint64_t>(nt, count);
      for (unsigned t = 0; t < k; t++)
        th.emplace_back([&, t] {
          for (uint64_t j = t; j < count; j += k) {
            const Affine<Fq> a = g1_of(proofs[j].a);
            const Fr c = Fr::from_mont(r[j]);
            Ps[j] = rho ? xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq>::from_affine(a), c.l, Fr::N)) : a;
            Qs[j] = g2_of(proofs[j].b);
          }
        });
      for (auto& x : th) x.join();
    }
    const Fr s0 = Fr::from_mont(coef[0]);
    const Affine<Fq> alpha = g1_of(vk->alpha_g1);
    Ps[count] = Affine<Fq>::neg(xyzz_to_affine(xyzz_mul_scalar(XYZZ<Fq>::from_affine(alpha), s0.l, Fr::N)));
    Qs[count] = g2_of(vk->beta_g2);
    Ps[count + 1] = acc.is_inf() ? acc : Affine<Fq>::neg(acc);
    Qs[count + 1] = g2_of(vk->gamma_g2);
    Ps[count + 2] = csum.is_inf() ? csum : Affine<Fq>::neg(csum);
    Qs[count + 2] = g2_of(vk->delta_g2);
    return PH::product_is_one(Ps, Qs, nt);
  }

  // ---- ark-serialize wire formats (wire_impl.cuh) ------------------------------------------------------------------
  static size_t point_size(int group, bool compressed) { return group == 1 ? W::g1_size(compressed) : W::g2_size(compressed); }
  static size_t raw_size(int group) { return group == 1 ? sizeof(Affine<Fq>) : sizeof(Affine<Fq2>); }

  // d_in: device byte stream of n encoded points -> d_out: n raw affine images (device).  Throws on a bad point.
  static void decode_dev(int group, const uint8_t* d_in, uint64_t n, bool compressed, int validate, void* d_out,
                         DevBuf& errbuf, hipStream_t st, const char* what) {
    if (n == 0) return;
    errbuf.ensure(8);
    ARK_CHECK_HIP(hipMemsetAsync(errbuf.p, 0, 8, st));
    const dim3 grid((uint32_t)((n + 127) / 128));
    if (group == 1)
      ARK_LAUNCH((wire_decode_kernel<Curve, 1>), grid, dim3(128), 0, st, d_in, n, compressed ? 1 : 0, validate, d_out,
                 errbuf.as<unsigned long long>());
    else
      ARK_LAUNCH((wire_decode_kernel<Curve, 2>), grid, dim3(128), 0, st, d_in, n, compressed ? 1 : 0, validate, d_out,
                 errbuf.as<unsigned long long>());
    ARK_CHECK_LAUNCH();
    unsigned long long e = 0;
    ARK_CHECK_HIP(hipMemcpyAsync(&e, errbuf.p, 8, hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
    if (e != 0)
      throw HipError{ARK355_EINVAL, std::string(what) + "[" + std::to_string((e >> 4) - 1) + "]: " + wire_status_name((int)(e & 15))};
  }

  static void points_decode(ark355_ctx* ctx, GenericScratch& g, int group, const uint8_t* in, uint64_t n, bool compressed,
                            int validate, uint8_t* out_raw) {
    hipStream_t st = ctx->stream;
    if (n == 0) return;
    g.a.ensure(n * point_size(group, compressed));
    g.b.ensure(n * raw_size(group));
    ARK_CHECK_HIP(hipMemcpyAsync(g.a.p, in, n * point_size(group, compressed), hipMemcpyHostToDevice, st));
    decode_dev(group, g.a.as<uint8_t>(), n, compressed, validate, g.b.p, g.c, st, "point");
    ARK_CHECK_HIP(hipMemcpyAsync(out_raw, g.b.p, n * raw_size(group), hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  static void points_encode(ark355_ctx* ctx, GenericScratch& g, int group, const uint8_t* in_raw, uint64_t n,
                            bool compressed, uint8_t* out) {
    hipStream_t st = ctx->stream;
    if (n == 0) return;
    g.a.ensure(n * raw_size(group));
    g.b.ensure(n * point_size(group, compressed));
    ARK_CHECK_HIP(hipMemcpyAsync(g.a.p, in_raw, n * raw_size(group), hipMemcpyHostToDevice, st));
    const dim3 grid((uint32_t)((n + 127) / 128));
    if (group == 1)
      ARK_LAUNCH((wire_encode_kernel<Curve, 1>), grid, dim3(128), 0, st, (const void*)g.a.p, n, compressed ? 1 : 0, g.b.as<uint8_t>());
    else
      ARK_LAUNCH((wire_encode_kernel<Curve, 2>), grid, dim3(128), 0, st, (const void*)g.a.p, n, compressed ? 1 : 0, g.b.as<uint8_t>());
    ARK_CHECK_LAUNCH();
    ARK_CHECK_HIP(hipMemcpyAsync(out, g.b.p, n * point_size(group, compressed), hipMemcpyDeviceToHost, st));
    ARK_CHECK_HIP(hipStreamSynchronize(st));
  }

  // Proof = a || b || c (three points: host code)
  static void proof_to_bytes(const ark355_proof_raw* p, bool compressed, uint8_t* out) {
    Affine<Fq> a, c;
    Affine<Fq2> b;
    memcpy(&a, p->a, sizeof(a));
    memcpy(&b, p->b, sizeof(b));
    memcpy(&c, p->c, sizeof(c));
    W::g1_encode(a, compressed, out);
    W::g2_encode(b, compressed, out + W::g1_size(compressed));
    W::g1_encode(c, compressed, out + W::g1_size(compressed) + W::g2_size(compressed));
  }
  static void proof_from_bytes(const uint8_t* in, uint64_t len, bool compressed, int validate, ark355_proof_raw* out) {
    ARK_REQUIRE(len == 2 * W::g1_size(compressed) + W::g2_size(compressed), ARK355_EINVAL, "bad proof length");
    Affine<Fq> a, c;
    Affine<Fq2> b;
    int st = W::g1_decode(in, compressed, validate, &a);
    if (st == WIRE_OK) st = W::g2_decode(in + W::g1_size(compressed), compressed, validate, &b);
    if (st == WIRE_OK) st = W::g1_decode(in + W::g1_size(compressed) + W::g2_size(compressed), compressed, validate, &c);
    if (st != WIRE_OK) throw HipError{ARK355_EINVAL, std::string("proof: ") + wire_status_name(st)};
    memset(out, 0, sizeof(*out));
    memcpy(out->a, &a, sizeof(a));
    memcpy(out->b, &b, sizeof(b));
    memcpy(out->c, &c, sizeof(c));
  }

  // ark_groth16::ProvingKey<E> stream (vk {alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1}, beta_g1, delta_g1,
  // a_query, b_g1_query, b_g2_query, h_query, l_query; Vec = u64 LE length + elements) -> resident key.  The host walks
  // the structure, the device decodes the points.
  static PkDev* pk_load_bytes(ark355_ctx* ctx, const uint8_t* bytes, uint64_t len, bool compressed, int validate) {
    hipStream_t st = ctx->stream;
    const size_t s1 = W::g1_size(compressed), s2 = W::g2_size(compressed);
    uint64_t off = 0;
    auto need = [&](uint64_t k) { ARK_REQUIRE(k <= len && off <= len - k, ARK355_EINVAL, "truncated proving key"); };
    struct Span {
      uint64_t off = 0, n = 0;
      int group = 1;
    };
    auto single = [&](int group) {
      Span sp;
      sp.group = group;
      sp.off = off;
      sp.n = 1;
      need(group == 1 ? s1 : s2);
      off += group == 1 ? s1 : s2;
      return sp;
    };
    auto vec = [&](int group) {
      need(8);
      uint64_t n = 0;
      memcpy(&n, bytes + off, 8);
      off += 8;
      const uint64_t sz = group == 1 ? s1 : s2;
      ARK_REQUIRE(n <= (len - off) / sz, ARK355_EINVAL, "vector length exceeds the stream");
      Span sp;
      sp.group = group;
      sp.off = off;
      sp.n = n;
      off += n * sz;
      return sp;
    };
    const Span alpha = single(1), beta2 = single(2), gamma2 = single(2), delta2 = single(2), gabc = vec(1);
    const Span beta1 = single(1), delta1 = single(1), aq = vec(1), b1q = vec(1), b2q = vec(2), hq = vec(1), lq = vec(1);
    (void)gamma2;
    ARK_REQUIRE(off == len, ARK355_EINVAL, "trailing bytes in proving key");
    const uint64_t ell = gabc.n, m = aq.n, w = lq.n, N = hq.n + 1;
    ARK_REQUIRE(ell >= 1 && m == ell + w && b1q.n == m && b2q.n == m && (N & (N - 1)) == 0, ARK355_EINVAL,
                "proving key: inconsistent query lengths");
    DevBuf d_bytes(len), d_err;
    ARK_CHECK_HIP(hipMemcpyAsync(d_bytes.p, bytes, len, hipMemcpyHostToDevice, st));
    auto decode_vec = [&](const Span& sp, DevBuf& out, const char* what) {
      out.alloc((sp.n ? sp.n : 1) * raw_size(sp.group));
      decode_dev(sp.group, d_bytes.as<uint8_t>() + sp.off, sp.n, compressed, validate, out.p, d_err, st, what);
    };
    DevBuf d_a, d_b1, d_b2, d_h, d_l;
    decode_vec(aq, d_a, "a_query");
    decode_vec(b1q, d_b1, "b_g1_query");
    decode_vec(b2q, d_b2, "b_g2_query");
    decode_vec(hq, d_h, "h_query");
    decode_vec(lq, d_l, "l_query");
    Affine<Fq> h_alpha, h_beta1, h_delta1;
    Affine<Fq2> h_beta2, h_delta2;
    auto dec1 = [&](const Span& sp, Affine<Fq>* o, const char* what) {
      const int e = W::g1_decode(bytes + sp.off, compressed, validate, o);
      if (e != WIRE_OK) throw HipError{ARK355_EINVAL, std::string(what) + ": " + wire_status_name(e)};
    };
    auto dec2 = [&](const Span& sp, Affine<Fq2>* o, const char* what) {
      const int e = W::g2_decode(bytes + sp.off, compressed, validate, o);
      if (e != WIRE_OK) throw HipError{ARK355_EINVAL, std::string(what) + ": " + wire_status_name(e)};
    };
    dec1(alpha, &h_alpha, "alpha_g1");
    dec1(beta1, &h_beta1, "beta_g1");
    dec1(delta1, &h_delta1, "delta_g1");
    dec2(beta2, &h_beta2, "beta_g2");
    dec2(delta2, &h_delta2, "delta_g2");
    ark355_pk_desc d{};
    d.num_instance = ell;
    d.num_witness = w;
    d.domain_size = N;
    d.a_query = d_a.as<uint8_t>();            // device pointers: pk_upload copies with hipMemcpyDefault
    d.b_g1_query = d_b1.as<uint8_t>();
    d.b_g2_query = d_b2.as<uint8_t>();
    d.h_query = d_h.as<uint8_t>();
    d.l_query = d_l.as<uint8_t>();
    d.alpha_g1 = reinterpret_cast<const uint8_t*>(&h_alpha);
    d.beta_g1 = reinterpret_cast<const uint8_t*>(&h_beta1);
    d.delta_g1 = reinterpret_cast<const uint8_t*>(&h_delta1);
    d.beta_g2 = reinterpret_cast<const uint8_t*>(&h_beta2);
    d.delta_g2 = reinterpret_cast<const uint8_t*>(&h_delta2);
    return pk_upload<Curve>(ctx->policy, &d, st);
  }
};

}  // namespace ark355
