/* TEST INFRASTRUCTURE (oracle) -- C restatement of the arkworks CPU algorithms on the Groth16 prove path.
 *
 * What it is: (i) the oracle at sizes the Python oracle cannot reach, (ii) the timed CPU baseline of
 * bench.py ("port": arkworks-algorithm CPU restatement, NOT arkworks itself -- the reference cannot be
 * compiled here: no Rust toolchain, and the arithmetic lives in un-vendored crates).
 * What it follows (published algorithms; SURVEY.md Appendix A is the normative spec):
 *   - ark-ff MontBackend (64-bit limbs, CIOS)                        -> fp_tmpl.h
 *   - ark-ec short_weierstrass Jacobian + VariableBaseMSM (Pippenger)  -> grp_tmpl.h
 *   - ark-poly Radix2EvaluationDomain fft/ifft/coset                   -> cb_ntt below
 *   - ark-groth16 r1cs_to_qap::witness_map_from_matrices, prover::create_proof_with_assignment
 *     behind ark_snark::SNARK::prove (/root/reference/snark/src/lib.rs:50-54)
 *   - mat_vec_mul (/root/reference/relations/src/utils/matrix.rs:26-36)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Memory images are the same as include/ark355.h (Montgomery LE limbs; affine infinity = zeros).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define NL 4
#define FP f4
#include "fp_tmpl.h"
#undef NL
#undef FP
#define NL 6
#define FP f6
#include "fp_tmpl.h"
#undef NL
#undef FP

#define BF f4
#define F2 f4x2
#include "fp2_tmpl.h"
#undef BF
#undef F2
#define BF f6
#define F2 f6x2
#include "fp2_tmpl.h"
#undef BF
#undef F2

#define FEP f4
#define GP bn_g1
#include "grp_tmpl.h"
#undef FEP
#undef GP
#define FEP f4x2
#define GP bn_g2
#include "grp_tmpl.h"
#undef FEP
#undef GP
#define FEP f6
#define GP bls_g1
#include "grp_tmpl.h"
#undef FEP
#undef GP
#define FEP f6x2
#define GP bls_g2
#include "grp_tmpl.h"
#undef FEP
#undef GP

typedef struct {
  int curve;             /* 0 = BLS12-381, 1 = BN254 */
  f4_params fr;
  f6_params fq6;         /* BLS Fq */
  f4_params fq4;         /* BN Fq  */
  int fr_bits, two_adicity;
  f4_t root;             /* 2^two_adicity-th root of unity (Montgomery) */
  f4_t gen;              /* multiplicative generator of Fr (Montgomery) */
} curve_ctx;

static curve_ctx g_curves[2];

/* params: all little-endian u64 arrays prepared by the Python side from the public constants */
void cb_init(int curve, const uint64_t* fr_mod, const uint64_t* fr_one, const uint64_t* fr_r2, uint64_t fr_inv,
             const uint64_t* fq_mod, const uint64_t* fq_one, const uint64_t* fq_r2, uint64_t fq_inv, int fr_bits,
             int two_adicity, const uint64_t* root_mont, const uint64_t* gen_mont) {
  curve_ctx* c = &g_curves[curve];
  c->curve = curve;
  memcpy(c->fr.mod, fr_mod, 32);
  memcpy(c->fr.one, fr_one, 32);
  memcpy(c->fr.r2, fr_r2, 32);
  c->fr.inv = fr_inv;
  if (curve == 0) {
    memcpy(c->fq6.mod, fq_mod, 48);
    memcpy(c->fq6.one, fq_one, 48);
    memcpy(c->fq6.r2, fq_r2, 48);
    c->fq6.inv = fq_inv;
  } else {
    memcpy(c->fq4.mod, fq_mod, 32);
    memcpy(c->fq4.one, fq_one, 32);
    memcpy(c->fq4.r2, fq_r2, 32);
    c->fq4.inv = fq_inv;
  }
  c->fr_bits = fr_bits;
  c->two_adicity = two_adicity;
  memcpy(c->root.l, root_mont, 32);
  memcpy(c->gen.l, gen_mont, 32);
}

int cb_num_threads(void) { return omp_get_max_threads(); }
void cb_set_threads(int n) { omp_set_num_threads(n); }

/* ---- radix-2 NTT over Fr, natural order in/out ------------------------------------------------------- */
static void fr_pow2k(f4_t* r, const f4_t* a, int k, const f4_params* P) {
  *r = *a;
  for (int i = 0; i < k; i++) f4_sqr(r, r, P);
}

/* p[i] = base^i * first, i < n, filled by chunks (each chunk starts from one exponentiation) */
static void fr_powers(f4_t* p, size_t n, const f4_t* base, const f4_t* first, const f4_params* P) {
  const int nt = omp_get_max_threads();
  const size_t chunk = (n + (size_t)nt - 1) / (size_t)nt;
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < nt; t++) {
    const size_t k0 = (size_t)t * chunk, k1 = (k0 + chunk < n) ? k0 + chunk : n;
    if (k0 >= k1) continue;
    uint64_t e[1] = {(uint64_t)k0};
    f4_t x;
    f4_pow(&x, base, e, 1, P);
    if (first) f4_mul(&x, &x, first, P);
    for (size_t k = k0; k < k1; k++) {
      p[k] = x;
      f4_mul(&x, &x, base, P);
    }
  }
}

static void ntt_core(f4_t* a, int log_n, const f4_t* w, const f4_params* P) {
  const size_t n = (size_t)1 << log_n;
  /* bit reversal */
#pragma omp parallel for schedule(static) if (n >= 4096)
  for (size_t i = 0; i < n; i++) {
    size_t j = 0;
    for (int b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
    if (i < j) {
      f4_t t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
  f4_t* tw = (f4_t*)malloc(sizeof(f4_t) * (n / 2 ? n / 2 : 1));
  /* tw[k] = w^k, k < n/2 */
  f4_set_one(&tw[0], P);
  if (n / 2 > 1) fr_powers(tw, n / 2, w, NULL, P);
  for (int s = 1; s <= log_n; s++) {
    const size_t len = (size_t)1 << s, half = len >> 1, step = n / len;
    /* one flat index over (block, k): late stages have few blocks but long ones (arkworks' parallel FFT splits
     * those as well) */
#pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t idx = 0; idx < n / 2; idx++) {
      const size_t blk = idx / half, k = idx % half;
      f4_t* x = a + blk * len;
      f4_t u = x[k], v;
      f4_mul(&v, &x[k + half], &tw[k * step], P);
      f4_add(&x[k], &u, &v, P);
      f4_sub(&x[k + half], &u, &v, P);
    }
  }
  free(tw);
}

/* data: 2^log_n Fr (Montgomery), in place */
int cb_ntt(int curve, uint64_t* data, int log_n, int inverse, int coset) {
  curve_ctx* c = &g_curves[curve];
  const f4_params* P = &c->fr;
  if (log_n > c->two_adicity) return -18;
  f4_t* a = (f4_t*)data;
  const size_t n = (size_t)1 << log_n;
  f4_t w, wi, gi, ninv;
  fr_pow2k(&w, &c->root, c->two_adicity - log_n, P);
  f4_inv(&wi, &w, P);
  f4_inv(&gi, &c->gen, P);
  f4_t* sc = (f4_t*)malloc(sizeof(f4_t) * n);
  if (!inverse) {
    if (coset) {
      fr_powers(sc, n, &c->gen, NULL, P);
#pragma omp parallel for schedule(static) if (n >= 4096)
      for (size_t i = 0; i < n; i++) f4_mul(&a[i], &a[i], &sc[i], P);
    }
    ntt_core(a, log_n, &w, P);
  } else {
    ntt_core(a, log_n, &wi, P);
    f4_t two, one;
    f4_set_one(&one, P);
    f4_add(&two, &one, &one, P);
    f4_inv(&ninv, &two, P);
    f4_t acc;
    f4_set_one(&acc, P);
    for (int i = 0; i < log_n; i++) f4_mul(&acc, &acc, &ninv, P);
    if (coset) {
      fr_powers(sc, n, &gi, &acc, P);
#pragma omp parallel for schedule(static) if (n >= 4096)
      for (size_t i = 0; i < n; i++) f4_mul(&a[i], &a[i], &sc[i], P);
    } else {
#pragma omp parallel for schedule(static) if (n >= 4096)
      for (size_t i = 0; i < n; i++) f4_mul(&a[i], &a[i], &acc, P);
    }
  }
  free(sc);
  return 0;
}

/* ---- R1CS: <row, z> ---------------------------------------------------------------------------------------- */
static void spmv(f4_t* out, size_t n, const uint64_t* rp, const uint32_t* col, const f4_t* coeff, const f4_t* z,
                 const f4_params* P) {
  f4_t one;
  f4_set_one(&one, P);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    f4_t acc;
    f4_set_zero(&acc);
    for (uint64_t k = rp[i]; k < rp[i + 1]; k++) {
      f4_t t;
      if (f4_eq(&coeff[k], &one)) t = z[col[k]];
      else f4_mul(&t, &coeff[k], &z[col[k]], P);
      f4_add(&acc, &acc, &t, P);
    }
    out[i] = acc;
  }
}

typedef struct {
  uint64_t n, ell, w;
  const uint64_t* rp[3];
  const uint32_t* col[3];
  const f4_t* coeff[3];
} r1cs_view;

static int domain_log(uint64_t need) {
  int lg = 0;
  while (((uint64_t)1 << lg) < need) lg++;
  return lg;
}

/* h[0..N) Montgomery */
static int witness_map(int curve, const r1cs_view* r, const f4_t* z, f4_t* h) {
  curve_ctx* c = &g_curves[curve];
  const f4_params* P = &c->fr;
  const int lg = domain_log(r->n + r->ell);
  if (lg > c->two_adicity) return -18;
  const size_t N = (size_t)1 << lg;
  f4_t* a = (f4_t*)calloc(N, sizeof(f4_t));
  f4_t* b = (f4_t*)calloc(N, sizeof(f4_t));
  f4_t* cc = (f4_t*)calloc(N, sizeof(f4_t));
  spmv(a, r->n, r->rp[0], r->col[0], r->coeff[0], z, P);
  spmv(b, r->n, r->rp[1], r->col[1], r->coeff[1], z, P);
  spmv(cc, r->n, r->rp[2], r->col[2], r->coeff[2], z, P);
  for (uint64_t j = 0; j < r->ell; j++) a[r->n + j] = z[j];
  cb_ntt(curve, (uint64_t*)a, lg, 1, 0);
  cb_ntt(curve, (uint64_t*)b, lg, 1, 0);
  cb_ntt(curve, (uint64_t*)cc, lg, 1, 0);
  cb_ntt(curve, (uint64_t*)a, lg, 0, 1);
  cb_ntt(curve, (uint64_t*)b, lg, 0, 1);
  cb_ntt(curve, (uint64_t*)cc, lg, 0, 1);
  f4_t gn, one, zinv;
  fr_pow2k(&gn, &c->gen, lg, P);
  f4_set_one(&one, P);
  f4_sub(&gn, &gn, &one, P);
  f4_inv(&zinv, &gn, P);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < N; i++) {
    f4_t t;
    f4_mul(&t, &a[i], &b[i], P);
    f4_sub(&t, &t, &cc[i], P);
    f4_mul(&h[i], &t, &zinv, P);
  }
  cb_ntt(curve, (uint64_t*)h, lg, 1, 1);
  free(a);
  free(b);
  free(cc);
  return lg;
}

int cb_witness_map(int curve, uint64_t n, uint64_t ell, uint64_t w, const uint64_t* rpa, const uint32_t* cola,
                   const uint64_t* cfa, const uint64_t* rpb, const uint32_t* colb, const uint64_t* cfb,
                   const uint64_t* rpc, const uint32_t* colc, const uint64_t* cfc, const uint64_t* z, uint64_t* h) {
  r1cs_view r = {n, ell, w, {rpa, rpb, rpc}, {cola, colb, colc}, {(const f4_t*)cfa, (const f4_t*)cfb, (const f4_t*)cfc}};
  int lg = witness_map(curve, &r, (const f4_t*)z, (f4_t*)h);
  return lg < 0 ? lg : 0;
}

int cb_mat_vec(int curve, uint64_t n, const uint64_t* rp, const uint32_t* col, const uint64_t* cf, const uint64_t* z,
               uint64_t* out) {
  spmv((f4_t*)out, n, rp, col, (const f4_t*)cf, (const f4_t*)z, &g_curves[curve].fr);
  return 0;
}

static void to_canon(uint64_t* dst, const f4_t* src, size_t n, const f4_params* P) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    f4_t t;
    f4_from_mont(&t, &src[i], P);
    memcpy(dst + 4 * i, t.l, 32);
  }
}

/* ---- MSM / fixed base: scalars canonical ---------------------------------------------------------------------- */
int cb_msm(int curve, int group, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out) {
  curve_ctx* c = &g_curves[curve];
  if (curve == 0 && group == 1) {
    bls_g1_jac r;
    bls_g1_msm(&r, (const bls_g1_aff*)bases, scalars, n, c->fr_bits, &c->fq6);
    bls_g1_to_affine((bls_g1_aff*)out, &r, &c->fq6);
  } else if (curve == 0) {
    bls_g2_jac r;
    bls_g2_msm(&r, (const bls_g2_aff*)bases, scalars, n, c->fr_bits, &c->fq6);
    bls_g2_to_affine((bls_g2_aff*)out, &r, &c->fq6);
  } else if (group == 1) {
    bn_g1_jac r;
    bn_g1_msm(&r, (const bn_g1_aff*)bases, scalars, n, c->fr_bits, &c->fq4);
    bn_g1_to_affine((bn_g1_aff*)out, &r, &c->fq4);
  } else {
    bn_g2_jac r;
    bn_g2_msm(&r, (const bn_g2_aff*)bases, scalars, n, c->fr_bits, &c->fq4);
    bn_g2_to_affine((bn_g2_aff*)out, &r, &c->fq4);
  }
  return 0;
}

int cb_fixed_base(int curve, int group, const uint64_t* base, const uint64_t* scalars, uint64_t n, uint64_t* out) {
  curve_ctx* c = &g_curves[curve];
  if (curve == 0 && group == 1) bls_g1_fixed_base((bls_g1_aff*)out, (const bls_g1_aff*)base, scalars, n, &c->fq6);
  else if (curve == 0) bls_g2_fixed_base((bls_g2_aff*)out, (const bls_g2_aff*)base, scalars, n, &c->fq6);
  else if (group == 1) bn_g1_fixed_base((bn_g1_aff*)out, (const bn_g1_aff*)base, scalars, n, &c->fq4);
  else bn_g2_fixed_base((bn_g2_aff*)out, (const bn_g2_aff*)base, scalars, n, &c->fq4);
  return 0;
}

/* ---- Groth16 prove (create_proof_with_reduction_and_matrices) ---------------------------------------------------- */
typedef struct {
  const uint64_t *a_query, *b_g1_query, *b_g2_query, *h_query, *l_query;
  const uint64_t *alpha_g1, *beta_g1, *delta_g1, *beta_g2, *delta_g2;
} pk_view;

#define PROVE_IMPL(NAME, G1, G2, FQP)                                                                                \
  static void NAME(curve_ctx* c, const pk_view* pk, uint64_t ell, uint64_t w, uint64_t N, const uint64_t* zc,      \
                   const uint64_t* hc, const uint64_t* r, const uint64_t* s, uint64_t* out_a, uint64_t* out_b,      \
                   uint64_t* out_c) {                                                                                \
    const uint64_t m = ell + w;                                                                                      \
    G1##_jac h_acc, l_acc, acc, g_a, g1_b, t, t2, g_c;                                                               \
    G2##_jac acc2, g2_b, u;                                                                                          \
    G1##_msm(&h_acc, (const G1##_aff*)pk->h_query, hc, N - 1, c->fr_bits, FQP);                                      \
    G1##_msm(&l_acc, (const G1##_aff*)pk->l_query, zc + 4 * ell, w, c->fr_bits, FQP);                                \
    const G1##_aff* aq = (const G1##_aff*)pk->a_query;                                                               \
    const G1##_aff* bq = (const G1##_aff*)pk->b_g1_query;                                                            \
    const G2##_aff* b2q = (const G2##_aff*)pk->b_g2_query;                                                           \
    /* g_a = r*delta + a_query[0] + MSM(a_query[1..], z[1..]) + alpha */                                             \
    G1##_msm(&acc, aq + 1, zc + 4, m - 1, c->fr_bits, FQP);                                                          \
    G1##_set_inf(&t, FQP);                                                                                           \
    G1##_madd(&t, &t, (const G1##_aff*)pk->delta_g1, FQP);                                                           \
    G1##_mul_scalar(&g_a, &t, r, FQP);                                                                               \
    G1##_madd(&g_a, &g_a, &aq[0], FQP);                                                                              \
    G1##_add(&g_a, &g_a, &acc, FQP);                                                                                 \
    G1##_madd(&g_a, &g_a, (const G1##_aff*)pk->alpha_g1, FQP);                                                       \
    /* g1_b */                                                                                                       \
    G1##_msm(&acc, bq + 1, zc + 4, m - 1, c->fr_bits, FQP);                                                          \
    G1##_mul_scalar(&g1_b, &t, s, FQP);                                                                              \
    G1##_madd(&g1_b, &g1_b, &bq[0], FQP);                                                                            \
    G1##_add(&g1_b, &g1_b, &acc, FQP);                                                                               \
    G1##_madd(&g1_b, &g1_b, (const G1##_aff*)pk->beta_g1, FQP);                                                      \
    /* g2_b */                                                                                                       \
    G2##_msm(&acc2, b2q + 1, zc + 4, m - 1, c->fr_bits, FQP);                                                        \
    G2##_set_inf(&u, FQP);                                                                                           \
    G2##_madd(&u, &u, (const G2##_aff*)pk->delta_g2, FQP);                                                           \
    G2##_mul_scalar(&g2_b, &u, s, FQP);                                                                              \
    G2##_madd(&g2_b, &g2_b, &b2q[0], FQP);                                                                           \
    G2##_add(&g2_b, &g2_b, &acc2, FQP);                                                                              \
    G2##_madd(&g2_b, &g2_b, (const G2##_aff*)pk->beta_g2, FQP);                                                      \
    /* g_c = s*g_a + r*g1_b - (r*s)*delta + l_acc + h_acc */                                                         \
    G1##_mul_scalar(&g_c, &g_a, s, FQP);                                                                             \
    G1##_mul_scalar(&t2, &g1_b, r, FQP);                                                                             \
    G1##_add(&g_c, &g_c, &t2, FQP);                                                                                  \
    {                                                                                                                \
      f4_t rm, sm, rs;                                                                                               \
      memcpy(rm.l, r, 32);                                                                                           \
      memcpy(sm.l, s, 32);                                                                                           \
      f4_to_mont(&rm, &rm, &c->fr);                                                                                  \
      f4_to_mont(&sm, &sm, &c->fr);                                                                                  \
      f4_mul(&rs, &rm, &sm, &c->fr);                                                                                 \
      f4_neg(&rs, &rs, &c->fr);                                                                                      \
      f4_from_mont(&rs, &rs, &c->fr);                                                                                \
      G1##_mul_scalar(&t2, &t, rs.l, FQP);                                                                           \
    }                                                                                                                \
    G1##_add(&g_c, &g_c, &t2, FQP);                                                                                  \
    G1##_add(&g_c, &g_c, &l_acc, FQP);                                                                               \
    G1##_add(&g_c, &g_c, &h_acc, FQP);                                                                               \
    G1##_to_affine((G1##_aff*)out_a, &g_a, FQP);                                                                     \
    G2##_to_affine((G2##_aff*)out_b, &g2_b, FQP);                                                                    \
    G1##_to_affine((G1##_aff*)out_c, &g_c, FQP);                                                                     \
  }

PROVE_IMPL(prove_bls, bls_g1, bls_g2, &c->fq6)
PROVE_IMPL(prove_bn, bn_g1, bn_g2, &c->fq4)

/* timings_out (optional, 4 doubles): witness map, MSMs + tail, total, spare */
int cb_prove(int curve, uint64_t n, uint64_t ell, uint64_t w, const uint64_t* rpa, const uint32_t* cola,
             const uint64_t* cfa, const uint64_t* rpb, const uint32_t* colb, const uint64_t* cfb, const uint64_t* rpc,
             const uint32_t* colc, const uint64_t* cfc, const uint64_t* z, const uint64_t* a_query,
             const uint64_t* b_g1_query, const uint64_t* b_g2_query, const uint64_t* h_query, const uint64_t* l_query,
             const uint64_t* alpha_g1, const uint64_t* beta_g1, const uint64_t* delta_g1, const uint64_t* beta_g2,
             const uint64_t* delta_g2, const uint64_t* r, const uint64_t* s, uint64_t* out_a, uint64_t* out_b,
             uint64_t* out_c, double* timings_out) {
  curve_ctx* c = &g_curves[curve];
  r1cs_view rv = {n, ell, w, {rpa, rpb, rpc}, {cola, colb, colc}, {(const f4_t*)cfa, (const f4_t*)cfb, (const f4_t*)cfc}};
  const int lg = domain_log(n + ell);
  if (lg > c->two_adicity) return -18;
  const uint64_t N = (uint64_t)1 << lg, m = ell + w;
  double t0 = omp_get_wtime();
  f4_t* h = (f4_t*)calloc(N, sizeof(f4_t));
  witness_map(curve, &rv, (const f4_t*)z, h);
  double t1 = omp_get_wtime();
  uint64_t* hc = (uint64_t*)malloc(32 * N);
  uint64_t* zc = (uint64_t*)malloc(32 * m);
  to_canon(hc, h, N, &c->fr);
  to_canon(zc, (const f4_t*)z, m, &c->fr);
  pk_view pk = {a_query, b_g1_query, b_g2_query, h_query, l_query, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2};
  if (curve == 0) prove_bls(c, &pk, ell, w, N, zc, hc, r, s, out_a, out_b, out_c);
  else prove_bn(c, &pk, ell, w, N, zc, hc, r, s, out_a, out_b, out_c);
  double t2 = omp_get_wtime();
  if (timings_out) {
    timings_out[0] = t1 - t0;
    timings_out[1] = t2 - t1;
    timings_out[2] = t2 - t0;
    timings_out[3] = 0;
  }
  free(h);
  free(hc);
  free(zc);
  return 0;
}

/* ---- Groth16 generator scalars (ark-groth16 generator.rs generate_parameters_with_qap, R1CSToQAP::
 * instance_map_with_evaluation; SURVEY.md Appendix A "Setup") for keys at sizes the Python oracle cannot reach.
 * td: tau, alpha, beta, gamma, delta (canonical, 4 limbs each).  Outputs (canonical scalars, 4 limbs each):
 *   u, v, w          m each            u_i(tau), v_i(tau), w_i(tau)
 *   l_s              w                 (beta u_i + alpha v_i + w_i) / delta,  i >= ell
 *   gabc_s           ell               (beta u_i + alpha v_i + w_i) / gamma,  i <  ell
 *   h_s              N - 1             Z(tau) tau^i / delta
 * The fixed-base multiplications are done by the caller with cb_fixed_base.  Checked against oracle/groth16.py
 * (qap_scalars / setup) in tests/test_oracle_c.py. */
static void fr_pow_u64(f4_t* r, const f4_t* a, uint64_t e, const f4_params* P) {
  uint64_t el[1] = {e};
  f4_pow(r, a, el, 1, P);
}

int cb_setup_scalars(int curve, uint64_t n, uint64_t ell, uint64_t wn, const uint64_t* rpa, const uint32_t* cola,
                     const uint64_t* cfa, const uint64_t* rpb, const uint32_t* colb, const uint64_t* cfb,
                     const uint64_t* rpc, const uint32_t* colc, const uint64_t* cfc, const uint64_t* td,
                     uint64_t* u_out, uint64_t* v_out, uint64_t* w_out, uint64_t* l_out, uint64_t* gabc_out,
                     uint64_t* h_out) {
  curve_ctx* c = &g_curves[curve];
  const f4_params* P = &c->fr;
  const int lg = domain_log(n + ell);
  if (lg > c->two_adicity) return -18;
  const size_t N = (size_t)1 << lg, m = ell + wn;
  f4_t tau, alpha, beta, gamma, delta, one;
  f4_set_one(&one, P);
  const f4_t* tdm[5] = {&tau, &alpha, &beta, &gamma, &delta};
  for (int i = 0; i < 5; i++) {
    f4_t t;
    memcpy(t.l, td + 4 * i, 32);
    f4_to_mont((f4_t*)tdm[i], &t, P);
  }
  f4_t omega, zt, ninv, cc;
  fr_pow2k(&omega, &c->root, c->two_adicity - lg, P);
  fr_pow_u64(&zt, &tau, (uint64_t)N, P);
  f4_sub(&zt, &zt, &one, P);
  {
    f4_t nn, t;
    uint64_t nl[4] = {(uint64_t)N, 0, 0, 0};
    memcpy(t.l, nl, 32);
    f4_to_mont(&nn, &t, P);
    f4_inv(&ninv, &nn, P);
  }
  f4_mul(&cc, &zt, &ninv, P);
  /* L_k(tau) = Z(tau)/N * omega^k / (tau - omega^k)  (tau outside H; the oracle never draws tau in H) */
  f4_t* L = (f4_t*)malloc(sizeof(f4_t) * N);
  f4_t* wk = (f4_t*)malloc(sizeof(f4_t) * N);
  if (f4_is_zero(&zt)) {
    free(L);
    free(wk);
    return -1;
  }
  const int nt = omp_get_max_threads();
  const size_t chunk = (N + (size_t)nt - 1) / (size_t)nt;
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < nt; t++) {
    const size_t k0 = (size_t)t * chunk, k1 = (k0 + chunk < N) ? k0 + chunk : N;
    if (k0 >= k1) continue;
    f4_t p;
    fr_pow_u64(&p, &omega, (uint64_t)k0, P);
    /* dens -> L (prefix products), then one inversion per chunk */
    f4_t acc = one;
    for (size_t k = k0; k < k1; k++) {
      wk[k] = p;
      f4_t d;
      f4_sub(&d, &tau, &p, P);
      f4_mul(&acc, &acc, &d, P);
      L[k] = acc;
      f4_mul(&p, &p, &omega, P);
    }
    f4_t inv;
    f4_inv(&inv, &acc, P);
    for (size_t k = k1; k-- > k0;) {
      f4_t d, iv;
      f4_sub(&d, &tau, &wk[k], P);
      if (k > k0) f4_mul(&iv, &inv, &L[k - 1], P);
      else iv = inv;
      f4_mul(&inv, &inv, &d, P);
      f4_mul(&iv, &iv, &wk[k], P);
      f4_mul(&L[k], &iv, &cc, P);
    }
  }
  free(wk);
  f4_t* uvw[3];
  for (int k = 0; k < 3; k++) uvw[k] = (f4_t*)calloc(m ? m : 1, sizeof(f4_t));
  for (uint64_t i = 0; i < ell; i++) uvw[0][i] = L[n + i];
  const uint64_t* rps[3] = {rpa, rpb, rpc};
  const uint32_t* cols[3] = {cola, colb, colc};
  const f4_t* cfs[3] = {(const f4_t*)cfa, (const f4_t*)cfb, (const f4_t*)cfc};
#pragma omp parallel for schedule(static, 1)
  for (int k = 0; k < 3; k++) {
    for (uint64_t i = 0; i < n; i++) {
      for (uint64_t t = rps[k][i]; t < rps[k][i + 1]; t++) {
        f4_t x;
        if (f4_eq(&cfs[k][t], &one)) x = L[i];
        else f4_mul(&x, &L[i], &cfs[k][t], P);
        f4_t* dst = &uvw[k][cols[k][t]];
        f4_add(dst, dst, &x, P);
      }
    }
  }
  free(L);
  f4_t gi, di;
  f4_inv(&gi, &gamma, P);
  f4_inv(&di, &delta, P);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < m; i++) {
    f4_t a, b, t;
    f4_mul(&a, &beta, &uvw[0][i], P);
    f4_mul(&b, &alpha, &uvw[1][i], P);
    f4_add(&a, &a, &b, P);
    f4_add(&a, &a, &uvw[2][i], P);
    if (i < ell) {
      f4_mul(&t, &a, &gi, P);
      f4_from_mont(&t, &t, P);
      memcpy(gabc_out + 4 * i, t.l, 32);
    } else {
      f4_mul(&t, &a, &di, P);
      f4_from_mont(&t, &t, P);
      memcpy(l_out + 4 * (i - ell), t.l, 32);
    }
  }
  to_canon(u_out, uvw[0], m, P);
  to_canon(v_out, uvw[1], m, P);
  to_canon(w_out, uvw[2], m, P);
  for (int k = 0; k < 3; k++) free(uvw[k]);
  f4_t h0;
  f4_mul(&h0, &zt, &di, P);
  const size_t hn = N - 1;
  const size_t hchunk = (hn + (size_t)nt - 1) / (size_t)nt;
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < nt; t++) {
    const size_t k0 = (size_t)t * hchunk, k1 = (k0 + hchunk < hn) ? k0 + hchunk : hn;
    if (k0 >= k1) continue;
    f4_t p, x;
    fr_pow_u64(&p, &tau, (uint64_t)k0, P);
    f4_mul(&p, &p, &h0, P);
    for (size_t k = k0; k < k1; k++) {
      f4_from_mont(&x, &p, P);
      memcpy(h_out + 4 * k, x.l, 32);
      f4_mul(&p, &p, &tau, P);
    }
  }
  return 0;
}

/* ---- ark-serialize point encodings (oracle side of the wire-format tests at BASELINE key sizes) -----------------------
 * Restates oracle/serialize.py (itself: ark-serialize + ark-bls12-381 curves/util.rs + ark-ec SWFlags; SURVEY.md
 * Appendix A "Serialisation") for VECTORS of raw affine points, so that a 2^20-constraint proving key (5.2 M points) can
 * be turned into a ProvingKey byte stream in seconds; checked against the Python encoders in tests/test_oracle_c.py.
 *   BLS12-381: big-endian coordinates, G2 as c1 || c0; flags in the top bits of the FIRST byte (0x80 compressed,
 *              0x40 infinity, 0x20 y is the lexicographically larger root);
 *   BN254:     little-endian coordinates, G2 as c0 || c1; flags in the top bits of the LAST byte (0x80 y > -y, 0x40 infinity).
 * raw: n points, x || y as Montgomery LE limbs (an all-zero point is infinity).  out: n * size bytes, size =
 * nb (G1 compressed), 2 nb (G1 uncompressed, G2 compressed), 4 nb (G2 uncompressed), nb = 48 / 32. */
static int limbs_gt(const uint64_t* a, const uint64_t* b, int nl) {
  for (int i = nl - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return 0;
}
static void limbs_out(uint8_t* dst, const uint64_t* l, int nl, int big_endian) {
  const int nb = 8 * nl;
  for (int i = 0; i < nb; i++) {
    const uint8_t v = (uint8_t)(l[i / 8] >> (8 * (i % 8)));
    dst[big_endian ? nb - 1 - i : i] = v;
  }
}
#define SER_BODY(FT, NLIMBS, PARAMS)                                                                                  \
  const int nl = NLIMBS, nb = 8 * NLIMBS, nc = group == 1 ? 1 : 2;                                                    \
  const size_t psz = (size_t)nb * nc * (compressed ? 1 : 2);                                                          \
  const int be = (curve == 0);                                                                                        \
  _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++) {                                       \
    const uint64_t* p = raw + i * (size_t)(2 * nc * nl);                                                              \
    uint8_t* o = out + i * psz;                                                                                       \
    memset(o, 0, psz);                                                                                                \
    int inf = 1;                                                                                                      \
    for (int k = 0; k < 2 * nc * nl; k++) inf &= (p[k] == 0);                                                         \
    uint8_t* flag = be ? o : o + psz - 1;                                                                             \
    if (inf) {                                                                                                        \
      *flag |= be ? (compressed ? 0xC0 : 0x40) : 0x40;                                                                \
      continue;                                                                                                       \
    }                                                                                                                 \
    FT##_t c[4], neg[2];                                                                                                 \
    for (int k = 0; k < 2 * nc; k++) {                                                                                \
      FT##_t m;                                                                                                          \
      memcpy(m.l, p + (size_t)k * nl, (size_t)nb);                                                                    \
      FT##_from_mont(&c[k], &m, PARAMS);                                                                              \
    }                                                                                                                 \
    /* y > -y, Fq2: c1 first, then c0 (canonical integers) */                                                         \
    int ygt;                                                                                                          \
    {                                                                                                                 \
      const FT##_t* y = c + nc;                                                                                          \
      for (int k = 0; k < nc; k++) {                                                                                  \
        int z = 1;                                                                                                    \
        for (int t = 0; t < nl; t++) z &= (y[k].l[t] == 0);                                                           \
        if (z) memset(neg[k].l, 0, (size_t)nb);                                                                       \
        else {                                                                                                        \
          uint64_t br = 0;                                                                                            \
          for (int t = 0; t < nl; t++) {                                                                              \
            const unsigned __int128 d = (unsigned __int128)(PARAMS)->mod[t] - y[k].l[t] - br;                         \
            neg[k].l[t] = (uint64_t)d;                                                                                \
            br = (uint64_t)(d >> 64) & 1;                                                                             \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
      if (nc == 2 && memcmp(y[1].l, neg[1].l, (size_t)nb) != 0) ygt = limbs_gt(y[1].l, neg[1].l, nl);                 \
      else ygt = limbs_gt(y[0].l, neg[0].l, nl);                                                                      \
    }                                                                                                                 \
    /* coordinate order on the wire */                                                                                \
    const int ncoord = compressed ? nc : 2 * nc;                                                                      \
    for (int k = 0; k < ncoord; k++) {                                                                                \
      const int comp = k / nc, idx = k % nc;                                                                          \
      const int src = comp * nc + (be && nc == 2 ? 1 - idx : idx);                                                    \
      limbs_out(o + (size_t)k * nb, c[src].l, nl, be);                                                                \
    }                                                                                                                 \
    if (be) {                                                                                                         \
      if (compressed) *flag |= 0x80 | (ygt ? 0x20 : 0);                                                               \
    } else if (ygt) {                                                                                                 \
      *flag |= 0x80;                                                                                                  \
    }                                                                                                                 \
  }                                                                                                                   \
  return 0;

int cb_points_serialize(int curve, int group, const uint64_t* raw, uint64_t n, int compressed, uint8_t* out) {
  if (curve == 0) {
    SER_BODY(f6, 6, &g_curves[0].fq6)
  } else {
    SER_BODY(f4, 4, &g_curves[1].fq4)
  }
}
