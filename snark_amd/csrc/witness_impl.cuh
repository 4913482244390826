// R1CS -> QAP witness map on the device (replaces the un-vendored
// ark-groth16 `LibsnarkReduction::witness_map_from_matrices`).
//
//   K8  CSR SpMV   a_i = <A_i, z>, b_i, c_i   -- semantics of mat_vec_mul
//                  (/root/reference/relations/src/utils/matrix.rs:26-36) and of
//                  Sr1csAdapter::evaluate_constraint (sr1cs/mod.rs:24-56, which skips the multiply when
//                  the coefficient is one :42-46).  Repeated columns in a row are summed.
//                  a_{n+j} = z_j for j < ell (input-consistency rows), everything else 0.
//   K6  6 NTTs     ntt_impl.cuh (the reference runs 7: see witness_map_run)
//   K7  pointwise  a'_i b'_i on the coset, then h_k = (rho_k - c_k) * (g^N - 1)^-1 on coefficients
//   a9  satisfaction check: first i with a_i b_i != c_i (which_constraint_is_unsatisfied,
//                  gr1cs/predicate/mod.rs:185-204)
#pragma once
#include <unordered_map>
#include "common.h"
#include "ntt_impl.cuh"

namespace ark355 {

struct R1csDev {
  int curve = 0;
  uint64_t n = 0, ell = 0, w = 0, m = 0, N = 0;
  uint32_t log_n = 0;
  uint64_t nnz[3] = {0, 0, 0};
  DevBuf row_ptr[3], col[3], cidx[3];
  DevBuf pool;          // interned coefficients; pool[0] == 1
  DevBuf zinv;          // (g^N - 1)^-1
};

template <class Fr>
__global__ void __launch_bounds__(256)
r1cs_spmv_kernel(const uint32_t* __restrict__ rp_a, const uint32_t* __restrict__ col_a, const uint32_t* __restrict__ ci_a,
                 const uint32_t* __restrict__ rp_b, const uint32_t* __restrict__ col_b, const uint32_t* __restrict__ ci_b,
                 const uint32_t* __restrict__ rp_c, const uint32_t* __restrict__ col_c, const uint32_t* __restrict__ ci_c,
                 const Fr* __restrict__ pool, const Fr* __restrict__ z, uint64_t n, uint64_t ell, uint64_t N,
                 Fr* __restrict__ out_a, Fr* __restrict__ out_b, Fr* __restrict__ out_c, uint64_t row0, uint64_t row_stride) {
  // out[j] = row (row0 + j * row_stride) for j < N: the whole domain (0, 1), or the residue class of one rank of the
  // distributed witness map (g, G; N = rows of that class)
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t mat = blockIdx.y;
  if (j >= N) return;
  const uint64_t i = row0 + j * row_stride;
  const uint32_t* rp = mat == 0 ? rp_a : (mat == 1 ? rp_b : rp_c);
  const uint32_t* col = mat == 0 ? col_a : (mat == 1 ? col_b : col_c);
  const uint32_t* ci = mat == 0 ? ci_a : (mat == 1 ? ci_b : ci_c);
  Fr* out = mat == 0 ? out_a : (mat == 1 ? out_b : out_c);
  Fr acc = Fr::zero();
  if (i < n) {
    const uint32_t lo = rp[i], hi = rp[i + 1];
    for (uint32_t k = lo; k < hi; k++) {
      Fr v = z[col[k]];
      const uint32_t c = ci[k];
      if (c != 0) v = Fr::mul(v, pool[c]);
      acc = Fr::add(acc, v);
    }
  } else if (mat == 0 && i - n < ell) {
    acc = z[i - n];
  }
  out[j] = acc;
}

template <class Fr>
__global__ void __launch_bounds__(256)
r1cs_check_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, uint64_t n,
                  unsigned long long* __restrict__ first_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (Fr::mul(a[i], b[i]) != c[i]) atomicMin(first_bad, (unsigned long long)i);
}

template <class Fr>
__global__ void __launch_bounds__(256)
qap_pointwise_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                     const Fr* __restrict__ zinv, uint64_t N, Fr* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  out[i] = Fr::mul(Fr::sub(Fr::mul(a[i], b[i]), c[i]), *zinv);
}

// the two elementwise steps of the six-transform map (witness_map_run)
template <class Fr>
__global__ void __launch_bounds__(256)
qap_mul_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b, uint64_t N, Fr* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  out[i] = Fr::mul(a[i], b[i]);
}
// h[k] = raw[k] * (g^-k / (N (g^N - 1))) - c[k] * cf     (raw: the inverse transform of a' b' without any scaling; c: the
// coefficients of c, or N times them, with cf = 1 / (g^N - 1) or that / N: ntt_quotient_tables).  In place on raw.
template <class Fr>
__global__ void __launch_bounds__(256)
qap_quotient_kernel(Fr* __restrict__ raw, const Fr* __restrict__ c, uint64_t N, const Fr* __restrict__ gi_lo,
                    const Fr* __restrict__ gz_hi, uint32_t lo_bits, const Fr* __restrict__ cf) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  raw[i] = Fr::sub(Fr::mul(raw[i], pow_lookup<Fr>(gi_lo, gz_hi, lo_bits, i)), Fr::mul(c[i], *cf));
}

template <class Curve>
static R1csDev* r1cs_upload(uint64_t n, uint64_t ell, uint64_t w, const uint64_t* const row_ptr[3],
                            const uint32_t* const col[3], const uint8_t* const coeff[3]) {
  using Fr = typename Curve::Fr;
  using P = typename Fr::Params;
  auto* r = new R1csDev();
  try {
    r->curve = Curve::ID;
    r->n = n;
    r->ell = ell;
    r->w = w;
    r->m = ell + w;
    ARK_REQUIRE(ell >= 1, ARK355_EINVAL, "num_instance must include the constant One");
    ARK_REQUIRE(r->m < (1ull << 31), ARK355_EINVAL, "too many variables");
    uint64_t need = n + ell;
    uint32_t lg = 0;
    while ((1ull << lg) < need) lg++;
    ARK_REQUIRE(lg <= (uint32_t)P::TWO_ADICITY, ARK355_E_POLY_DEGREE_TOO_LARGE,
                "n + ell exceeds the largest radix-2 domain of Fr");
    r->log_n = lg;
    r->N = 1ull << lg;
    // intern coefficients (pool[0] = one): mirrors the idea of field_interner.rs:24-57
    struct Key {
      uint32_t l[Fr::N];
      bool operator==(const Key& o) const { return memcmp(l, o.l, sizeof(l)) == 0; }
    };
    struct KeyHash {
      size_t operator()(const Key& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < Fr::N; i++) h = (h ^ k.l[i]) * 1099511628211ull;
        return (size_t)h;
      }
    };
    std::unordered_map<Key, uint32_t, KeyHash> interner;
    std::vector<Fr> pool;
    pool.push_back(Fr::one());
    {
      Key k;
      memcpy(k.l, pool[0].l, sizeof(k.l));
      interner.emplace(k, 0u);
    }
    for (int mtx = 0; mtx < 3; mtx++) {
      const uint64_t nnz = row_ptr[mtx][n];
      ARK_REQUIRE(nnz < (1ull << 32), ARK355_EINVAL, "nnz must be < 2^32");
      r->nnz[mtx] = nnz;
      std::vector<uint32_t> rp(n + 1), ci(nnz);
      for (uint64_t i = 0; i <= n; i++) {
        ARK_REQUIRE(row_ptr[mtx][i] <= nnz && (i == 0 || row_ptr[mtx][i] >= row_ptr[mtx][i - 1]), ARK355_EINVAL,
                    "row_ptr must be non-decreasing");
        rp[i] = (uint32_t)row_ptr[mtx][i];
      }
      for (uint64_t k = 0; k < nnz; k++) {
        ARK_REQUIRE(col[mtx][k] < r->m, ARK355_EINVAL, "column index out of range");
        Key key;
        memcpy(key.l, coeff[mtx] + k * sizeof(Fr), sizeof(key.l));
        auto it = interner.find(key);
        if (it == interner.end()) {
          Fr f;
          memcpy(f.l, key.l, sizeof(key.l));
          it = interner.emplace(key, (uint32_t)pool.size()).first;
          pool.push_back(f);
        }
        ci[k] = it->second;
      }
      r->row_ptr[mtx].alloc((n + 1) * 4);
      r->col[mtx].alloc(nnz * 4);
      r->cidx[mtx].alloc(nnz * 4);
      ARK_CHECK_HIP(hipMemcpy(r->row_ptr[mtx].p, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
      if (nnz) {
        ARK_CHECK_HIP(hipMemcpy(r->col[mtx].p, col[mtx], nnz * 4, hipMemcpyHostToDevice));
        ARK_CHECK_HIP(hipMemcpy(r->cidx[mtx].p, ci.data(), nnz * 4, hipMemcpyHostToDevice));
      }
    }
    r->pool.alloc(pool.size() * sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpy(r->pool.p, pool.data(), pool.size() * sizeof(Fr), hipMemcpyHostToDevice));
    // (g^N - 1)^-1
    Fr g = fr_from_params<Fr>(&P::gen);
    Fr gn = fr_pow2k(g, lg);
    Fr zinv = Fr::inv(Fr::sub(gn, Fr::one()));
    r->zinv.alloc(sizeof(Fr));
    ARK_CHECK_HIP(hipMemcpy(r->zinv.p, &zinv, sizeof(Fr), hipMemcpyHostToDevice));
  } catch (...) {
    delete r;
    throw;
  }
  return r;
}

// Scratch of one witness map: a, b, c and their ping-pong partners (N Fr each).
struct WitnessScratch {
  // ONE allocation of 6 N elements, buf[2v] / buf[2v + 1] = vector v (a, b, c) and its ping-pong partner: the three
  // vectors sit at a uniform stride of 2 N elements, so every NTT pass over them is one launch (ntt_inverse_then_coset)
  struct View {
    void* p = nullptr;
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
  };
  DevBuf all;
  View buf[6];
  DevBuf first_bad;
  void ensure(size_t vec_bytes) {
    all.ensure(6 * vec_bytes);
    for (int i = 0; i < 6; i++) buf[i].p = static_cast<uint8_t*>(all.p) + (size_t)i * vec_bytes;
  }
};

template <class Curve>
static void spmv_run(const R1csDev& r, const void* d_z, WitnessScratch& ws, hipStream_t stream, uint64_t row0 = 0,
                     uint64_t row_stride = 1) {
  using Fr = typename Curve::Fr;
  const uint64_t rows = r.N / row_stride;
  ws.ensure(rows * sizeof(Fr));
  const uint32_t grid = (uint32_t)((rows + 255) / 256);
  ARK_LAUNCH((r1cs_spmv_kernel<Fr>), dim3(grid, 3), dim3(256), 0, stream, r.row_ptr[0].as<uint32_t>(),
             r.col[0].as<uint32_t>(), r.cidx[0].as<uint32_t>(), r.row_ptr[1].as<uint32_t>(), r.col[1].as<uint32_t>(),
             r.cidx[1].as<uint32_t>(), r.row_ptr[2].as<uint32_t>(), r.col[2].as<uint32_t>(), r.cidx[2].as<uint32_t>(),
             r.pool.as<Fr>(), (const Fr*)d_z, r.n, r.ell, rows, ws.buf[0].as<Fr>(), ws.buf[2].as<Fr>(),
             ws.buf[4].as<Fr>(), row0, row_stride);
  ARK_CHECK_LAUNCH();
}

// returns device pointer to h[0..N) (Montgomery)
//
// Six transforms where the reference's LibsnarkReduction runs seven (un-vendored ark-groth16 r1cs_to_qap.rs: ifft a, b, c;
// coset_fft a, b, c; ab - c divided by Z(g) pointwise; coset_ifft): the coset evaluations of c are never needed.  With
// Z = x^N - 1 and a b = h Z + c (deg h <= N - 2, deg c <= N - 1), the evaluations of a b on g H determine
//     rho = a b mod (x^N - g^N) = (g^N - 1) h + c        (the high half of a b is h, the low half c - h),
// and rho is what the inverse coset transform of a' b' returns; c's coefficients are already there after its inverse
// transform.  So h_k = (rho_k - c_k) / (g^N - 1) -- the same field elements as the reference's h for ANY assignment (the
// map is linear: coset_ifft(a'b' - c') = coset_ifft(a'b') - c), hence the same bytes.  Domains of fewer than 8 points keep
// the seven-transform form (one-lane kernels).
// check_rows: also compare a_i b_i with c_i on the rows the SpMV has just written (policy CHECK_SATISFIED); the index of
// the first unsatisfied constraint (or ~0) is left in ws.first_bad for the caller to fetch.
template <class Curve>
static void* witness_map_run(ark355_ctx* ctx, const R1csDev& r, const void* d_z, WitnessScratch& ws,
                             hipStream_t stream, bool check_rows = false) {
  using Fr = typename Curve::Fr;
  spmv_run<Curve>(r, d_z, ws, stream);
  if (check_rows) {
    ws.first_bad.ensure(8);
    ARK_CHECK_HIP(hipMemsetAsync(ws.first_bad.p, 0xFF, 8, stream));
    if (r.n) {
      ARK_LAUNCH((r1cs_check_kernel<Fr>), dim3((uint32_t)((r.n + 255) / 256)), dim3(256), 0, stream, (const Fr*)ws.buf[0].as<Fr>(),
                 (const Fr*)ws.buf[2].as<Fr>(), (const Fr*)ws.buf[4].as<Fr>(), r.n, ws.first_bad.as<unsigned long long>());
      ARK_CHECK_LAUNCH();
    }
  }
  const uint32_t grid = (uint32_t)((r.N + 255) / 256);
  const uint64_t stride = 2 * r.N;
  void* cur[3];
  void* oth[3];
  if (r.log_n < 3) {
    Fr* res0 = (Fr*)ntt_inverse_then_coset<Curve>(ctx, ws.buf[0].p, ws.buf[1].p, r.log_n, stream, 3, stride);
    const bool swapped = res0 != ws.buf[0].as<Fr>();
    for (int v = 0; v < 3; v++) {
      cur[v] = ws.buf[2 * v + (swapped ? 1 : 0)].p;
      oth[v] = ws.buf[2 * v + (swapped ? 0 : 1)].p;
    }
    ARK_LAUNCH((qap_pointwise_kernel<Fr>), dim3(grid), dim3(256), 0, stream, (const Fr*)cur[0], (const Fr*)cur[1],
               (const Fr*)cur[2], r.zinv.as<Fr>(), r.N, (Fr*)oth[0]);
    ARK_CHECK_LAUNCH();
    return ntt_run<Curve>(ctx, oth[0], cur[0], r.log_n, /*inverse=*/true, /*coset=*/true, stream);
  }
  // evaluations on H -> coefficients (a, b, c) -> evaluations on g H (a, b): inverse NTT and coset NTT with the seam fused
  // (ntt_impl.cuh), the vectors in one launch per pass
  void* c_coef = nullptr;
  bool c_scaled = false;
  Fr* res0 = (Fr*)ntt_inverse_then_coset<Curve>(ctx, ws.buf[0].p, ws.buf[1].p, r.log_n, stream, 3, stride, 1, &c_coef, &c_scaled);
  const bool swapped = res0 != ws.buf[0].as<Fr>();
  for (int v = 0; v < 2; v++) {
    cur[v] = ws.buf[2 * v + (swapped ? 1 : 0)].p;
    oth[v] = ws.buf[2 * v + (swapped ? 0 : 1)].p;
  }
  ARK_LAUNCH((qap_mul_kernel<Fr>), dim3(grid), dim3(256), 0, stream, (const Fr*)cur[0], (const Fr*)cur[1], r.N, (Fr*)oth[0]);
  ARK_CHECK_LAUNCH();
  Fr* raw = (Fr*)ntt_passes<Curve>(ctx, oth[0], cur[0], r.log_n, /*inverse=*/true, stream);
  NttTables* t = get_ntt_tables<Curve>(ctx, r.log_n);
  const Fr* gz_hi = nullptr;
  const Fr* zconst = nullptr;
  ntt_quotient_tables<Fr>(t, &gz_hi, &zconst);
  ARK_LAUNCH((qap_quotient_kernel<Fr>), dim3(grid), dim3(256), 0, stream, raw, (const Fr*)c_coef, r.N, (const Fr*)t->gi_lo.as<Fr>(), gz_hi,
             t->lo_bits, zconst + (c_scaled ? 0 : 1));
  ARK_CHECK_LAUNCH();
  return raw;
}

}  // namespace ark355
