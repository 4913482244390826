// The R1CS -> QAP witness map SHARDED over the G GPUs of a sharded proof (BASELINE.json configs[2]).
//
// Rounds 1-3 sharded only the MSM terms; every rank repeated the whole witness map (SpMV + 7 NTTs of N points), which at
// n = 2^22 is ~10 ms of chip-filling work of the 25 ms a rank of eight spends on a proof (profiles/r04_runA_shard_rank_22.json:
// 3.9x on eight GPUs).  Here a rank owns 1/G of every vector through the whole map, and what crosses xGMI is three
// all-to-all exchanges of N/G elements per vector (3 + 2 + 1 vectors: 6 N/G x 32 B per rank and proof, 25 MB at
// N = 2^23, G = 8, spread over all seven links of a rank at once).
//
// N = G M, w = primitive N-th root.  Two layouts of a length-N vector over the ranks:
//   R (residue):  rank g holds x[g + G i2], i2 < M                         (local index i2)
//   B (block):    rank r holds x[k1 + M k2], k1 in [r M/G, (r+1) M/G), k2 < G
// A transform with root u (w or w^-1) moves a vector between them with ONE exchange (four-step FFT, the G-point
// transforms across the ranks being the "rows"):
//   R -> B:  local M-point transform (root u^G), all-to-all, then per k1:  X[k1 + M k2] = sum_g (u^M)^(g k2) u^(g k1) Y_g[k1]
//   B -> R:  per k1: Z_k1[g] = u^(g k1) sum_k2 (u^M)^(k2 g) X[k1 + M k2], all-to-all, local M-point transform (root u^G)
// The witness map chains them so that the rank-crossing G-point transforms of two neighbouring transforms meet in one
// kernel (dwm_seam_kernel: inverse transform's columns, g^k / N, forward transform's columns) and the vectors change
// hands only where the arithmetic needs it:
//   SpMV of this rank's rows (layout R)                                      a, b, c: M elements each
//   local inverse M-NTT x3 | all-to-all (a, b, c) | seam kernel | all-to-all (a, b) | local forward M-NTT x2   -> a', b' on the coset (R)
//   pointwise a' b'                                                           (R, local)
//   local inverse M-NTT | all-to-all | final kernel (columns, g^-k / N, minus c, 1 / Z)      -> h in layout B
// c takes part in the inverse transform only (round 6; the six-transform form of the map, witness_impl.cuh
// witness_map_run: h_k = (rho_k - c_k) / (g^N - 1) with rho the inverse coset transform of a' b'): the seam kernel leaves
// c's coefficients in layout B on the rank that will hold the same coefficients of h, so c never travels back.
// So rank r ends up with h[k1 + M k2] for its k1 range, stored as h_loc[k2 (M/G) + (k1 - r M/G)], and its shard of
// h_query holds the bases of exactly those coefficients in that order (pk_upload, PkDev::h_dist).  Three exchanges (3 + 2 + 1
// vectors), six local transforms of N/G points.  G is a power of two with G^2 | N/8; anything else keeps the replicated map.
//
// The exchange is RCCL (grouped ncclSend / ncclRecv pairs: an all-to-all over the rank's direct xGMI links) on the
// witness-map stream; it is ordered against the prover's other collectives by data dependence (plan check before, the
// all-gather of the partial sums after the H MSM), so one communicator serves them all.  `DwmSim` runs all G ranks of the
// same code on ONE device with device-to-device copies as the exchange: that is how the kernels are checked on a single
// MI355X (ark355_witness_map_dist_sim) -- the real multi-rank path runs under the RCCL emulator in the CPU tier.
#pragma once
#include "witness_impl.cuh"
#include "comm_impl.cuh"

namespace ark355 {

constexpr int DWM_MAX_LOG_G = 4;      // up to 16 ranks (the emulator's and a node's limit)

template <class Fr>
struct DwmRoots {
  Fr fwd[1 << (DWM_MAX_LOG_G - 1)];   // (w^M)^e, e < G/2: twiddles of the G-point forward transform
  Fr inv[1 << (DWM_MAX_LOG_G - 1)];   // (w^-M)^e
};

// in-register radix-2 transform of 2^LG values (decimation in time; x in natural order, result in natural order)
template <class Fr, int LG>
ARK_D void dwm_small_dft(Fr (&x)[1 << LG], const Fr* roots) {
  constexpr int G = 1 << LG;
  if constexpr (LG == 0) {
    (void)roots;
    return;
  } else {
    // bit-reversal permutation
#pragma unroll
    for (int i = 0; i < G; i++) {
      int j = 0;
#pragma unroll
      for (int b = 0; b < LG; b++) j |= ((i >> b) & 1) << (LG - 1 - b);
      if (j > i) {
        const Fr t = x[i];
        x[i] = x[j];
        x[j] = t;
      }
    }
#pragma unroll
    for (int s = 0; s < LG; s++) {
      const int half = 1 << s;
#pragma unroll
      for (int blk = 0; blk < G; blk += 2 * half) {
#pragma unroll
        for (int j = 0; j < half; j++) {
          const int e = j * (G >> (s + 1));            // exponent of the G-th root
          Fr t = x[blk + j + half];
          if (e != 0) t = Fr::mul(t, roots[e]);
          const Fr u = x[blk + j];
          x[blk + j] = Fr::add(u, t);
          x[blk + j + half] = Fr::sub(u, t);
        }
      }
    }
  }
}

// The seam of the distributed inverse -> coset pair, one lane per (vector, k1) of this rank's block:
//   recv[v][g][j]  = Y_g[k1]          (rank g's local inverse transform, k1 = r Mc + j)
//   send[v][g'][j] = w^(g' k1) sum_k2 (w^M)^(k2 g') (g^k / N) X[k], k = k1 + M k2, X[k] = sum_g (w^-M)^(g k2) w^-(g k1) Y_g[k1]
// Vector 2 (c) stops after the inverse transform's columns: c_loc[k2 Mc + j] = N c[k1 + M k2] (unscaled; the final kernel
// folds the 1/N in), the layout h_loc will have.
template <class Fr, int LG>
__global__ void __launch_bounds__(256)
dwm_seam_kernel(const Fr* __restrict__ recv, Fr* __restrict__ send, uint64_t vec_stride, uint32_t mc, uint32_t rank,
                uint64_t m_local, DwmRoots<Fr> roots, const Fr* __restrict__ w_lo, const Fr* __restrict__ w_hi,
                const Fr* __restrict__ wi_lo, const Fr* __restrict__ wi_hi, uint32_t lo_bits, const Fr* __restrict__ seam,
                Fr* __restrict__ c_loc) {
  constexpr int G = 1 << LG;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mc) return;
  const uint64_t k1 = (uint64_t)rank * mc + j;
  const Fr* in = recv + (uint64_t)blockIdx.y * vec_stride;
  Fr* out = send + (uint64_t)blockIdx.y * vec_stride;
  Fr x[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    x[g] = in[(uint64_t)g * mc + j];
    if (g != 0) x[g] = Fr::mul(x[g], pow_lookup<Fr>(wi_lo, wi_hi, lo_bits, (uint64_t)g * k1));
  }
  dwm_small_dft<Fr, LG>(x, roots.inv);
  if (blockIdx.y == 2) {
#pragma unroll
    for (int k2 = 0; k2 < G; k2++) c_loc[(uint64_t)k2 * mc + j] = x[k2];
    return;
  }
#pragma unroll
  for (int k2 = 0; k2 < G; k2++) x[k2] = Fr::mul(x[k2], seam[k1 + m_local * (uint64_t)k2]);
  dwm_small_dft<Fr, LG>(x, roots.fwd);
#pragma unroll
  for (int g = 0; g < G; g++) {
    if (g != 0) x[g] = Fr::mul(x[g], pow_lookup<Fr>(w_lo, w_hi, lo_bits, (uint64_t)g * k1));
    out[(uint64_t)g * mc + j] = x[g];
  }
}

// The columns of the last (inverse coset) transform and the quotient step:
//   h[k1 + M k2] = (g^-k / (N (g^N - 1))) sum_g (w^-M)^(g k2) w^-(g k1) Y_g[k1]  -  c_loc[k2 Mc + j] / (N (g^N - 1)),
// stored at h_loc[k2 Mc + j]  (gz_hi, zc_n: ntt_quotient_tables; c_loc: what dwm_seam_kernel left).
template <class Fr, int LG>
__global__ void __launch_bounds__(256)
dwm_final_kernel(const Fr* __restrict__ recv, Fr* __restrict__ h_loc, uint32_t mc, uint32_t rank, uint64_t m_local,
                 DwmRoots<Fr> roots, const Fr* __restrict__ wi_lo, const Fr* __restrict__ wi_hi,
                 const Fr* __restrict__ gi_lo, const Fr* __restrict__ gz_hi, uint32_t lo_bits, const Fr* __restrict__ c_loc,
                 const Fr* __restrict__ zc_n) {
  constexpr int G = 1 << LG;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mc) return;
  const uint64_t k1 = (uint64_t)rank * mc + j;
  Fr x[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    x[g] = recv[(uint64_t)g * mc + j];
    if (g != 0) x[g] = Fr::mul(x[g], pow_lookup<Fr>(wi_lo, wi_hi, lo_bits, (uint64_t)g * k1));
  }
  dwm_small_dft<Fr, LG>(x, roots.inv);
  const Fr cf = *zc_n;
#pragma unroll
  for (int k2 = 0; k2 < G; k2++)
    h_loc[(uint64_t)k2 * mc + j] = Fr::sub(Fr::mul(x[k2], pow_lookup<Fr>(gi_lo, gz_hi, lo_bits, k1 + m_local * (uint64_t)k2)),
                                           Fr::mul(c_loc[(uint64_t)k2 * mc + j], cf));
}

// h in the distributed layout from a whole h (replicated witness map over a key shard loaded for the distributed one):
// h_loc[k2 Mc + j] = h[(r Mc + j) + M k2]
template <class Fr>
__global__ void __launch_bounds__(256)
dwm_gather_kernel(const Fr* __restrict__ h, Fr* __restrict__ h_loc, uint32_t mc, uint32_t rank, uint64_t m_local, uint32_t world) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)mc * world) return;
  const uint64_t k2 = i / mc, j = i % mc;
  h_loc[i] = h[(uint64_t)rank * mc + j + m_local * k2];
}

// self_one: policy RCCL_SELF -- world size 1 runs the same stages with 1-point "rank-crossing" transforms and the rank as its
// own peer (the exchange code of an 8-GPU proof on the one GPU of a test box)
static inline bool dwm_supported(uint32_t log_n, uint32_t world, bool self_one = false) {
  if (world == 1) return self_one && log_n >= 3;
  if (world < 2 || (world & (world - 1)) != 0) return false;
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  return lg <= (uint32_t)DWM_MAX_LOG_G && log_n >= 2 * lg + 3;       // local transforms of >= 8 points, G | M
}

// Scratch of one rank of the distributed map: the three vectors with their ping-pong partners (ws, M elements each), the
// landing zone / staging area of the exchanges (3 M / 2 M), the local h and the local coefficients of c (M each).
struct DwmScratch {
  WitnessScratch ws;
  DevBuf recv, send, h, c_loc;
};

template <class Fr>
static DwmRoots<Fr> dwm_roots(uint32_t log_n, uint32_t log_g) {
  DwmRoots<Fr> r;
  const Fr one = Fr::one();
  for (auto& v : r.fwd) v = one;
  for (auto& v : r.inv) v = one;
  if (log_g == 0) return r;
  // w^M is the primitive G-th root of the same tower: ntt_root(log_g)
  const Fr wf = ntt_root<Fr>(log_g, false), wi = ntt_root<Fr>(log_g, true);
  (void)log_n;
  Fr f = one, i = one;
  for (uint32_t e = 0; e < (1u << log_g) / 2; e++) {
    r.fwd[e] = f;
    r.inv[e] = i;
    f = Fr::mul(f, wf);
    i = Fr::mul(i, wi);
  }
  return r;
}

// ---- the stages of ONE rank (between them: an exchange) ------------------------------------------------------------------
// stage A: SpMV of the rank's rows + local inverse transforms.  Returns vector 0 of the three results (stride 2 M).
template <class Curve>
static typename Curve::Fr* dwm_stage_a(ark355_ctx* ctx, const R1csDev& r, const void* d_z, uint32_t world, uint32_t rank,
                                       DwmScratch& sc, hipStream_t stream) {
  using Fr = typename Curve::Fr;
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  spmv_run<Curve>(r, d_z, sc.ws, stream, rank, world);
  sc.recv.ensure(3 * M * sizeof(Fr));
  sc.send.ensure(2 * M * sizeof(Fr));
  sc.h.ensure(M * sizeof(Fr));
  sc.c_loc.ensure(M * sizeof(Fr));
  return (Fr*)ntt_passes<Curve>(ctx, sc.ws.buf[0].p, sc.ws.buf[1].p, r.log_n - lg, /*inverse=*/true, stream, 3, 2 * M);
}

// stage B: the seam kernel, recv (3 vectors, stride M) -> send (a, b: 2 vectors, stride M) and c_loc
template <class Curve>
static void dwm_stage_b(ark355_ctx* ctx, const R1csDev& r, uint32_t world, uint32_t rank, DwmScratch& sc, hipStream_t stream) {
  using Fr = typename Curve::Fr;
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  const uint32_t mc = (uint32_t)(M >> lg);
  NttTables* t = get_ntt_tables<Curve>(ctx, r.log_n);
  const DwmRoots<Fr> roots = dwm_roots<Fr>(r.log_n, lg);
  const Fr* seam = ntt_seam_table<Fr>(t);
  const dim3 grid((mc + 255) / 256, 3);
#define ARK_DWM_SEAM(LGV)                                                                                                     \
  ARK_LAUNCH((dwm_seam_kernel<Fr, LGV>), grid, dim3(256), 0, stream, (const Fr*)sc.recv.as<Fr>(), sc.send.as<Fr>(), M, mc, rank, M,  \
             roots, (const Fr*)t->w_lo.as<Fr>(), (const Fr*)t->w_hi.as<Fr>(), (const Fr*)t->wi_lo.as<Fr>(),                      \
             (const Fr*)t->wi_hi.as<Fr>(), t->lo_bits, seam, sc.c_loc.as<Fr>())
  switch (lg) {
    case 0: ARK_DWM_SEAM(0); break;
    case 1: ARK_DWM_SEAM(1); break;
    case 2: ARK_DWM_SEAM(2); break;
    case 3: ARK_DWM_SEAM(3); break;
    case 4: ARK_DWM_SEAM(4); break;
    default: throw HipError{ARK355_EINVAL, "distributed witness map: world size out of range"};
  }
#undef ARK_DWM_SEAM
  ARK_CHECK_LAUNCH();
}

// stage C: local forward transforms of the two vectors that the second exchange landed in ws.buf[0 / 2], their pointwise
// product, its local inverse transform.  Returns the result (M elements).
template <class Curve>
static typename Curve::Fr* dwm_stage_c(ark355_ctx* ctx, const R1csDev& r, uint32_t world, DwmScratch& sc, hipStream_t stream) {
  using Fr = typename Curve::Fr;
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  Fr* res0 = (Fr*)ntt_passes<Curve>(ctx, sc.ws.buf[0].p, sc.ws.buf[1].p, r.log_n - lg, /*inverse=*/false, stream, 2, 2 * M);
  const bool swapped = res0 != sc.ws.buf[0].as<Fr>();
  Fr* cur[2];
  Fr* oth[2];
  for (int v = 0; v < 2; v++) {
    cur[v] = sc.ws.buf[2 * v + (swapped ? 1 : 0)].as<Fr>();
    oth[v] = sc.ws.buf[2 * v + (swapped ? 0 : 1)].as<Fr>();
  }
  const uint32_t grid = (uint32_t)((M + 255) / 256);
  ARK_LAUNCH((qap_mul_kernel<Fr>), dim3(grid), dim3(256), 0, stream, (const Fr*)cur[0], (const Fr*)cur[1], M, oth[0]);
  ARK_CHECK_LAUNCH();
  return (Fr*)ntt_passes<Curve>(ctx, oth[0], cur[0], r.log_n - lg, /*inverse=*/true, stream);
}

// stage D: the last transform's columns, recv (1 vector) -> h_loc
template <class Curve>
static void dwm_stage_d(ark355_ctx* ctx, const R1csDev& r, uint32_t world, uint32_t rank, DwmScratch& sc, hipStream_t stream) {
  using Fr = typename Curve::Fr;
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  const uint32_t mc = (uint32_t)(M >> lg);
  NttTables* t = get_ntt_tables<Curve>(ctx, r.log_n);
  const DwmRoots<Fr> roots = dwm_roots<Fr>(r.log_n, lg);
  const Fr* gz_hi = nullptr;
  const Fr* zconst = nullptr;
  ntt_quotient_tables<Fr>(t, &gz_hi, &zconst);
  const dim3 grid((mc + 255) / 256);
#define ARK_DWM_FINAL(LGV)                                                                                                  \
  ARK_LAUNCH((dwm_final_kernel<Fr, LGV>), grid, dim3(256), 0, stream, (const Fr*)sc.recv.as<Fr>(), sc.h.as<Fr>(), mc, rank, M, roots, \
             (const Fr*)t->wi_lo.as<Fr>(), (const Fr*)t->wi_hi.as<Fr>(), (const Fr*)t->gi_lo.as<Fr>(), gz_hi, t->lo_bits,       \
             (const Fr*)sc.c_loc.as<Fr>(), zconst + 1)
  switch (lg) {
    case 0: ARK_DWM_FINAL(0); break;
    case 1: ARK_DWM_FINAL(1); break;
    case 2: ARK_DWM_FINAL(2); break;
    case 3: ARK_DWM_FINAL(3); break;
    case 4: ARK_DWM_FINAL(4); break;
    default: throw HipError{ARK355_EINVAL, "distributed witness map: world size out of range"};
  }
#undef ARK_DWM_FINAL
  ARK_CHECK_LAUNCH();
}

// ---- the exchange -----------------------------------------------------------------------------------------------------------
// all-to-all of `nvec` vectors: chunk p (mc elements) of send vector v goes to rank p, which stores it as chunk (my rank)
// of its recv vector v.  send / recv vector strides in elements.  loopback (diagnostic policy DWM_LOOPBACK: the per-rank
// COST of a G-rank map measured on one GPU, results meaningless): every chunk is copied locally instead.
// self_rccl (policy RCCL_SELF): the rank's own chunk travels through ncclSend / ncclRecv as well (peer = own rank, inside the
// same group) instead of a device-to-device copy; at world size 1 that is the whole exchange.
template <class Fr>
static void dwm_all_to_all(CommDev* cm, uint32_t world, uint32_t rank, const Fr* send, uint64_t send_stride, Fr* recv,
                           uint64_t recv_stride, uint32_t nvec, uint32_t mc, hipStream_t stream, bool loopback, bool self_rccl = false) {
  const size_t bytes = (size_t)mc * sizeof(Fr);
  for (uint32_t v = 0; v < nvec; v++) {
    const Fr* s = send + (uint64_t)v * send_stride;
    Fr* d = recv + (uint64_t)v * recv_stride;
    if (loopback) {
      ARK_CHECK_HIP(hipMemcpyAsync(d, s, bytes * world, hipMemcpyDeviceToDevice, stream));
      continue;
    }
    if (!self_rccl)
      ARK_CHECK_HIP(hipMemcpyAsync(d + (uint64_t)rank * mc, s + (uint64_t)rank * mc, bytes, hipMemcpyDeviceToDevice, stream));
  }
  if (loopback) return;
  ARK_REQUIRE(cm && cm->world == (int)world && cm->rank == (int)rank, ARK355_EINVAL, "distributed witness map: communicator mismatch");
  if (world == 1 && !self_rccl) return;
  ARK_CHECK_NCCL(ncclGroupStart());
  for (uint32_t p = 0; p < world; p++) {
    if (p == rank && !self_rccl) continue;
    for (uint32_t v = 0; v < nvec; v++) {
      ARK_CHECK_NCCL(ncclSend(send + (uint64_t)v * send_stride + (uint64_t)p * mc, bytes, ncclUint8, (int)p, cm->comm, stream));
      ARK_CHECK_NCCL(ncclRecv(recv + (uint64_t)v * recv_stride + (uint64_t)p * mc, bytes, ncclUint8, (int)p, cm->comm, stream));
    }
  }
  ARK_CHECK_NCCL(ncclGroupEnd());
}

// One rank's distributed witness map.  Returns h_loc (M = N / world elements, layout of the header comment).
template <class Curve>
static void* witness_map_dist_run(ark355_ctx* ctx, const R1csDev& r, const void* d_z, DwmScratch& sc, CommDev* cm, uint32_t world,
                                  uint32_t rank, hipStream_t stream, bool loopback = false, bool self_rccl = false) {
  using Fr = typename Curve::Fr;
  ARK_REQUIRE(dwm_supported(r.log_n, world, self_rccl), ARK355_EINVAL, "distributed witness map: world size / domain not supported");
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  const uint32_t mc = (uint32_t)(M >> lg);
  Fr* y = dwm_stage_a<Curve>(ctx, r, d_z, world, rank, sc, stream);
  dwm_all_to_all<Fr>(cm, world, rank, y, 2 * M, sc.recv.as<Fr>(), M, 3, mc, stream, loopback, self_rccl);
  dwm_stage_b<Curve>(ctx, r, world, rank, sc, stream);
  dwm_all_to_all<Fr>(cm, world, rank, sc.send.as<Fr>(), M, sc.ws.buf[0].as<Fr>(), 2 * M, 2, mc, stream, loopback, self_rccl);
  Fr* q = dwm_stage_c<Curve>(ctx, r, world, sc, stream);
  dwm_all_to_all<Fr>(cm, world, rank, q, M, sc.recv.as<Fr>(), M, 1, mc, stream, loopback, self_rccl);
  dwm_stage_d<Curve>(ctx, r, world, rank, sc, stream);
  return sc.h.p;
}

// All `world` ranks of the distributed map on ONE device, the exchanges as device-to-device copies: h (N elements,
// natural order) into d_h.  Test / diagnostic entry (ark355_witness_map_dist_sim): every kernel of the distributed path on
// a single MI355X against the oracle's witness map.
template <class Curve>
static void witness_map_dist_sim(ark355_ctx* ctx, const R1csDev& r, const void* d_z, uint32_t world, void* d_h, hipStream_t stream) {
  using Fr = typename Curve::Fr;
  ARK_REQUIRE(dwm_supported(r.log_n, world), ARK355_EINVAL, "distributed witness map: world size / domain not supported");
  uint32_t lg = 0;
  while ((1u << lg) < world) lg++;
  const uint64_t M = r.N >> lg;
  const uint32_t mc = (uint32_t)(M >> lg);
  std::vector<std::unique_ptr<DwmScratch>> ranks;
  for (uint32_t g = 0; g < world; g++) ranks.emplace_back(new DwmScratch());
  auto exchange = [&](std::vector<const Fr*>& send, uint64_t send_stride, std::vector<Fr*>& recv, uint64_t recv_stride, uint32_t nvec) {
    for (uint32_t g = 0; g < world; g++)
      for (uint32_t p = 0; p < world; p++)
        for (uint32_t v = 0; v < nvec; v++)
          ARK_CHECK_HIP(hipMemcpyAsync(recv[p] + (uint64_t)v * recv_stride + (uint64_t)g * mc,
                                       send[g] + (uint64_t)v * send_stride + (uint64_t)p * mc, (size_t)mc * sizeof(Fr),
                                       hipMemcpyDeviceToDevice, stream));
  };
  std::vector<const Fr*> snd(world);
  std::vector<Fr*> rcv(world);
  for (uint32_t g = 0; g < world; g++) snd[g] = dwm_stage_a<Curve>(ctx, r, d_z, world, g, *ranks[g], stream);
  for (uint32_t g = 0; g < world; g++) rcv[g] = ranks[g]->recv.template as<Fr>();
  exchange(snd, 2 * M, rcv, M, 3);
  for (uint32_t g = 0; g < world; g++) dwm_stage_b<Curve>(ctx, r, world, g, *ranks[g], stream);
  for (uint32_t g = 0; g < world; g++) {
    snd[g] = ranks[g]->send.template as<Fr>();
    rcv[g] = ranks[g]->ws.buf[0].template as<Fr>();
  }
  exchange(snd, M, rcv, 2 * M, 2);
  for (uint32_t g = 0; g < world; g++) snd[g] = dwm_stage_c<Curve>(ctx, r, world, *ranks[g], stream);
  for (uint32_t g = 0; g < world; g++) rcv[g] = ranks[g]->recv.template as<Fr>();
  exchange(snd, M, rcv, M, 1);
  for (uint32_t g = 0; g < world; g++) {
    dwm_stage_d<Curve>(ctx, r, world, g, *ranks[g], stream);
    // h_loc[k2 Mc + j] -> h[(g Mc + j) + M k2]
    for (uint32_t k2 = 0; k2 < world; k2++)
      ARK_CHECK_HIP(hipMemcpyAsync((Fr*)d_h + (uint64_t)g * mc + M * (uint64_t)k2, ranks[g]->h.template as<Fr>() + (uint64_t)k2 * mc,
                                   (size_t)mc * sizeof(Fr), hipMemcpyDeviceToDevice, stream));
  }
  ARK_CHECK_HIP(hipStreamSynchronize(stream));       // the per-rank scratch is freed on return
}

}  // namespace ark355
