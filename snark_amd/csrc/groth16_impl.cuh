// Groth16 prover on the device: everything behind `SNARK::prove`
// (/root/reference/snark/src/lib.rs:50-54) after synthesis -- the un-vendored ark-groth16
// `create_proof_with_reduction_and_matrices` / `create_proof_with_assignment` (SURVEY.md 3.1, 3.2,
// Appendix A "Proof").
//
// All constant terms are folded into the MSMs so that no 255-bit scalar multiplication of a fixed key
// element is ever done on its own.  With the extended scalar vector
//     zx = [ z_0 .. z_{m-1},  -r*s,  1,  r,  s ]                       (Montgomery Fr, device)
// and base vectors extended at pk-load time by
//     a_ext   = [ a_query,    O, alpha_1, delta_1, O       ]
//     b1_ext  = [ b_g1_query, O, beta_1,  O,       delta_1 ]
//     b2_ext  = [ b_g2_query, O, beta_2,  O,       delta_2 ]
//     l_ext   = [ O x ell,    l_query, delta_1, O, O, O ]              (aligned with zx as well)
// one gets  A = MSM(a_ext, zx), B1 = MSM(b1_ext, zx), B2 = MSM(b2_ext, zx), L' = MSM(l_ext, zx) = l_acc - r s delta_1
// (ONE digit/sort pass shared by the four), H = MSM(h_query, h[0..N-1)), and
//     C = s*A + r*B1 + L' + H.
// The only remaining sequential work is s*A and r*B1 (two 255-bit double-and-add chains) and three affine
// normalisations: O(1) work, independent of the circuit size.  A single GPU lane needs ~25 ms for it
// (measured, round 1), the host ~1 ms, so the MSM results are copied back and the library's own host-compiled field code
// finishes the proof (the device flavour of rounds 1-5, groth16_finalize_kernel, is gone).  Since round 6 the same host
// code also does the last 2c group operations of every bucket reduction: the tails leave c partial sums per MSM
// (tails28_impl.cuh), ~15 KB per proof, and msm_parts_finish is a Horner pass over them (~50 us per MSM).
#pragma once
#include <atomic>
#include <chrono>
#include <future>
#include "common.h"
#include "msm_impl.cuh"
#include "witness_impl.cuh"
#include "comm_impl.cuh"
#include "witness_dist_impl.cuh"
#include "diag_impl.cuh"
#include "sched_tuner.h"
#include <thread>

namespace ark355 {

struct PkDev {
  int curve = 0;
  uint64_t ell = 0, w = 0, m = 0, N = 0;
  // per-window tables T[w*n + i] = 2^(c*w) * P_i of the five (extended) query vectors, built at load
  PrecompTable a_ext, b1_ext, b2_ext, h_query, l_ext;
  // MSM term-range sharding across GPUs: this handle holds terms [lo, lo+cnt) of each
  // (extended) query vector; shard_count == 1 is the whole key
  uint32_t shard_index = 0, shard_count = 1;
  uint64_t z_lo = 0, z_cnt = 0;     // of the m+4 extended a/b/l terms
  uint64_t h_lo = 0, h_cnt = 0;     // of the N-1 h terms
  uint32_t wstride = 1;             // MsmPlan::wstride of the five tables (1 = every window has its table)
  // h_query shard in the layout of the distributed witness map (witness_dist_impl.cuh): local row k2 (M/G) + j holds the
  // base of coefficient (shard_index M/G + j) + M k2, M = N / shard_count; the row of coefficient N - 1 (which the proof
  // does not use) is the point at infinity.  h_cnt = M on every rank then.
  bool h_dist = false;
  uint64_t table_bytes() const {
    return a_ext.table.bytes + b1_ext.table.bytes + b2_ext.table.bytes + h_query.table.bytes + l_ext.table.bytes;
  }
};

static inline void shard_range(uint64_t total, uint32_t idx, uint32_t cnt, uint64_t* lo, uint64_t* n) {
  const uint64_t a = total * idx / cnt, b = total * (idx + 1) / cnt;
  *lo = a;
  *n = b - a;
}

// Wait for the proof's last event WITHOUT burning a core.  On the boxes this was built on, every wait of the HIP runtime
// spins -- hipStreamSynchronize, hipEventSynchronize on a hipEventBlockingSync event, torch.cuda.synchronize alike
// (profiles/r02_host_wait.txt: 1.00 cores per waiting thread, 4.8 cores for a bench with four proofs in flight) -- and
// those are the cores the synthesis threads of an end-to-end prover and the other ranks of a multi-GPU node need.  A proof
// takes tens of milliseconds, so the proving thread polls the event and sleeps 100 us in between.  spin (policy
// WAIT_SPIN=1, or the SCHED_ONE_STREAM_SPIN schedule) hands the wait back to the runtime.
// expect_ms: how long proofs of this shape took to drain on this context lately (0 = unknown).  The thread sleeps through
// most of that in ONE go before it starts polling: a 2^20 proof with four in flight is ~95 ms of waiting, i.e. ~900 polls
// at 100 us, each a runtime call -- a fifth of a host core per proof in flight that the synthesis threads of an end-to-end
// prover and the other ranks of a node need (bench.py host_cpu_threads).
static inline void wait_event_polite(hipEvent_t ev, bool spin, double expect_ms = 0.0, bool* overslept = nullptr) {
  if (spin) {
    ARK_CHECK_HIP(hipEventSynchronize(ev));
    return;
  }
  if (overslept) *overslept = false;
  const bool slept = expect_ms > 0.5;
  if (slept) std::this_thread::sleep_for(std::chrono::microseconds((long long)(expect_ms * 1000.0)));
  for (bool first = true;; first = false) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) {
      if (overslept) *overslept = slept && first;
      return;
    }
    if (e != hipErrorNotReady) ARK_CHECK_HIP(e);
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}

static inline uint64_t prove_shape(int curve, uint64_t N, uint64_t m) { return (N << 8) ^ (m << 1) ^ (uint64_t)curve; }

// small proofs (N < 2^20) poll from the start: with eight 2^18 proofs in flight the long sleep cost 4 % (profiles/r03_epilogue_ab.txt)
static inline bool epi_sleep_ok(uint64_t domain) { return domain >= (1ull << 20); }

struct ProverScratch {
  // Drain times (all work queued -> last event) of the last four proofs of one shape on this context.  The wait sleeps
  // through HALF of the shortest of them before polling (wait_event_polite): with several proofs in flight consecutive
  // drains of one context differ by 2x (46 .. 107 ms in one trace), and a thread that oversleeps starts its next proof late
  // (BN254 2^20 x 4 in flight lost 10 % with a sleep of 3/4 of the LAST drain).  A wait that found the event complete
  // when it woke up halves the history instead of trusting its (sleep-dominated) reading.
  double drain_hist[4] = {0, 0, 0, 0};
  uint32_t drain_n = 0;
  uint64_t drain_shape = 0;
  double drain_hint(bool adapt, uint64_t shape) const {
    // OFF by default (policy WAIT_ADAPT=1 turns it on): with the synchronises gone the proving threads cost 0.07-0.1 host
    // cores with or without the sleep, and BN254 2^20 x 4 in flight ran 17.8 ms per proof with it against 16.6 without
    // (profiles/r03_epilogue_ab.txt) -- a thread that wakes up late starts its next proof late.
    if (!adapt || shape != drain_shape || drain_n < 4) return 0.0;
    double mn = drain_hist[0];
    for (int i = 1; i < 4; i++) mn = drain_hist[i] < mn ? drain_hist[i] : mn;
    return 0.5 * mn;
  }
  void drain_record(uint64_t shape, double ms, bool overslept) {
    if (shape != drain_shape) {
      drain_shape = shape;
      drain_n = 0;
    }
    if (overslept) {
      for (double& d : drain_hist) d *= 0.5;
      return;
    }
    drain_hist[drain_n & 3] = ms;
    drain_n++;
  }
  // one sort per distinct scalar vector (zx for A/B1/B2/L', h for H) and one bucket set per MSM: the five MSMs of
  // a proof are in flight together on separate streams (sized for 288 GB of HBM, not for reuse)
  MsmSort sortZ, sortH;
  MsmBuckets bkA, bkB1, bkB2, bkL, bkH;
  // feeder streams of the five-stream pipeline (witness map, sorts, reductions); the accumulations run on the
  // context's own stream.  Created in ONE block under a process-wide lock (ensure_streams): the runtime hands out its
  // few hardware queues per priority class round-robin in creation order, so three streams created back to back land
  // on three different queues -- whereas four proving threads that create theirs concurrently can interleave so that
  // all three feeder streams of one context share ONE in-order hardware queue (a reduction kernel waiting for its
  // accumulation then blocks the NTT passes queued behind it): the box-to-box spread of the pipelined schedule.
  hipStream_t sW = nullptr, sS = nullptr, sR = nullptr;
  hipStream_t lane = nullptr;       // this context's stream for one-stream proofs (LanePool: pairwise different hardware queues)
  bool lane_asked = false;
  uint64_t warm_shape = 0;          // shape of the last proof on this context (its scratch and tables exist)
  void ensure_streams(bool prio) {
    if (sW) return;
    static std::mutex create_mu;
    std::lock_guard<std::mutex> lk(create_mu);
    int prio_lo = 0, prio_hi = 0;
#if !defined(ARK_EMUL)
    if (prio) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);     // hi is the numerically smaller value
#endif
    for (hipStream_t* st : {&sW, &sS, &sR}) {
#if !defined(ARK_EMUL)
      if (prio && prio_hi != prio_lo) {
        ARK_CHECK_HIP(hipStreamCreateWithPriority(st, hipStreamDefault, prio_hi));
        continue;
      }
#endif
      ARK_CHECK_HIP(hipStreamCreate(st));
    }
  }
  // events of one proof, created once per context (prove_run used to create and destroy 23 of them per proof)
  static constexpr int N_EVENTS = 32;
  hipEvent_t events[N_EVENTS] = {};
  bool have_events = false;
  void ensure_events() {
    if (have_events) return;
    for (auto& e : events) ARK_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventBlockingSync));
    have_events = true;
  }
  ~ProverScratch() {
    release_pinned();
    release_stage();
    for (hipStream_t st : {sW, sS, sR})
      if (st) (void)hipStreamDestroy(st);
    for (auto& e : events)
      if (e) (void)hipEventDestroy(e);
  }
  WitnessScratch ws;
  DwmScratch dwm;   // one rank of the distributed witness map (sharded proofs)
  DevBuf zx;        // extended scalar vector
  DevBuf results;   // XYZZ results: A, B1, L, H (G1) then B2 (G2)
  DevBuf proof;     // raw affine proof A | B | C
  DevBuf rs;        // canonical r, s (2 x Fr)
  // Page-locked landing zone of the proof's last copy (five XYZZ sums, or the three affine points).  A D2H copy into
  // pageable memory is not asynchronous: hipMemcpyAsync then blocks -- spinning -- until everything queued before it on
  // the stream has run, i.e. for the whole proof (profiles/r02_host_wait.txt: one host core per proof in flight).
  void* h_pinned = nullptr;
  size_t h_pinned_bytes = 0;
  // page-locked staging area of the proof's small H2D copies (tail scalars, r, s): from pageable memory (the stack) such a
  // copy is not asynchronous either -- the runtime stages it and waits for the stream
  void* h_stage = nullptr;
  static constexpr size_t STAGE_BYTES = 1024;
  void* stage() {
    if (!h_stage) {
#if defined(ARK_EMUL)
      h_stage = malloc(STAGE_BYTES);
      if (!h_stage) throw HipError{ARK355_ENOMEM, "host allocation failed"};
#else
      if (hipHostMalloc(&h_stage, STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
        h_stage = nullptr;
        throw HipError{ARK355_ENOMEM, "hipHostMalloc failed"};
      }
#endif
    }
    return h_stage;
  }
  void* pinned(size_t bytes) {
    if (bytes > h_pinned_bytes) {
      release_pinned();
#if defined(ARK_EMUL)
      h_pinned = malloc(bytes);
      if (!h_pinned) throw HipError{ARK355_ENOMEM, "host allocation failed"};
#else
      if (hipHostMalloc(&h_pinned, bytes, hipHostMallocDefault) != hipSuccess) {
        h_pinned = nullptr;
        throw HipError{ARK355_ENOMEM, "hipHostMalloc failed"};
      }
#endif
      h_pinned_bytes = bytes;
    }
    return h_pinned;
  }
  void release_pinned() {
    if (!h_pinned) return;
#if defined(ARK_EMUL)
    free(h_pinned);
#else
    (void)hipHostFree(h_pinned);
#endif
    h_pinned = nullptr;
    h_pinned_bytes = 0;
  }
  void release_stage() {
    if (!h_stage) return;
#if defined(ARK_EMUL)
    free(h_stage);
#else
    (void)hipHostFree(h_stage);
#endif
    h_stage = nullptr;
  }
};

// host tail: C = s*A + r*B1 + L' + H ; three affine normalisations (O(1) work, library's own host field code)
template <class Curve>
static void finalize_host(const XYZZ<typename Curve::Fq> g1[4], const XYZZ<typename Curve::Fq2>& g2,
                          const typename Curve::Fr& r_canon, const typename Curve::Fr& s_canon, ark355_proof_raw* out) {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using Fr = typename Curve::Fr;
  // four independent pieces on four host threads: s*A, r*B1, affine(A), affine(B2); then C
  auto f_sa = std::async(std::launch::async, [&] { return xyzz_mul_scalar(g1[0], s_canon.l, Fr::N); });
  auto f_rb = std::async(std::launch::async, [&] { return xyzz_mul_scalar(g1[1], r_canon.l, Fr::N); });
  auto f_pb = std::async(std::launch::async, [&] { return xyzz_to_affine(g2); });
  Affine<Fq> pa = xyzz_to_affine(g1[0]);
  XYZZ<Fq> c = xyzz_add(f_sa.get(), f_rb.get());
  c = xyzz_add(c, g1[2]);
  c = xyzz_add(c, g1[3]);
  Affine<Fq> pc = xyzz_to_affine(c);
  Affine<Fq2> pb = f_pb.get();
  memset(out, 0, sizeof(*out));
  memcpy(out->a, &pa, sizeof(pa));
  memcpy(out->b, &pb, sizeof(pb));
  memcpy(out->c, &pc, sizeof(pc));
}

// What the five MSMs of a proof leave on the device: per MSM msm_parts_count() partial sums, A, B1, L', H (G1) then B2 (G2).
template <class Curve>
struct ProofParts {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  MsmPlan plan[5];          // A, B1, L', H, B2
  int fmt[5] = {0, 0, 0, 0, 0};
  uint32_t count[5] = {0, 0, 0, 0, 0};
  size_t off[5] = {0, 0, 0, 0, 0};          // byte offsets in the results buffer
  size_t bytes = 0;
  void layout() {
    bytes = 0;
    for (int i = 0; i < 5; i++) {
      count[i] = msm_parts_count(plan[i], fmt[i]);
      off[i] = bytes;
      bytes += (size_t)count[i] * (i < 4 ? sizeof(XYZZ<Fq>) : sizeof(XYZZ<Fq2>));
    }
  }
  // host: the five sums from a copy of the results buffer (the four G1 Horner passes beside the G2 one)
  void finish(const uint8_t* land, XYZZ<Fq> g1[4], XYZZ<Fq2>& g2) const {
    auto f_g2 = std::async(std::launch::async, [&] {
      return msm_parts_finish<Fq2>(reinterpret_cast<const XYZZ<Fq2>*>(land + off[4]), plan[4], fmt[4]);
    });
    for (int i = 0; i < 4; i++) g1[i] = msm_parts_finish<Fq>(reinterpret_cast<const XYZZ<Fq>*>(land + off[i]), plan[i], fmt[i]);
    g2 = f_g2.get();
  }
};

// sum `count` shard partials (each: 4 G1 XYZZ + 1 G2 XYZZ, as prove_run hands back) and finish the proof
template <class Curve>
static void combine_partials_host(const uint8_t* partials, uint64_t count, const uint8_t r_canon[32],
                                  const uint8_t s_canon[32], ark355_proof_raw* out) {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  using Fr = typename Curve::Fr;
  const size_t stride = 4 * sizeof(XYZZ<Fq>) + sizeof(XYZZ<Fq2>);
  XYZZ<Fq> g1[4];
  XYZZ<Fq2> g2 = XYZZ<Fq2>::inf();
  for (auto& p : g1) p = XYZZ<Fq>::inf();
  for (uint64_t k = 0; k < count; k++) {
    XYZZ<Fq> t1[4];
    XYZZ<Fq2> t2;
    memcpy(t1, partials + k * stride, sizeof(t1));
    memcpy(&t2, partials + k * stride + sizeof(t1), sizeof(t2));
    for (int i = 0; i < 4; i++) g1[i] = xyzz_add(g1[i], t1[i]);
    g2 = xyzz_add(g2, t2);
  }
  Fr rc, sc;
  memcpy(rc.l, r_canon, sizeof(Fr));
  memcpy(sc.l, s_canon, sizeof(Fr));
  finalize_host<Curve>(g1, g2, rc, sc, out);
}

template <class Curve>
static PkDev* pk_upload(const TunePolicy& pol, const ark355_pk_desc* d, hipStream_t stream, uint32_t shard_index = 0,
                        uint32_t shard_count = 1) {
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  constexpr size_t G1 = sizeof(Affine<Fq>), G2 = sizeof(Affine<Fq2>);
  auto* pk = new PkDev();
  try {
    pk->curve = Curve::ID;
    pk->ell = d->num_instance;
    pk->w = d->num_witness;
    pk->m = pk->ell + pk->w;
    pk->N = d->domain_size;
    ARK_REQUIRE(pk->ell >= 1 && pk->N >= 1 && (pk->N & (pk->N - 1)) == 0, ARK355_EINVAL, "bad pk dimensions");
    ARK_REQUIRE(d->a_query && d->b_g1_query && d->b_g2_query && (d->h_query || pk->N == 1) && (d->l_query || pk->w == 0) &&
                    d->alpha_g1 && d->beta_g1 && d->delta_g1 && d->beta_g2 && d->delta_g2,
                ARK355_EINVAL, "null pointer in pk descriptor");
    const uint64_t m = pk->m;
    using Fr = typename Curve::Fr;
    DevBuf stage;
    auto ext = [&](const uint8_t* query, size_t psz, const uint8_t* t1, const uint8_t* t2, const uint8_t* t3) {
      // [query (m), O, t1, t2, t3]; null tail pointers mean infinity
      stage.ensure((m + 4) * psz);
      ARK_CHECK_HIP(hipMemcpy(stage.p, query, m * psz, hipMemcpyDefault));      // host, or device (pk_load_bytes)
      std::vector<uint8_t> tail(4 * psz, 0);
      if (t1) memcpy(tail.data() + 1 * psz, t1, psz);
      if (t2) memcpy(tail.data() + 2 * psz, t2, psz);
      if (t3) memcpy(tail.data() + 3 * psz, t3, psz);
      ARK_CHECK_HIP(hipMemcpy((uint8_t*)stage.p + m * psz, tail.data(), 4 * psz, hipMemcpyHostToDevice));
    };
    ARK_REQUIRE(shard_count >= 1 && shard_index < shard_count, ARK355_EINVAL, "bad shard index");
    pk->shard_index = shard_index;
    pk->shard_count = shard_count;
    const uint64_t hn = pk->N ? pk->N - 1 : 0;
    shard_range(m + 4, shard_index, shard_count, &pk->z_lo, &pk->z_cnt);
    shard_range(hn, shard_index, shard_count, &pk->h_lo, &pk->h_cnt);
    {
      uint32_t lgN = 0;
      while ((1ull << lgN) < pk->N) lgN++;
      pk->h_dist = pol.shard_dist_wm != 0 && dwm_supported(lgN, shard_count, pol.rccl_self != 0);
      if (pk->h_dist) {
        pk->h_lo = 0;
        pk->h_cnt = pk->N / shard_count;
      }
    }
    // every shard of a key uses the window size of the largest shard (see precomp_build)
    const uint64_t z_plan = shard_count > 1 ? (m + 4 + shard_count - 1) / shard_count : 0;
    const uint64_t h_plan = shard_count > 1 ? (pk->h_dist ? pk->h_cnt : (hn + shard_count - 1) / shard_count) : 0;
    bool packed_rows = false;
    {
      // Window stride of the five tables: 1 (a table per window) whenever that fits next to the scratch of the proving
      // contexts that will work on this key (four of them: two sort areas of 16 B per (term, window), nine N-element NTT
      // buffers / tables each), otherwise the smallest stride that does; ARK355_ENOMEM when not even the bare vectors fit.
      // The shards of one key must reach the SAME stride on every rank (the bucket-level exchange adds bucket arrays of
      // different ranks): they plan from the device size alone and with the LARGEST shard's lengths -- the actual shard
      // lengths differ by one between ranks.
      const uint64_t zn = shard_count > 1 ? z_plan : pk->z_cnt, hn_ = shard_count > 1 ? h_plan : pk->h_cnt;
      const TableNeed need[5] = {{zn, z_plan, false}, {zn, z_plan, false}, {zn, z_plan, true},
                                 {hn_, h_plan, false, pol.msm_c_h}, {zn, z_plan, false}};
      const size_t scratch = 4 * ((size_t)16 * 17 * (zn + hn_) + (size_t)9 * 32 * pk->N);
      std::string why;
      pk->wstride = table_stride_plan<Fq, Fq2, Fr>(pol, need, 5, table_budget_bytes(pol, scratch, shard_count == 1), &why, &packed_rows);
      if (pk->wstride == 0) throw HipError{ARK355_ENOMEM, "proving key: " + why};
    }
    const uint32_t ws = pk->wstride;
    const int pkd = packed_rows ? 1 : 0;           // one row format for the five tables of a key (table_pack_default)
    ext(d->a_query, G1, d->alpha_g1, d->delta_g1, nullptr);
    precomp_build<Fq, Fr>(pol, pk->a_ext, (uint8_t*)stage.p + pk->z_lo * G1, pk->z_cnt, stream, z_plan, ws, 0, pkd);
    ext(d->b_g1_query, G1, d->beta_g1, nullptr, d->delta_g1);
    precomp_build<Fq, Fr>(pol, pk->b1_ext, (uint8_t*)stage.p + pk->z_lo * G1, pk->z_cnt, stream, z_plan, ws, 0, pkd);
    ext(d->b_g2_query, G2, d->beta_g2, nullptr, d->delta_g2);
    precomp_build<Fq2, Fr>(pol, pk->b2_ext, (uint8_t*)stage.p + pk->z_lo * G2, pk->z_cnt, stream, z_plan, ws, 0, pkd);
    stage.ensure((pk->h_cnt ? pk->h_cnt : 1) * G1);
    if (pk->h_dist) {
      const uint64_t M = pk->h_cnt, mc = M / shard_count;
      ARK_CHECK_HIP(hipMemset(stage.p, 0, M * G1));            // (the unused coefficient N - 1: infinity)
      for (uint32_t k2 = 0; k2 < shard_count; k2++) {
        const uint64_t first = (uint64_t)shard_index * mc + M * k2;
        uint64_t cnt = mc;
        if (first + cnt > hn) cnt = hn > first ? hn - first : 0;
        if (cnt) ARK_CHECK_HIP(hipMemcpy((uint8_t*)stage.p + (uint64_t)k2 * mc * G1, d->h_query + first * G1, cnt * G1, hipMemcpyDefault));
      }
    } else if (pk->h_cnt) {
      ARK_CHECK_HIP(hipMemcpy(stage.p, d->h_query + pk->h_lo * G1, pk->h_cnt * G1, hipMemcpyDefault));
    }
    precomp_build<Fq, Fr>(pol, pk->h_query, stage.p, pk->h_cnt, stream, h_plan, ws, pol.msm_c_h, pkd);
    // l_ext aligned with zx: ell leading infinities (instance variables carry no l term), l_query, delta_1 at
    // the -rs slot, three trailing infinities
    stage.ensure((m + 4) * G1);
    ARK_CHECK_HIP(hipMemset(stage.p, 0, (m + 4) * G1));
    if (pk->w) ARK_CHECK_HIP(hipMemcpy((uint8_t*)stage.p + pk->ell * G1, d->l_query, pk->w * G1, hipMemcpyDefault));
    ARK_CHECK_HIP(hipMemcpy((uint8_t*)stage.p + m * G1, d->delta_g1, G1, hipMemcpyHostToDevice));
    precomp_build<Fq, Fr>(pol, pk->l_ext, (uint8_t*)stage.p + pk->z_lo * G1, pk->z_cnt, stream, z_plan, ws, 0, pkd);
    stage.release();
  } catch (...) {
    delete pk;
    throw;
  }
  return pk;
}

// z_src: host or device pointer to m Fr (Montgomery); z_on_device selects the copy kind.
//
// Stream plan (MI355X: 256 CUs; the accumulation kernels fill the chip, everything else is small or
// latency-bound and is tucked underneath them):
//   sM (ctx stream)  H2D of z and the tail scalars                                  -> evZ
//   sW               witness map: SpMV, 7 NTTs, pointwise                            -> evH
//   sS               digits/scan/scatter of zx, then (after evH) of h               -> evSort[0], evSort[2]
//   sA               bucket accumulation: A, B1, B2, L' (share sort 0), H            -> evAcc[0..4]
//   sR               merge + bucket reduction + combine per MSM as evAcc[i] fires; D2H of the five XYZZ results
template <class Curve>
static void prove_run(ark355_ctx* ctx, ProverScratch& sc, const PkDev& pk, const R1csDev& r1, const void* z_src,
                      bool z_on_device, const uint8_t r_canon[32], const uint8_t s_canon[32], ark355_proof_raw* out,
                      uint8_t* partials_out = nullptr, CommDev* cm = nullptr, int shard_mode = 0) {
  using Fr = typename Curve::Fr;
  using Fq = typename Curve::Fq;
  using Fq2 = typename Curve::Fq2;
  ARK_REQUIRE(pk.curve == Curve::ID && r1.curve == Curve::ID, ARK355_EINVAL, "curve mismatch");
  ARK_REQUIRE(pk.ell == r1.ell && pk.w == r1.w && pk.N == r1.N, ARK355_EINVAL,
              "proving key and R1CS dimensions differ");
  const TunePolicy& pol = ctx->policy;          // (the context's mutex is held: the policy cannot change under this proof)
  const auto t_enter = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  const bool trace_host = pol.trace_host != 0;
  double t_launched = 0, t_synced = 0, t_tail = 0;
  hipStream_t sM = ctx->stream;
  // Schedule.  A proof ALONE on the device runs best as the five-stream pipeline below (the latency-bound tails of one MSM
  // hide under the next MSM's accumulation: 26.9 against 31.7 ms for a single 2^20 proof).  With OTHER proofs in flight the
  // same proof runs best on ONE stream: the gaps are filled by the other proofs' kernels anyway, and the pipeline's
  // cross-stream waits cost more than they buy -- HIP maps its streams onto a few in-order hardware queues, so a kernel that
  // waits for an event of its own proof blocks the ready kernels of other proofs queued behind it (the 2^18 x 8 timeline
  // shows a kernel running 99.9 % of the time but an accumulation only 77 %).  Measured, same box
  // (profiles/r03_one_stream_ab.txt): BLS12-381 2^20 23.81 -> 23.42 ms per proof at four in flight on 0.14 instead of 0.93
  // host cores, BN254 2^20 16.1 -> 15.0 ms, 2^18 8.28 -> 7.38 ms at six in flight.  Those are the STATIC defaults; with
  // policy SCHED = AUTO (the default) they are only the starting point of a measured choice per class (SchedTuner above),
  // and SCHED = 0..3 forces one schedule for every proof of the context.
  struct InFlight {
    std::atomic<int>& c;
    int mine;
    explicit InFlight(std::atomic<int>& counter) : c(counter), mine(++counter) {}
    ~InFlight() { --c; }
  };
  static std::atomic<int> g_inflight[64];
  InFlight inflight(g_inflight[(unsigned)ctx->device & 63u]);
  const bool concurrent = inflight.mine >= 2;
  const uint64_t shape = prove_shape(Curve::ID, pk.N, pk.m);
  // measured regimes of the pipeline's epilogue (profiles/r03_epilogue_ab.txt): BLS12-381 N = 2^21 / 2^22 / 2^23 win without
  // the stream synchronises, small proofs and BN254 with them
  // (only a sharded proof still starts from the pipeline: its witness-map exchanges and the copy of the assignment overlap the
  // accumulations of the rank.  Since round 4 a proof alone runs best on one stream as well -- batched G1 tails, G2 tails on
  // the side stream: 26.0-26.6 against 27.0-27.4 ms at 2^20 on every box measured -- so one stream is the default everywhere
  // and the measured choice only leaves it for a 5 % win.)
  const int static_alone = (pk.N >= (1ull << 20) && sizeof(Fq) >= 48) ? SCHED_PIPELINE : SCHED_PIPELINE_SYNC;
  const int static_sched = cm ? static_alone : SCHED_ONE_STREAM;
  int sched = pol.sched;
  bool exploring = false;
  const uint64_t tune_key = SchedTuner::key(shape, concurrent);
  if (cm) {
    // a sharded proof is a collective: every rank must queue the same operations in the same order on the same kind of
    // stream, so the choice cannot depend on one rank's measurements
    if (sched < 0 || sched >= SCHED_COUNT) sched = static_sched;
  } else if (sched < 0 || sched >= SCHED_COUNT) {
    const bool warm = sc.warm_shape == shape;
    sched = warm ? SchedTuner::of(ctx->device).pick(tune_key, concurrent, pol.sched_explore, static_sched, &exploring)
                 : static_sched;
  }
  sc.warm_shape = shape;
  ctx->last_sched = sched;
  // an exploring proof that leaves by exception must hand its sample slot back, or the class never latches
  struct ExploreGuard {
    int device;
    uint64_t key;
    int sched;
    bool armed;
    ~ExploreGuard() {
      if (armed) SchedTuner::of(device).unstart(key, sched);
    }
  } explore_guard{ctx->device, tune_key, sched, exploring};
  const bool one_stream = sched == SCHED_ONE_STREAM || sched == SCHED_ONE_STREAM_SPIN;
  const bool spin = pol.wait_spin != 0 || sched == SCHED_ONE_STREAM_SPIN;
  // The short kernels that feed the accumulations (witness map, sorts) and the latency-bound reductions outrank the
  // long accumulation launches: when workgroup slots free up, a waiting NTT pass or sort of ANOTHER proof in flight
  // is dispatched before the next round of accumulation workgroups, which keeps an accumulation queued at all times.
  // Policy STREAM_PRIO=0 turns it off (A/B).
  if (!one_stream) sc.ensure_streams(pol.stream_prio != 0);
  if (one_stream) {
    // one-stream proofs run on a "lane": a stream of the device's pool, probed at start-up to sit on its own hardware queue
    if (!sc.lane_asked) {
      sc.lane = LanePool::of(ctx->device).acquire();
      sc.lane_asked = true;
    }
    if (sc.lane) sM = sc.lane;
  }
  // A one-stream proof that is ALONE on the device (round 5, policy SIDE_WM): nothing but the H MSM needs h, so the witness map
  // and the sort of h go to the context's witness-map stream and run BESIDE the sort of z and the accumulations of B2, A, B1 and
  // L' instead of in front of them (2.4 ms of a 25 ms proof); the proof's stream meets them again at the H accumulation.  With
  // other proofs in flight everything stays on the one stream.
  const bool side_wm = one_stream && !concurrent && !cm && !partials_out && pol.side_wm != 0;
  if (side_wm) sc.ensure_streams(pol.stream_prio != 0);
  hipStream_t sW = one_stream ? (side_wm ? sc.sW : sM) : sc.sW, sS = one_stream ? sM : sc.sS, sA = sM, sR = one_stream ? sM : sc.sR;
  hipStream_t sSH = side_wm ? sc.sW : sS;            // the stream the sort of h runs on
  const uint64_t m = pk.m, ell = pk.ell;
  // policy CHECK_SATISFIED: whole proofs of a whole key only (a rank of a sharded proof sees 1/G of the rows, and the ranks of a
  // collective must not disagree on its outcome)
  const bool check_sat = pol.check_satisfied != 0 && !cm && !partials_out && pk.shard_count <= 1;
  if (cm && cm->world > 1) {
    // Every rank must have planned the same window size and table stride for its shard: the bucket-level exchange adds
    // bucket arrays of different ranks element by element.  The planners are deterministic functions of the key's
    // dimensions and the device size, but a policy override on one rank (MSM_C, TABLE_STRIDE, HBM_BUDGET_MB) would break
    // that silently -- so the ranks compare notes over the communicator, on EVERY sharded proof: a per-process "already
    // checked" flag could differ between ranks (one rank reloaded its shard, or failed before it set the flag), and then
    // only some ranks would enter this all-gather while the others went on to the ring steps -- mismatched collectives on
    // one communicator.  16 bytes per rank and one stream synchronise: ~50 us against a proof of tens of milliseconds.
    // (the layout of the h_query shard rides along: a rank that loaded its shard with policy SHARD_DIST_WM = 0 would skip the
    // witness map's all-to-alls while the others wait in them)
    const uint32_t mine[4] = {pk.a_ext.plan.c, pk.a_ext.plan.wstride, pk.h_query.plan.c,
                              pk.h_query.plan.wstride | (pk.h_dist ? 0x10000u : 0u)};
    cm->gather.ensure(sizeof(mine) * (size_t)(cm->world + 1));
    uint8_t* d_mine = cm->gather.as<uint8_t>() + sizeof(mine) * (size_t)cm->world;
    uint8_t* stg = static_cast<uint8_t*>(sc.stage());
    memcpy(stg + ProverScratch::STAGE_BYTES - sizeof(mine), mine, sizeof(mine));
    ARK_CHECK_HIP(hipMemcpyAsync(d_mine, stg + ProverScratch::STAGE_BYTES - sizeof(mine), sizeof(mine), hipMemcpyHostToDevice, sM));
    ARK_CHECK_NCCL(ncclAllGather(d_mine, cm->gather.p, sizeof(mine), ncclUint8, cm->comm, sM));
    uint32_t* all = reinterpret_cast<uint32_t*>(sc.pinned(sizeof(mine) * (size_t)cm->world));
    ARK_CHECK_HIP(hipMemcpyAsync(all, cm->gather.p, sizeof(mine) * (size_t)cm->world, hipMemcpyDeviceToHost, sM));
    ARK_CHECK_HIP(hipStreamSynchronize(sM));
    for (int g = 0; g < cm->world; g++)
      for (int k = 0; k < 4; k++)
        ARK_REQUIRE(all[4 * (size_t)g + k] == mine[k], ARK355_EINVAL,
                    "key shards of different ranks were planned with different window sizes / table strides / witness-map layouts");
  }
  enum { E_START, E_Z, E_ZS, E_H, E_SORT0, E_SORT1, E_SORT2, E_G2T, E_FILL, E_HT, E_ACC_DONE0, E_END = E_ACC_DONE0 + 5, E_COUNT };
  static_assert(E_COUNT + 10 <= ProverScratch::N_EVENTS, "event pool too small");
  sc.ensure_events();
  hipEvent_t* ev = sc.events;
  hipEvent_t* acc0 = sc.events + E_COUNT;
  hipEvent_t* acc1 = sc.events + E_COUNT + 5;
  bool h_tails_aside = false;       // the tails of the H MSM were queued on the sort stream (E_HT is its last event then)
  {
    // r, s -> Montgomery on the host (the library's own field code); tail = [-rs, 1, r, s]
    Fr rc, scn;
    memcpy(rc.l, r_canon, sizeof(Fr));
    memcpy(scn.l, s_canon, sizeof(Fr));
    Fr rm = Fr::to_mont(rc), sm = Fr::to_mont(scn);
    Fr tail[4] = {Fr::neg(Fr::mul(rm, sm)), Fr::one(), rm, sm};
    Fr rs_c[2] = {rc, scn};
    sc.zx.ensure((m + 4) * sizeof(Fr));
    sc.rs.ensure(2 * sizeof(Fr));
    sc.proof.ensure(2 * sizeof(Affine<Fq>) + sizeof(Affine<Fq2>));

    ARK_CHECK_HIP(hipEventRecord(ev[E_START], sM));
    const auto zkind = z_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // (the previous proof of this context has drained: its staging area is free)
    static_assert(sizeof(tail) + sizeof(rs_c) <= ProverScratch::STAGE_BYTES, "staging area too small");
    uint8_t* stg = static_cast<uint8_t*>(sc.stage());
    memcpy(stg, tail, sizeof(tail));
    memcpy(stg + sizeof(tail), rs_c, sizeof(rs_c));
    // A key shard sorts only ITS slice of zx: that slice (and the four tail scalars) goes first and releases the sort and the
    // accumulations (E_ZS); the rest of the assignment, which only the witness map needs, follows (E_Z).  From host memory a
    // 2^22-constraint assignment is 134 MB = 2.5 ms of PCIe time in front of a rank's 17 ms (section 5 of DESIGN.md).
    const uint64_t s_lo = pk.z_lo < m ? pk.z_lo : m;
    const uint64_t s_hi = (pk.z_lo + pk.z_cnt) < m ? (pk.z_lo + pk.z_cnt) : m;
    const bool split = pk.shard_count > 1 && (s_lo > 0 || s_hi < m);
    auto copy_z = [&](uint64_t lo, uint64_t hi, hipStream_t st) {
      if (hi > lo)
        ARK_CHECK_HIP(hipMemcpyAsync((uint8_t*)sc.zx.p + lo * sizeof(Fr), (const uint8_t*)z_src + lo * sizeof(Fr), (hi - lo) * sizeof(Fr), zkind, st));
    };
    if (split) copy_z(s_lo, s_hi, sM);
    else copy_z(0, m, sM);
    ARK_CHECK_HIP(hipMemcpyAsync((uint8_t*)sc.zx.p + m * sizeof(Fr), stg, sizeof(tail), hipMemcpyHostToDevice, sM));
    ARK_CHECK_HIP(hipMemcpyAsync(sc.rs.p, stg + sizeof(tail), sizeof(rs_c), hipMemcpyHostToDevice, sM));
    ARK_CHECK_HIP(hipEventRecord(ev[E_ZS], sM));
    if (split) {
      // the rest travels on the WITNESS-MAP stream (its only consumer): the context's stream carries the accumulations, which
      // must not queue up behind 2 ms of copies they do not need
      ARK_CHECK_HIP(hipStreamWaitEvent(sW, ev[E_ZS], 0));
      copy_z(0, s_lo, sW);
      copy_z(s_hi, m, sW);
      ARK_CHECK_HIP(hipEventRecord(ev[E_Z], sW));
    } else {
      ARK_CHECK_HIP(hipEventRecord(ev[E_Z], sM));
    }

    // witness map -> h
    ARK_CHECK_HIP(hipStreamWaitEvent(sW, ev[E_Z], 0));
    void* d_h;
    if (pk.h_dist && ((cm && shard_mode != ARK355_SHARD_BUCKET_RING) || (!cm && pol.dwm_loopback))) {
      // the witness map sharded like the MSMs: this rank's 1/G of every vector, three all-to-all exchanges on the
      // witness-map stream (witness_dist_impl.cuh).  They are ordered against the prover's other collectives by data
      // dependence (the plan check above has completed; the all-gather of the partial sums waits for the H MSM), so the one
      // communicator serves them all.  The bucket-ring exchange interleaves its own send / receive steps with the MSMs
      // and therefore keeps the replicated map.
      d_h = witness_map_dist_run<Curve>(ctx, r1, sc.zx.p, sc.dwm, cm, pk.shard_count, pk.shard_index, sW, /*loopback=*/!cm,
                                        // a whole key in the distributed layout exists only when it was loaded under policy
                                        // RCCL_SELF: the KEY says that the rank is its own peer, whatever the policy reads now
                                        /*self_rccl=*/cm && pk.shard_count == 1);
      if (trace_host)
        fprintf(stderr, "[ark355] witness map distributed over %u ranks (rank %u: N / G = %llu elements per vector)%s\n", pk.shard_count,
                pk.shard_index, (unsigned long long)(pk.N / pk.shard_count), cm ? "" : " -- LOOPBACK exchange, timing only");
    } else {
      d_h = witness_map_run<Curve>(ctx, r1, sc.zx.p, sc.ws, sW, check_sat);
      if (pk.h_dist) {
        // a key shard in the distributed layout under the replicated map: pick this rank's coefficients out of h
        const uint64_t M = pk.h_cnt;
        const uint32_t mc = (uint32_t)(M / pk.shard_count);
        sc.dwm.h.ensure(M * sizeof(Fr));
        ARK_LAUNCH((dwm_gather_kernel<Fr>), dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, sW, (const Fr*)d_h, sc.dwm.h.as<Fr>(), mc,
                   pk.shard_index, M, pk.shard_count);
        ARK_CHECK_LAUNCH();
        d_h = sc.dwm.h.p;
      }
    }
    ARK_CHECK_HIP(hipEventRecord(ev[E_H], sW));

    // sorts.  Everything the two sorts and the five bucket sets need cleared is planned first and cleared by ONE fill
    // dispatch (FillBatch, msm_impl.cuh) at the head of the sort stream: the scratch belongs to this context and the
    // previous proof on it has drained.
    {
      FillBatch fb(sS);
      msm_sort_plan<Fr>(ctx, sc.sortZ, pk.z_cnt, sS, &pk.a_ext, &fb);
      // bucket sets are sized and cleared here as well (msm_prepare_phase)
      msm_prepare_phase<Fq2>(pol, sc.sortZ, sc.bkB2, sS, pk.b2_ext.fmt(), &fb);
      msm_prepare_phase<Fq>(pol, sc.sortZ, sc.bkA, sS, pk.a_ext.fmt(), &fb);
      msm_prepare_phase<Fq>(pol, sc.sortZ, sc.bkB1, sS, pk.b1_ext.fmt(), &fb);
      msm_prepare_phase<Fq>(pol, sc.sortZ, sc.bkL, sS, pk.l_ext.fmt(), &fb);
      msm_sort_plan<Fr>(ctx, sc.sortH, pk.h_cnt, sS, &pk.h_query, &fb);
      msm_prepare_phase<Fq>(pol, sc.sortH, sc.bkH, sS, pk.h_query.fmt(), &fb);
      fb.flush();
    }
    // what the five MSMs leave for the host (c partial sums per bucket set and MSM: tails28_impl.cuh), A, B1, L', H, then B2
    ProofParts<Curve> parts;
    {
      const MsmSort* so[5] = {&sc.sortZ, &sc.sortZ, &sc.sortZ, &sc.sortH, &sc.sortZ};
      const PrecompTable* tb[5] = {&pk.a_ext, &pk.b1_ext, &pk.l_ext, &pk.h_query, &pk.b2_ext};
      for (int i = 0; i < 5; i++) {
        parts.plan[i] = so[i]->plan;
        parts.fmt[i] = tb[i]->fmt();
      }
      parts.layout();
    }
    sc.results.ensure(parts.bytes);
    uint8_t* const res_base = sc.results.as<uint8_t>();
    XYZZ<Fq>* const g1res[4] = {reinterpret_cast<XYZZ<Fq>*>(res_base + parts.off[0]), reinterpret_cast<XYZZ<Fq>*>(res_base + parts.off[1]),
                                reinterpret_cast<XYZZ<Fq>*>(res_base + parts.off[2]), reinterpret_cast<XYZZ<Fq>*>(res_base + parts.off[3])};
    XYZZ<Fq2>* const g2res = reinterpret_cast<XYZZ<Fq2>*>(res_base + parts.off[4]);
    if (sSH != sS) ARK_CHECK_HIP(hipEventRecord(ev[E_FILL], sS));      // (the sort of h on another stream must see its counters cleared)
    ARK_CHECK_HIP(hipStreamWaitEvent(sS, ev[E_ZS], 0));
    msm_sort_run<Fr>(ctx, sc.sortZ, (const uint8_t*)sc.zx.p + pk.z_lo * sizeof(Fr), pk.z_cnt, 1, sS, &pk.a_ext);
    ARK_CHECK_HIP(hipEventRecord(ev[E_SORT0], sS));
    ARK_CHECK_HIP(hipEventRecord(ev[E_SORT1], sS));     // (L' shares the sort of zx)
    if (sSH != sS) ARK_CHECK_HIP(hipStreamWaitEvent(sSH, ev[E_FILL], 0));
    ARK_CHECK_HIP(hipStreamWaitEvent(sSH, ev[E_H], 0));
    msm_sort_run<Fr>(ctx, sc.sortH, (const uint8_t*)d_h + pk.h_lo * sizeof(Fr), pk.h_cnt, 1, sSH, &pk.h_query);
    ARK_CHECK_HIP(hipEventRecord(ev[E_SORT2], sSH));

    // accumulations (A, B1, B2 share the sort of zx) and, per MSM, its reduction on sR
    struct Job {
      int sort_ev;
      const MsmSort* sort;
      MsmBuckets* bk;
      bool g2;
      const PrecompTable* tab;
      int res;
    } jobs[5] = {
        // G2 first: its bucket reduction is the longest latency-bound tail (~4-6 ms on a few workgroups) and
        // hides under the four G1 accumulations that follow
        {E_SORT0, &sc.sortZ, &sc.bkB2, true, &pk.b2_ext, 0},
        {E_SORT0, &sc.sortZ, &sc.bkA, false, &pk.a_ext, 0},
        {E_SORT0, &sc.sortZ, &sc.bkB1, false, &pk.b1_ext, 1},
        {E_SORT0, &sc.sortZ, &sc.bkL, false, &pk.l_ext, 2},
        {E_SORT2, &sc.sortH, &sc.bkH, false, &pk.h_query, 3},
    };
    uint64_t pts = 0;
    // one wave per workgroup for a proof alone on one stream, 256 lanes otherwise (msm_accumulate_phase; policy ACC_THREADS)
    struct HintGuard {
      ark355_ctx* c;
      ~HintGuard() { c->acc_threads_hint = 0; }
    } hint_guard{ctx};
    ctx->acc_threads_hint = (one_stream && !concurrent) ? 64 : 256;
    // One-stream proofs: the five accumulations first, then the G2 tails and the tails of the four G1 MSMs as ONE launch
    // per step (msm_reduce_phase_batch) -- 8 tail dispatches instead of 20, and the four latency-bound G1 chains side by
    // side instead of one after the other.  (The pipeline hides each MSM's tails under the next accumulation instead.)
    const bool batch_tails = one_stream && !cm && pol.batch_tails != 0;
    for (int j = 0; j < 5; j++) {
      const Job& jb = jobs[j];
      ARK_CHECK_HIP(hipStreamWaitEvent(sA, ev[jb.sort_ev], 0));
      const MsmSort* red_sort = jb.sort;          // what the reduction reads offsets / counts from
      if (jb.g2) {
        msm_accumulate_phase<Fq2>(ctx, *jb.sort, *jb.bk, jb.tab->table.template as<Affine<Fq2>>(), sA, acc0[j], acc1[j],
                                  jb.tab->fmt());
      } else {
        msm_accumulate_phase<Fq>(ctx, *jb.sort, *jb.bk, jb.tab->table.template as<Affine<Fq>>(), sA, acc0[j], acc1[j],
                                 jb.tab->fmt());
      }
      ARK_CHECK_HIP(hipEventRecord(ev[E_ACC_DONE0 + j], sA));
      pts += (uint64_t)jb.sort->plan.windows * jb.sort->plan.n;
      if (batch_tails) continue;
      // Multi-stream schedules: the tails of the LAST MSM (H) go to the sort stream, which has been idle since the sort of h,
      // instead of queueing behind the tails of L' on the reduction stream.  The tails run starved under the accumulations
      // (2.4 ms per MSM instead of 1.2), so with short accumulations -- a rank of a sharded proof: 2 ms each -- the reduction
      // stream falls behind and the tails of H, the end of the proof's critical path, started 0.8 ms after its accumulation had
      // ended (kernel trace of run M).  Not with the bucket ring: its grouped sends must be queued in one order on every rank.
      const bool ring = cm && shard_mode == ARK355_SHARD_BUCKET_RING;
      const bool side_tail = j == 4 && !one_stream && !ring && sS != sR && pol.side_h_tails != 0;
      hipStream_t sT = side_tail ? sS : sR;
      ARK_CHECK_HIP(hipStreamWaitEvent(sT, ev[E_ACC_DONE0 + j], 0));
      if (ring) {
        // bucket-level exchange: the ranks run their MSMs in the same order, so the ring steps pair up
        const int bfmt = jb.bk->fmt;
        if (jb.g2)
          msm_reduce_phase<Fq2>(ctx, *red_sort, *jb.bk, g2res, 0, sT, [&](void* bk, uint32_t nb, hipStream_t st) {
            ring_reduce_scatter_buckets<Fq2>(*cm, bk, nb, bfmt, st, pol.rccl_self != 0);
          });
        else
          msm_reduce_phase<Fq>(ctx, *red_sort, *jb.bk, g1res[jb.res], 0, sT, [&](void* bk, uint32_t nb, hipStream_t st) {
            ring_reduce_scatter_buckets<Fq>(*cm, bk, nb, bfmt, st, pol.rccl_self != 0);
          });
      } else if (jb.g2) {
        msm_reduce_phase<Fq2>(ctx, *red_sort, *jb.bk, g2res, 0, sT);
      } else {
        msm_reduce_phase<Fq>(ctx, *red_sort, *jb.bk, g1res[jb.res], 0, sT);
      }
      if (side_tail) {
        h_tails_aside = true;
        ARK_CHECK_HIP(hipEventRecord(ev[E_HT], sT));
        ARK_CHECK_HIP(hipStreamWaitEvent(sR, ev[E_HT], 0));      // everything still drains into sR
      }
    }
    if (batch_tails) {
      // A one-stream proof that is ALONE on the device has nobody to fill the gaps of its latency-bound tails: its G2 tails
      // (the longest chain: merge + reduction + combination on lane pairs, 1.9 ms at 2^20) go to the context's reduction
      // stream and run underneath the four G1 accumulations; the proof's stream picks the result up before its last copy.
      // With other proofs in flight everything stays on the one stream (a second stream per proof is exactly what the
      // one-stream schedule exists to avoid).  Policy SIDE_G2_TAILS=0: always on the proof's stream.
      const bool side_g2 = !concurrent && pol.side_g2_tails != 0;
      if (side_g2) {
        sc.ensure_streams(pol.stream_prio != 0);
        ARK_CHECK_HIP(hipStreamWaitEvent(sc.sR, ev[E_ACC_DONE0 + 0], 0));
        msm_reduce_phase<Fq2>(ctx, sc.sortZ, sc.bkB2, g2res, 0, sc.sR);
        ARK_CHECK_HIP(hipEventRecord(ev[E_G2T], sc.sR));
      } else {
        msm_reduce_phase<Fq2>(ctx, sc.sortZ, sc.bkB2, g2res, 0, sR);
      }
      const MsmSort* sorts[4] = {&sc.sortZ, &sc.sortZ, &sc.sortZ, &sc.sortH};
      MsmBuckets* bks[4] = {&sc.bkA, &sc.bkB1, &sc.bkL, &sc.bkH};
      XYZZ<Fq>* outs[4] = {g1res[0], g1res[1], g1res[2], g1res[3]};
      // Policy SIDE_G1_TAILS (lone proofs): the tails of A, B1, L' as a batch of three on the second stream, behind the G2 tails and
      // under the H accumulation; the proof's stream ends with H's own tails alone.
      const bool side_g1 = side_g2 && pol.side_g1_tails != 0;
      bool batched;
      if (side_g1) {
        ARK_CHECK_HIP(hipStreamWaitEvent(sc.sR, ev[E_ACC_DONE0 + 3], 0));
        batched = msm_reduce_phase_batch<Fq>(ctx, 3, sorts, bks, outs, sc.sR);
        if (!batched)
          for (int i = 0; i < 3; i++) msm_reduce_phase<Fq>(ctx, *sorts[i], *bks[i], outs[i], 0, sc.sR);
        ARK_CHECK_HIP(hipEventRecord(ev[E_G2T], sc.sR));        // (re-recorded: now behind the G2 tails AND the batch of three)
        msm_reduce_phase<Fq>(ctx, *sorts[3], *bks[3], outs[3], 0, sR);
      } else {
        batched = msm_reduce_phase_batch<Fq>(ctx, 4, sorts, bks, outs, sR);
        if (!batched)
          for (int i = 0; i < 4; i++) msm_reduce_phase<Fq>(ctx, *sorts[i], *bks[i], outs[i], 0, sR);
      }
      if (trace_host)
        fprintf(stderr, "[ark355] G1 tails: %s%s\n", batched ? "one launch per step" : "per MSM", side_g1 ? " (A, B1, L' aside, H at the end)" : " for the four MSMs");
      if (side_g2) ARK_CHECK_HIP(hipStreamWaitEvent(sR, ev[E_G2T], 0));
    }

    if (out) memset(out, 0, sizeof(*out));
    // The proof's last copy: the partial sums of the five MSMs (~15 KB at c = 17) into page-locked memory; the host then runs the
    // Horner pass of every bucket reduction and the O(1) tail (ProofParts::finish, finalize_host).
    const size_t psz = 4 * sizeof(XYZZ<Fq>) + sizeof(XYZZ<Fq2>);
    uint8_t* land = static_cast<uint8_t*>(sc.pinned(parts.bytes + 8 + (cm ? psz * (size_t)(cm->world + 1) : 0)));
    ARK_CHECK_HIP(hipMemcpyAsync(land, res_base, parts.bytes, hipMemcpyDeviceToHost, sR));
    uint8_t* const land_sat = land + parts.bytes + (cm ? psz * (size_t)(cm->world + 1) : 0);
    // (sR has everything of the witness-map stream behind it: the H accumulation waited for the sort of h, which waited for h)
    if (check_sat) ARK_CHECK_HIP(hipMemcpyAsync(land_sat, sc.ws.first_bad.p, 8, hipMemcpyDeviceToHost, sR));
    ARK_CHECK_HIP(hipEventRecord(ev[E_END], sR));
    t_launched = since(t_enter);
    bool overslept = false;
    const bool plain = !cm && !partials_out;
    wait_event_polite(ev[E_END], spin, (plain && epi_sleep_ok(pk.N)) ? sc.drain_hint(pol.wait_adapt != 0, shape) : 0.0, &overslept);
    t_synced = since(t_enter);
    if (plain) sc.drain_record(shape, t_synced - t_launched, overslept);
    XYZZ<Fq> h1[4];
    XYZZ<Fq2> h2;
    parts.finish(land, h1, h2);
    if (cm) {
      // sharded prove: this rank's five XYZZ sums (A, B1, L', H in G1, then B2 in G2; 960 B for BLS12-381) go back to HBM, ONE
      // ncclAllGather on the reduction stream, then the O(world) additions and the O(1) tail on the host -- on every rank
      uint8_t* mine = land + parts.bytes;
      memcpy(mine, h1, sizeof(h1));
      memcpy(mine + sizeof(h1), &h2, sizeof(h2));
      cm->gather.ensure(psz * (size_t)(cm->world + 1));
      uint8_t* d_mine = cm->gather.as<uint8_t>() + psz * (size_t)cm->world;
      ARK_CHECK_HIP(hipMemcpyAsync(d_mine, mine, psz, hipMemcpyHostToDevice, sR));
      ARK_CHECK_NCCL(ncclAllGather(d_mine, cm->gather.p, psz, ncclUint8, cm->comm, sR));
      const size_t all_bytes = psz * (size_t)cm->world;
      uint8_t* all = mine + psz;
      ARK_CHECK_HIP(hipMemcpyAsync(all, cm->gather.p, all_bytes, hipMemcpyDeviceToHost, sR));
      ARK_CHECK_HIP(hipEventRecord(ev[E_END], sR));
      wait_event_polite(ev[E_END], spin);
      combine_partials_host<Curve>(all, (uint64_t)cm->world, r_canon, s_canon, out);
    } else if (partials_out) {
      // sharded prove: hand back the five XYZZ partial sums (A, B1, L', H in G1, then B2 in G2)
      memcpy(partials_out, h1, sizeof(h1));
      memcpy(partials_out + sizeof(h1), &h2, sizeof(h2));
    } else {
      finalize_host<Curve>(h1, h2, rc, scn, out);
    }
    t_tail = since(t_enter);
    // Every stream has drained into sR through the event chain: E_END completes only after the last event of sW (E_H),
    // sS (E_SORT2) and the accumulation stream (E_ACC_DONE0 + 4); nothing else is queued on them.  The prover used to call
    // hipStreamSynchronize on the feeder streams here.  That is NOT free: HIP streams share a handful of hardware queues, and
    // a synchronise on an idle stream of this proof waits -- spinning -- for the other proofs' kernels in the same queue: 22
    // and 82 ms in two of six traced 2^20 proofs with four in flight (profiles/r03_host_cpu.txt).  Measured, same box
    // (profiles/r03_epilogue_ab.txt): without them 2^20 x 4 in flight ran 23.9 instead of 24.7 ms per proof on 0.8 instead of
    // 1.5 host cores -- but 2^18 x 8 in flight 10.8 instead of 8.5 ms.  Both epilogues are therefore SCHEDULES
    // (SCHED_PIPELINE checks the three events, SCHED_PIPELINE_SYNC synchronises) and the choice between them is measured
    // per class like the rest; a one-stream proof has nothing to synchronise.
    if (sched == SCHED_PIPELINE_SYNC) {
      ARK_CHECK_HIP(hipStreamSynchronize(sS));
      ARK_CHECK_HIP(hipStreamSynchronize(sW));
    } else if (!one_stream) {
      for (hipEvent_t last : {ev[E_ACC_DONE0 + 4], h_tails_aside ? ev[E_HT] : ev[E_SORT2], ev[E_H]}) {
        const hipError_t q = hipEventQuery(last);
        if (q == hipErrorNotReady) wait_event_polite(last, spin);
        else if (q != hipSuccess) ARK_CHECK_HIP(q);
      }
    }
    if (exploring) {
      explore_guard.armed = false;
      const int now_in_flight = inflight.c.load();
      if (!concurrent && now_in_flight > 1) {
        SchedTuner::of(ctx->device).unstart(tune_key, sched);
      } else {
        // in flight: the proof's wall time over the number of proofs that shared the device with it (mean of the counts at its
        // start and at its end) -- an estimate of the time per proof that does not reward a phase for running while the
        // caller's batch ramps down (run G: six in flight, a phase at the end of a batch read 107 ms of latency against
        // 129 ms and was latched; the timed region then ran 7 % slower)
        const double share = concurrent ? 0.5 * (double)(inflight.mine + now_in_flight) : 1.0;
        SchedTuner::of(ctx->device).report(tune_key, concurrent, sched, since(t_enter) / (share < 1.0 ? 1.0 : share), pol.sched_explore,
                                           static_sched);
      }
    }
    auto el = [&](hipEvent_t a, hipEvent_t b) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      return ms;
    };
    // phases overlap across streams: the entries are elapsed times of the respective stream segments
    if (trace_host)
      fprintf(stderr, "[ark355] prove host wall: all work queued %.2f ms, GPU drained %.2f ms, host tail done %.2f ms, "
                      "epilogue %.2f ms; GPU span %.2f ms\n",
              t_launched, t_synced, t_tail, since(t_enter), el(ev[E_START], ev[E_END]));
    ctx->timings.total_ms = el(ev[E_START], ev[E_END]);
    ctx->timings.h2d_ms = el(ev[E_START], ev[E_Z]);
    ctx->timings.witness_map_ms = el(ev[E_Z], ev[E_H]);
    ctx->timings.msm_b_g2_ms = el(acc0[0], ev[E_ACC_DONE0 + 0]);
    ctx->timings.msm_ab_g1_ms = el(acc0[1], ev[E_ACC_DONE0 + 2]);
    ctx->timings.msm_l_ms = el(acc0[3], ev[E_ACC_DONE0 + 3]);
    ctx->timings.msm_h_ms = el(acc0[4], ev[E_ACC_DONE0 + 4]);
    ctx->timings.finalize_ms = el(ev[E_ACC_DONE0 + 4], ev[E_END]);
    float acc_ms = 0;
    for (int i = 0; i < 5; i++) acc_ms += el(acc0[i], acc1[i]);
    ctx->acc_ms = acc_ms;
    ctx->acc_launches = 5;
    ctx->acc_points = pts;
    if (check_sat) {
      unsigned long long fb;
      memcpy(&fb, land_sat, 8);
      if (fb != ~0ull) {
        if (out) memset(out, 0, sizeof(*out));
        char msg[160];
        snprintf(msg, sizeof(msg), "constraint %llu is not satisfied by the assignment (policy CHECK_SATISFIED; the first of them)", fb);
        throw HipError{ARK355_E_UNSATISFIABLE, msg};
      }
    }
  }
}

}  // namespace ark355
