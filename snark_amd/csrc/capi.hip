// extern "C" surface of libark355.so (include/ark355.h).  No C++ exception crosses this boundary:
// every entry point catches and maps to an error code (the reference builds with panic='abort' for
// exactly that reason, /root/reference/Cargo.toml:33,45).
#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include "api_impl.cuh"

namespace ark355 {
extern template struct Api<BlsCurve>;
extern template struct Api<BnCurve>;
}  // namespace ark355

using namespace ark355;

struct ark355_pk {
  PkDev* d;
};
struct ark355_r1cs {
  R1csDev* d;
};
struct ark355_bases {
  BasesDev* d;
};
struct ark355_comm {
  CommDev* d;
};

namespace {
struct CtxExtra {
  ProverScratch prover;
  GenericScratch generic;
  std::vector<ark355_ctx*> lanes;     // child contexts of ark355_prove_batch (same device, private streams/scratch)
  ~CtxExtra();
};
std::mutex g_extra_mu;
std::map<ark355_ctx*, CtxExtra*> g_extra;

CtxExtra& extra(ark355_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_extra_mu);
  auto it = g_extra.find(ctx);
  if (it == g_extra.end()) it = g_extra.emplace(ctx, new CtxExtra()).first;
  return *it->second;
}

CtxExtra::~CtxExtra() {
  for (ark355_ctx* c : lanes) ark355_ctx_destroy(c);
}

// locked: the caller already holds ctx->mu (ark355_diag_streams takes every context's mutex in one sorted try-lock pass)
template <class Fn>
int32_t guarded(ark355_ctx* ctx, Fn&& fn, bool locked = false) {
  try {
    if (ctx && !locked) {
      std::lock_guard<std::mutex> lk(ctx->mu);
      (void)hipSetDevice(ctx->device);
      fn();
    } else {
      if (ctx) (void)hipSetDevice(ctx->device);
      fn();
    }
    return ARK355_OK;
  } catch (const HipError& e) {
    if (ctx) ctx->last_error = e.what;
    return e.code;
  } catch (const std::bad_alloc&) {
    if (ctx) ctx->last_error = "host allocation failed";
    return ARK355_ENOMEM;
  } catch (const std::exception& e) {
    if (ctx) ctx->last_error = e.what();
    return ARK355_EINVAL;
  } catch (...) {
    if (ctx) ctx->last_error = "unknown error";
    return ARK355_EINVAL;
  }
}

#define CURVE_DISPATCH(curve, CALL)                                           \
  do {                                                                        \
    if ((curve) == ARK355_BLS12_381) {                                        \
      using A = Api<BlsCurve>;                                                \
      CALL;                                                                   \
    } else if ((curve) == ARK355_BN254) {                                     \
      using A = Api<BnCurve>;                                                 \
      CALL;                                                                   \
    } else {                                                                  \
      throw HipError{ARK355_EINVAL, "unknown curve id"};                      \
    }                                                                         \
  } while (0)
}  // namespace

extern "C" {

uint32_t ark355_version(void) { return (0u << 16) | 1u; }

int32_t ark355_sizes(int32_t curve, uint32_t what[4]) {
  return guarded(nullptr, [&] { CURVE_DISPATCH(curve, A::sizes(what)); });
}

int32_t ark355_ctx_create(int32_t device_id, ark355_ctx** out) {
  if (!out) return ARK355_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ARK355_ENODEV;
  if (device_id < 0 || device_id >= count) return ARK355_EINVAL;
  ark355_ctx* ctx = new (std::nothrow) ark355_ctx();
  if (!ctx) return ARK355_ENOMEM;
  ctx->device = device_id;
  ctx->policy = TunePolicy::from_env();       // the ONLY place the environment is read (policy.h)
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
    delete ctx;
    return ARK355_EHIP;
  }
  // streams for one-stream proofs, probed (once per device, now: the device is idle) to sit on different hardware queues
  try {
    uint32_t want = 4;
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) {
      const int v = atoi(e);
      if (v >= 1 && v <= 16) want = (uint32_t)v;
    }
    LanePool::of(device_id).build(want);
  } catch (...) {
  }
  *out = ctx;
  return ARK355_OK;
}

void ark355_ctx_destroy(ark355_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  CtxExtra* ex = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_extra_mu);
    auto it = g_extra.find(ctx);
    if (it != g_extra.end()) {
      ex = it->second;
      g_extra.erase(it);
    }
  }
  delete ex;       // outside the lock: it destroys the batch worker contexts, which come back through here
  ctx->ntt_tables.clear();          // shared tables live on while another context of the device holds them
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int32_t ark355_host_alloc(uint64_t bytes, void** out) {
  if (!out) return ARK355_EINVAL;
  *out = nullptr;
  void* p = nullptr;
#if defined(ARK_EMUL)
  p = malloc(bytes ? bytes : 1);
  if (!p) return ARK355_ENOMEM;
#else
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return ARK355_ENOMEM;
#endif
  *out = p;
  return ARK355_OK;
}
void ark355_host_free(void* p) {
  if (!p) return;
#if defined(ARK_EMUL)
  free(p);
#else
  (void)hipHostFree(p);
#endif
}

const char* ark355_last_error(const ark355_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int32_t ark355_ctx_set_policy(ark355_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return ARK355_EINVAL;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (ctx->policy.set(name, value) != 0) {
    ctx->last_error = std::string("unknown policy name: ") + name;
    return ARK355_EINVAL;
  }
  return ARK355_OK;
}
int32_t ark355_ctx_get_policy(ark355_ctx* ctx, const char* name, int64_t* value) {
  if (!ctx || !name || !value) return ARK355_EINVAL;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ctx->policy.get(name, value) == 0 ? ARK355_OK : ARK355_EINVAL;
}
int32_t ark355_sched_info(const ark355_ctx* ctx, const ark355_pk* pk, int32_t in_flight, ark355_sched_report* out) {
  if (!ctx || !pk || !pk->d || !out) return ARK355_EINVAL;
  memset(out, 0, sizeof(*out));
  out->latched = -1;
  out->last = ctx->last_sched;
  double mean[SCHED_COUNT];
  uint32_t n[SCHED_COUNT];
  int latched = -1;
  const uint64_t key = SchedTuner::key(prove_shape(pk->d->curve, pk->d->N, pk->d->m), in_flight != 0);
  if (SchedTuner::of(ctx->device).info(key, &latched, mean, n)) {
    out->latched = latched;
    static_assert(SCHED_COUNT == 4, "ark355_sched_report holds four schedules");
    for (int v = 0; v < SCHED_COUNT; v++) {
      out->mean_ms[v] = mean[v];
      out->samples[v] = n[v];
    }
  }
  return ARK355_OK;
}
int32_t ark355_diag_streams(ark355_ctx** ctxs, uint32_t count, int8_t* serialised) {
  if (!ctxs || !serialised || count == 0 || count > 16) return ARK355_EINVAL;
  for (uint32_t i = 0; i < count; i++)
    if (!ctxs[i]) return ARK355_EINVAL;
  // The probe synchronises and launches on EVERY passed context's stream, so every one of them is locked for its duration --
  // ctxs[0] included, all in ONE pass of try-locks in address order: no caller ever blocks while it holds a mutex, so two callers
  // with overlapping sets cannot deadlock (ctxs[0] used to be locked blockingly after the others), and a context
  // that is proving right now makes the call fail instead of racing with it.  The bookkeeping allocates: it
  // sits inside the same exception mapping as everything else.
  std::vector<std::unique_lock<std::mutex>> held;
  try {
    std::vector<ark355_ctx*> all(ctxs, ctxs + count);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (ark355_ctx* c : all) {
      std::unique_lock<std::mutex> lk(c->mu, std::try_to_lock);
      if (!lk.owns_lock()) return ARK355_EINVAL;          // busy: the diagnostic wants an idle device
      held.push_back(std::move(lk));
    }
  } catch (const std::bad_alloc&) {
    return ARK355_ENOMEM;
  } catch (...) {
    return ARK355_EINVAL;
  }
  // streams probed: every context's own stream, then the three feeder streams of ctxs[0] (created if need be)
  return guarded(ctxs[0], [&] {
    CtxExtra& ex = extra(ctxs[0]);
    ex.prover.ensure_streams(ctxs[0]->policy.stream_prio != 0);
    std::vector<hipStream_t> st;
    for (uint32_t i = 0; i < count; i++) st.push_back(ctxs[i]->stream);
    st.push_back(ex.prover.sW);
    st.push_back(ex.prover.sS);
    st.push_back(ex.prover.sR);
    const size_t n = st.size();
    for (size_t i = 0; i < n; i++)
      for (size_t j = 0; j < n; j++)
        serialised[i * n + j] = (int8_t)(i == j ? 1 : diag_streams_serialised(st[i], st[j]));
  }, /*locked=*/true);
}
int32_t ark355_diag_dispatch(ark355_ctx* ctx, uint32_t launches, uint32_t spin_us, float* gap_us, uint32_t* lanes) {
  if (!ctx || !gap_us || launches == 0 || launches > 100000 || spin_us > 100000) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    *gap_us = diag_dispatch_gap_us(ctx->stream, launches, spin_us);
    if (lanes) *lanes = (uint32_t)LanePool::of(ctx->device).size();
  });
}
int32_t ark355_diag_mad_rate(ark355_ctx* ctx, float target_ms, float* tmad_per_s, float* elapsed_ms) {
  if (!ctx || !tmad_per_s || !(target_ms >= 0.1f) || target_ms > 1000.f) return ARK355_EINVAL;
  return guarded(ctx, [&] { *tmad_per_s = diag_mad_rate_t(ctx->stream, ctx->device, target_ms, elapsed_ms); });
}
int32_t ark355_diag_clocks(ark355_ctx* ctx, uint64_t* pairs, uint32_t capacity, uint32_t* count) {
  if (!ctx || !pairs || !count || capacity < DIAG_CLOCK_SLOTS) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    diag_clocks(ctx->stream, pairs);
    *count = DIAG_CLOCK_SLOTS;
  });
}
int32_t ark355_sched_reset(const ark355_ctx* ctx) {
  if (!ctx) return ARK355_EINVAL;
  SchedTuner::of(ctx->device).reset();
  return ARK355_OK;
}

int32_t ark355_pk_load(ark355_ctx* ctx, int32_t curve, const ark355_pk_desc* desc, ark355_pk** out) {
  if (!ctx || !desc || !out) return ARK355_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    PkDev* d = nullptr;
    CURVE_DISPATCH(curve, d = A::pk_load(ctx->policy, desc, ctx->stream));
    *out = new ark355_pk{d};
  });
}
int32_t ark355_pk_load_shard(ark355_ctx* ctx, int32_t curve, const ark355_pk_desc* desc, uint32_t shard_index,
                             uint32_t shard_count, ark355_pk** out) {
  if (!ctx || !desc || !out) return ARK355_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    PkDev* d = nullptr;
    CURVE_DISPATCH(curve, d = A::pk_load(ctx->policy, desc, ctx->stream, shard_index, shard_count));
    *out = new ark355_pk{d};
  });
}
void ark355_pk_free(ark355_pk* pk) {
  if (!pk) return;
  delete pk->d;
  delete pk;
}

int32_t ark355_r1cs_load(ark355_ctx* ctx, int32_t curve, uint64_t n, uint64_t ell, uint64_t w,
                         const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                         const uint8_t* const coeff[3], ark355_r1cs** out) {
  if (!ctx || !row_ptr || !col || !coeff || !out) return ARK355_EINVAL;
  *out = nullptr;
  for (int i = 0; i < 3; i++)
    if (!row_ptr[i]) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    R1csDev* d = nullptr;
    CURVE_DISPATCH(curve, d = A::r1cs_load(n, ell, w, row_ptr, col, coeff));
    *out = new ark355_r1cs{d};
  });
}
void ark355_r1cs_free(ark355_r1cs* r) {
  if (!r) return;
  delete r->d;
  delete r;
}
uint64_t ark355_r1cs_domain_size(const ark355_r1cs* r) { return r ? r->d->N : 0; }

static int32_t prove_common(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1, const void* z, uint64_t z_len,
                            bool on_dev, const uint8_t* r, const uint8_t* s, ark355_proof_raw* out) {
  if (!ctx || !pk || !r1 || !z || !r || !s || !out) return ARK355_EINVAL;
  if (z_len < r1->d->m) {
    ctx->last_error = "assignment shorter than num_instance + num_witness";
    return ARK355_E_ASSIGNMENT_MISSING;
  }
  if (pk->d->shard_count != 1) {
    ctx->last_error = "this key handle is an MSM shard: use ark355_prove_shard + ark355_prove_combine";
    return ARK355_EINVAL;
  }
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(pk->d->curve, A::prove(ctx, ex.prover, *pk->d, *r1->d, z, on_dev, r, s, out));
  });
}

int32_t ark355_prove(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len,
                     const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out) {
  return prove_common(ctx, pk, r1, z, z_len, false, r, s, out);
}
int32_t ark355_prove_dev(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1, const void* d_z, uint64_t z_len,
                         const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out) {
  return prove_common(ctx, pk, r1, d_z, z_len, true, r, s, out);
}

int32_t ark355_prove_batch(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1, const uint8_t* const* z,
                           uint64_t z_len, const uint8_t* r, const uint8_t* s, uint64_t count, uint32_t inflight,
                           ark355_proof_raw* out) {
  if (!ctx || !pk || !r1 || (count && (!z || !r || !s || !out))) return ARK355_EINVAL;
  if (inflight < 1 || inflight > 16) return ARK355_EINVAL;
  if (count == 0) return ARK355_OK;
#if defined(ARK_EMUL)
  inflight = 1;                        // the test-only emulator is single-threaded
#endif
  if (inflight > count) inflight = (uint32_t)count;
  int32_t first_err = ARK355_OK;
  std::string first_msg;
  const int32_t rc = guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    while (ex.lanes.size() + 1 < inflight) {
      ark355_ctx* c = nullptr;
      const int32_t e = ark355_ctx_create(ctx->device, &c);
      if (e != ARK355_OK) throw HipError{e, "ark355_prove_batch: cannot create a worker context"};
      ex.lanes.push_back(c);
    }
    for (ark355_ctx* c : ex.lanes) c->policy = ctx->policy;       // the workers prove under the caller's policy
    std::atomic<uint64_t> next{0};
    std::mutex err_mu;
    auto work = [&](ark355_ctx* lane) {
      for (;;) {
        const uint64_t i = next.fetch_add(1);
        if (i >= count) return;
        int32_t e;
        if (lane == ctx) {
          // the parent's mutex is already held by this call: run its share on the calling thread, unguarded path
          e = ARK355_OK;
          try {
            if (!z[i]) throw HipError{ARK355_EINVAL, "null assignment pointer"};
            if (z_len < r1->d->m) throw HipError{ARK355_E_ASSIGNMENT_MISSING, "assignment shorter than num_instance + num_witness"};
            if (pk->d->shard_count != 1) throw HipError{ARK355_EINVAL, "this key handle is an MSM shard"};
            (void)hipSetDevice(ctx->device);
            CURVE_DISPATCH(pk->d->curve, A::prove(ctx, ex.prover, *pk->d, *r1->d, z[i], false, r + 32 * i, s + 32 * i, out + i));
          } catch (const HipError& he) {
            e = he.code;
            ctx->last_error = he.what;
          } catch (const std::exception& se) {
            e = ARK355_EINVAL;
            ctx->last_error = se.what();
          }
        } else {
          e = z[i] ? prove_common(lane, pk, r1, z[i], z_len, false, r + 32 * i, s + 32 * i, out + i) : ARK355_EINVAL;
        }
        if (e != ARK355_OK) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (first_err == ARK355_OK) {
            first_err = e;
            first_msg = "proof " + std::to_string(i) + ": " + lane->last_error;
          }
        }
      }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 1; k < inflight; k++) th.emplace_back(work, ex.lanes[k - 1]);
    work(ctx);
    for (auto& t : th) t.join();
  });
  if (rc != ARK355_OK) return rc;
  if (first_err != ARK355_OK) ctx->last_error = first_msg;
  return first_err;
}

uint64_t ark355_partial_size(int32_t curve) {
  size_t n = 0;
  try {
    CURVE_DISPATCH(curve, n = A::partial_size());
  } catch (...) {
    return 0;
  }
  return n;
}

int32_t ark355_prove_shard(ark355_ctx* ctx, const ark355_pk* pk, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len,
                           const uint8_t r[32], const uint8_t s[32], uint8_t* out_partials) {
  if (!ctx || !pk || !r1 || !z || !r || !s || !out_partials) return ARK355_EINVAL;
  if (z_len < r1->d->m) return ARK355_E_ASSIGNMENT_MISSING;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(pk->d->curve, A::prove(ctx, ex.prover, *pk->d, *r1->d, z, false, r, s, nullptr, out_partials));
  });
}

int32_t ark355_prove_combine(ark355_ctx* ctx, int32_t curve, const uint8_t* partials, uint64_t count,
                             const uint8_t r[32], const uint8_t s[32], ark355_proof_raw* out) {
  if (!ctx || !partials || !count || !r || !s || !out) return ARK355_EINVAL;
  return guarded(ctx, [&] { CURVE_DISPATCH(curve, A::combine(partials, count, r, s, out)); });
}

int32_t ark355_comm_unique_id(uint8_t id[ARK355_COMM_ID_BYTES]) {
  if (!id) return ARK355_EINVAL;
  static_assert(sizeof(ncclUniqueId) <= ARK355_COMM_ID_BYTES, "communicator id does not fit");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return ARK355_ERCCL;
  memset(id, 0, ARK355_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return ARK355_OK;
}

int32_t ark355_comm_init(ark355_ctx* ctx, const uint8_t id[ARK355_COMM_ID_BYTES], int32_t rank, int32_t world,
                         ark355_comm** out) {
  if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return ARK355_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    auto* d = new CommDev();
    d->rank = rank;
    d->world = world;
    d->device = ctx->device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    const ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) {
      d->comm = nullptr;
      delete d;
      throw HipError{ARK355_ERCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)};
    }
    *out = new ark355_comm{d};
  });
}

void ark355_comm_destroy(ark355_comm* comm) {
  if (!comm) return;
  if (comm->d) (void)hipSetDevice(comm->d->device);
  delete comm->d;
  delete comm;
}

static int32_t prove_sharded_common(ark355_ctx* ctx, ark355_comm* comm, const ark355_pk* pk, const ark355_r1cs* r1,
                                    const void* z, bool on_dev, uint64_t z_len, const uint8_t* r, const uint8_t* s,
                                    int32_t mode, ark355_proof_raw* out) {
  if (!ctx || !comm || !pk || !r1 || !z || !r || !s || !out) return ARK355_EINVAL;
  if (mode != ARK355_SHARD_WINDOW && mode != ARK355_SHARD_BUCKET_RING) return ARK355_EINVAL;
  if (z_len < r1->d->m) {
    ctx->last_error = "assignment shorter than num_instance + num_witness";
    return ARK355_E_ASSIGNMENT_MISSING;
  }
  if ((int)pk->d->shard_count != comm->d->world || (int)pk->d->shard_index != comm->d->rank) {
    ctx->last_error = "key shard (index, count) does not match the communicator's (rank, world)";
    return ARK355_EINVAL;
  }
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(pk->d->curve, A::prove(ctx, ex.prover, *pk->d, *r1->d, z, on_dev, r, s, out, nullptr, comm->d, mode));
  });
}
int32_t ark355_prove_sharded(ark355_ctx* ctx, ark355_comm* comm, const ark355_pk* pk, const ark355_r1cs* r1,
                             const uint8_t* z, uint64_t z_len, const uint8_t r[32], const uint8_t s[32], int32_t mode,
                             ark355_proof_raw* out) {
  return prove_sharded_common(ctx, comm, pk, r1, z, false, z_len, r, s, mode, out);
}
int32_t ark355_prove_sharded_dev(ark355_ctx* ctx, ark355_comm* comm, const ark355_pk* pk, const ark355_r1cs* r1,
                                 const void* d_z, uint64_t z_len, const uint8_t r[32], const uint8_t s[32], int32_t mode,
                                 ark355_proof_raw* out) {
  return prove_sharded_common(ctx, comm, pk, r1, d_z, true, z_len, r, s, mode, out);
}

int32_t ark355_witness_map(ark355_ctx* ctx, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len, uint8_t* h_out) {
  if (!ctx || !r1 || !z || !h_out) return ARK355_EINVAL;
  if (z_len < r1->d->m) return ARK355_E_ASSIGNMENT_MISSING;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(r1->d->curve, A::witness_map(ctx, ex.prover, *r1->d, z, h_out));
  });
}

int32_t ark355_witness_map_dist_sim(ark355_ctx* ctx, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len, uint32_t world,
                                    uint8_t* h_out) {
  if (!ctx || !r1 || !z || !h_out) return ARK355_EINVAL;
  if (z_len < r1->d->m) return ARK355_E_ASSIGNMENT_MISSING;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(r1->d->curve, A::witness_map_dist_sim(ctx, ex.prover, *r1->d, z, world, h_out));
  });
}

int32_t ark355_is_satisfied(ark355_ctx* ctx, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len,
                            int64_t* first_bad) {
  if (!ctx || !r1 || !z || !first_bad) return ARK355_EINVAL;
  if (z_len < r1->d->m) return ARK355_E_ASSIGNMENT_MISSING;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(r1->d->curve, A::mat_vec(ctx, ex.prover, *r1->d, z, nullptr, nullptr, nullptr, first_bad));
  });
}

int32_t ark355_r1cs_mat_vec(ark355_ctx* ctx, const ark355_r1cs* r1, const uint8_t* z, uint64_t z_len, uint8_t* az,
                            uint8_t* bz, uint8_t* cz) {
  if (!ctx || !r1 || !z) return ARK355_EINVAL;
  if (z_len < r1->d->m) return ARK355_E_ASSIGNMENT_MISSING;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(r1->d->curve, A::mat_vec(ctx, ex.prover, *r1->d, z, az, bz, cz, nullptr));
  });
}

int32_t ark355_ntt_fr(ark355_ctx* ctx, int32_t curve, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset) {
  if (!ctx || !data) return ARK355_EINVAL;
  if (log_n > 40) return ARK355_E_POLY_DEGREE_TOO_LARGE;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::ntt_host(ctx, ex.generic, data, log_n, inverse != 0, coset != 0));
  });
}

int32_t ark355_ntt_fr_dev(ark355_ctx* ctx, int32_t curve, void* d_data, void* d_scratch, uint32_t log_n,
                          int32_t inverse, int32_t coset, void* stream) {
  if (!ctx || !d_data || !d_scratch) return ARK355_EINVAL;
  if (log_n > 40) return ARK355_E_POLY_DEGREE_TOO_LARGE;
  return guarded(ctx, [&] {
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    CURVE_DISPATCH(curve, A::ntt_dev(ctx, d_data, d_scratch, log_n, inverse != 0, coset != 0, st));
  });
}

static int32_t msm_host_common(ark355_ctx* ctx, int32_t curve, int group, const uint8_t* bases, const uint8_t* scalars,
                               uint64_t n, uint8_t* out) {
  if (!ctx || !out || (n && (!bases || !scalars))) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::msm_host(ctx, ex.generic, group, bases, scalars, n, out));
  });
}
int32_t ark355_msm_g1(ark355_ctx* ctx, int32_t curve, const uint8_t* bases, const uint8_t* scalars, uint64_t n,
                      uint8_t* out) {
  return msm_host_common(ctx, curve, 1, bases, scalars, n, out);
}
int32_t ark355_msm_g2(ark355_ctx* ctx, int32_t curve, const uint8_t* bases, const uint8_t* scalars, uint64_t n,
                      uint8_t* out) {
  return msm_host_common(ctx, curve, 2, bases, scalars, n, out);
}

int32_t ark355_bases_load(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* bases, uint64_t n,
                          ark355_bases** out) {
  if (!ctx || !out || (n && !bases) || (group != 1 && group != 2)) return ARK355_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    BasesDev* d = nullptr;
    CURVE_DISPATCH(curve, d = A::bases_load(ctx->policy, group, bases, n, ctx->stream));
    *out = new ark355_bases{d};
  });
}
void ark355_bases_free(ark355_bases* b) {
  if (!b) return;
  delete b->d;
  delete b;
}

int32_t ark355_msm_dev(ark355_ctx* ctx, const ark355_bases* bases, const void* d_scalars, uint64_t n,
                       int32_t scalars_mont, uint8_t* out_affine) {
  if (!ctx || !bases || !out_affine || (n && !d_scalars)) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(bases->d->curve, A::msm_dev(ctx, ex.generic, *bases->d, d_scalars, n, scalars_mont, out_affine, true));
  });
}
int32_t ark355_msm_dev_partial(ark355_ctx* ctx, const ark355_bases* bases, const void* d_scalars, uint64_t n,
                               int32_t scalars_mont, uint8_t* out_xyzz) {
  if (!ctx || !bases || !out_xyzz || (n && !d_scalars)) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(bases->d->curve, A::msm_dev(ctx, ex.generic, *bases->d, d_scalars, n, scalars_mont, out_xyzz, false));
  });
}

int32_t ark355_xyzz_sum(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* partials, uint64_t count,
                        uint8_t* out_affine) {
  if (!ctx || !out_affine || (count && !partials) || (group != 1 && group != 2) || count >= (1ull << 31))
    return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::xyzz_sum(ctx, ex.generic, group, partials, count, out_affine));
  });
}

int32_t ark355_fixed_base_mul(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* base,
                              const uint8_t* scalars, uint64_t n, uint8_t* out_affine) {
  if (!ctx || !base || (n && (!scalars || !out_affine)) || (group != 1 && group != 2)) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::fixed_base(ctx, ex.generic, group, base, scalars, n, out_affine));
  });
}

uint64_t ark355_point_size(int32_t curve, int32_t group, int32_t compressed) {
  if (group != 1 && group != 2) return 0;
  size_t n = 0;
  try {
    CURVE_DISPATCH(curve, n = A::point_size(group, compressed != 0));
  } catch (...) {
    return 0;
  }
  return n;
}

int32_t ark355_pk_load_bytes(ark355_ctx* ctx, int32_t curve, const uint8_t* bytes, uint64_t len, int32_t compressed,
                             int32_t validate, ark355_pk** out) {
  if (!ctx || !bytes || !out || validate < 0 || validate > 2) return ARK355_EINVAL;
  *out = nullptr;
  return guarded(ctx, [&] {
    PkDev* d = nullptr;
    CURVE_DISPATCH(curve, d = A::pk_load_bytes(ctx, bytes, len, compressed != 0, validate));
    *out = new ark355_pk{d};
  });
}

int32_t ark355_pk_dims(const ark355_pk* pk, uint64_t* num_instance, uint64_t* num_witness, uint64_t* domain_size) {
  if (!pk || !pk->d) return ARK355_EINVAL;
  if (num_instance) *num_instance = pk->d->ell;
  if (num_witness) *num_witness = pk->d->w;
  if (domain_size) *domain_size = pk->d->N;
  return ARK355_OK;
}

int32_t ark355_pk_table_info(const ark355_pk* pk, uint32_t* window_bits, uint32_t* windows, uint32_t* table_stride,
                             uint64_t* table_bytes) {
  if (!pk || !pk->d) return ARK355_EINVAL;
  const MsmPlan& p = pk->d->a_ext.plan;
  if (window_bits) *window_bits = p.c;
  if (windows) *windows = p.windows;
  if (table_stride) *table_stride = pk->d->wstride;
  if (table_bytes) *table_bytes = pk->d->table_bytes();
  return ARK355_OK;
}

int32_t ark355_points_decode(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* in, uint64_t n,
                             int32_t compressed, int32_t validate, uint8_t* out_raw) {
  if (!ctx || (group != 1 && group != 2) || (n && (!in || !out_raw)) || validate < 0 || validate > 2) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::points_decode(ctx, ex.generic, group, in, n, compressed != 0, validate, out_raw));
  });
}

int32_t ark355_points_encode(ark355_ctx* ctx, int32_t curve, int32_t group, const uint8_t* in_raw, uint64_t n,
                             int32_t compressed, uint8_t* out) {
  if (!ctx || (group != 1 && group != 2) || (n && (!in_raw || !out))) return ARK355_EINVAL;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    CURVE_DISPATCH(curve, A::points_encode(ctx, ex.generic, group, in_raw, n, compressed != 0, out));
  });
}

int32_t ark355_proof_to_bytes(int32_t curve, const ark355_proof_raw* proof, int32_t compressed, uint8_t* out) {
  if (!proof || !out) return ARK355_EINVAL;
  return guarded(nullptr, [&] { CURVE_DISPATCH(curve, A::proof_to_bytes(proof, compressed != 0, out)); });
}

int32_t ark355_proof_from_bytes(int32_t curve, const uint8_t* in, uint64_t len, int32_t compressed, int32_t validate,
                                ark355_proof_raw* out) {
  if (!in || !out || validate < 0 || validate > 2) return ARK355_EINVAL;
  return guarded(nullptr, [&] { CURVE_DISPATCH(curve, A::proof_from_bytes(in, len, compressed != 0, validate, out)); });
}

int32_t ark355_verify_batch(ark355_ctx* ctx, int32_t curve, const ark355_vk_desc* vk, const ark355_proof_raw* proofs,
                            const uint8_t* public_inputs, const uint8_t* rho, uint64_t count, int32_t* ok) {
  if (!ctx || !vk || !proofs || !ok || !vk->alpha_g1 || !vk->beta_g2 || !vk->gamma_g2 || !vk->delta_g2 || !vk->gamma_abc_g1 ||
      (vk->num_instance > 1 && !public_inputs))
    return ARK355_EINVAL;
  *ok = 0;
  return guarded(ctx, [&] {
    CtxExtra& ex = extra(ctx);
    bool good = false;
    CURVE_DISPATCH(curve, good = A::verify_batch(ctx, ex.generic, vk, proofs, public_inputs, rho, count));
    *ok = good ? 1 : 0;
  });
}

int32_t ark355_setup_scalars(int32_t curve, uint64_t n, uint64_t ell, uint64_t w, const uint64_t* const row_ptr[3],
                             const uint32_t* const col[3], const uint8_t* const coeff[3], const uint8_t* trapdoor,
                             uint8_t* out_u, uint8_t* out_v, uint8_t* out_w, uint8_t* out_l, uint8_t* out_gamma_abc,
                             uint8_t* out_h) {
  // zero-length outputs may be NULL (l when num_witness == 0, h when the domain has one point)
  if (!row_ptr || !col || !coeff || !trapdoor || !out_u || !out_v || !out_w || (w && !out_l) || !out_gamma_abc ||
      (n + ell > 1 && !out_h))
    return ARK355_EINVAL;
  for (int i = 0; i < 3; i++)
    if (!row_ptr[i] || (row_ptr[i][n] && (!col[i] || !coeff[i]))) return ARK355_EINVAL;
  return guarded(nullptr, [&] {
    CURVE_DISPATCH(curve, A::setup_scalars(n, ell, w, row_ptr, col, coeff, trapdoor, out_u, out_v, out_w, out_l, out_gamma_abc, out_h));
  });
}

int32_t ark355_get_timings(const ark355_ctx* ctx, ark355_timings* out) {
  if (!ctx || !out) return ARK355_EINVAL;
  *out = ctx->timings;
  return ARK355_OK;
}

int32_t ark355_get_kernel_stats(const ark355_ctx* ctx, float* accumulate_ms, uint64_t* launches, uint64_t* points) {
  if (!ctx) return ARK355_EINVAL;
  if (accumulate_ms) *accumulate_ms = ctx->acc_ms;
  if (launches) *launches = ctx->acc_launches;
  if (points) *points = ctx->acc_points;
  return ARK355_OK;
}

}  // extern "C"
