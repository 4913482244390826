// TunePolicy: every runtime switch of the library in ONE struct per context.
//
// The environment (ARK355_<NAME>) is read exactly once, when a context is created (ark355_ctx_create); afterwards a
// context's policy changes only through ark355_ctx_set_policy(ctx, "<NAME>", value).  Nothing on the proving path calls
// getenv, and two contexts of one process can run different policies (bench.py's in-run A/B does).  Child contexts of
// ark355_prove_batch inherit the parent's policy at every call.
//
// Three groups:
//   per proof      SCHED, SCHED_EXPLORE, WAIT_SPIN, WAIT_ADAPT, STREAM_PRIO, BATCH_TAILS, SIDE_G2_TAILS,
//                  SIDE_WM, SIDE_H_TAILS, TRACE_HOST (+ the legacy spellings SERIAL and EPILOGUE_SYNC, which map onto SCHED);
//                  sharded proofs: DWM_LOOPBACK, RCCL_SELF (window / bucket-ring combining is the `mode` argument of
//                  ark355_prove_sharded, not a policy)
//   per key load   MSM_C, MSM_C_H, PACK_ROWS, TABLE_STRIDE, HBM_BUDGET_MB, SHARD_DIST_WM
//                  (read when a key / base set is loaded through the context: the tables are built for them)
//   per call       MSM_SEG, ACC_THREADS, MSM_TWO_LEVEL_MIN, NTT_RMAX, NTT_DIRECT_MAX, NTT_NOFUSE (A/B and test knobs of the kernels'
//                  host drivers), CHECK_SATISFIED
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace ark355 {

// How one proof is laid onto HIP streams and how its thread waits for it (groth16_impl.cuh, prove_run).
enum Sched : int32_t {
  SCHED_AUTO = -1,          // measured choice: the first proofs of a (device, shape, alone / in flight) class try the
                            // candidates below, the class then keeps the fastest (SchedTuner)
  SCHED_ONE_STREAM = 0,     // every kernel of the proof on the context's stream; the thread polls the last event
  SCHED_PIPELINE = 1,       // five-stream pipeline; epilogue checks the (complete by construction) last events
  SCHED_PIPELINE_SYNC = 2,  // five-stream pipeline; epilogue synchronises the three feeder streams (round-2 behaviour)
  SCHED_ONE_STREAM_SPIN = 3,// one stream; the thread waits inside the HIP runtime (which spins: one host core per proof)
  SCHED_COUNT = 4
};

struct TunePolicy {
  // ---- per proof
  int32_t sched = SCHED_AUTO;
  int32_t wait_spin = 0;          // 1: every wait of a proof inside the HIP runtime (hipEventSynchronize)
  int32_t wait_adapt = 0;         // 1: sleep through half of the shortest recent drain before polling
  int32_t stream_prio = 1;        // feeder streams (witness map, sort, reduction) at the higher stream priority
  int32_t trace_host = 0;         // host wall-clock phases on stderr
  int32_t sched_explore = 3;      // samples per candidate before SCHED_AUTO latches (0: static default, no exploration)
  int32_t batch_tails = 1;        // one-stream proofs: merge / reduce / combine of the four G1 MSMs as ONE launch each
  int32_t side_g2_tails = 1;      // a one-stream proof ALONE on the device: its G2 tails on a second stream, under the G1 accumulations
  int32_t side_wm = 1;            // ... and its witness map + the sort of h on a second stream, beside the sort of z and the first four accumulations
  int32_t side_g1_tails = 0;      // a one-stream proof ALONE on the device: the tails of A, B1, L' as a batch of three on the second stream, under the
                                  // H accumulation; only H's own tails at the end of the proof (round 6 re-run of round 5's run S on the 28-bit tails)
  int32_t side_h_tails = 1;       // multi-stream schedules: the tails of the last MSM (H) on the sort stream instead of behind the tails of L'
  int32_t dwm_loopback = 0;       // DIAGNOSTIC (timing only, wrong proofs): ark355_prove_shard runs the distributed witness map of its
                                  // rank with the exchanges as local copies -- the per-rank cost of a G-GPU proof on one GPU
  int32_t rccl_self = 0;          // DIAGNOSTIC / TEST (read when a key shard is loaded and per proof): at world size 1 a sharded proof runs
                                  // the distributed witness map and the bucket ring with every exchange as a grouped ncclSend / ncclRecv to
                                  // the rank ITSELF -- the point-to-point calls of an 8-GPU proof on the one GPU a test box has; same proof bytes
  // ---- per key load
  int32_t msm_c = 0;              // window size of resident tables (0: planner)
  int32_t msm_c_h = 0;            // window size of the h_query table alone (0: same rule as the others)
  int32_t pack_rows = -1;         // rows of the 28-bit tables: 1 bit-packed, 0 one word per limb, -1 (default) see table_pack_default (msm_impl.cuh)
  int32_t table_stride = 0;       // 0: planner
  int64_t hbm_budget_mb = 0;      // 0: 80 % of the device
  int32_t shard_dist_wm = 1;      // key shards: h_query in the layout of the distributed witness map when the world size allows it
  // ---- per call
  int32_t msm_seg = 0;            // entries per accumulation lane (0: msm_seg_len)
  int32_t acc_threads = 0;        // workgroup size of the accumulation kernels without LDS: 64 / 128 / 256; 0: the call's own choice (a proof alone
                                  // on one stream: 64, everything else: 256; msm_accumulate_phase)
  int64_t msm_two_level_min = -1; // bucket count from which the two-level reduction runs (-1: ARK_MSM_TWO_LEVEL_MIN)
  int32_t ntt_rmax = 0;           // 0: NTT_RMAX_LOG
  int32_t ntt_direct_max = -1;    // -1: NTT_DIRECT_MAX_LOG
  int32_t ntt_nofuse = 0;
  int32_t check_satisfied = 0;    // 1: ark355_prove / _dev / _batch also check a_i b_i == c_i on the rows the witness map has just computed
                                  // (one elementwise kernel, the verdict travels with the proof's last copy) and return
                                  // ARK355_E_UNSATISFIABLE instead of a proof that cannot verify; 0 (default): prove whatever z is, as the
                                  // reference's release build does

  struct Field {
    const char* name;
    int kind;                     // 0: int32, 1: int64
    size_t off;
  };
  static const Field* fields(int* count);

  // 0 on success, -1 for an unknown name
  int set(const char* name, int64_t v);
  int get(const char* name, int64_t* v) const;
  static TunePolicy from_env();
};

#define ARK_POLICY_FIELD32(n, f) {n, 0, offsetof(TunePolicy, f)}
#define ARK_POLICY_FIELD64(n, f) {n, 1, offsetof(TunePolicy, f)}

inline const TunePolicy::Field* TunePolicy::fields(int* count) {
  static const Field tab[] = {
      ARK_POLICY_FIELD32("SCHED", sched),
      ARK_POLICY_FIELD32("WAIT_SPIN", wait_spin),
      ARK_POLICY_FIELD32("WAIT_ADAPT", wait_adapt),
      ARK_POLICY_FIELD32("STREAM_PRIO", stream_prio),
      ARK_POLICY_FIELD32("TRACE_HOST", trace_host),
      ARK_POLICY_FIELD32("SCHED_EXPLORE", sched_explore),
      ARK_POLICY_FIELD32("BATCH_TAILS", batch_tails),
      ARK_POLICY_FIELD32("SIDE_G2_TAILS", side_g2_tails),
      ARK_POLICY_FIELD32("SIDE_WM", side_wm),
      ARK_POLICY_FIELD32("SIDE_H_TAILS", side_h_tails),
      ARK_POLICY_FIELD32("SIDE_G1_TAILS", side_g1_tails),
      ARK_POLICY_FIELD32("MSM_C", msm_c),
      ARK_POLICY_FIELD32("MSM_C_H", msm_c_h),
      ARK_POLICY_FIELD32("PACK_ROWS", pack_rows),
      ARK_POLICY_FIELD32("TABLE_STRIDE", table_stride),
      ARK_POLICY_FIELD64("HBM_BUDGET_MB", hbm_budget_mb),
      ARK_POLICY_FIELD32("SHARD_DIST_WM", shard_dist_wm),
      ARK_POLICY_FIELD32("DWM_LOOPBACK", dwm_loopback),
      ARK_POLICY_FIELD32("RCCL_SELF", rccl_self),
      ARK_POLICY_FIELD32("MSM_SEG", msm_seg),
      ARK_POLICY_FIELD32("ACC_THREADS", acc_threads),
      ARK_POLICY_FIELD64("MSM_TWO_LEVEL_MIN", msm_two_level_min),
      ARK_POLICY_FIELD32("NTT_RMAX", ntt_rmax),
      ARK_POLICY_FIELD32("NTT_DIRECT_MAX", ntt_direct_max),
      ARK_POLICY_FIELD32("CHECK_SATISFIED", check_satisfied),
      ARK_POLICY_FIELD32("NTT_NOFUSE", ntt_nofuse),
  };
  *count = (int)(sizeof(tab) / sizeof(tab[0]));
  return tab;
}

inline int TunePolicy::set(const char* name, int64_t v) {
  if (!name) return -1;
  // legacy spellings of the schedule
  if (strcmp(name, "SERIAL") == 0) {
    sched = v < 0 ? SCHED_AUTO : (v ? SCHED_ONE_STREAM : SCHED_PIPELINE);
    return 0;
  }
  if (strcmp(name, "EPILOGUE_SYNC") == 0) {
    if (sched == SCHED_PIPELINE || sched == SCHED_PIPELINE_SYNC) sched = v ? SCHED_PIPELINE_SYNC : SCHED_PIPELINE;
    return 0;
  }
  int n = 0;
  const Field* f = fields(&n);
  for (int i = 0; i < n; i++) {
    if (strcmp(name, f[i].name) != 0) continue;
    char* base = reinterpret_cast<char*>(this) + f[i].off;
    if (f[i].kind == 0) *reinterpret_cast<int32_t*>(base) = (int32_t)v;
    else *reinterpret_cast<int64_t*>(base) = v;
    return 0;
  }
  return -1;
}

inline int TunePolicy::get(const char* name, int64_t* v) const {
  if (!name || !v) return -1;
  int n = 0;
  const Field* f = fields(&n);
  for (int i = 0; i < n; i++) {
    if (strcmp(name, f[i].name) != 0) continue;
    const char* base = reinterpret_cast<const char*>(this) + f[i].off;
    *v = f[i].kind == 0 ? (int64_t)*reinterpret_cast<const int32_t*>(base) : *reinterpret_cast<const int64_t*>(base);
    return 0;
  }
  return -1;
}

inline TunePolicy TunePolicy::from_env() {
  TunePolicy p;
  int n = 0;
  const Field* f = fields(&n);
  char var[64];
  for (int i = 0; i < n; i++) {
    snprintf(var, sizeof(var), "ARK355_%s", f[i].name);
    if (const char* e = getenv(var)) {
      if (e[0]) p.set(f[i].name, strtoll(e, nullptr, 10));
    }
  }
  // legacy spellings (rounds 1-3): ARK355_SERIAL=1|0, ARK355_EPILOGUE_SYNC=0|1
  if (const char* e = getenv("ARK355_SERIAL")) {
    if (e[0]) p.set("SERIAL", e[0] == '1' ? 1 : 0);
  }
  if (const char* e = getenv("ARK355_EPILOGUE_SYNC")) {
    if (e[0]) p.set("EPILOGUE_SYNC", e[0] == '1' ? 1 : 0);
  }
  return p;
}

}  // namespace ark355
