// Measured choice of the per-proof schedule (policy SCHED = AUTO): which of the schedules of prove_run (groth16_impl.cuh) a class
// of proofs keeps.  Plain host C++ (no HIP): the logic has its own unit test, tests/cpp/test_sched_tuner.cpp.
#pragma once
#include <stdint.h>
#include <chrono>
#include <map>
#include <mutex>
#include "policy.h"

namespace ark355 {

// ---- measured choice of the per-proof schedule (policy SCHED = AUTO) -------------------------------------------------------
// Rounds 1-3 picked the schedule from thresholds fitted on two or three boxes (one stream when other proofs are in
// flight, the pipeline for a proof alone; epilogue synchronises for everything but large BLS12-381 proofs) -- and the
// round-3 driver box ran 30-40 % slower than any box those thresholds were fitted on, in exactly the phases where
// streams overlap.  So the library measures, per (device, proof shape, alone | in flight) class:
//   * a proof ALONE: the first warm proofs run the candidate schedules in turn (policy SCHED_EXPLORE samples each,
//     default 3) and the class keeps the one with the smallest mean wall time inside prove_run -- nothing else runs, so a
//     proof's own latency is the objective;
//   * proofs IN FLIGHT: latency of one proof is the WRONG objective while the candidates are mixed (round-4 run A: a
//     pipeline proof, whose feeder streams have the higher priority, finished sooner at the expense of the one-stream
//     proofs running beside it; the tuner latched the schedule that the bench's own throughput A/B then showed to be 4 %
//     slower).  The class therefore explores in PHASES: every proof started during a phase runs the phase's candidate, the
//     first SKIP completions of a phase are ignored (proofs of the previous phase are still draining), and the phase's
//     score is the mean wall time of the next 4 * SCHED_EXPLORE + 4 proofs -- in a HOMOGENEOUS phase throughput is the
//     number of proofs in flight over exactly that.  The static default (one stream) runs the first AND the last phase (its better one counts) and is
//     only abandoned for a candidate that beats it by 5 %.
// A context's first proof of a shape (allocations, table builds) is neither explored nor recorded.  The spinning wait is
// no candidate of the automatic choice (it costs a host core per proof in flight); policy SCHED = 3 forces it.
struct SchedTuner {
  using Clock = std::chrono::steady_clock;
  static constexpr uint32_t PHASE_SKIP = 8;
  struct Entry {
    int latched = -1;
    uint32_t started[SCHED_COUNT] = {}, done[SCHED_COUNT] = {};
    double sum_ms[SCHED_COUNT] = {};      // alone: summed wall time; in flight: span of the phase's scored completions
    uint32_t ncand = 0;
    int cand[SCHED_COUNT] = {};
    // in-flight classes
    uint32_t phase = 0, phase_done = 0;
    double phase_sum = 0.0;
  };
  std::mutex mu;
  std::map<uint64_t, Entry> entries;
  static SchedTuner& of(int device) {
    static SchedTuner t[64];
    return t[(unsigned)device & 63u];
  }
  static uint64_t key(uint64_t shape, bool concurrent) { return (shape << 1) | (concurrent ? 1u : 0u); }
  static uint32_t phase_len(int explore_n) { return 4u * (uint32_t)explore_n + 4u; }
  // the schedule for a proof of class `k`; *explore = this proof is a sample and must be reported
  int pick(uint64_t k, bool concurrent, int explore_n, int fallback, bool* explore) {
    *explore = false;
    if (explore_n <= 0) return fallback;
    std::lock_guard<std::mutex> lk(mu);
    Entry& e = entries[k];
    if (e.latched >= 0) return e.latched;
    if (e.ncand == 0) {
      if (concurrent) {
        // the static default first AND last: its score is the better of its two phases, so that whatever disturbs the
        // first phase of a process (run C of round 4: 26.3 ms per completion in phase one, 23.1 ms in the bench's own A/B a
        // second later) cannot hand the class to another schedule
        const int c[] = {SCHED_ONE_STREAM, SCHED_PIPELINE, SCHED_PIPELINE_SYNC, SCHED_ONE_STREAM};
        for (int v : c) e.cand[e.ncand++] = v;
      } else {
        const int c[] = {SCHED_ONE_STREAM, SCHED_PIPELINE, SCHED_PIPELINE_SYNC};
        for (int v : c) e.cand[e.ncand++] = v;
      }
    }
    if (concurrent) {
      const int v = e.cand[e.phase < e.ncand ? e.phase : 0];
      e.started[v]++;
      *explore = true;
      return v;
    }
    int best = -1;
    for (uint32_t i = 0; i < e.ncand; i++) {
      const int v = e.cand[i];
      if (e.started[v] >= (uint32_t)explore_n) continue;
      if (best < 0 || e.started[v] < e.started[best]) best = v;
    }
    if (best < 0) return fallback;          // every sample is under way: wait for the reports
    e.started[best]++;
    *explore = true;
    return best;
  }
  void report(uint64_t k, bool concurrent, int variant, double ms, int explore_n, int fallback) {
    std::lock_guard<std::mutex> lk(mu);
    Entry& e = entries[k];
    if (e.latched >= 0 || variant < 0 || variant >= SCHED_COUNT || e.ncand == 0) return;
    if (concurrent) {
      if (e.phase >= e.ncand || variant != e.cand[e.phase]) return;       // a straggler of an earlier phase
      const uint32_t len = phase_len(explore_n);
      e.phase_done++;
      if (e.phase_done <= PHASE_SKIP) return;
      // The score of a phase is the summed (wall time / proofs in flight) of its scored proofs (prove_run divides).  Inside a
      // phase every proof in flight runs the same schedule, and with threads that start their next proof as soon as one
      // returns, throughput = proofs in flight / mean latency -- so this orders the schedules exactly as throughput does.  (The first
      // version timed the span between the 8th and the 24th completion: proofs in flight complete in lockstep waves of four,
      // and where the window's ends fell inside a wave moved the reading by +-6 % -- run D latched a schedule on a 21.7 ms
      // phase whose timed region then ran at 23.7 ms.)
      e.phase_sum += ms;
      if (e.phase_done < PHASE_SKIP + len) return;
      if (e.done[variant] == 0 || e.phase_sum < e.sum_ms[variant]) e.sum_ms[variant] = e.phase_sum;      // a schedule's best phase counts
      e.done[variant] = len;
      e.phase++;
      e.phase_done = 0;
      e.phase_sum = 0.0;
      if (e.phase < e.ncand) return;
      int best = -1;
      for (uint32_t i = 0; i < e.ncand; i++)
        if (best < 0 || e.sum_ms[e.cand[i]] < e.sum_ms[best]) best = e.cand[i];
      // the static default stays unless a candidate is clearly better
      if (fallback >= 0 && fallback < SCHED_COUNT && e.done[fallback] && e.sum_ms[best] > 0.95 * e.sum_ms[fallback]) best = fallback;
      e.latched = best;
      return;
    }
    e.done[variant]++;
    e.sum_ms[variant] += ms;
    int best = -1;
    for (uint32_t i = 0; i < e.ncand; i++) {
      const int v = e.cand[i];
      if (e.done[v] < (uint32_t)explore_n) return;
      if (best < 0 || e.sum_ms[v] / e.done[v] < e.sum_ms[best] / e.done[best]) best = v;
    }
    // three samples per candidate separate schedules that differ by several per cent, not by one: the static default stays
    // unless another candidate is clearly better
    if (fallback >= 0 && fallback < SCHED_COUNT && e.done[fallback] &&
        e.sum_ms[best] / e.done[best] > 0.95 * e.sum_ms[fallback] / e.done[fallback])
      best = fallback;
    e.latched = best;
  }
  // an "alone" sample during which another proof started on the device says nothing about the schedule: take it back
  void unstart(uint64_t k, int variant) {
    std::lock_guard<std::mutex> lk(mu);
    Entry& e = entries[k];
    if (e.latched < 0 && variant >= 0 && variant < SCHED_COUNT && e.started[variant] > e.done[variant]) e.started[variant]--;
  }
  // forget what was measured (ark355_sched_reset: tests, benches that change the load pattern)
  void reset() {
    std::lock_guard<std::mutex> lk(mu);
    entries.clear();
  }
  // mean_ms: alone = mean wall time of a proof; in flight = mean of (wall time / proofs sharing the device) over the scored
  // proofs of the schedule's best phase, i.e. an estimate of the time per proof
  bool info(uint64_t k, int* latched, double mean_ms[SCHED_COUNT], uint32_t samples[SCHED_COUNT]) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = entries.find(k);
    if (it == entries.end()) return false;
    *latched = it->second.latched;
    for (int v = 0; v < SCHED_COUNT; v++) {
      samples[v] = it->second.done[v];
      mean_ms[v] = it->second.done[v] ? it->second.sum_ms[v] / it->second.done[v] : 0.0;
    }
    return true;
  }
};

}  // namespace ark355
