// Unit test of snark_amd/csrc/sched_tuner.h (host-only logic): the measured choice of the per-proof schedule.
// Four times in round 4 the tuner latched a schedule that the bench's own A/B then showed to be slower; every one of those
// readings is replayed here against the final scoring rules.  g++ -std=c++17 -I snark_amd/csrc tests/cpp/test_sched_tuner.cpp
#include <cassert>
#include <cstdio>
#include <vector>
#include "sched_tuner.h"

using namespace ark355;

static int failures = 0;
#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      failures++;                                                  \
    }                                                              \
  } while (0)

// one in-flight exploration: every started proof reports `per_proof_ms[variant]` (the caller has already divided the wall time
// by the proofs in flight); returns what the class latched
static int explore_in_flight(SchedTuner& t, uint64_t key, const double per_proof_ms[SCHED_COUNT], int explore_n, int fallback,
                             double first_phase_penalty = 0.0, int stragglers = 0) {
  int latched = -1;
  double mean[SCHED_COUNT];
  uint32_t samples[SCHED_COUNT];
  int phase_starts = 0, last_variant = -1;
  for (int i = 0; i < 400; i++) {
    bool explore = false;
    const int v = t.pick(key, /*concurrent=*/true, explore_n, fallback, &explore);
    if (!explore) {
      bool found = t.info(key, &latched, mean, samples);
      CHECK(found);
      CHECK(latched == v);
      return latched;
    }
    if (v != last_variant) {
      phase_starts++;
      // stragglers of the previous phase report AFTER the phase changed: they must be ignored
      for (int s = 0; s < stragglers && last_variant >= 0; s++) t.report(key, true, last_variant, 1.0, explore_n, fallback);
      last_variant = v;
    }
    const double ms = per_proof_ms[v] + (phase_starts == 1 ? first_phase_penalty : 0.0);
    t.report(key, true, v, ms, explore_n, fallback);
  }
  CHECK(!"the class never latched");
  return -1;
}

int main() {
  SchedTuner& t = SchedTuner::of(63);
  const int DEF = SCHED_ONE_STREAM;
  // (1) the usual box: one stream 22.4, pipeline 24.3, pipeline + sync 23.0 ms per proof -> the default stays
  {
    const double ms[SCHED_COUNT] = {22.4, 24.3, 23.0, 22.4};
    CHECK(explore_in_flight(t, SchedTuner::key(1, true), ms, 3, DEF) == SCHED_ONE_STREAM);
  }
  // (2) run C: the FIRST phase of the process was disturbed (26.3 against 23.1 a second later); the default runs the last phase
  //     too and its better phase counts
  {
    const double ms[SCHED_COUNT] = {23.1, 25.1, 25.8, 23.1};
    CHECK(explore_in_flight(t, SchedTuner::key(2, true), ms, 3, DEF, /*first_phase_penalty=*/3.2) == SCHED_ONE_STREAM);
  }
  // (3) a candidate that is 2 % better does not displace the default; one that is 10 % better does
  {
    const double close[SCHED_COUNT] = {22.0, 21.6, 23.0, 22.0};
    CHECK(explore_in_flight(t, SchedTuner::key(3, true), close, 3, DEF) == SCHED_ONE_STREAM);
    const double clear[SCHED_COUNT] = {33.0, 24.0, 29.0, 33.0};
    CHECK(explore_in_flight(t, SchedTuner::key(4, true), clear, 3, DEF) == SCHED_PIPELINE);
  }
  // (4) stragglers of the previous phase (reports with the old variant after the phase changed) are ignored, even absurd ones
  {
    const double ms[SCHED_COUNT] = {22.4, 24.3, 23.0, 22.4};
    CHECK(explore_in_flight(t, SchedTuner::key(5, true), ms, 3, DEF, 0.0, /*stragglers=*/5) == SCHED_ONE_STREAM);
    double mean[SCHED_COUNT];
    uint32_t samples[SCHED_COUNT];
    int latched = -1;
    CHECK(t.info(SchedTuner::key(5, true), &latched, mean, samples));
    CHECK(samples[SCHED_ONE_STREAM] == SchedTuner::phase_len(3) && mean[SCHED_PIPELINE] > 24.0 && mean[SCHED_PIPELINE] < 24.6);
  }
  // (5) the spinning wait is no candidate of the automatic choice
  {
    double mean[SCHED_COUNT];
    uint32_t samples[SCHED_COUNT];
    int latched = -1;
    CHECK(t.info(SchedTuner::key(1, true), &latched, mean, samples));
    CHECK(samples[SCHED_ONE_STREAM_SPIN] == 0);
  }
  // (6) a proof alone: three samples per candidate in turn; run F: pipeline 26.18 against one stream 26.45 ms (1 %) must NOT take
  //     the class (its latency reading afterwards was 27.4 ms); 10 % must
  {
    const uint64_t k = SchedTuner::key(6, false);
    const double ms[SCHED_COUNT] = {26.45, 26.18, 27.66, 0};
    for (int i = 0; i < 9; i++) {
      bool explore = false;
      const int v = t.pick(k, false, 3, DEF, &explore);
      CHECK(explore);
      t.report(k, false, v, ms[v], 3, DEF);
    }
    bool explore = true;
    CHECK(t.pick(k, false, 3, DEF, &explore) == SCHED_ONE_STREAM && !explore);
    const uint64_t k2 = SchedTuner::key(7, false);
    const double ms2[SCHED_COUNT] = {30.0, 26.0, 27.0, 0};
    for (int i = 0; i < 9; i++) {
      const int v = t.pick(k2, false, 3, DEF, &explore);
      t.report(k2, false, v, ms2[v], 3, DEF);
    }
    CHECK(t.pick(k2, false, 3, DEF, &explore) == SCHED_PIPELINE);
  }
  // (7) a polluted "alone" sample is taken back and measured again
  {
    const uint64_t k = SchedTuner::key(8, false);
    bool explore = false;
    const int v = t.pick(k, false, 1, DEF, &explore);
    t.unstart(k, v);
    const int v2 = t.pick(k, false, 1, DEF, &explore);
    CHECK(explore && v2 == v);
  }
  // (8) SCHED_EXPLORE = 0: the static default without measuring; reset() forgets
  {
    bool explore = true;
    CHECK(t.pick(SchedTuner::key(9, true), true, 0, SCHED_PIPELINE_SYNC, &explore) == SCHED_PIPELINE_SYNC && !explore);
    t.reset();
    double mean[SCHED_COUNT];
    uint32_t samples[SCHED_COUNT];
    int latched = -1;
    CHECK(!t.info(SchedTuner::key(1, true), &latched, mean, samples));
  }
  if (failures) return 1;
  printf("sched tuner: all checks passed\n");
  return 0;
}
