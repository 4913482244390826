// Diagnostics behind ark355_diag_streams: which HIP streams of this library share an in-order hardware queue.
//
// The HIP runtime maps its streams onto a few hardware queues per priority class (GPU_MAX_HW_QUEUES, default 4),
// handing them out in creation order.  Two streams on ONE queue are serialised whatever their events say: a proof on
// one of them stalls the other proof's ready kernels.  The mapping is not exposed by the API, so it is measured: a
// kernel that spins for `spin_us` on stream A, a trivial kernel on stream B behind it; when B's kernel completes while
// A's is still spinning the two streams sit on different queues.  Run it on an idle device.
#pragma once
#include "common.h"

namespace ark355 {

#if !defined(ARK_EMUL)
static __global__ void diag_spin_kernel(uint64_t ticks) {       // wall_clock64: constant 100 MHz counter
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static __global__ void diag_nop_kernel(uint32_t* sink) {
  if (sink) *sink = 1;
}
#endif

// 1: a kernel on `b` waited for the spinning kernel on `a` (same hardware queue), 0: it did not, -1: not measurable here
static inline int diag_streams_serialised(hipStream_t a, hipStream_t b, uint32_t spin_us = 1500) {
#if defined(ARK_EMUL)
  (void)a; (void)b; (void)spin_us;
  return -1;
#else
  if (a == b) return 1;
  hipEvent_t ea = nullptr, eb = nullptr;
  ARK_CHECK_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
  ARK_CHECK_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  int res = -1;
  try {
    ARK_CHECK_HIP(hipStreamSynchronize(a));
    ARK_CHECK_HIP(hipStreamSynchronize(b));
    ARK_LAUNCH(diag_spin_kernel, dim3(1), dim3(64), 0, a, (uint64_t)spin_us * 100ull);
    ARK_CHECK_LAUNCH();
    ARK_CHECK_HIP(hipEventRecord(ea, a));
    ARK_LAUNCH(diag_nop_kernel, dim3(1), dim3(1), 0, b, (uint32_t*)nullptr);
    ARK_CHECK_LAUNCH();
    ARK_CHECK_HIP(hipEventRecord(eb, b));
    ARK_CHECK_HIP(hipEventSynchronize(eb));
    const hipError_t q = hipEventQuery(ea);          // still spinning => b overtook a => different queues
    res = q == hipErrorNotReady ? 0 : 1;
    ARK_CHECK_HIP(hipEventSynchronize(ea));
  } catch (...) {
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    throw;
  }
  (void)hipEventDestroy(ea);
  (void)hipEventDestroy(eb);
  return res;
#endif
}


// GPU-side cost of one dispatch in an in-order stream: `launches` kernels that each spin for `spin_us`, back to back on
// `st`; (elapsed between the first and the last event) / launches - spin_us.  A few microseconds on most boxes of the
// pool this was developed on, 50-90 us on some (round 4: the same library proves 30 % slower there, with every kernel as
// fast as elsewhere) -- the one-number fingerprint bench.py prints so that a slow BOX can be told from a slow TREE.
static inline float diag_dispatch_gap_us(hipStream_t st, uint32_t launches = 200, uint32_t spin_us = 20) {
#if defined(ARK_EMUL)
  (void)st; (void)launches; (void)spin_us;
  return -1.f;
#else
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ARK_CHECK_HIP(hipEventCreate(&e0));
  ARK_CHECK_HIP(hipEventCreate(&e1));
  float ms = 0.f;
  try {
    for (int warm = 0; warm < 2; warm++) {
      ARK_CHECK_HIP(hipEventRecord(e0, st));
      for (uint32_t i = 0; i < launches; i++) {
        ARK_LAUNCH(diag_spin_kernel, dim3(1), dim3(64), 0, st, (uint64_t)spin_us * 100ull);
        ARK_CHECK_LAUNCH();
      }
      ARK_CHECK_HIP(hipEventRecord(e1, st));
      ARK_CHECK_HIP(hipEventSynchronize(e1));
      ARK_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    }
  } catch (...) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    throw;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms * 1000.f / (float)launches - (float)spin_us;
#endif
}

// The chip's sustained v_mad_u64_u32 issue rate on THIS box, NOW: the instruction the bucket-accumulation kernels are made of
// (field28.cuh; 71 % of their instructions), at their occupancy (two waves per SIMD), eight independent chains per lane so that
// the multiplier's latency is covered.  The rate follows the gfx clock the box sustains under its power cap (2.1-2.3 GHz on the
// pool this was developed on: a 4-5 % spread in every proof time), so bench.py prices `roofline.alu` against THIS number instead
// of a constant read off another box, and reports its times in units of it.  Two launches: a short one to size the second,
// which runs for about target_ms.  Returns T (10^12) multiply-adds per second; < 0: not measurable on this build.
#if !defined(ARK_EMUL)
static __global__ void __launch_bounds__(256, 2) diag_mad_kernel(uint64_t* sink, uint32_t a0, uint32_t iters) {
  const uint32_t a = a0 + threadIdx.x, b = (a0 * 2654435761u) ^ (blockIdx.x * 256u + threadIdx.x);
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (uint64_t)(i * 77 + a) << 7;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i];
  if (s == 0x5a5a5a5a5a5a5a5aull) *sink = s;        // (never true in practice: keeps the chains alive)
}
#endif
static inline float diag_mad_rate_t(hipStream_t st, int device, float target_ms, float* elapsed_ms) {
#if defined(ARK_EMUL)
  (void)st; (void)device; (void)target_ms;
  if (elapsed_ms) *elapsed_ms = 0.f;
  return -1.f;
#else
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
  const uint32_t grid = (uint32_t)cus * 2u;          // two workgroups of four waves per CU: two waves per SIMD
  DevBuf sink(8);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ARK_CHECK_HIP(hipEventCreate(&e0));
  ARK_CHECK_HIP(hipEventCreate(&e1));
  float ms = 0.f;
  uint32_t iters = 4000;
  try {
    // (an untimed first launch: the one-off cost of loading the kernel must not end up in the sizing pass -- it did in the first
    // version, the measuring launch came out ten times too short and read 7 % low)
    ARK_LAUNCH(diag_mad_kernel, dim3(grid), dim3(256), 0, st, sink.as<uint64_t>(), 1u, 64u);
    ARK_CHECK_LAUNCH();
    for (int pass = 0; pass < 2; pass++) {
      ARK_CHECK_HIP(hipEventRecord(e0, st));
      ARK_LAUNCH(diag_mad_kernel, dim3(grid), dim3(256), 0, st, sink.as<uint64_t>(), 12345u + (uint32_t)pass, iters);
      ARK_CHECK_LAUNCH();
      ARK_CHECK_HIP(hipEventRecord(e1, st));
      ARK_CHECK_HIP(hipEventSynchronize(e1));
      ARK_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (pass == 0) {
        const double scale = (ms > 1e-3 ? (double)target_ms / ms : 1.0);
        double it2 = (double)iters * scale;
        if (it2 < 1000.0) it2 = 1000.0;
        if (it2 > 4.0e7) it2 = 4.0e7;
        iters = (uint32_t)it2;
      }
    }
  } catch (...) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    throw;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (elapsed_ms) *elapsed_ms = ms;
  const double mads = (double)grid * 256.0 * (double)iters * 32.0;
  return ms > 0.f ? (float)(mads / ((double)ms * 1e-3) / 1e12) : -1.f;
#endif
}

// The two free-running counters of the GPU, read by one lane at the same moment: s_memtime counts SHADER-CLOCK cycles (it slows
// down and speeds up with the gfx clock the power management grants), s_memrealtime a constant 100 MHz reference.  Two readings
// bracket a region: (shader cycles elapsed) / (work done) is a time in units that do not depend on which clock THIS box sustained
// under its power cap, and (shader cycles) / (reference ticks) x 100 MHz is the mean gfx clock of the region, measured on the
// chip itself rather than sampled over SMI.  out[0] = shader cycles, out[1] = 100 MHz ticks; zeros on builds that cannot tell.
// The shader-cycle counter is NOT one counter: readings taken on different XCDs (and, as run G showed, not only XCDs) differ by
// arbitrary offsets, so two readings are comparable only when they come from the same place.  The probe therefore covers the
// chip (2 048 one-wave workgroups) and records one (cycles, ticks) pair PER COMPUTE UNIT, keyed by (XCC_ID, HW_ID.se / sh / cu);
// a caller brackets a region with two probes and takes, per compute unit present in both, the ratio of the two deltas -- the
// median over the compute units is the region's mean gfx clock (diag_clocks_delta).
constexpr uint32_t DIAG_CLOCK_SLOTS = 8 * 128;       // 8 XCCs x (se, sh, cu) of HW_REG_HW_ID packed into 7 bits
#if !defined(ARK_EMUL)
static __global__ void diag_clocks_kernel(uint64_t* out) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3u << 11) | 20u) & 7u;        // HW_REG_XCC_ID (20), bits [3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((15u << 11) | (0u << 6) | 4u);   // HW_REG_HW_ID (4), bits [15:0]: cu_id [11:8], sh_id [12], se_id [15:13]
    const uint32_t slot = xcc * 128u + ((hw >> 8) & 127u);
    const uint64_t a = __builtin_readcyclecounter();        // s_memtime
    const uint64_t b = wall_clock64();                      // s_memrealtime
    out[2 * slot] = a;
    out[2 * slot + 1] = b;
  }
}
#endif
// out: DIAG_CLOCK_SLOTS pairs (zeros where no workgroup landed)
static inline void diag_clocks(hipStream_t st, uint64_t* out) {
  memset(out, 0, (size_t)DIAG_CLOCK_SLOTS * 16);
#if defined(ARK_EMUL)
  (void)st;
#else
  DevBuf d((size_t)DIAG_CLOCK_SLOTS * 16);
  ARK_CHECK_HIP(hipMemsetAsync(d.p, 0, (size_t)DIAG_CLOCK_SLOTS * 16, st));
  ARK_LAUNCH(diag_clocks_kernel, dim3(2048), dim3(64), 0, st, d.as<uint64_t>());
  ARK_CHECK_LAUNCH();
  ARK_CHECK_HIP(hipMemcpyAsync(out, d.p, (size_t)DIAG_CLOCK_SLOTS * 16, hipMemcpyDeviceToHost, st));
  ARK_CHECK_HIP(hipStreamSynchronize(st));
#endif
}

// Streams for one-stream proofs ("lanes"), one set per device: created back to back at the first request and then
// PROBED so that the set handed out sits on pairwise different hardware queues (diag_streams_serialised).  Round 4 found
// two of bench.py's four context streams on one queue on a fresh box -- two of the four proofs "in flight" took turns --
// because the runtime assigns queues by creation order across everything the process creates (torch's streams, the null
// stream, ...).  Contexts take lanes round-robin; more contexts than lanes share (that is what a hardware queue would
// make them do anyway).  The probe is timing-based and wants an idle device: when it cannot establish a FULL set (another
// process on the GPU, work of the host application in flight, a slow first launch), the pool stays EMPTY and every
// context keeps its own stream -- a short pool would funnel all proofs "in flight" into one or two in-order streams.
struct LanePool {
  std::mutex mu;
  std::vector<hipStream_t> lanes;        // pairwise on different hardware queues (as far as that could be established)
  std::vector<hipStream_t> parked;       // created, found to collide, kept alive so that the queue stays "taken"
  uint32_t next = 0;
  bool built = false;
  static LanePool& of(int device) {
    static LanePool p[64];
    return p[(unsigned)device & 63u];
  }
  // call with the device idle (ark355_ctx_create of the first context)
  void build(uint32_t want = 4, uint32_t max_create = 12) {
    std::lock_guard<std::mutex> lk(mu);
    if (built) return;
    built = true;
#if defined(ARK_EMUL)
    (void)want; (void)max_create;
#else
    bool failed = false;
    for (uint32_t made = 0; made < max_create && lanes.size() < want && !failed; made++) {
      hipStream_t st = nullptr;
      if (hipStreamCreate(&st) != hipSuccess) break;
      bool clash = false;
      try {
        for (hipStream_t other : lanes) {
          if (diag_streams_serialised(other, st, 400) == 1 || diag_streams_serialised(st, other, 400) == 1) {
            clash = true;
            break;
          }
        }
      } catch (...) {
        failed = true;       // the probe itself failed: no pool
        clash = true;
      }
      (clash ? parked : lanes).push_back(st);
    }
    if (failed || lanes.size() < want) {   // no full set: no pool (see above)
      for (hipStream_t st : lanes) parked.push_back(st);
      lanes.clear();
    }
#endif
  }
  // nullptr: no pool on this build / device (the caller falls back to the context's own stream)
  hipStream_t acquire() {
    std::lock_guard<std::mutex> lk(mu);
    if (lanes.empty()) return nullptr;
    return lanes[next++ % lanes.size()];
  }
  size_t size() {
    std::lock_guard<std::mutex> lk(mu);
    return lanes.size();
  }
};

}  // namespace ark355
