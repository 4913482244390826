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

}  // namespace ark355
