// TEST INFRASTRUCTURE ONLY -- a tiny single-threaded emulator for the subset of HIP the library
// uses, so that the *kernel and host-orchestration logic* (indexing, scans, barriers, shuffles,
// buffer sizes) can be exercised by `pytest -m "not gpu"` in a container with no GPU.
//
// It is never built into, linked by, or loaded from the product: `snark_amd` loads only
// libark355.so (hipcc, gfx950) and fails loudly without it.  tests/emul/build_emul.py compiles the
// same sources with g++ -DARK_EMUL into tests/emul/libark355_emul.so, which only tests open.
//
// Model: blocks run one after another; the threads of a block are coroutines (a register-only x86-64 context switch: glibc's
// swapcontext makes a sigprocmask system call per switch, a third of the CPU tier's time until round 6); a wave is
// 64 consecutive threads.  __syncthreads / wave shuffles yield to a scheduler that releases a
// barrier once every live thread of the block / wave has arrived.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <functional>
#include <vector>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct alignas(8) uint2 {
  unsigned x, y;
};
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

namespace emu {
enum State { RUN = 0, WAVE_WAIT = 1, BLOCK_WAIT = 2, DONE = 3 };
struct Thread {
  void* sp;          // saved stack pointer of the coroutine (callee-saved registers are on its stack)
  int state;
  dim3 tid;
};
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern Thread* g_cur;
extern void* g_sched_sp;
extern uint64_t g_xchg[64];
extern unsigned char* g_dyn_smem;
// mailbox of the lane-pair exchange (models a DPP quad_perm [1,0,3,2] move): 2-deep ring indexed by sequence parity
struct PairBox {
  uint32_t val[2] = {0, 0};
  uint64_t seq = 0;
};
extern PairBox g_pairbox[1024];
void yield(int st);
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

static inline void __syncthreads() { emu::yield(emu::BLOCK_WAIT); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T>
static inline T __emu_xchg(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle of <= 8 bytes");
  int lane = (int)(emu::g_threadIdx.x & 63);
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  emu::g_xchg[lane] = bits;
  emu::yield(emu::WAVE_WAIT);
  uint64_t got = emu::g_xchg[src_lane & 63];
  emu::yield(emu::WAVE_WAIT);
  T r;
  memcpy(&r, &got, sizeof(T));
  return r;
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  int lane = (int)(emu::g_threadIdx.x & 63);
  int base = lane & ~(width - 1);
  return __emu_xchg(v, base + (src & (width - 1)));
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = (int)(emu::g_threadIdx.x & 63);
  (void)width;
  return __emu_xchg(v, lane ^ mask);
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = (int)(emu::g_threadIdx.x & 63);
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return __emu_xchg(v, src);
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = (int)(emu::g_threadIdx.x & 63);
  int src = lane - (int)d;
  if (src < (lane & ~(width - 1))) src = lane;
  return __emu_xchg(v, src);
}
// value of lane (lane ^ 1); unlike the wave shuffles this pairs exactly the two lanes of a pair, whatever the
// other lanes of the wave are doing (on hardware: v_mov_b32_dpp quad_perm:[1,0,3,2])
static inline uint32_t __emu_pair_xchg(uint32_t v) {
  const unsigned me = emu::g_threadIdx.x & 1023u, other = me ^ 1u;
  emu::PairBox& mine = emu::g_pairbox[me];
  emu::PairBox& theirs = emu::g_pairbox[other];
  const uint64_t k = mine.seq + 1;
  mine.val[k & 1] = v;
  mine.seq = k;
  while (theirs.seq < k) emu::yield(emu::RUN);
  return theirs.val[k & 1];
}

static inline unsigned long long __ballot(int pred) {
  int lane = (int)(emu::g_threadIdx.x & 63);
  // lanes that exited or are not participating contribute 0: clear, barrier, set, barrier, read
  emu::g_xchg[lane] = 0;
  emu::yield(emu::WAVE_WAIT);
  emu::g_xchg[lane] = pred ? 1 : 0;
  emu::yield(emu::WAVE_WAIT);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) m |= (unsigned long long)(emu::g_xchg[i] & 1) << i;
  emu::yield(emu::WAVE_WAIT);
  return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

template <class T>
static inline T atomicAdd(T* p, T v) {
  T old = *p;
  *p = old + v;
  return old;
}
template <class T>
static inline T atomicMax(T* p, T v) {
  T old = *p;
  if (v > old) *p = v;
  return old;
}
template <class T>
static inline T atomicMin(T* p, T v) {
  T old = *p;
  if (v < old) *p = v;
  return old;
}
template <class T>
static inline T atomicOr(T* p, T v) {
  T old = *p;
  *p = old | v;
  return old;
}
template <class T>
static inline T atomicExch(T* p, T v) {
  T old = *p;
  *p = v;
  return old;
}
template <class T>
static inline T atomicCAS(T* p, T cmp, T v) {
  T old = *p;
  if (old == cmp) *p = v;
  return old;
}

// ---- runtime API subset -----------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct EmuEvent {
  std::chrono::steady_clock::time_point t;
};
typedef EmuEvent* hipEvent_t;
#define hipSuccess 0
#define hipErrorOutOfMemory 2
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) {
  return hipMalloc((void**)p, n);
}
static inline hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) {
  memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
  memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) {
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) {
  *s = nullptr;
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
static inline hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new EmuEvent();
  return hipSuccess;
}
constexpr unsigned hipEventBlockingSync = 1u;
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
#ifndef hipErrorNotReady
#define hipErrorNotReady ((hipError_t)600)
#endif
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
