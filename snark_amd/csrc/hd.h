// Host/device qualifiers + the kernel-launch macro shared by every source of libark355.
//
// Product builds are hipcc for gfx950 only.  ARK_EMUL is a *test-only* configuration
// (tests/emul/, g++): it swaps the HIP runtime for a single-threaded emulator so that kernel and
// orchestration logic can be exercised on a machine without a GPU; it is never shipped or loaded by
// the snark_amd package.
#pragma once
#if defined(ARK_EMUL)
#include "hip_emul.h"
#define ARK_HD inline
#define ARK_HD_NOINLINE __attribute__((noinline))
#define ARK_D inline
#define ARK_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
#define ARK_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(emu::g_dyn_smem)
// value held by the other lane of this lane's pair (lane ^ 1)
static inline uint32_t ark_pair_xchg(uint32_t v) { return __emu_pair_xchg(v); }
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ARK_HD __host__ __device__ __forceinline__
#define ARK_HD_NOINLINE __host__ __device__ __noinline__
#define ARK_D __device__ __forceinline__
#define ARK_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (stream), __VA_ARGS__)
#define ARK_DYN_SMEM(T, name)                                  \
  extern __shared__ __align__(16) unsigned char _ark_smem[];   \
  T* name = reinterpret_cast<T*>(_ark_smem)
// value held by the other lane of this lane's pair (lane ^ 1): one v_mov_b32_dpp quad_perm:[1,0,3,2]
__device__ __forceinline__ uint32_t ark_pair_xchg(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // bound_ctrl: no lane of this pattern reads out of bounds, and with it hipcc needs no v_mov to initialise the destination
#else
  return v;   // never executed on the host pass
#endif
}
#else
// plain host translation unit (no kernels): arithmetic headers only
#define ARK_HD inline
#define ARK_HD_NOINLINE __attribute__((noinline))
#define ARK_D inline
#define ARK_PLAIN_HOST 1      // no lanes here: lane-pair types (Fp2L) are not declared
#endif
