// Cross-GPU exchange of the sharded prover (BASELINE.json configs[2]) -- RCCL behind the C ABI.
//
// One process per GPU; every rank holds MSM terms [T*g/G, T*(g+1)/G) of each query vector (ark355_pk_load_shard) and
// the partial results combine by group addition.  RCCL has no elliptic-curve reduction operator, so the "all-reduce of
// partial sums" is built from the primitives it does have:
//   * window level (default): every rank finishes its five bucket reductions; ONE ncclAllGather of the five XYZZ
//     partial sums (960 B per rank for BLS12-381) on the reduction stream, straight from HBM; the O(G) additions and
//     the O(1) proof tail run on the host.  Latency-bound (a few microseconds over xGMI), not link-bound.
//   * bucket level (ARK355_SHARD_BUCKET_RING, the literal "all-reduce of partial bucket sums"): after the local
//     accumulation + merge, the bucket array of every MSM goes through a ring reduce-scatter -- G-1 steps of
//     ncclSend to the right neighbour / ncclRecv from the left one, grouped, each followed by an element-wise EC-add
//     kernel -- so that rank g owns the fully summed bucket range g+1 (mod G); it zeroes the other ranges, runs the
//     usual weighted bucket reduction over its range, and the all-gather above finishes.  Per proof and rank this moves
//     (G-1)/G of 5 bucket arrays (4 x 6.3 MB + 12.6 MB at c = 16, BLS12-381) over one xGMI link (~153 GB/s): ~0.25 ms
//     of wire time at G = 8 plus 5 x (G-1) grouped steps of latency; it exists to be measured against the default.
// Both give byte-identical proofs (tests/test_distributed_gloo.py at world sizes 2, 3, 8 over the emulated RCCL; on the GPU at
// world size 1, where policy RCCL_SELF makes the rank exchange with ITSELF so that the grouped ncclSend / ncclRecv steps and the
// EC-add kernel of the ring really run on RCCL: tests/rccl_single_rank.py).
#pragma once
#include "common.h"
#include "curve.cuh"
#include "msm_impl.cuh"
#if defined(ARK_EMUL)
#include "rccl_emul.h"
#else
#include <rccl/rccl.h>
#endif

namespace ark355 {

#define ARK_CHECK_NCCL(expr)                                                                        \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess)                                                                          \
      throw ::ark355::HipError{ARK355_ERCCL, std::string(#expr) + ": " + ncclGetErrorString(_r)};   \
  } while (0)

struct CommDev {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  DevBuf gather;       // world * partial_size bytes
  DevBuf ring_tmp;     // one bucket chunk
  ~CommDev() {
    if (comm) (void)ncclCommDestroy(comm);
  }
};

// dst[i] += src[i] over canonical XYZZ points (buckets of the 32-bit accumulation kernels)
template <class F>
__global__ void __launch_bounds__(256)
xyzz_add_inplace_kernel(XYZZ<F>* __restrict__ dst, const XYZZ<F>* __restrict__ src, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = xyzz_add(dst[i], src[i]);
}
// ... and over 28-bit bucket slots (every MSM over resident tables; a lane pair per G2 slot): tails28_impl.cuh
template <class P, bool G2>
__global__ void __launch_bounds__(256, ARK_TAIL28_WAVES)
slot28_add_inplace_kernel(void* __restrict__ dst_, const void* __restrict__ src_, uint32_t count) {
  using T = Tail28<P, G2>;
  using Slot = typename T::Slot;
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / T::LPI;
  if (i >= count) return;
  Slot* dst = static_cast<Slot*>(dst_);
  const Slot* src = static_cast<const Slot*>(src_);
  Acc28<P> a, b;
  bool ae, be;
  T::load(&dst[i], a, ae);
  T::load(&src[i], b, be);
  T::add(a, ae, b, be);
  T::store(&dst[i], a, ae);
}

static inline void ring_chunk(uint32_t total, int world, int c, uint32_t* lo, uint32_t* n) {
  const uint64_t a = (uint64_t)total * (uint32_t)c / (uint32_t)world, b = (uint64_t)total * ((uint32_t)c + 1) / (uint32_t)world;
  *lo = (uint32_t)a;
  *n = (uint32_t)(b - a);
}

// Ring reduce-scatter of `total` buckets (fmt: the slot format of the bucket array, MsmBuckets::fmt -- canonical XYZZ or
// 28-bit slots); afterwards rank g's range (g+1) mod G holds the sum over all ranks and every other range is zeroed
// (= infinity in both formats), so the ordinary bucket reduction over the whole array yields this rank's share of
// sum_b (b+1) B_b.
// self_exchange (policy RCCL_SELF, world size 1 only): one ring step of the rank with itself -- the lower half of the
// array travels through ncclSend / ncclRecv (peer = own rank, grouped) into the staging buffer, is cleared in place and
// comes back through the same EC-add kernel: 0 + x = x, the array is unchanged.
template <class F>
static void ring_reduce_scatter_buckets(CommDev& cm, void* buckets_, uint32_t total, int fmt, hipStream_t stream, bool self_exchange = false) {
  using P = typename Tail28Of<F>::P;
  constexpr bool G2 = Tail28Of<F>::G2;
  const size_t slot = msm_slot_bytes<F>(fmt);
  uint8_t* buckets = static_cast<uint8_t*>(buckets_);
  auto add_inplace = [&](uint8_t* dst, const uint8_t* src, uint32_t n) {
    if (fmt) {
      constexpr uint32_t lpi = Tail28<P, G2>::LPI;
      ARK_LAUNCH((slot28_add_inplace_kernel<P, G2>), dim3((uint32_t)(((uint64_t)n * lpi + 255) / 256)), dim3(256), 0, stream, (void*)dst,
                 (const void*)src, n);
    } else {
      ARK_LAUNCH((xyzz_add_inplace_kernel<F>), dim3((n + 255) / 256), dim3(256), 0, stream, reinterpret_cast<XYZZ<F>*>(dst),
                 reinterpret_cast<const XYZZ<F>*>(src), n);
    }
    ARK_CHECK_LAUNCH();
  };
  const int G = cm.world, g = cm.rank;
  if (G == 1) {
    const uint32_t half = total / 2;
    if (!self_exchange || half == 0) return;
    cm.ring_tmp.ensure((size_t)half * slot);
    uint8_t* tmp = cm.ring_tmp.as<uint8_t>();
    ARK_CHECK_NCCL(ncclGroupStart());
    ARK_CHECK_NCCL(ncclSend(buckets, (size_t)half * slot, ncclUint8, g, cm.comm, stream));
    ARK_CHECK_NCCL(ncclRecv(tmp, (size_t)half * slot, ncclUint8, g, cm.comm, stream));
    ARK_CHECK_NCCL(ncclGroupEnd());
    ARK_CHECK_HIP(hipMemsetAsync(buckets, 0, (size_t)half * slot, stream));
    add_inplace(buckets, tmp, half);
    return;
  }
  uint32_t max_n = 0;
  for (int c = 0; c < G; c++) {
    uint32_t lo, n;
    ring_chunk(total, G, c, &lo, &n);
    if (n > max_n) max_n = n;
  }
  cm.ring_tmp.ensure((size_t)max_n * slot);
  uint8_t* tmp = cm.ring_tmp.as<uint8_t>();
  const int right = (g + 1) % G, left = (g + G - 1) % G;
  for (int t = 0; t < G - 1; t++) {
    const int cs = ((g - t) % G + G) % G, cr = ((g - t - 1) % G + G) % G;
    uint32_t slo, sn, rlo, rn;
    ring_chunk(total, G, cs, &slo, &sn);
    ring_chunk(total, G, cr, &rlo, &rn);
    ARK_CHECK_NCCL(ncclGroupStart());
    ARK_CHECK_NCCL(ncclSend(buckets + (size_t)slo * slot, (size_t)sn * slot, ncclUint8, right, cm.comm, stream));
    ARK_CHECK_NCCL(ncclRecv(tmp, (size_t)rn * slot, ncclUint8, left, cm.comm, stream));
    ARK_CHECK_NCCL(ncclGroupEnd());
    if (rn) add_inplace(buckets + (size_t)rlo * slot, tmp, rn);
  }
  const int own = (g + 1) % G;
  uint32_t olo, on;
  ring_chunk(total, G, own, &olo, &on);
  if (olo) ARK_CHECK_HIP(hipMemsetAsync(buckets, 0, (size_t)olo * slot, stream));
  if (olo + on < total)
    ARK_CHECK_HIP(hipMemsetAsync(buckets + (size_t)(olo + on) * slot, 0, (size_t)(total - olo - on) * slot, stream));
}

}  // namespace ark355
