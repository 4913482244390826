// Internal plumbing of libark355: context, device buffers, error handling, curve traits.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <cstring>
#include <initializer_list>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "hd.h"
#include "policy.h"
#include "curve.cuh"
#include "../../include/ark355.h"

namespace ark355 {

struct BlsCurve {
  static constexpr int ID = ARK355_BLS12_381;
  using Fr = BlsFr;
  using Fq = BlsFq;
  using Fq2 = BlsFq2;
  using Consts = BlsCurveConsts;
};
struct BnCurve {
  static constexpr int ID = ARK355_BN254;
  using Fr = BnFr;
  using Fq = BnFq;
  using Fq2 = BnFq2;
  using Consts = BnCurveConsts;
};

struct HipError {
  int code;
  std::string what;
};

#define ARK_CHECK_HIP(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw ::ark355::HipError{ARK355_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
  } while (0)

#define ARK_CHECK_LAUNCH() ARK_CHECK_HIP(hipGetLastError())

#define ARK_REQUIRE(cond, code, msg)                                 \
  do {                                                               \
    if (!(cond)) throw ::ark355::HipError{(code), std::string(msg)}; \
  } while (0)

// RAII device allocation
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      o.p = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      throw HipError{ARK355_ENOMEM, "hipMalloc(" + std::to_string(n) + ") failed"};
    }
    bytes = n;
  }
  // grow-only scratch
  void ensure(size_t n) {
    if (n > bytes) alloc(n);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

struct NttTables;   // ntt_impl.cuh

}  // namespace ark355

struct ark355_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  std::mutex mu;
  ark355::TunePolicy policy;    // every runtime switch; read from the environment once, at ark355_ctx_create (policy.h)
  ark355_timings timings{};
  float acc_ms = 0.f;           // bucket-accumulation kernel time of the last MSM/prove
  uint64_t acc_launches = 0;
  uint64_t acc_points = 0;
  int acc_threads_hint = 0;     // workgroup size the running call wants for the LDS-free accumulation kernels (policy ACC_THREADS = 0); 0: 256
  std::atomic<int> last_sched{-1};   // schedule the last prove on this context ran as (ark355::Sched); read without the lock by ark355_sched_info
  // NTT twiddle tables keyed by (curve << 8 | log_n)
  std::map<uint32_t, std::shared_ptr<ark355::NttTables>> ntt_tables;   // shared with the other contexts of the device
  // grow-only scratch buffers reused across calls (sized for 288 GB HBM: never shrunk)
  ark355::DevBuf scratch[12];
};
