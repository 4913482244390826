// Host cost of the HIP runtime calls a proof is made of (dev tool): wall and CPU microseconds per call, with one and with
// four calling threads, plus the CPU the process burns in threads that are NOT the callers (the runtime's own).
//   hipcc -O2 --offload-arch=gfx950 tools/host_api_cost.hip -o tools/host_api_cost.bin -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu_of(clockid_t id) { timespec ts; clock_gettime(id, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
constexpr int NCASE = 6;
static const char* NAMES[NCASE] = {"kernel launch", "event record (timing)", "event record (no timing)",
                                   "record + wait on other stream + launch there", "memsetAsync 4 KiB", "launch + eventQuery"};
struct Res { double wall_us[NCASE], cpu_us[NCASE]; };
static void worker(int iters, Res* out) {
  hipStream_t a, b;
  CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
  std::vector<hipEvent_t> evt(iters), evn(iters);
  for (auto& e : evt) CK(hipEventCreateWithFlags(&e, hipEventBlockingSync));
  for (auto& e : evn) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  void* buf; CK(hipMalloc(&buf, 1 << 20));
  auto run = [&](int which, auto&& fn) {
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    const double w0 = now(), c0 = cpu_of(CLOCK_THREAD_CPUTIME_ID);
    for (int i = 0; i < iters; i++) fn(i);
    out->wall_us[which] = (now() - w0) / iters * 1e6;
    out->cpu_us[which] = (cpu_of(CLOCK_THREAD_CPUTIME_ID) - c0) / iters * 1e6;
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
  };
  run(0, [&](int) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, (int*)nullptr); });
  run(1, [&](int i) { CK(hipEventRecord(evt[i], a)); });
  run(2, [&](int i) { CK(hipEventRecord(evn[i], a)); });
  run(3, [&](int i) { CK(hipEventRecord(evn[i], a)); CK(hipStreamWaitEvent(b, evn[i], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, (int*)nullptr); });
  run(4, [&](int) { CK(hipMemsetAsync(buf, 0, 4096, a)); });
  run(5, [&](int i) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, (int*)nullptr); (void)hipEventQuery(evt[i]); });
  for (auto& e : evt) (void)hipEventDestroy(e);
  for (auto& e : evn) (void)hipEventDestroy(e);
  (void)hipFree(buf); (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  CK(hipSetDevice(0)); CK(hipFree(nullptr));
  for (int nt : {1, 4}) {
    std::vector<Res> res(nt);
    const double w0 = now(), p0 = cpu_of(CLOCK_PROCESS_CPUTIME_ID);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back(worker, iters, &res[t]);
    for (auto& t : th) t.join();
    const double wall = now() - w0, proc = cpu_of(CLOCK_PROCESS_CPUTIME_ID) - p0;
    printf("%d calling thread(s): %d iterations per case, wall %.2f s, process CPU %.2f s\n", nt, iters, wall, proc);
    for (int c = 0; c < NCASE; c++) {
      double w = 0, u = 0;
      for (auto& r : res) { w += r.wall_us[c]; u += r.cpu_us[c]; }
      printf("  %-46s wall %7.1f us/call   caller CPU %7.1f us/call\n", NAMES[c], w / nt, u / nt);
    }
  }
  return 0;
}
