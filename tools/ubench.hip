// Dev tool: instruction-throughput and field-op microbenchmarks on gfx950 (guides optimisation of field.cuh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../snark_amd/csrc/curve.cuh"
#include "../snark_amd/csrc/field28.cuh"
using namespace ark355;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k_ops(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
  uint64_t acc[8];
  double f[8];
  for (int i = 0; i < 8; i++) { acc[i] = i * 77 + a; f[i] = 1.0 + i + a * 1e-9; }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) acc[i] = (uint64_t)a * (uint32_t)acc[i] + acc[i];                 // v_mad_u64_u32
      if (OP == 1) acc[i] = (uint32_t)acc[i] * b + i;                                // v_mul_lo_u32 (+add)
      if (OP == 2) acc[i] = __umulhi((uint32_t)acc[i], b) + 3;                       // v_mul_hi_u32
      if (OP == 3) acc[i] = acc[i] + ((uint64_t)b << 13 | a);                        // 64-bit add (add_co + addc)
      if (OP == 4) acc[i] = __umul24((uint32_t)acc[i], b) + 1;       // v_mul_u32_u24 / mad
      if (OP == 5) f[i] = __builtin_fma(f[i], 1.0000001, 0.5);                       // v_fma_f64
      if (OP == 6) acc[i] = (uint32_t)acc[i] + b;                                    // v_add_u32
      if (OP == 7) acc[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc[i], 0xB1, 0xF, 0xF, false) + b;   // v_mov_b32_dpp quad_perm
    }
  }
  uint64_t s = 0; double fs = 0;
  for (int i = 0; i < 8; i++) { s += acc[i]; fs += f[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s + (uint32_t)(s >> 32) + (uint32_t)fs;
}

template <class F, bool NI>
__global__ void __launch_bounds__(256) k_fmul(F* out, const F* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = in[t & 1023], b = in[(t + 1) & 1023];
  for (int it = 0; it < iters; it++) { a = fmul<NI>(a, b); b = F::add(b, a); }
  out[t] = a;
}

template <class F>
__global__ void __launch_bounds__(256) k_fmul28(F* out, const F* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = in[t & 1023], b = in[(t + 1) & 1023];
  for (int it = 0; it < iters; it++) { a = F::mul(a, b); b = F::norm(F::add(b, a)); }
  out[t] = a;
}

template <class F, bool NI>
__global__ void __launch_bounds__(256) k_madd(XYZZ<F>* out, const Affine<F>* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int it = 0; it < iters; it++) {
    Affine<F> p = in[(t * 7 + it * 13) & 1023];
    if (NI) xyzz_madd_ni(acc, p); else xyzz_madd(acc, p);
  }
  out[t] = acc;
}

template <class F>
__global__ void k_check(const uint32_t* in, int n, unsigned* bad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  F a, b;
  for (int i = 0; i < F::N; i++) { a.l[i] = in[(2 * t) * 16 + i]; b.l[i] = in[(2 * t + 1) * 16 + i]; }
  F x = F::mul(a, b), y = F::mul_c(a, b);
  F s1 = F::mul(a, a), s2 = F::mul_c(a, a);
  if (x != y || s1 != s2) atomicAdd(bad, 1u);
}

template <class F>
static int check_mul(const char* name) {
  const int n = 1 << 16;
  uint32_t* h = (uint32_t*)calloc((size_t)n * 2 * 16, 4);
  uint64_t st = 88172645463325252ull;
  for (int t = 0; t < 2 * n; t++) {
    for (int i = 0; i < F::N; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[t * 16 + i] = (uint32_t)(st >> 16); }
    h[t * 16 + F::N - 1] %= F::Params::mod(F::N - 1);                 // strictly below the modulus
    if (t % 97 == 0) for (int i = 0; i < F::N; i++) h[t * 16 + i] = F::Params::mod(i) - (i == 0 ? 1 : 0);   // p - 1
    if (t % 101 == 0) for (int i = 0; i < F::N; i++) h[t * 16 + i] = 0;
    if (t % 103 == 0) for (int i = 0; i < F::N; i++) h[t * 16 + i] = (i < F::N - 1) ? 0xFFFFFFFFu : F::Params::mod(i) - 1;
  }
  uint32_t* d; unsigned* bad; unsigned hb = 0;
  hipMalloc(&d, (size_t)n * 2 * 16 * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  hipMemcpy(d, h, (size_t)n * 2 * 16 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_check<F>), dim3(n / 256), dim3(256), 0, 0, d, n, bad);
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("asm-vs-C Montgomery mul check %-8s: %s (%u mismatches of %d)\n", name, hb ? "MISMATCH" : "MATCH", hb, n);
  hipFree(d); hipFree(bad); free(h);
  return hb != 0;
}

template <class P>
__global__ void __launch_bounds__(256, 2) k_madd_g2l(XYZZ<Fp2<P>>* out, const Affine<Fp2<P>>* in, int iters) {
  using FL = Fp2L<P>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int seg = t >> 1, par = t & 1;
  XYZZ<FL> acc = XYZZ<FL>::inf();
  for (int it = 0; it < iters; it++) {
    const Fp<P>* b = reinterpret_cast<const Fp<P>*>(in + ((seg * 7 + it * 13) & 511));
    Affine<FL> p;
    p.x.c = b[par];
    p.y.c = b[2 + par];
    xyzz_madd(acc, p);
  }
  Fp<P>* d = reinterpret_cast<Fp<P>*>(out + seg);
  d[par] = acc.x.c; d[2 + par] = acc.y.c; d[4 + par] = acc.zz.c; d[6 + par] = acc.zzz.c;
}
template <class P> static void l_madd_g2l(void* c);

static float time_it(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(ctx); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch(ctx);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

struct Ctx { void* out; void* in; int iters; int blocks; };
template <int OP> static void l_ops(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL(k_ops<OP>, dim3(x->blocks), dim3(256), 0, 0, (uint32_t*)x->out, 12345u, 6789u, x->iters); }
template <class F, bool NI> static void l_fmul(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((k_fmul<F, NI>), dim3(x->blocks), dim3(256), 0, 0, (F*)x->out, (const F*)x->in, x->iters); }
template <class F> static void l_fmul28(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((k_fmul28<F>), dim3(x->blocks), dim3(256), 0, 0, (F*)x->out, (const F*)x->in, x->iters); }
template <class F, bool NI> static void l_madd(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((k_madd<F, NI>), dim3(x->blocks), dim3(256), 0, 0, (XYZZ<F>*)x->out, (const Affine<F>*)x->in, x->iters); }

template <class P> static void l_madd_g2l(void* c) { Ctx* x = (Ctx*)c; hipLaunchKernelGGL((k_madd_g2l<P>), dim3(x->blocks), dim3(256), 0, 0, (XYZZ<Fp2<P>>*)x->out, (const Affine<Fp2<P>>*)x->in, x->iters); }

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  LDS/block=%zu  regs/block=%d\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.sharedMemPerBlock, prop.regsPerBlock);
  int bad = check_mul<BlsFq>("BlsFq") | check_mul<BlsFr>("BlsFr") | check_mul<BnFq>("BnFq") | check_mul<BnFr>("BnFr");
  if (bad) printf("!!! asm multiplication is WRONG\n");
  Ctx c; c.blocks = prop.multiProcessorCount * 8; c.iters = 2000;
  CHECK(hipMalloc(&c.out, (size_t)c.blocks * 256 * 1024)); CHECK(hipMalloc(&c.in, 1024 * 512));
  // fill inputs with valid-ish field elements: small integers (valid residues)
  { uint32_t* h = (uint32_t*)calloc(1024 * 512 / 4, 4); for (int i = 0; i < 1024 * 128; i++) h[i] = (i * 2654435761u) >> 4; 
    // clear top limbs to stay below the modulus for every layout used
    CHECK(hipMemcpy(c.in, h, 1024 * 512, hipMemcpyHostToDevice)); free(h); }
  const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "add_u64", "v_mul_u32_u24", "v_fma_f64", "v_add_u32", "v_mov_dpp+add"};
  void (*ls[])(void*) = {l_ops<0>, l_ops<1>, l_ops<2>, l_ops<3>, l_ops<4>, l_ops<5>, l_ops<6>, l_ops<7>};
  for (int op = 0; op < 8; op++) {
    float ms = time_it(ls[op], &c, 5);
    double ops = (double)c.blocks * 256 * c.iters * 8;
    printf("%-18s %8.3f ms  %8.2f Gop/s/chip  (%.2f ops/clk/CU @2.4GHz)\n", names[op], ms, ops / ms / 1e6, ops / ms / 1e6 / 2.4 / prop.multiProcessorCount);
  }
  c.iters = 200;
  struct { const char* n; void (*l)(void*); double muls; } fm[] = {
    {"BlsFq mul inline", l_fmul<BlsFq, false>, 1}, {"BlsFq mul noinline", l_fmul<BlsFq, true>, 1},
    {"BlsFr mul inline", l_fmul<BlsFr, false>, 1}, {"BnFq mul inline", l_fmul<BnFq, false>, 1},
    {"BlsFq 28-bit limbs", l_fmul28<BlsFq28>, 1}, {"BnFq 28-bit limbs", l_fmul28<BnFq28>, 1},
    {"BlsFq2 mul inline", l_fmul<BlsFq2, false>, 1}, {"BlsFq2 mul noinline", l_fmul<BlsFq2, true>, 1}};
  for (auto& f : fm) {
    float ms = time_it(f.l, &c, 3);
    double ops = (double)c.blocks * 256 * c.iters;
    printf("%-22s %8.3f ms  %8.2f Gmul/s\n", f.n, ms, ops / ms / 1e6);
  }
  c.iters = 64;
  struct { const char* n; void (*l)(void*); } md[] = {
    {"G1 BLS madd inline", l_madd<BlsFq, false>}, {"G1 BLS madd noinline", l_madd<BlsFq, true>},
    {"G2 BLS madd inline", l_madd<BlsFq2, false>}, {"G2 BLS madd noinline", l_madd<BlsFq2, true>},
    {"G1 BN madd inline", l_madd<BnFq, false>}, {"G2 BN madd noinline", l_madd<BnFq2, true>},
    {"G2 BLS madd lane-split (x2 lanes)", l_madd_g2l<BlsFqParams>}, {"G2 BN madd lane-split (x2 lanes)", l_madd_g2l<BnFqParams>}};
  for (auto& f : md) {
    float ms = time_it(f.l, &c, 3);
    double ops = (double)c.blocks * 256 * c.iters;
    printf("%-22s %8.3f ms  %8.3f Gadd/s\n", f.n, ms, ops / ms / 1e6);
  }
  return 0;
}
