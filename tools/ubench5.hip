// Dev tool (round 5): what ONE instruction of the bucket-accumulation kernels costs on gfx950, and what the whole
// Montgomery product / mixed addition costs with and without the Karatsuba level of field28.cuh.
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I snark_amd/csrc [-DARK_F28_KARATSUBA=0] tools/ubench5.hip -o ubench5
//
// Part 1: issue cost per wave-instruction, measured INSIDE the kernel (s_memtime) with 8 independent chains per lane, at
//         one and at two waves per SIMD (the occupancy of the accumulation kernels), plus the wall-clock rate of the chip.
// Part 2: the go / no-go number for a floating-point multiplier: the instruction skeleton of a 381-bit Montgomery product
//         on 8 x 52-bit limbs (128 limb products, each two v_fma_f64, one v_add_f64 and two 64-bit integer additions --
//         Emmart's scheme; splitting, carries and the final conversion NOT included, i.e. a lower bound) against the real
//         28-bit product of field28.cuh.
// Part 3: Fp28 product and G1 / G2 mixed addition (madd28, madd28_g2 of msm28_impl.cuh) in a register-resident loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "msm_impl.cuh"
using namespace ark355;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { OP_MAD_U64, OP_MAD_I64, OP_LSHL_ADD_U64, OP_LSHR_B64, OP_AND, OP_SUB, OP_MUL_LO, OP_MUL_HI, OP_CNDMASK, OP_DPP, OP_FMA64,
       OP_ADD64_PAIR, OP_ALIGNBIT, OP_MAD_U24, OP_MIX_MAD_AND, OP_MIX_MAD_ADD64, OP_MAD_DEP, OP_ADD3, OP_XAD, OP_CNDMASK_SGPR, OP_BFI, OP_DS_WRITE, OP_COUNT };
static const char* OP_NAME[OP_COUNT] = {"v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64", "v_lshrrev_b64", "v_and_b32", "v_sub_u32",
  "v_mul_lo_u32", "v_mul_hi_u32", "v_cndmask_b32", "v_mov_b32_dpp", "v_fma_f64", "v_add_co+v_addc_co (pair)", "v_alignbit_b32",
  "v_mad_u32_u24", "mix 3 mad_u64 : 1 and (4)", "mix 1 mad_u64 : 1 lshl_add_u64 (2)", "v_mad_u64_u32 ONE dependent chain", "v_add3_u32",
  "v_xad_u32", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_bfi_b32", "ds_write_b32 (stride 1 KiB, own lane)"};
// instructions per unrolled group of 8 chains
static const int OP_PER_GROUP[OP_COUNT] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 16, 8, 8, 32, 16, 8, 8, 8, 8, 8, 8};

template <int OP, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_instr(uint64_t* out, uint64_t* cyc, uint32_t a0, int iters) {
  uint32_t a = a0 + threadIdx.x, b = (a0 * 2654435761u) ^ threadIdx.x;
  uint64_t acc[8];
  double f[8];
  uint32_t w[8], w2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { acc[i] = (uint64_t)(i * 77 + a) << 7; f[i] = 1.0 + i + a * 1e-9; w[i] = a * (i + 3); w2[i] = b + i; }
  const double fc = 1.0000001, fd = 0.5;
  const uint64_t smask = __ballot((threadIdx.x & 1) != 0);       // lane-parity mask in an SGPR pair (what Pair28::sel selects on)
  __shared__ uint32_t lds_buf[OP == OP_DS_WRITE ? 256 : 1];
  const uint32_t lds_addr = (uint32_t)(threadIdx.x * 4);
  (void)lds_buf;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == OP_MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (OP == OP_MAD_I64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        if (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
        if (OP == OP_LSHR_B64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
        if (OP == OP_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
        if (OP == OP_SUB) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
        if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
        if (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
        if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(w[i]) : "v"(b) : );
        if (OP == OP_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(w[i]));
        if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fc), "v"(fd));
        if (OP == OP_ADD64_PAIR) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc"
                                              : "+v"(w[i]), "+v"(w2[i]) : "v"(a), "v"(b) : "vcc");
        if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(w[i]) : "v"(b));
        if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
        if (OP == OP_MIX_MAD_AND) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[(i + 3) & 7]) : "v"(b), "v"(a) : "vcc");
          asm volatile("v_and_b32 %0, %1, %0" : "+v"(w[i]) : "v"(b));
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[(i + 5) & 7]) : "v"(a), "v"(a) : "vcc");
        }
        if (OP == OP_MIX_MAD_ADD64) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[(i + 4) & 7]) : "v"(acc[(i + 5) & 7]));
        }
        if (OP == OP_MAD_DEP) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
        if (OP == OP_XAD) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
        if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(w[i]) : "v"(b), "s"(smask));
        if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
        if (OP == OP_DS_WRITE) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(lds_addr), "v"(w[i]), "n"(1024 * 0) : "memory");
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0;
  double fs = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { s += acc[i] + w[i] + w2[i]; fs += f[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (uint64_t)fs;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Part 2: the skeleton of a 52-bit-limb floating-point Montgomery product (lower bound, see the header)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_dfma_skeleton(uint64_t* out, uint64_t* cyc, uint32_t a0, int iters) {
  double a[8], b[8];
  for (int i = 0; i < 8; i++) { a[i] = (double)((a0 + threadIdx.x) * (i + 1)); b[i] = (double)((a0 ^ threadIdx.x) + i); }
  const double c1 = 0x1p104, c2 = 0x1p104 + 0x1p52;
  uint64_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    // 2 x 64 limb products (operand product + m p product); every product: hi = fma(a,b,c1); lo = fma(a,b,c2-hi);
    // both bit patterns accumulated as integers
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          double h, d, l;
          asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(h) : "v"(a[i]), "v"(b[j]), "v"(c1));
          asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d) : "v"(c2), "v"(h));
          asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(l) : "v"(a[i]), "v"(b[j]), "v"(d));
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(hi[(i + j) & 3]) : "v"(h));
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(lo[(i + j) & 3]) : "v"(l));
        }
      }
      a[half] = __longlong_as_double((long long)(hi[0] & 0xFFFFFFFFFFFFFull) | 0x4330000000000000ll);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = hi[0] + hi[1] + hi[2] + hi[3] + lo[0] + lo[1] + lo[2] + lo[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Part 3
template <class F, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_fmul28(F* out, uint64_t* cyc, const F* in, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  F a = in[t & 1023], b = in[(t + 1) & 1023];
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) { a = F::mul(a, b); b = F::mul(b, a); }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[t] = F::add(a, b);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class P>
__global__ void __launch_bounds__(256, 2) k_madd28(Acc28<P>* out, uint64_t* cyc, const Affine28<P>* in, int iters) {
  using F = Fp28<P>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Acc28<P> acc;
  acc.x = acc.y = acc.zz = acc.zzz = F::zero();
  bool empty = true;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const Affine28<P>& row = in[(t * 7 + it * 13) & 1023];
    const F px = Affine28<P>::unpack(row.w), py = Affine28<P>::unpack(row.w + Affine28<P>::NB);
    madd28<P>(acc, empty, px, py, (it & 1) != 0);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[t] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class P>
__global__ void __launch_bounds__(256, ARK_G2L28_WAVES) k_madd28_g2(Acc28<P>* out, uint64_t* cyc, const Affine28G2<P>* in, int iters) {
  using F = Fp28<P>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int seg = t >> 1, par = t & 1;
  Acc28<P> acc;
  acc.x = acc.y = acc.zz = acc.zzz = F::zero();
  bool empty = true;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const Affine28U<P>& row = in[(seg * 7 + it * 13) & 511].half[par];
    F px, py;
#pragma unroll
    for (int k = 0; k < F::N; k++) { px.l[k] = row.w[k] & F::MASK; py.l[k] = row.w[F::N + k] & F::MASK; }
    madd28_g2<P>(acc, empty, px, py, (it & 1) != 0);
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[t] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Run {
  void* out; uint64_t* cyc; void* in; int iters; int blocks;
};
static double avg_cycles(const Run& r) {
  uint64_t* h = (uint64_t*)malloc(r.blocks * 8);
  hipMemcpy(h, r.cyc, r.blocks * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < r.blocks; i++) s += (double)h[i];
  free(h);
  return s / r.blocks;
}
template <class L>
static float time_it(L launch, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

template <int OP, int WAVES>
static void run_instr(Run r, int cus) {
  r.blocks = cus * WAVES;                      // one wave per SIMD and block; WAVES blocks per CU
  const float ms = time_it([&] { hipLaunchKernelGGL((k_instr<OP, WAVES>), dim3(r.blocks), dim3(256), 0, 0, (uint64_t*)r.out, r.cyc, 12345u, r.iters); }, 3);
  const double n = (double)r.iters * 4 * OP_PER_GROUP[OP];        // wave-instructions per wave
  const double cyc = avg_cycles(r);
  printf("%-36s waves/SIMD=%d  %7.3f ms  %6.2f counter ticks per instruction and wave  %8.2f G wave-instr/s/chip  (%6.2f T lane-op/s)\n",
         OP_NAME[OP], WAVES, ms, cyc / n, n * r.blocks * 4 / ms / 1e6, n * r.blocks * 256 / ms / 1e9);
}
template <int OP>
static void run_instr_both(const Run& r, int cus) { run_instr<OP, 1>(r, cus); run_instr<OP, 2>(r, cus); }

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clock=%d MHz   ARK_F28_KARATSUBA=%d ARK_G2L28_WAVES=%d\n", prop.name, cus, prop.clockRate / 1000, ARK_F28_KARATSUBA, ARK_G2L28_WAVES);
  Run r; r.iters = 4000; r.blocks = cus * 2;
  CHECK(hipMalloc(&r.out, (size_t)cus * 16 * 256 * 256)); CHECK(hipMalloc(&r.cyc, cus * 16 * 8)); CHECK(hipMalloc(&r.in, 1024 * 512));
  {   // counter frequency: ticks of the in-kernel counter per second of wall clock
    run_instr<OP_AND, 1>(r, cus);
  }
  printf("--- part 1: cost per instruction (counter ticks; see the counter rate below) ---\n");
  run_instr_both<OP_MAD_U64>(r, cus); run_instr_both<OP_MAD_I64>(r, cus); run_instr_both<OP_LSHL_ADD_U64>(r, cus);
  run_instr_both<OP_LSHR_B64>(r, cus); run_instr_both<OP_AND>(r, cus); run_instr_both<OP_SUB>(r, cus); run_instr_both<OP_MUL_LO>(r, cus);
  run_instr_both<OP_MUL_HI>(r, cus); run_instr_both<OP_CNDMASK>(r, cus); run_instr_both<OP_DPP>(r, cus); run_instr_both<OP_FMA64>(r, cus);
  run_instr_both<OP_ADD64_PAIR>(r, cus); run_instr_both<OP_ALIGNBIT>(r, cus); run_instr_both<OP_MAD_U24>(r, cus);
  run_instr_both<OP_MIX_MAD_AND>(r, cus); run_instr_both<OP_MIX_MAD_ADD64>(r, cus); run_instr_both<OP_MAD_DEP>(r, cus);
  run_instr_both<OP_ADD3>(r, cus); run_instr_both<OP_XAD>(r, cus); run_instr_both<OP_CNDMASK_SGPR>(r, cus);
  run_instr_both<OP_BFI>(r, cus); run_instr_both<OP_DS_WRITE>(r, cus);
  {   // counter rate: a kernel of known duration
    Run q = r; q.blocks = cus * 2;
    const float ms = time_it([&] { hipLaunchKernelGGL((k_instr<OP_MAD_U64, 2>), dim3(q.blocks), dim3(256), 0, 0, (uint64_t*)q.out, q.cyc, 1u, q.iters); }, 1);
    printf("counter: %.1f ticks per microsecond of kernel time (kernel %.3f ms incl. launch)\n", avg_cycles(q) / (ms * 1e3), ms);
  }
  printf("--- part 2: floating-point multiplier, go / no-go ---\n");
  {
    Run q = r; q.iters = 400;
    for (int waves = 1; waves <= 2; waves++) {
      q.blocks = cus * waves;
      const float ms = waves == 1
        ? time_it([&] { hipLaunchKernelGGL((k_dfma_skeleton<1>), dim3(q.blocks), dim3(256), 0, 0, (uint64_t*)q.out, q.cyc, 7u, q.iters); }, 3)
        : time_it([&] { hipLaunchKernelGGL((k_dfma_skeleton<2>), dim3(q.blocks), dim3(256), 0, 0, (uint64_t*)q.out, q.cyc, 7u, q.iters); }, 3);
      printf("52-bit-limb DFMA Montgomery skeleton (128 limb products: 256 fma_f64 + 128 add_f64 + 256 add_u64)  waves/SIMD=%d  %7.3f ms  %8.1f ticks per product and wave  %7.2f G products/s/chip (lower bound of the cost)\n",
             waves, ms, avg_cycles(q) / q.iters, (double)q.iters * q.blocks * 256 / ms / 1e6);
    }
  }
  printf("--- part 3: Fp28 product and mixed additions ---\n");
  {   // valid-ish inputs: small canonical limbs
    uint32_t* h = (uint32_t*)calloc(1024 * 512 / 4, 4);
    for (int i = 0; i < 1024 * 128; i++) h[i] = ((uint32_t)i * 2654435761u);
    for (int i = 0; i < 1024 * 128; i += 4) h[i + 3] &= 0xFFFFF;      // every coordinate (8 or 12 words) stays below its modulus
    hipMemcpy(r.in, h, 1024 * 512, hipMemcpyHostToDevice); free(h);
  }
  {
    Run q = r; q.iters = 1000;
    for (int waves = 1; waves <= 2; waves++) {
      q.blocks = cus * waves * 4;
      const float ms = waves == 1
        ? time_it([&] { hipLaunchKernelGGL((k_fmul28<BlsFq28, 1>), dim3(q.blocks), dim3(256), 0, 0, (BlsFq28*)q.out, q.cyc, (const BlsFq28*)q.in, q.iters); }, 3)
        : time_it([&] { hipLaunchKernelGGL((k_fmul28<BlsFq28, 2>), dim3(q.blocks), dim3(256), 0, 0, (BlsFq28*)q.out, q.cyc, (const BlsFq28*)q.in, q.iters); }, 3);
      printf("BlsFq28 mul (dependent pair)   launch_bounds waves=%d  %7.3f ms  %8.1f ticks per product and wave  %7.2f G products/s/chip\n", waves, ms,
             avg_cycles(q) / (2.0 * q.iters), 2.0 * q.iters * q.blocks * 256 / ms / 1e6);
    }
    q.blocks = cus * 8; q.iters = 400;
    float ms = time_it([&] { hipLaunchKernelGGL((k_madd28<BlsFqParams>), dim3(q.blocks), dim3(256), 0, 0, (Acc28<BlsFqParams>*)q.out, q.cyc, (const Affine28<BlsFqParams>*)q.in, q.iters); }, 3);
    printf("G1 BLS madd28                  %7.3f ms  %8.1f ticks per addition and wave  %7.3f G additions/s/chip\n", ms, avg_cycles(q) / q.iters,
           (double)q.iters * q.blocks * 256 / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL((k_madd28_g2<BlsFqParams>), dim3(q.blocks), dim3(256), 0, 0, (Acc28<BlsFqParams>*)q.out, q.cyc, (const Affine28G2<BlsFqParams>*)q.in, q.iters); }, 3);
    printf("G2 BLS madd28_g2 (lane pairs)  %7.3f ms  %8.1f ticks per addition and wave  %7.3f G additions/s/chip\n", ms, avg_cycles(q) / q.iters,
           (double)q.iters * q.blocks * 128 / ms / 1e6);
    ms = time_it([&] { hipLaunchKernelGGL((k_madd28<BnFqParams>), dim3(q.blocks), dim3(256), 0, 0, (Acc28<BnFqParams>*)q.out, q.cyc, (const Affine28<BnFqParams>*)q.in, q.iters); }, 3);
    printf("G1 BN254 madd28                %7.3f ms  %8.1f ticks per addition and wave  %7.3f G additions/s/chip\n", ms, avg_cycles(q) / q.iters,
           (double)q.iters * q.blocks * 256 / ms / 1e6);
  }
  return 0;
}
