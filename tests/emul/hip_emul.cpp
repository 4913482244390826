// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP emulator (see hip_emul.h).
#include "hip_emul.h"

namespace emu {
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
Thread* g_cur = nullptr;
void* g_sched_sp = nullptr;
uint64_t g_xchg[64];
unsigned char* g_dyn_smem = nullptr;
PairBox g_pairbox[1024];

static const std::function<void()>* g_body = nullptr;
static const size_t STACK = 1 << 20;
static std::vector<char*> g_stacks;

// Register-only context switch (System V x86-64): push the callee-saved registers, swap stack pointers, pop, return.
// No signal mask, no floating-point environment: the coroutines are plain integer / SSE code of one thread.
#if !defined(__x86_64__)
#error "the HIP emulator's context switch is written for x86-64"
#endif
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

void yield(int st) {
  Thread* me = g_cur;
  me->state = st;
  emu_switch(&me->sp, g_sched_sp);
}

static void trampoline() {
  (*g_body)();
  g_cur->state = DONE;
  emu_switch(&g_cur->sp, g_sched_sp);
  abort();                               // a finished coroutine is never resumed
}

static void resume(Thread& t) {
  g_cur = &t;
  g_threadIdx = t.tid;
  emu_switch(&g_sched_sp, t.sp);
}

// a fresh coroutine: six zeroed callee-saved registers, then the address emu_switch "returns" to; the slot above it is where a
// caller's return address would sit, so that the entry sees the stack alignment the ABI promises (rsp = 16 k + 8)
static void* fresh_stack(char* stack, size_t size) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                  // (fake return address of the trampoline)
  *--sp = reinterpret_cast<void*>(&trampoline);
  for (int i = 0; i < 6; i++) *--sp = nullptr;      // rbp rbx r12 r13 r14 r15
  return sp;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || grid.x * grid.y * grid.z == 0) return;
  g_body = &body;
  g_blockDim = block;
  g_gridDim = grid;
  while (g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(STACK));
  std::vector<Thread> th(nthreads);
  std::vector<unsigned char> dyn(smem + 64);
  g_dyn_smem = dyn.data();
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        g_blockIdx = dim3(bx, by, bz);
        unsigned k = 0;
        for (unsigned tz = 0; tz < block.z; tz++)
          for (unsigned ty = 0; ty < block.y; ty++)
            for (unsigned tx = 0; tx < block.x; tx++, k++) {
              Thread& t = th[k];
              t.sp = fresh_stack(g_stacks[k], STACK);
              t.state = RUN;
              t.tid = dim3(tx, ty, tz);
              g_pairbox[k % 1024] = PairBox();
            }
        const unsigned nwaves = (nthreads + 63) / 64;
        for (;;) {
          bool any_run = false;
          for (unsigned w = 0; w < nwaves; w++) {
            const unsigned lo = w * 64, hi = (lo + 64 < nthreads) ? lo + 64 : nthreads;
            for (;;) {
              bool ran = false;
              for (unsigned i = lo; i < hi; i++)
                if (th[i].state == RUN) {
                  resume(th[i]);
                  ran = true;
                }
              // lanes that yielded with RUN (pair exchange polling) are simply resumed on the next pass
              bool any_running = false, any_wave_wait = false;
              for (unsigned i = lo; i < hi; i++) {
                any_running |= (th[i].state == RUN);
                any_wave_wait |= (th[i].state == WAVE_WAIT);
              }
              if (any_running) {
                if (!ran) break;
                continue;
              }
              if (!any_wave_wait) break;
              for (unsigned i = lo; i < hi; i++)
                if (th[i].state == WAVE_WAIT) th[i].state = RUN;
            }
            for (unsigned i = lo; i < hi; i++) any_run |= (th[i].state == RUN);
          }
          if (any_run) continue;
          bool any_block_wait = false;
          for (unsigned i = 0; i < nthreads; i++) any_block_wait |= (th[i].state == BLOCK_WAIT);
          if (!any_block_wait) break;
          for (unsigned i = 0; i < nthreads; i++)
            if (th[i].state == BLOCK_WAIT) th[i].state = RUN;
        }
      }
  g_dyn_smem = nullptr;
}
}  // namespace emu
