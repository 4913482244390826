// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP emulator (see hip_emul.h).
#include "hip_emul.h"

namespace emu {
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
Thread* g_cur = nullptr;
ucontext_t g_sched;
uint64_t g_xchg[64];
unsigned char* g_dyn_smem = nullptr;
PairBox g_pairbox[1024];

static const std::function<void()>* g_body = nullptr;
static const size_t STACK = 1 << 20;
static std::vector<char*> g_stacks;

void yield(int st) {
  Thread* me = g_cur;
  me->state = st;
  swapcontext(&me->ctx, &g_sched);
}

static void trampoline() {
  (*g_body)();
  g_cur->state = DONE;
  swapcontext(&g_cur->ctx, &g_sched);
}

static void resume(Thread& t) {
  g_cur = &t;
  g_threadIdx = t.tid;
  swapcontext(&g_sched, &t.ctx);
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
              getcontext(&t.ctx);
              t.ctx.uc_stack.ss_sp = g_stacks[k];
              t.ctx.uc_stack.ss_size = STACK;
              t.ctx.uc_link = nullptr;
              makecontext(&t.ctx, trampoline, 0);
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
