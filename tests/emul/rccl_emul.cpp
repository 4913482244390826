// TEST INFRASTRUCTURE ONLY -- see rccl_emul.h.  One shared-memory segment per communicator:
//   header (world, per-rank "joined" flags) + world*world mailboxes, mailbox (src -> dst) = {full flag, length, data}.
// Sends and receives are queued while a group is open and progressed together at ncclGroupEnd (a blocking
// call outside a group is a group of one; several operations with one peer complete in posting order), which is what makes a ring step (send to the right, receive from the left)
// deadlock-free exactly as in RCCL.
#include "rccl_emul.h"

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {
constexpr size_t BOX_CAP = 1u << 16;
constexpr int MAX_WORLD = 16;

struct Mailbox {
  std::atomic<uint32_t> full;
  uint32_t len;
  unsigned char data[BOX_CAP];
};
struct Segment {
  std::atomic<uint32_t> ready;
  std::atomic<uint32_t> joined;
  std::atomic<uint32_t> left;
  uint32_t world;
  Mailbox box[1];      // world * world, row = src
};

struct Op {
  bool send;
  unsigned char* p;
  size_t remaining;
  int peer;
};
}  // namespace

struct emuNcclComm {
  Segment* seg = nullptr;
  size_t seg_bytes = 0;
  int rank = 0, world = 1;
  char name[80];
  std::vector<Op> pending;
};

namespace {
thread_local int g_group_depth = 0;
thread_local std::vector<emuNcclComm*> g_group_comms;

Mailbox& box(emuNcclComm* c, int src, int dst) { return c->seg->box[(size_t)src * c->world + dst]; }

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

ncclResult_t progress(emuNcclComm* c) {
  const double deadline = now_s() + 300.0;
  for (;;) {
    bool all_done = true, moved = false;
    // Several sends to (receives from) ONE peer inside a group are matched in the order they were posted, as in RCCL: only the
    // first unfinished send / receive per peer may touch the pair's mailbox in a pass.  (Without this, a later send could slip
    // into a mailbox that the peer emptied between two checks of the same pass and overtake an earlier one -- the all-to-all of
    // the distributed witness map posts three sends per peer and hit exactly that about once in six runs at world size 8.)
    bool send_busy[MAX_WORLD] = {}, recv_busy[MAX_WORLD] = {};
    for (Op& op : c->pending) {
      if (op.remaining == 0) continue;
      all_done = false;
      bool* busy = op.send ? send_busy : recv_busy;
      if (busy[op.peer]) continue;
      busy[op.peer] = true;
      if (op.send) {
        Mailbox& b = box(c, c->rank, op.peer);
        if (b.full.load(std::memory_order_acquire)) continue;
        const size_t k = op.remaining < BOX_CAP ? op.remaining : BOX_CAP;
        memcpy(b.data, op.p, k);
        b.len = (uint32_t)k;
        b.full.store(1, std::memory_order_release);
        op.p += k;
        op.remaining -= k;
        moved = true;
      } else {
        Mailbox& b = box(c, op.peer, c->rank);
        if (!b.full.load(std::memory_order_acquire)) continue;
        const size_t k = b.len;
        if (k > op.remaining) return ncclInvalidUsage;       // mismatched send/recv sizes
        memcpy(op.p, b.data, k);
        b.full.store(0, std::memory_order_release);
        op.p += k;
        op.remaining -= k;
        moved = true;
      }
    }
    if (all_done) break;
    if (!moved) {
      if (now_s() > deadline) return ncclSystemError;
      usleep(100);          // ranks may outnumber cores: do not spin against the peer we are waiting for
    }
  }
  c->pending.clear();
  return ncclSuccess;
}

ncclResult_t enqueue(emuNcclComm* c, Op op) {
  if (!c || op.peer < 0 || op.peer >= c->world) return ncclInvalidArgument;
  if (op.peer == c->rank && g_group_depth == 0) return ncclInvalidUsage;   // a send to self needs its receive in the same group (as in RCCL)
  c->pending.push_back(op);
  if (g_group_depth == 0) return progress(c);
  bool known = false;
  for (auto* k : g_group_comms) known |= (k == c);
  if (!known) g_group_comms.push_back(c);
  return ncclSuccess;
}
}  // namespace

extern "C++" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id->internal, 0, sizeof(id->internal));
  unsigned char rnd[16];
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, rnd, sizeof(rnd)) != (ssize_t)sizeof(rnd)) {
    if (fd >= 0) close(fd);
    return ncclSystemError;
  }
  close(fd);
  char* o = id->internal;
  o += sprintf(o, "/ark355emu_");
  for (unsigned char b : rnd) o += sprintf(o, "%02x", b);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_WORLD || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  if (id.internal[0] != '/') return ncclInvalidArgument;
  auto* c = new emuNcclComm();
  c->rank = rank;
  c->world = nranks;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->seg_bytes = sizeof(Segment) + sizeof(Mailbox) * ((size_t)nranks * nranks);
  bool creator = true;
  int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 && errno == EEXIST) {
    creator = false;
    fd = shm_open(c->name, O_RDWR, 0600);
  }
  if (fd < 0) {
    delete c;
    return ncclSystemError;
  }
  if (creator && ftruncate(fd, (off_t)c->seg_bytes) != 0) {
    close(fd);
    delete c;
    return ncclSystemError;
  }
  if (!creator) {
    // wait until the creator has sized the segment
    const double deadline = now_s() + 120.0;
    struct stat st;
    while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->seg_bytes) {
      if (now_s() > deadline) {
        close(fd);
        delete c;
        return ncclSystemError;
      }
      sched_yield();
    }
  }
  void* p = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->seg = (Segment*)p;
  if (creator) {
    c->seg->world = (uint32_t)nranks;       // fresh segments are zero-filled: flags and counters start at 0
    c->seg->ready.store(1, std::memory_order_release);
  }
  const double deadline = now_s() + 120.0;
  while (!c->seg->ready.load(std::memory_order_acquire)) {
    if (now_s() > deadline) return ncclSystemError;
    sched_yield();
  }
  if (c->seg->world != (uint32_t)nranks) return ncclInvalidArgument;
  c->seg->joined.fetch_add(1);
  while (c->seg->joined.load() < (uint32_t)nranks) {       // ncclCommInitRank is collective
    if (now_s() > deadline) return ncclSystemError;
    usleep(200);
  }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->seg) {
    const uint32_t gone = c->seg->left.fetch_add(1) + 1;
    const bool last = gone == (uint32_t)c->world;
    munmap(c->seg, c->seg_bytes);
    if (last) shm_unlink(c->name);
  }
  delete c;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclSystemError: return "emulated RCCL: system error / timeout";
    case ncclInvalidArgument: return "emulated RCCL: invalid argument";
    case ncclInvalidUsage: return "emulated RCCL: invalid usage";
    default: return "emulated RCCL: error";
  }
}

ncclResult_t ncclGroupStart(void) {
  g_group_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (auto* c : g_group_comms) {
    const ncclResult_t e = progress(c);
    if (rc == ncclSuccess) rc = e;
  }
  g_group_comms.clear();
  return rc;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t, int peer, ncclComm_t comm, void*) {
  return enqueue(comm, Op{true, (unsigned char*)sendbuff, count, peer});
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t, int peer, ncclComm_t comm, void*) {
  return enqueue(comm, Op{false, (unsigned char*)recvbuff, count, peer});
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           void* stream) {
  if (!comm) return ncclInvalidArgument;
  unsigned char* out = (unsigned char*)recvbuff;
  if (out + (size_t)comm->rank * sendcount != sendbuff) memmove(out + (size_t)comm->rank * sendcount, sendbuff, sendcount);
  if (comm->world == 1) return ncclSuccess;
  ncclResult_t rc = ncclGroupStart();
  for (int p = 0; p < comm->world && rc == ncclSuccess; p++) {
    if (p == comm->rank) continue;
    rc = ncclSend(out + (size_t)comm->rank * sendcount, sendcount, dt, p, comm, stream);
    if (rc == ncclSuccess) rc = ncclRecv(out + (size_t)p * sendcount, sendcount, dt, p, comm, stream);
  }
  const ncclResult_t e = ncclGroupEnd();
  return rc != ncclSuccess ? rc : e;
}

}  // extern "C++"
