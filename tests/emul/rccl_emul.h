// TEST INFRASTRUCTURE ONLY -- the subset of the RCCL API that libark355's sharded prover uses
// (snark_amd/csrc/comm_impl.cuh), emulated between PROCESSES of one host over a POSIX shared-memory segment, so that
// the multi-rank code path (unique id, communicator, all-gather, grouped send/recv ring) runs under
// `pytest -m "not gpu"` with world sizes 2, 3, 8 on a machine without GPUs.  The product links the real
// librccl.so (hipcc build); this header is only seen by tests/emul/build_emul.py (g++ -DARK_EMUL).
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef struct emuNcclComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
  char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, void* stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, void* stream);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           void* stream);
