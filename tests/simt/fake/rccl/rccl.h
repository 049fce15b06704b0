// tests/simt/fake/rccl/rccl.h -- TEST INFRASTRUCTURE: the few RCCL types csrc/dhqr_comm.h names, so the library
// host-compiles for the CPU emulator.  The emulated builds never create an RCCL communicator (the tests use the
// in-process LOCAL transport or the CALLBACK transport); librccl.so itself is only ever dlopen()ed by the product.
#pragma once
#include <cstddef>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1 } ncclResult_t;
typedef enum { ncclChar = 0, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
