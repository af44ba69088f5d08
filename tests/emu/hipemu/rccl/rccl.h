/* TEST INFRASTRUCTURE -- hipemu: the few RCCL types csrc/comm_host.inc names.  The library resolves RCCL's functions at run time
 * (dlsym), so the emulation build needs declarations only; in an emulated process no librccl is loaded and zk_comm_create reports
 * "RCCL is not available" -- the host transport (zk_comm_create_host) is what multi-rank emulation runs on. */
#ifndef HIPEMU_RCCL_H
#define HIPEMU_RCCL_H
#include <hip/hip_runtime_api.h>
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
#endif
