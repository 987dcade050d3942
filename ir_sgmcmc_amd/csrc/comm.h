// Internal view of irs_comm (include/irsgmcmc.h) for slab.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/irsgmcmc.h"

namespace irs {
struct IpcState;  // ipc.hip
}

struct irs_comm {
    int kind = 0;  // 0 RCCL, 1 callbacks, 2 peer-mapped landing buffers (ipc.hip)
    int rank = 0, world = 1;
    void* nccl = nullptr;  // ncclComm_t
    irs_exchange_fn ex = nullptr;
    irs_allreduce_fn ar = nullptr;
    void* user = nullptr;
    irs::IpcState* ipc = nullptr;
};

namespace irs {
// one grouped round of point-to-point transfers, asynchronous on `st`
int comm_exchange(irs_comm* c, const irs_xfer* x, int n, hipStream_t st);
// in-place all-reduce on `st`; kind 0: SUM of `count` doubles, 1: MAX of `count` uint32 (non-negative float bits order like
// integers), 2: SUM of `count` floats
int comm_allreduce(irs_comm* c, void* buf, size_t count, int max_u32, hipStream_t st);
// room for exchanges of up to `xbytes` per neighbour and all-reduces of up to `arbytes` (the peer-mapped transport allocates and
// publishes its landing area here; collective and blocking there, nothing for the other transports)
int comm_reserve(irs_comm* c, size_t xbytes, size_t arbytes);
// a timeout raised on the device by a waiting kernel of the peer-mapped transport (0: none)
int comm_check(irs_comm* c);
// the sticky DEVICE word such a timeout also sets (nullptr for the other transports): kernels that modify persistent state test
// it, so a transition that ran on a timed-out exchange is a no-op (scalar_kernels.h: comm_bad)
const unsigned* comm_error_flag(const irs_comm* c);

// ipc.hip
int ipc_create(const char* name, int rank, int world, irs_comm** out);
void ipc_destroy(irs_comm* c);
int ipc_reserve(irs_comm* c, size_t xbytes, size_t arbytes);
int ipc_exchange(irs_comm* c, const irs_xfer* x, int n, hipStream_t st);
int ipc_allreduce(irs_comm* c, void* buf, size_t count, int kind, hipStream_t st);
int ipc_check(irs_comm* c);
void ipc_describe(const irs_comm* c, char* out, size_t n);
const unsigned* ipc_error_flag(const irs_comm* c);
}  // namespace irs
