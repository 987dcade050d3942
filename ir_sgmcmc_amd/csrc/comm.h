// Internal view of irs_comm (include/irsgmcmc.h) for slab.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/irsgmcmc.h"

struct irs_comm {
    int kind = 0;  // 0 RCCL, 1 callbacks
    int rank = 0, world = 1;
    void* nccl = nullptr;  // ncclComm_t
    irs_exchange_fn ex = nullptr;
    irs_allreduce_fn ar = nullptr;
    void* user = nullptr;
};

namespace irs {
// one grouped round of point-to-point transfers, asynchronous on `st`
int comm_exchange(irs_comm* c, const irs_xfer* x, int n, hipStream_t st);
// in-place all-reduce on `st`; kind 0: SUM of `count` doubles, 1: MAX of `count` uint32 (non-negative float bits order like
// integers), 2: SUM of `count` floats
int comm_allreduce(irs_comm* c, void* buf, size_t count, int max_u32, hipStream_t st);
}  // namespace irs
