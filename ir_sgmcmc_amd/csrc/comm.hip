// Transport of the z-slab decomposition (include/irsgmcmc.h: irs_comm_*): ghost-plane point-to-point exchanges between
// neighbouring ranks and the small all-reduces of the partial sums.
//
//   RCCL        ncclSend / ncclRecv inside one group per exchange, ncclAllReduce, all enqueued on a HIP stream (the
//               communication stream of slab.hip) -- xGMI point-to-point between the neighbouring GPUs of one node.  librccl
//               is bound at RUN time (dlopen): the library an application already has in its process (PyTorch ships one) is
//               reused, and a host without RCCL can still load this library for everything that is not multi-GPU.
//   ipc         peer-mapped landing buffers written by the producer, sequence flags instead of a rendezvous (ipc.hip): xGMI stores
//               on a node, and the one transport with which several ranks can share a device ASYNCHRONOUSLY.
//   callbacks   the same two operations handed to caller-supplied functions.  Exists so that the slab schedule can be
//               rehearsed with several ranks SHARING one GPU (RCCL refuses two ranks on one device); tests only.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <time.h>

#include "comm.h"
#include "ctx.h"

namespace irs {

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.handle) return 0;
    const char* names[] = {getenv("IRS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // an RCCL that is already in the process first (two copies of the library in one process would each own their own
    // bootstrap / proxy threads)
    for (const char* n : names)
        if (n && *n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
        if (n && *n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("RCCL not found (librccl.so.1): %s", dlerror());
    RcclApi a;
    a.handle = h;
#define IRS_SYM(field, name)                                                      \
    *(void**)(&a.field) = dlsym(h, name);                                         \
    if (!a.field) return fail("librccl lacks %s", name)
    IRS_SYM(GetUniqueId, "ncclGetUniqueId");
    IRS_SYM(CommInitRank, "ncclCommInitRank");
    IRS_SYM(CommDestroy, "ncclCommDestroy");
    IRS_SYM(Send, "ncclSend");
    IRS_SYM(Recv, "ncclRecv");
    IRS_SYM(AllReduce, "ncclAllReduce");
    IRS_SYM(GroupStart, "ncclGroupStart");
    IRS_SYM(GroupEnd, "ncclGroupEnd");
    IRS_SYM(GetErrorString, "ncclGetErrorString");
#undef IRS_SYM
    g_rccl = a;
    return 0;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r_));           \
    } while (0)

}  // namespace

int comm_exchange(irs_comm* cm, const irs_xfer* x, int n, hipStream_t st) {
    if (!cm) return fail("exchange without a communicator");
    if (n <= 0) return 0;
    if (cm->kind == 1) {
        if (cm->ex(cm->user, x, n, (void*)st)) return fail("exchange callback failed");
        return 0;
    }
    if (cm->kind == 2) return ipc_exchange(cm, x, n, st);
    NCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        if (x[i].peer < 0 || x[i].peer >= cm->world || x[i].peer == cm->rank) {
            (void)g_rccl.GroupEnd();
            return fail("exchange: bad peer %d", x[i].peer);
        }
        const ncclResult_t r = x[i].recv ? g_rccl.Recv(x[i].ptr, x[i].bytes, ncclChar, x[i].peer, (ncclComm_t)cm->nccl, st)
                                         : g_rccl.Send(x[i].ptr, x[i].bytes, ncclChar, x[i].peer, (ncclComm_t)cm->nccl, st);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail("ncclSend/ncclRecv failed: %s", g_rccl.GetErrorString(r));
        }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

int comm_allreduce(irs_comm* cm, void* buf, size_t count, int max_u32, hipStream_t st) {
    if (!cm) return fail("all-reduce without a communicator");
    if (cm->world == 1 || count == 0) return 0;
    if (cm->kind == 1) {
        if (cm->ar(cm->user, buf, count, max_u32, (void*)st)) return fail("all-reduce callback failed");
        return 0;
    }
    if (cm->kind == 2) return ipc_allreduce(cm, buf, count, max_u32, st);
    if (max_u32 == 1) NCCL_TRY(g_rccl.AllReduce(buf, buf, count, ncclUint32, ncclMax, (ncclComm_t)cm->nccl, st));
    else if (max_u32 == 2) NCCL_TRY(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, (ncclComm_t)cm->nccl, st));
    else NCCL_TRY(g_rccl.AllReduce(buf, buf, count, ncclFloat64, ncclSum, (ncclComm_t)cm->nccl, st));
    return 0;
}

int comm_reserve(irs_comm* cm, size_t xbytes, size_t arbytes) {
    if (!cm || cm->kind != 2 || cm->world == 1) return 0;
    return ipc_reserve(cm, xbytes, arbytes);
}

int comm_check(irs_comm* cm) { return cm && cm->kind == 2 ? ipc_check(cm) : 0; }
const unsigned* comm_error_flag(const irs_comm* cm) { return cm && cm->kind == 2 ? ipc_error_flag(cm) : nullptr; }

}  // namespace irs

using namespace irs;

extern "C" {

int irs_comm_unique_id(uint8_t id[IRS_COMM_ID_BYTES]) {
    if (!id) return fail("irs_comm_unique_id: null argument");
    if (load_rccl()) return 1;
    static_assert(sizeof(ncclUniqueId) == IRS_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

int irs_comm_create_rccl(const uint8_t id[IRS_COMM_ID_BYTES], int rank, int world, irs_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail("irs_comm_create_rccl: bad arguments");
    if (load_rccl()) return 1;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, world, u, rank));
    irs_comm* c = new (std::nothrow) irs_comm();
    if (!c) return fail("irs_comm_create_rccl: out of host memory");
    c->kind = 0;
    c->rank = rank;
    c->world = world;
    c->nccl = (void*)comm;
    *out = c;
    return 0;
}

int irs_comm_create_callbacks(irs_exchange_fn ex, irs_allreduce_fn ar, void* user, int rank, int world, irs_comm** out) {
    if (!ex || !ar || !out || world < 1 || rank < 0 || rank >= world) return fail("irs_comm_create_callbacks: bad arguments");
    irs_comm* c = new (std::nothrow) irs_comm();
    if (!c) return fail("irs_comm_create_callbacks: out of host memory");
    c->kind = 1;
    c->rank = rank;
    c->world = world;
    c->ex = ex;
    c->ar = ar;
    c->user = user;
    *out = c;
    return 0;
}

int irs_comm_create_ipc(const char* name, int rank, int world, irs_comm** out) { return ipc_create(name, rank, world, out); }

void irs_comm_destroy(irs_comm* c) {
    if (!c) return;
    if (c->kind == 2) ipc_destroy(c);
    if (c->kind == 0 && c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->nccl);
    delete c;
}

int irs_comm_describe(const irs_comm* c, char* out, size_t n) {
    if (!c || !out || n == 0) return fail("irs_comm_describe: bad arguments");
    if (c->kind == 2) ipc_describe(c, out, n);
    else snprintf(out, n, "%s: %d ranks", c->kind == 0 ? "rccl" : "callbacks", c->world);
    return 0;
}

int irs_comm_rank(const irs_comm* c) { return c ? c->rank : -1; }
int irs_comm_world(const irs_comm* c) { return c ? c->world : -1; }

// All-reduces (SUM of doubles, MAX of uint32) and grouped exchanges with the slab neighbours (rank - 1 and rank + 1, a send and a
// receive with each: the shape of every ghost-plane exchange) on device scratch, checked word by word on the host: the first thing
// a multi-GPU run executes, so that a broken transport fails here and not as a wrong chain.  Several passes of different sizes
// back to back with no synchronisation in between but the final one of each pass: warm caches, both landing slots of the
// peer-mapped transport, its 16-byte and its 4-byte copy paths, one and many workgroups per run.  With one rank the exchange is
// skipped and the all-reduce is the identity.  blocking.
int irs_comm_selftest(irs_comm* c, void* stream) {
    if (!c) return fail("irs_comm_selftest: null communicator");
    hipStream_t st = (hipStream_t)stream;
    const int sizes[5] = {1024, 1001, 1 << 18, 333, 1024};  // doubles per message
    const int nmax = 1 << 18;
    if (comm_reserve(c, 2 * (size_t)nmax * sizeof(double), (size_t)nmax * sizeof(double))) return 1;
    char* dev = nullptr;
    HIP_TRY(hipMalloc((void**)&dev, 5 * (size_t)nmax * sizeof(double)));
    double* d = (double*)dev;
    unsigned* u = (unsigned*)(d + nmax);
    double* sendb = d + 2 * (size_t)nmax;
    double* recv_lo = d + 3 * (size_t)nmax;
    double* recv_hi = d + 4 * (size_t)nmax;
    double* hd = (double*)malloc(3 * (size_t)nmax * sizeof(double));
    unsigned* hu = (unsigned*)malloc((size_t)nmax * sizeof(unsigned));
    auto done = [&](int code) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(dev);
        free(hd);
        free(hu);
        return code;
    };
    if (!hd || !hu) return done(fail("selftest: out of host memory"));
    double *rl = hd + nmax, *rh = hd + 2 * (size_t)nmax;
    const bool has_hi = c->rank + 1 < c->world, has_lo = c->rank > 0;
    for (int pass = 0; pass < 5; ++pass) {
        const int n = sizes[pass];
        const size_t nb = (size_t)n * sizeof(double);
        for (int i = 0; i < n; ++i) {
            hd[i] = (double)(c->rank + 1 + pass) * (i + 1);
            hu[i] = (unsigned)(c->rank * 7 + i + pass);
        }
        if (hipMemcpyAsync(d, hd, nb, hipMemcpyHostToDevice, st) != hipSuccess) return done(fail("selftest: copy failed"));
        (void)hipMemcpyAsync(u, hu, n * sizeof(unsigned), hipMemcpyHostToDevice, st);
        (void)hipMemcpyAsync(sendb, hd, nb, hipMemcpyHostToDevice, st);
        (void)hipMemsetAsync(recv_lo, 0, nb, st);
        (void)hipMemsetAsync(recv_hi, 0, nb, st);
        int rc = 0;
        if (c->kind == 0) {  // through RCCL even with one rank (comm_allreduce short-cuts that case): the bound entry points get used
            if (g_rccl.AllReduce(d, d, n, ncclFloat64, ncclSum, (ncclComm_t)c->nccl, st) != ncclSuccess ||
                g_rccl.AllReduce(u, u, n, ncclUint32, ncclMax, (ncclComm_t)c->nccl, st) != ncclSuccess)
                return done(fail("selftest: ncclAllReduce failed"));
        } else {
            rc |= comm_allreduce(c, d, n, 0, st);
            rc |= comm_allreduce(c, u, n, 1, st);
        }
        if (c->world > 1) {  // the order of exchange_planes (slab.hip): the same on both sides of a link
            irs_xfer x[4];
            int m = 0;
            if (has_hi) {
                x[m++] = irs_xfer{sendb, nb, c->rank + 1, 0};
                x[m++] = irs_xfer{recv_hi, nb, c->rank + 1, 1};
            }
            if (has_lo) {
                x[m++] = irs_xfer{sendb, nb, c->rank - 1, 0};
                x[m++] = irs_xfer{recv_lo, nb, c->rank - 1, 1};
            }
            rc |= comm_exchange(c, x, m, st);
        }
        if (rc) return done(1);
        (void)hipMemcpyAsync(hd, d, nb, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(hu, u, n * sizeof(unsigned), hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(rl, recv_lo, nb, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(rh, recv_hi, nb, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) return done(fail("selftest: stream failed"));
        if (comm_check(c)) return done(1);
        const double tri = 0.5 * c->world * (c->world + 1) + (double)pass * c->world;
        for (int i = 0; i < n; ++i) {
            if (hd[i] != tri * (i + 1)) return done(fail("selftest pass %d: all-reduce SUM wrong at %d (%g, expected %g)", pass, i, hd[i], tri * (i + 1)));
            if (hu[i] != (unsigned)((c->world - 1) * 7 + i + pass)) return done(fail("selftest pass %d: all-reduce MAX wrong at %d", pass, i));
            if (has_lo && rl[i] != (double)(c->rank + pass) * (i + 1)) return done(fail("selftest pass %d: exchange with rank %d wrong at %d", pass, c->rank - 1, i));
            if (has_hi && rh[i] != (double)(c->rank + 2 + pass) * (i + 1)) return done(fail("selftest pass %d: exchange with rank %d wrong at %d", pass, c->rank + 1, i));
        }
    }
    return done(0);
}

// Timing hook (tools/comm_probe.py): `iters` grouped neighbour exchanges of `bytes` per direction and link, then `iters` all-reduces of
// `ar_doubles` doubles, back to back on `stream`, host-timed around a stream synchronisation: what ONE hand-over of the transport costs
// when nothing else runs (the software part of an exchange; on a node the wire time comes on top).  usec[0]: per exchange, usec[1]:
// per all-reduce.  Collective, blocking.
int irs_comm_probe(irs_comm* c, size_t bytes, size_t ar_doubles, int iters, void* stream, double usec[2]) {
    if (!c || !usec || iters < 1 || (bytes & 15u)) return fail("irs_comm_probe: bad arguments (bytes a multiple of 16)");
    hipStream_t st = (hipStream_t)stream;
    if (comm_reserve(c, bytes, ar_doubles * sizeof(double))) return 1;
    char* dev = nullptr;
    HIP_TRY(hipMalloc((void**)&dev, 4 * bytes + (ar_doubles + 1) * sizeof(double)));
    HIP_TRY(hipMemsetAsync(dev, 0, 4 * bytes + (ar_doubles + 1) * sizeof(double), st));
    auto done = [&](int code) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(dev);
        return code;
    };
    const bool has_hi = c->rank + 1 < c->world, has_lo = c->rank > 0;
    irs_xfer x[4];
    int m = 0;
    if (has_hi) {
        x[m++] = irs_xfer{dev, bytes, c->rank + 1, 0};
        x[m++] = irs_xfer{dev + bytes, bytes, c->rank + 1, 1};
    }
    if (has_lo) {
        x[m++] = irs_xfer{dev + 2 * bytes, bytes, c->rank - 1, 0};
        x[m++] = irs_xfer{dev + 3 * bytes, bytes, c->rank - 1, 1};
    }
    double* ar = (double*)(dev + 4 * bytes);
    auto now = []() {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
    };
    for (int phase = 0; phase < 2; ++phase) {
        for (int warm = 0; warm < 2; ++warm) {  // one untimed round first
            if (hipStreamSynchronize(st) != hipSuccess) return done(fail("irs_comm_probe: stream failed"));
            const double t0 = now();
            const int n = warm ? iters : 3;
            for (int i = 0; i < n; ++i)
                if (phase == 0 ? (c->world > 1 && bytes ? comm_exchange(c, x, m, st) : 0) : comm_allreduce(c, ar, ar_doubles, 0, st)) return done(1);
            if (hipStreamSynchronize(st) != hipSuccess) return done(fail("irs_comm_probe: stream failed"));
            if (warm) usec[phase] = (now() - t0) / n;
        }
    }
    if (comm_check(c)) return done(1);
    return done(0);
}

}  // extern "C"
