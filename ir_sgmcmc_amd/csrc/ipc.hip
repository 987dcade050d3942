// Peer-mapped transport of the z-slab decomposition (include/irsgmcmc.h: irs_comm_create_ipc): ghost planes are WRITTEN BY THE
// PRODUCER into a landing buffer of the consuming rank that the producer has mapped with hipIpcOpenMemHandle, and ordered across
// processes by sequence flags -- no host or stream synchronisation inside an exchange or an all-reduce, no send / recv kernels
// of a communication library competing for CUs.  On a node the stores travel over xGMI; several ranks may also share ONE
// device (the handles open there too), which is how this transport -- and with it the asynchronous event plumbing of the slab
// schedule, csrc/slab.hip -- is exercised on a one-GPU box.  The sharded body is the reference's single-device loop body,
// trainer/trainer.py:291-356 (the reference has no multi-device code, base/base_trainer.py:16).
//
//   bootstrap   a POSIX shared-memory segment named by the caller: every rank publishes the hipIpcMemHandle_t of ONE device
//               allocation (its landing area) there and opens its peers'.  The same segment, page-locked and mapped into every
//               rank's device address space (hipHostRegister), holds the sequence flags: uncached system memory, so a store
//               of one process's kernel is what the next poll of another's reads.  (Round 4: flags inside landing areas of plain
//               hipMalloc memory -- polled locally, written through the peer mapping -- did NOT work: a system-scope poll never saw
//               the other process's store, even with both ranks on one device.)
//               IRS_IPC_FLAGS=device (round 5) places them in the header of every rank's LANDING AREA instead -- polled locally,
//               raised by the peer through its mapping: on a node a hand-over is then one xGMI store and local polls instead of
//               two PCIe round trips.  With the landing area in UNCACHED device memory this works with two ranks on one device
//               (tests/test_gpu_slab.py) and a hand-over is 10-16 % cheaper (profiles/r05_comm_probe_flags.json).  Whether a poll
//               of device memory sees another DEVICE's store is exactly what a one-GPU box cannot tell: the pre-flight child (ir_sgmcmc_amd/ipc_preflight.py) probes it with a short timeout --
//               a flag that never arrives is a clean error since round 5 -- and bench.py uses device flags only where every
//               rank's child proved them.
//   exchange    push kernel: packs this rank's strips into slot (seq & 1) of each neighbour's landing area; when its last
//               workgroup has drained its stores (system-scope release) it stores `seq` into the neighbours' flags.
//               wait kernel (ONE wavefront): polls this rank's flags until `seq` has arrived -- bounded: a timeout raises a STICKY
//               error word (pinned host memory for ipc_check, a device word for the kernels), after which every drain / reduce
//               kernel of this communicator does nothing and the slab transition in flight ends as a no-op on this rank
//               (scalar_kernels.h: comm_bad -- no parameter, moment, counter or velocity changes), so no stale landing slot is
//               ever consumed; the host gets the error from its next call.  drain kernel behind it: unpacks the landing slot
//               into the ghost planes.  Two slots suffice without credits because every
//               exchange is symmetric on a link: a rank's push of seq + 2 follows its own drain of seq + 1, which waited for the
//               neighbour's push of seq + 1, which that neighbour enqueued behind ITS drain of seq (checked: comm_exchange
//               refuses an asymmetric list for this transport).
//   all-reduce  push kernel: this rank's contribution into slot [seq & 1][rank] of EVERY rank's landing area (its own too), then
//               the flags; wait kernel, then the reduce kernel: combines the contributions IN RANK ORDER (every rank obtains the same
//               bits) and writes the result in place.  An all-reduce is a barrier, so two slots suffice here as well.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <new>

#include "comm.h"
#include "ctx.h"

namespace irs {

namespace {

constexpr uint32_t kMagic = 0x49525331u;  // "IRS1"
constexpr int kMaxWorld = 8;
constexpr int kMaxRuns = 2 * 3 * IRS_MAX_CHAINS;  // sends (or receives) of one exchange_planes call (slab.hip): both sides, planar runs
constexpr size_t kLandHeader = 4096;              // device flags + completion counters in front of the landing slots
constexpr size_t kLine = 128;

// ---- the shared segment ----------------------------------------------------------------------------------------------------
struct ShmRank {
    hipIpcMemHandle_t handle;
    uint64_t land_bytes, x_slot, ar_slot;  // size of the allocation; bytes of one exchange slot (one side); of one all-reduce contribution
    uint32_t gen;                          // generation of the published allocation (0: none)
    int32_t device;
    char pad[kLine - ((sizeof(hipIpcMemHandle_t) + 3 * 8 + 8) % kLine)];
};
struct ShmFlags {  // one rank's incoming flags, each on a line of its own; written by the peer's device, polled by this rank's
    struct { uint32_t v; char pad[kLine - 4]; } x[2];            // exchange sequence from the lower / upper neighbour
    struct { uint32_t v; char pad[kLine - 4]; } ar[kMaxWorld];   // all-reduce sequence from every rank
};
struct Shm {
    uint32_t magic, world;
    uint32_t arrived;  // host barrier (monotonic counter)
    uint32_t failed;   // a rank gave up during bootstrap: the others stop waiting
    uint32_t flags_device_votes;  // ranks that asked for IRS_IPC_FLAGS=device (all or none)
    char pad[kLine - 20];
    ShmRank rank[kMaxWorld];
    ShmFlags flags[kMaxWorld];
};

constexpr size_t kShmBytes = (sizeof(Shm) + 65535) & ~(size_t)65535;
static_assert(sizeof(ShmRank) == kLine && sizeof(ShmFlags) + 128 <= kLandHeader, "shared-segment layout");

double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

}  // namespace

struct IpcState {
    char name[96];
    Shm* shm = nullptr;            // host mapping
    Shm* shm_dev = nullptr;        // the same bytes in the device address space
    bool registered = false;
    int rank = 0, world = 1;
    uint32_t barriers = 0;         // host barriers passed
    uint32_t gen = 0;
    char* land = nullptr;          // my landing area (device)
    size_t land_bytes = 0, x_slot = 0, ar_slot = 0;
    char* peer[kMaxWorld] = {};    // peers' landing areas as mapped here (peer[rank] == land)
    uint64_t peer_x_slot[kMaxWorld] = {}, peer_ar_slot[kMaxWorld] = {};
    bool land_uncached = false;    // the landing area is uncached device memory (else plain hipMalloc)
    char* retired[8] = {};         // landing areas outgrown by a later reservation (freed with the communicator)
    int n_retired = 0;
    uint32_t xseq = 0, arseq = 0;
    unsigned* err = nullptr;       // pinned, device-visible: first timeout (code) raised by a waiting kernel
    unsigned* err_dev = nullptr;
    unsigned* err_flag = nullptr;  // the same fact in DEVICE memory: what the consumer kernels test (no PCIe round trip per launch)
    bool bailing = false;          // this rank failed during the bootstrap: it takes no further part in the arrival counter
    bool device_flags = false;     // IRS_IPC_FLAGS=device: sequence flags in the landing-area headers (else in the host segment)
    unsigned long long timeout_ticks = 0;
    uint64_t exchanges = 0, allreduces = 0;
};

namespace {

// how long a rank waits for the others at a bootstrap step (creation, reservation of the landing area): IRS_IPC_BOOT_TIMEOUT_S,
// default 120 s -- a peer that DIED (it cannot raise `failed` any more) costs the survivors this long
double boot_timeout_s() {
    static double t = 0.0;
    if (t <= 0.0) {
        const char* e = getenv("IRS_IPC_BOOT_TIMEOUT_S");
        t = e && atof(e) > 0.0 ? atof(e) : 120.0;
    }
    return t;
}

int host_barrier(IpcState* s, double timeout_s = 0.0) {
    if (timeout_s <= 0.0) timeout_s = boot_timeout_s();
    Shm* m = s->shm;
    __atomic_add_fetch(&m->arrived, 1u, __ATOMIC_ACQ_REL);
    const uint32_t want = (uint32_t)s->world * (++s->barriers);
    const double t0 = now_s();
    while ((int32_t)(__atomic_load_n(&m->arrived, __ATOMIC_ACQUIRE) - want) < 0) {
        if (__atomic_load_n(&m->failed, __ATOMIC_ACQUIRE)) return fail("ipc transport: another rank failed during the bootstrap");
        if (now_s() - t0 > timeout_s) {
            __atomic_store_n(&m->failed, 1u, __ATOMIC_RELEASE);
            return fail("ipc transport: rank %d waited %.0f s for the other ranks (%u of %u arrivals)", s->rank, timeout_s,
                        __atomic_load_n(&m->arrived, __ATOMIC_ACQUIRE), want);
        }
        usleep(200);
    }
    // (a rank that fails AFTER its arrival has completed this barrier is seen here, not a barrier later)
    if (__atomic_load_n(&m->failed, __ATOMIC_ACQUIRE)) return fail("ipc transport: another rank failed during the bootstrap");
    return 0;
}

int give_up(IpcState* s, int rc) {  // tell the peers before returning an error from a collective step
    if (rc && s->shm) __atomic_store_n(&s->shm->failed, 1u, __ATOMIC_RELEASE);
    return rc;
}

// flag words: where this rank POLLS (its own, written by `src`) and where it SIGNALS (the peer's, naming itself)
inline size_t xflag_off(int side) { return offsetof(ShmFlags, x) + (size_t)side * kLine; }
inline size_t arflag_off(int src) { return offsetof(ShmFlags, ar) + (size_t)src * kLine; }
// (device flags: the first sizeof(ShmFlags) bytes of the owner's landing area, reached through this rank's mapping of it)
unsigned* flag_ptr(IpcState* s, int owner, size_t off) {
    if (s->device_flags) return (unsigned*)(s->peer[owner] + off);
    return (unsigned*)((char*)&s->shm_dev->flags[owner] + off);
}

// ---- device side -----------------------------------------------------------------------------------------------------------
struct Run {
    const char* src;
    char* dst;
    uint64_t bytes;
};
struct Runs {
    int n;
    Run r[kMaxRuns];
};
struct Signals {
    int n;
    unsigned* flag[kMaxWorld];
};

__device__ __forceinline__ unsigned load_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ __forceinline__ bool comm_failed(const unsigned* err_flag) {
    return __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

__device__ __forceinline__ void copy_bytes(const char* __restrict__ src, char* __restrict__ dst, uint64_t bytes, unsigned part, unsigned parts) {
    const unsigned tid = threadIdx.x, nt = blockDim.x;
    if ((((uintptr_t)src | (uintptr_t)dst | bytes) & 15u) == 0) {
        const uint4* s = (const uint4*)src;
        uint4* d = (uint4*)dst;
        const uint64_t n = bytes >> 4;
        uint64_t i = (uint64_t)part * nt + tid;
        const uint64_t step = (uint64_t)parts * nt;
        for (; i + 3 * step < n; i += 4 * step) {  // four independent 16-byte loads in flight per lane
            const uint4 a = s[i], b = s[i + step], c = s[i + 2 * step], e = s[i + 3 * step];
            d[i] = a;
            d[i + step] = b;
            d[i + 2 * step] = c;
            d[i + 3 * step] = e;
        }
        for (; i < n; i += step) d[i] = s[i];
    } else {  // ragged volumes: planes are only 4-byte aligned
        const uint32_t* s = (const uint32_t*)src;
        uint32_t* d = (uint32_t*)dst;
        const uint64_t n = bytes >> 2;
        for (uint64_t i = (uint64_t)part * nt + tid; i < n; i += (uint64_t)parts * nt) d[i] = s[i];
    }
}

// every workgroup: stores drained, system-scope release; the LAST one to arrive stores `seq` into the flags.
// (MI355X_MICROARCH.md, inter-workgroup visibility: every storing wave's vmcnt(0), the barrier, lane-0 release, an explicit
// vmcnt(0) the compiler cannot drop, then the counter / flag.)
// A rank whose communicator has FAILED (a wait of its own timed out) keeps its flags to itself: what it would push derives from
// stale ghost planes, and a peer that never sees the flag times out in turn instead of consuming it.
__device__ void publish(unsigned* done, unsigned total, const Signals& sig, unsigned seq, const unsigned* err_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == total - 1 ? 1u : 0u;
        if (last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
    }
    __syncthreads();
    if (last && threadIdx.x == 0 && !comm_failed(err_flag))
        for (int i = 0; i < sig.n; ++i) __hip_atomic_store(sig.flag[i], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid (parts, runs)
__global__ void __launch_bounds__(256) ipc_push_kernel(Runs runs, Signals sig, unsigned seq, unsigned* done, const unsigned* err_flag) {
    const Run r = runs.r[blockIdx.y];
    copy_bytes(r.src, r.dst, r.bytes, blockIdx.x, gridDim.x);
    publish(done, gridDim.x * gridDim.y, sig, seq, err_flag);
}

// Waiting is ONE wavefront's business: a kernel of its own in front of the consumer, so that what spins while a peer is late is
// 64 lanes with a handful of registers -- not the consumer's whole grid parked on every CU, taking registers and wave slots from
// the interior launches the wait is supposed to overlap with.  The consumer kernel then starts behind it in stream order (its
// kernel-start acquire sees what the producer released before it raised the flag).
struct Waits {
    int n;
    const unsigned* flag[kMaxWorld];
};
__global__ void __launch_bounds__(64) ipc_wait_kernel(Waits w, unsigned seq, unsigned long long timeout, unsigned* err, unsigned* err_flag, unsigned code) {
    // one lane per flag, ALL polling in the same loop: the bound is one timeout whatever the number of peers (wrap-safe
    // sequence comparison; 100 MHz wall clock).  The flag pointers are picked per lane by selects (kernel arguments stay scalar).
    const int lane = (int)threadIdx.x;
    const unsigned* p = w.flag[0];
#pragma unroll
    for (int i = 1; i < kMaxWorld; ++i) p = lane == i ? w.flag[i] : p;
    bool pending = lane < w.n;
    if (comm_failed(err_flag)) return;  // an earlier wait of this communicator timed out: nothing behind it consumes anything
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (pending && (int)(load_flag(p) - seq) >= 0) pending = false;
        if (!__any(pending)) break;
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > timeout) {
            if (pending) {
                __hip_atomic_store(err, code | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(err_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void __launch_bounds__(256) ipc_drain_kernel(Runs runs, const unsigned* err_flag) {
    if (comm_failed(err_flag)) return;  // the wait in front timed out: the landing slot is stale
    const Run r = runs.r[blockIdx.y];
    copy_bytes(r.src, r.dst, r.bytes, blockIdx.x, gridDim.x);
}

struct ArDst {
    char* dst[kMaxWorld];  // slot [seq & 1][me] of every rank's landing area
};
// grid (parts, world): this rank's contribution to every rank
__global__ void __launch_bounds__(256) ipc_ar_push_kernel(const char* buf, uint64_t bytes, ArDst to, Signals sig, unsigned seq, unsigned* done,
                                                          const unsigned* err_flag) {
    copy_bytes(buf, to.dst[blockIdx.y], bytes, blockIdx.x, gridDim.x);
    publish(done, gridDim.x * gridDim.y, sig, seq, err_flag);
}

// kind 0: SUM of doubles, 1: MAX of uint32, 2: SUM of floats; contributions combined in rank order (launched behind ipc_wait_kernel)
__global__ void __launch_bounds__(256) ipc_ar_reduce_kernel(void* buf, uint64_t count, int kind, const char* slot, uint64_t stride, int world,
                                                            const unsigned* err_flag) {
    if (comm_failed(err_flag)) return;  // the wait in front timed out: contributions are missing
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        if (kind == 0) {
            double a = ((const double*)slot)[i];
            for (int r = 1; r < world; ++r) a += ((const double*)(slot + (uint64_t)r * stride))[i];
            ((double*)buf)[i] = a;
        } else if (kind == 1) {
            unsigned a = ((const unsigned*)slot)[i];
            for (int r = 1; r < world; ++r) {
                const unsigned b = ((const unsigned*)(slot + (uint64_t)r * stride))[i];
                a = b > a ? b : a;
            }
            ((unsigned*)buf)[i] = a;
        } else {
            float a = ((const float*)slot)[i];
            for (int r = 1; r < world; ++r) a += ((const float*)(slot + (uint64_t)r * stride))[i];
            ((float*)buf)[i] = a;
        }
    }
}

inline unsigned parts_for(uint64_t bytes) {  // workgroups per run: 16 KB each, at most 64
    const uint64_t p = (bytes + 16383) / 16384;
    return (unsigned)(p < 1 ? 1 : (p > 64 ? 64 : p));
}

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// landing-area geometry (identical arithmetic on both sides of a link, from the OWNER's published slot sizes)
inline size_t x_off(size_t x_slot, int slot, int side) { return kLandHeader + ((size_t)slot * 2 + side) * x_slot; }
inline size_t ar_off(size_t x_slot, size_t ar_slot, int world, int slot, int src) {
    return kLandHeader + 4 * x_slot + ((size_t)slot * world + src) * ar_slot;
}
inline unsigned* done_ptr(IpcState* s, int which) { return (unsigned*)(s->land + sizeof(ShmFlags) + 64 * (size_t)which); }

void close_peers(IpcState* s) {
    for (int r = 0; r < s->world; ++r) {
        if (r != s->rank && s->peer[r]) (void)hipIpcCloseMemHandle(s->peer[r]);
        s->peer[r] = nullptr;
    }
}

}  // namespace

void ipc_describe(const irs_comm* cm, char* out, size_t n) {
    const IpcState* s = cm->ipc;
    snprintf(out, n, "ipc: %d ranks, landing area %.1f MiB %s, sequence flags in %s, %llu exchanges / %llu all-reduces so far",
             s->world, (double)s->land_bytes / 1048576.0, s->land ? (s->land_uncached ? "uncached device memory" : "hipMalloc") : "(not reserved yet)",
             s->device_flags ? "the landing areas (device memory)" : "host shared memory",
             (unsigned long long)s->exchanges, (unsigned long long)s->allreduces);
}

int ipc_check(irs_comm* cm) {
    IpcState* s = cm->ipc;
    if (!s || !s->err) return 0;
    const unsigned e = *(volatile unsigned*)s->err;
    if (!e) return 0;
    return fail("ipc transport: rank %d timed out waiting for a peer (code 0x%x: %s %u, from rank / side %u)", s->rank, e,
                (e >> 28) == 1 ? "exchange" : "all-reduce", (e >> 4) & 0xffffffu, e & 15u);
}

const unsigned* ipc_error_flag(const irs_comm* cm) { return cm && cm->ipc ? cm->ipc->err_flag : nullptr; }

// (Re)allocate the landing area for exchanges of up to `xbytes` per side and all-reduces of up to `arbytes`; collective, blocking.
int ipc_reserve(irs_comm* cm, size_t xbytes, size_t arbytes) {
    IpcState* s = cm->ipc;
    if (!s) return fail("ipc_reserve: not an ipc communicator");
    xbytes = align16(xbytes + 16 * kMaxRuns);  // every run starts on a 16-byte boundary of its slot
    arbytes = align16(arbytes < 256 ? 256 : arbytes);
    if (!s->land) {
        // The first reservation is generous -- 8 MiB per side and slot (a 256^3 slab with the widest ghost zone needs 6.3), 2 MiB per
        // all-reduce contribution; IRS_IPC_SLOT_MB overrides -- so that a run normally never reserves twice: exporting a SECOND
        // allocation after the first one was freed has failed here now and then (hipIpcGetMemHandle: invalid argument).
        const char* mb = getenv("IRS_IPC_SLOT_MB");
        const size_t first = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 8) << 20;
        if (xbytes < first) xbytes = first;
        if (arbytes < ((size_t)2 << 20)) arbytes = (size_t)2 << 20;
    }
    // every rank must take the same decision: the sizes come from the (identical) configuration, and a rank that already holds
    // enough still takes part in the barriers of one that does not -- so all of them re-publish whenever ANY call grows
    const bool grow = xbytes > s->x_slot || arbytes > s->ar_slot || !s->land;
    if (!grow) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return give_up(s, fail("ipc_reserve: hipDeviceSynchronize failed"));
    if (give_up(s, host_barrier(s))) return 1;  // nobody is still writing into an old area
    close_peers(s);
    if (give_up(s, host_barrier(s))) return 1;  // nobody still maps an old area
    // an outgrown area is retired, not freed: a fresh allocation must not land on the address range of one that was exported
    if (s->land) {
        if (s->n_retired >= 8) return give_up(s, fail("ipc_reserve: the landing area was outgrown %d times", s->n_retired));
        s->retired[s->n_retired++] = s->land;
    }
    s->land = nullptr;
    s->x_slot = xbytes > s->x_slot ? xbytes : s->x_slot;
    s->ar_slot = arbytes > s->ar_slot ? arbytes : s->ar_slot;
    s->land_bytes = kLandHeader + 4 * s->x_slot + 2 * (size_t)s->world * s->ar_slot;
    // The landing area is written by OTHER devices' kernels (xGMI stores) and read by this one's: asked for as uncached device
    // memory first (no line of it can sit stale in this device's L2 whatever the fabric does about probes -- what RCCL does with
    // its own buffers); plain hipMalloc when that kind cannot be allocated or exported (IRS_IPC_LANDING=default skips the attempt).
    ShmRank& me = s->shm->rank[s->rank];
    hipError_t e = hipErrorUnknown;
    const char* kind = getenv("IRS_IPC_LANDING");
    s->land_uncached = false;
    if (!(kind && !strcmp(kind, "default"))) {
        e = hipExtMallocWithFlags((void**)&s->land, s->land_bytes, hipDeviceMallocUncached);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&me.handle, s->land);
        if (e == hipSuccess) s->land_uncached = true;
        else {
            if (s->land) (void)hipFree(s->land);
            s->land = nullptr;
            (void)hipGetLastError();
        }
    }
    if (!s->land) {
        e = hipMalloc((void**)&s->land, s->land_bytes);
        if (e != hipSuccess) return give_up(s, fail("ipc_reserve: hipMalloc of %zu bytes failed: %s", s->land_bytes, hipGetErrorString(e)));
        e = hipIpcGetMemHandle(&me.handle, s->land);
    }
    if (hipMemset(s->land, 0, kLandHeader) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return give_up(s, fail("ipc_reserve: clearing the landing header failed"));  // (the peers stop waiting at once)
    // (device flags live in that header: a fresh area starts at zero while the sequence numbers go on -- every later flag value is
    // larger than anything waited for before, and nobody waits across a reservation: it is collective and drains the device first)
    if (e != hipSuccess) return give_up(s, fail("hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed where the driver only has dmabuf IPC)", hipGetErrorString(e)));
    me.land_bytes = s->land_bytes;
    me.x_slot = s->x_slot;
    me.ar_slot = s->ar_slot;
    (void)hipGetDevice(&me.device);
    __atomic_store_n(&me.gen, ++s->gen, __ATOMIC_RELEASE);
    if (give_up(s, host_barrier(s))) return 1;  // every handle is published
    for (int r = 0; r < s->world; ++r) {
        const ShmRank& pr = s->shm->rank[r];
        s->peer_x_slot[r] = pr.x_slot;
        s->peer_ar_slot[r] = pr.ar_slot;
        if (r == s->rank) {
            s->peer[r] = s->land;
            continue;
        }
        void* p = nullptr;
        hipIpcMemHandle_t h = pr.handle;
        e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return give_up(s, fail("hipIpcOpenMemHandle (rank %d <- rank %d, device %d) failed: %s", s->rank, r, pr.device, hipGetErrorString(e)));
        s->peer[r] = (char*)p;
    }
    return give_up(s, host_barrier(s));  // every mapping is open: the area may be written
}

int ipc_exchange(irs_comm* cm, const irs_xfer* x, int n, hipStream_t st) {
    IpcState* s = cm->ipc;
    if (!s->land) return fail("ipc transport: nothing reserved (irs_slab_create / irs_comm_selftest reserve the landing area)");
    if (n > 4 * kMaxRuns) return fail("ipc exchange: %d transfers", n);
    if (ipc_check(cm)) return 1;
    const uint32_t seq = ++s->xseq;
    const int slot = (int)(seq & 1u);
    Runs push, drain;
    push.n = drain.n = 0;
    Signals sig;
    sig.n = 0;
    Waits wt;
    wt.n = 0;
    for (int i = 0; i < kMaxWorld; ++i) wt.flag[i] = nullptr;
    // my neighbours: side 0 = the lower one (rank - 1), side 1 = the upper one.  At the UPPER neighbour I am its lower side (0).
    for (int side = 0; side < 2; ++side) {
        const int peer = side == 0 ? s->rank - 1 : s->rank + 1;
        size_t so = 0, ro = 0;
        int ns = 0, nr = 0;
        for (int i = 0; i < n; ++i) {
            if (x[i].peer != peer) continue;
            if (peer < 0 || peer >= s->world) return fail("ipc exchange: peer %d out of range", peer);
            if (x[i].bytes & 3u) return fail("ipc exchange: transfer of %zu bytes", x[i].bytes);
            if (!x[i].recv) {
                if (push.n >= kMaxRuns) return fail("ipc exchange: too many sends");
                if (so + x[i].bytes > s->peer_x_slot[peer]) return fail("ipc exchange: %zu bytes for a landing slot of %llu (rank %d)", so + x[i].bytes, (unsigned long long)s->peer_x_slot[peer], peer);
                push.r[push.n++] = Run{(const char*)x[i].ptr, s->peer[peer] + x_off(s->peer_x_slot[peer], slot, 1 - side) + so, x[i].bytes};
                so += align16(x[i].bytes);
                ++ns;
            } else {
                if (drain.n >= kMaxRuns) return fail("ipc exchange: too many receives");
                if (ro + x[i].bytes > s->x_slot) return fail("ipc exchange: %zu bytes from a landing slot of %zu", ro + x[i].bytes, s->x_slot);
                drain.r[drain.n++] = Run{s->land + x_off(s->x_slot, slot, side) + ro, (char*)x[i].ptr, x[i].bytes};
                ro += align16(x[i].bytes);
                ++nr;
            }
        }
        // two landing slots need no credits only while every exchange is symmetric on a link (header of this file)
        if (ns != nr || so != ro) return fail("ipc exchange: %d sends (%zu bytes) but %d receives (%zu bytes) with rank %d: this transport carries symmetric neighbour exchanges", ns, so, nr, ro, peer);
        if (ns) {
            sig.flag[sig.n++] = flag_ptr(s, peer, xflag_off(1 - side));
            wt.flag[wt.n++] = flag_ptr(s, s->rank, xflag_off(side));
        }
    }
    for (int i = 0; i < n; ++i)
        if (x[i].peer != s->rank - 1 && x[i].peer != s->rank + 1) return fail("ipc exchange: rank %d is not a neighbour of rank %d", x[i].peer, s->rank);
    if (!push.n) return 0;
    uint64_t most = 0;
    for (int i = 0; i < push.n; ++i) most = push.r[i].bytes > most ? push.r[i].bytes : most;
    const unsigned parts = parts_for(most);
    hipLaunchKernelGGL(ipc_push_kernel, dim3(parts, push.n), dim3(256), 0, st, push, sig, seq, done_ptr(s, 0), (const unsigned*)s->err_flag);
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, st, wt, seq, s->timeout_ticks, s->err_dev, s->err_flag, (1u << 28) | ((seq & 0xffffffu) << 4));
    hipLaunchKernelGGL(ipc_drain_kernel, dim3(parts, drain.n), dim3(256), 0, st, drain, (const unsigned*)s->err_flag);
    LAUNCH_CHECK();
    ++s->exchanges;
    return 0;
}

int ipc_allreduce(irs_comm* cm, void* buf, size_t count, int kind, hipStream_t st) {
    IpcState* s = cm->ipc;
    if (!s->land) return fail("ipc transport: nothing reserved (irs_slab_create / irs_comm_selftest reserve the landing area)");
    if (ipc_check(cm)) return 1;
    const size_t bytes = count * (kind == 0 ? 8 : 4);
    for (int r = 0; r < s->world; ++r)
        if (bytes > s->peer_ar_slot[r]) return fail("ipc all-reduce: %zu bytes for a slot of %llu (rank %d)", bytes, (unsigned long long)s->peer_ar_slot[r], r);
    const uint32_t seq = ++s->arseq;
    const int slot = (int)(seq & 1u);
    ArDst to;
    Waits wt;
    Signals sig;
    sig.n = wt.n = 0;
    for (int r = 0; r < kMaxWorld; ++r) to.dst[r] = nullptr, wt.flag[r] = nullptr;
    for (int r = 0; r < s->world; ++r) {
        to.dst[r] = s->peer[r] + ar_off(s->peer_x_slot[r], s->peer_ar_slot[r], s->world, slot, s->rank);
        if (r == s->rank) continue;
        sig.flag[sig.n++] = flag_ptr(s, r, arflag_off(s->rank));
        wt.flag[wt.n++] = flag_ptr(s, s->rank, arflag_off(r));
    }
    const unsigned parts = parts_for(bytes);
    hipLaunchKernelGGL(ipc_ar_push_kernel, dim3(parts, s->world), dim3(256), 0, st, (const char*)buf, (uint64_t)bytes, to, sig, seq, done_ptr(s, 1), (const unsigned*)s->err_flag);
    const uint64_t per_block = 256 * 8;
    unsigned blocks = (unsigned)((count + per_block - 1) / per_block);
    blocks = blocks < 1 ? 1 : (blocks > 128 ? 128 : blocks);
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, st, wt, seq, s->timeout_ticks, s->err_dev, s->err_flag, (2u << 28) | ((seq & 0xffffffu) << 4));
    hipLaunchKernelGGL(ipc_ar_reduce_kernel, dim3(blocks), dim3(256), 0, st, buf, (uint64_t)count, kind,
                       (const char*)(s->land + ar_off(s->x_slot, s->ar_slot, s->world, slot, 0)), (uint64_t)s->ar_slot, s->world,
                       (const unsigned*)s->err_flag);
    LAUNCH_CHECK();
    ++s->allreduces;
    return 0;
}

void ipc_destroy(irs_comm* cm) {
    IpcState* s = cm->ipc;
    if (!s) return;
    (void)hipDeviceSynchronize();
    close_peers(s);
    if (s->shm) {
        // the owner of a landing area frees it only when nobody maps it any more; a peer that has already died is not waited for long
        // (a rank that failed during the bootstrap only raises `failed`: an extra arrival could complete the barrier the healthy
        // ranks are in and let them return success next to a dead peer)
        if (s->bailing) __atomic_store_n(&s->shm->failed, 1u, __ATOMIC_RELEASE);
        else (void)host_barrier(s, 10.0);
        if (s->registered) (void)hipHostUnregister(s->shm);
        (void)munmap(s->shm, kShmBytes);
    }
    if (s->land) (void)hipFree(s->land);
    for (int i = 0; i < s->n_retired; ++i) (void)hipFree(s->retired[i]);
    if (s->err) (void)hipHostFree(s->err);
    if (s->err_flag) (void)hipFree(s->err_flag);
    delete s;
    cm->ipc = nullptr;
}

int ipc_create(const char* name, int rank, int world, irs_comm** out) {
    if (!name || !*name || !out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
        return fail("irs_comm_create_ipc: bad arguments (1 <= world <= %d)", kMaxWorld);
    if (strlen(name) >= sizeof(IpcState::name) - 1) return fail("irs_comm_create_ipc: name too long");
    IpcState* s = new (std::nothrow) IpcState();
    irs_comm* c = new (std::nothrow) irs_comm();
    if (!s || !c) {
        delete s;
        delete c;
        return fail("irs_comm_create_ipc: out of host memory");
    }
    c->kind = 2;
    c->rank = s->rank = rank;
    c->world = s->world = world;
    c->ipc = s;
    snprintf(s->name, sizeof(s->name), "%s%s", name[0] == '/' ? "" : "/", name);
    auto bail = [&](int rc) {
        s->bailing = true;
        ipc_destroy(c);
        delete c;
        return rc;
    };
    // rank 0 creates the segment, the others wait for it to appear and to carry the magic word
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        (void)shm_unlink(s->name);  // a leftover of a run that died
        fd = shm_open(s->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return bail(fail("shm_open(%s) failed: %s", s->name, strerror(errno)));
        if (ftruncate(fd, kShmBytes) != 0) {
            close(fd);
            return bail(fail("ftruncate(%s) failed: %s", s->name, strerror(errno)));
        }
    } else {
        while ((fd = shm_open(s->name, O_RDWR, 0600)) < 0) {
            if (now_s() - t0 > 120.0) return bail(fail("shm_open(%s): rank 0 has not created the segment after 120 s", s->name));
            usleep(1000);
        }
        struct stat sb;
        while (fstat(fd, &sb) == 0 && (size_t)sb.st_size < kShmBytes) {
            if (now_s() - t0 > 120.0) {
                close(fd);
                return bail(fail("ipc transport: the segment %s never reached its size", s->name));
            }
            usleep(1000);
        }
    }
    void* p = mmap(nullptr, kShmBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return bail(fail("mmap(%s) failed: %s", s->name, strerror(errno)));
    s->shm = (Shm*)p;
    if (rank == 0) {
        s->shm->world = (uint32_t)world;
        __atomic_store_n(&s->shm->magic, kMagic, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&s->shm->magic, __ATOMIC_ACQUIRE) != kMagic) {
            if (now_s() - t0 > 120.0) return bail(fail("ipc transport: the segment %s was never initialised", s->name));
            usleep(200);
        }
        if (s->shm->world != (uint32_t)world) return bail(give_up(s, fail("ipc transport: world %d here, %u on rank 0", world, s->shm->world)));
    }
    if (give_up(s, host_barrier(s))) return bail(1);  // everyone has the segment mapped ...
    if (rank == 0) (void)shm_unlink(s->name);         // ... so the name can go: nothing is left behind in /dev/shm
    // the segment in the device address space (sequence flags), the error word, the timeout of a waiting kernel
    hipError_t e = hipHostRegister(s->shm, kShmBytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) return bail(give_up(s, fail("hipHostRegister of the shared segment failed: %s", hipGetErrorString(e))));
    s->registered = true;
    e = hipHostGetDevicePointer((void**)&s->shm_dev, s->shm, 0);
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->err, 64, hipHostMallocMapped | hipHostMallocPortable);
    if (e == hipSuccess) {
        *s->err = 0;
        e = hipHostGetDevicePointer((void**)&s->err_dev, s->err, 0);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&s->err_flag, 64);
    if (e == hipSuccess) e = hipMemset(s->err_flag, 0, 64);
    if (e != hipSuccess) return bail(give_up(s, fail("ipc transport: pinned host memory / error word failed: %s", hipGetErrorString(e))));
    // How long a kernel waits for a peer's flag.  It bounds the SKEW between ranks, host pauses included: a rank whose host stops
    // enqueueing for longer than this between two transitions (checkpoint, metrics I/O) makes its neighbours give up -- raise
    // IRS_IPC_TIMEOUT_S for such runs.  A timeout is fail-safe (header of this file), never a corrupted chain.
    const char* fl = getenv("IRS_IPC_FLAGS");
    s->device_flags = fl && !strcmp(fl, "device");
    const char* to = getenv("IRS_IPC_TIMEOUT_S");
    const double secs = to && atof(to) > 0.0 ? atof(to) : 60.0;
    s->timeout_ticks = (unsigned long long)(secs * 100.0e6);  // wall_clock64: 100 MHz
    // every rank must have chosen the same flag placement (an environment variable: easy to set on one rank only)
    __atomic_fetch_add(&s->shm->flags_device_votes, s->device_flags ? 1u : 0u, __ATOMIC_ACQ_REL);
    if (give_up(s, host_barrier(s))) return bail(1);
    const uint32_t votes = __atomic_load_n(&s->shm->flags_device_votes, __ATOMIC_ACQUIRE);
    if (votes != 0u && votes != (uint32_t)world)
        return bail(give_up(s, fail("ipc transport: IRS_IPC_FLAGS=device on %u of %d ranks -- every rank must place the flags alike", votes, world)));
    if (give_up(s, host_barrier(s))) return bail(1);
    *out = c;
    return 0;
}

}  // namespace irs
