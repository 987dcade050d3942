"""z-slab decomposition of ONE chain over the ranks of a node (BASELINE.json config 4).

The schedule -- ghost-plane exchanges, communication-avoiding rounds of squaring steps, interior / boundary overlap, the
small all-reduces -- lives in the library (`irs_slab_transition`, csrc/slab.hip): this module only creates the
communicator, cuts the caller's arrays to the planes a rank holds and hands pointers over.  What is sharded is the
single-device loop body of the reference, trainer/trainer.py:291-356.

    comm = SlabComm.rccl()                      # one process per GPU, torch.distributed initialised (backend nccl = RCCL)
    eng = SlabEngine(cfg, device, comm)
    fixed_l, moving = eng.prepare(fixed, moving) # fixed image / mask cut to the held planes, moving image stays whole
    v_l = eng.local(v)                           # (C, 3, hi - lo, H, W)
    eng.transition(fixed_l, moving, v_l)

`SlabComm.ipc()` is the peer-mapped transport (landing buffers exported with hipIpcGetMemHandle, written by the producer,
ordered by sequence flags): asynchronous like RCCL, and several ranks may share ONE GPU, which RCCL refuses.
`SlabComm.rehearsal()` is the same schedule over a transport of Python callbacks (torch.distributed point-to-point with
host staging, synchronous): tests only.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import EngineConfig, TransitionEngine, _on_device, irs_config


class _RawDevice:
    """__cuda_array_interface__ shim: lets torch view device memory owned by the C library"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False), 'version': 2}


class SlabComm:
    """owner of an `irs_comm` (include/irsgmcmc.h)"""

    def __init__(self, handle, rank, world, keep=None):
        self.handle, self.rank, self.world, self._keep = handle, rank, world, keep
        self.lib = L.load()

    @classmethod
    def rccl(cls, device=None):
        """RCCL communicator over the ranks of the initialised torch.distributed group: rank 0 draws the unique id, the
        group broadcasts it, every rank joins on its current device (collective, blocking)."""
        lib = L.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = (C.c_uint8 * L.IRS_COMM_ID_BYTES)()
        if rank == 0:
            L.check(lib.irs_comm_unique_id(C.byref(uid)))
        on_gpu = dist.get_backend() == 'nccl'
        t = torch.tensor(list(uid), dtype=torch.uint8, device=(device or torch.device('cuda', torch.cuda.current_device())) if on_gpu else 'cpu')
        dist.broadcast(t, src=0)
        uid = (C.c_uint8 * L.IRS_COMM_ID_BYTES)(*t.cpu().tolist())
        h = C.c_void_p()
        L.check(lib.irs_comm_create_rccl(C.byref(uid), rank, world, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def ipc(cls, name=None):
        """peer-mapped transport (csrc/ipc.hip): every rank exports a landing area with hipIpcGetMemHandle, producers store ghost
        planes straight into their neighbours' (xGMI stores on a node), sequence flags order the processes -- no host or stream
        synchronisation inside an exchange.  Ranks may share a device: the asynchronous schedule then runs for real on a one-GPU
        box.  Rank 0 picks the name of the bootstrap segment, the initialised torch.distributed group (any backend) carries it."""
        import os
        lib = L.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [name or f'irs_ipc_{os.getpid()}_{int.from_bytes(os.urandom(4), "little"):08x}'] if rank == 0 else [None]
        if dist.get_backend() == 'nccl':  # (object collectives of the nccl backend stage through the current device)
            dist.broadcast_object_list(box, src=0, device=torch.device('cuda', torch.cuda.current_device()))
        else:
            dist.broadcast_object_list(box, src=0)
        h = C.c_void_p()
        L.check(lib.irs_comm_create_ipc(box[0].encode(), rank, world, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def create(cls, transport, device=None):
        """by name: 'rccl' | 'ipc' | 'rehearsal'"""
        if transport in ('rccl', 'nccl'):
            return cls.rccl(device)
        if transport == 'ipc':
            return cls.ipc()
        if transport == 'rehearsal':
            return cls.rehearsal(device or torch.device('cuda', torch.cuda.current_device()))
        raise L.IrsError(f'unknown slab transport {transport!r}')

    @classmethod
    def rehearsal(cls, device):
        """callback transport: the library's exchanges / all-reduces are carried out by torch.distributed (any backend) with
        host staging, synchronously.  Tests only."""
        lib = L.load()
        rank, world = dist.get_rank(), dist.get_world_size()
        device = torch.device(device)

        def view(ptr, nbytes):
            return torch.as_tensor(_RawDevice(ptr, nbytes), device=device)

        def wait(stream):  # a NULL hipStream_t arrives as None
            if stream:
                torch.cuda.ExternalStream(stream, device=device).synchronize()
            else:
                torch.cuda.synchronize(device)

        def exchange(user, xfers, n, stream):
            try:
                wait(stream)
                ops, landing = [], []
                for i in range(n):
                    x = xfers[i]
                    if x.recv:
                        buf = torch.empty(x.bytes, dtype=torch.uint8)
                        ops.append(dist.P2POp(dist.irecv, buf, x.peer))
                        landing.append((x.ptr, buf))
                    else:
                        ops.append(dist.P2POp(dist.isend, view(x.ptr, x.bytes).cpu(), x.peer))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                for ptr, buf in landing:
                    view(ptr, buf.numel()).copy_(buf)
                torch.cuda.synchronize(device)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                print(f'[slab rehearsal] exchange failed: {e!r}', flush=True)
                return 1

        def allreduce(user, buf, count, max_u32, stream):
            try:
                wait(stream)
                # kind 0: SUM of doubles, 1: MAX of uint32 (non-negative float bits order like int32), 2: SUM of floats
                dtype = {0: torch.float64, 1: torch.int32, 2: torch.float32}[max_u32]
                raw = view(buf, count * (8 if max_u32 == 0 else 4))
                host = raw.cpu().view(dtype)
                dist.all_reduce(host, op=dist.ReduceOp.MAX if max_u32 == 1 else dist.ReduceOp.SUM)
                raw.copy_(host.view(torch.uint8))
                torch.cuda.synchronize(device)
                return 0
            except Exception as e:
                print(f'[slab rehearsal] all-reduce failed: {e!r}', flush=True)
                return 1

        ex, ar = L.EXCHANGE_FN(exchange), L.ALLREDUCE_FN(allreduce)
        h = C.c_void_p()
        L.check(lib.irs_comm_create_callbacks(ex, ar, None, rank, world, C.byref(h)))
        return cls(h, rank, world, keep=(ex, ar))

    def describe(self):
        buf = C.create_string_buffer(256)
        L.check(self.lib.irs_comm_describe(self.handle, buf, 256))
        return buf.value.decode()

    def probe(self, nbytes, ar_doubles=21, iters=200):
        """microseconds per neighbour exchange of `nbytes` per direction and per all-reduce of `ar_doubles` doubles (back to back)"""
        us = (C.c_double * 2)()
        L.check(self.lib.irs_comm_probe(self.handle, nbytes, ar_doubles, iters, L.stream_ptr(), C.byref(us)))
        return us[0], us[1]

    def selftest(self):
        L.check(self.lib.irs_comm_selftest(self.handle, L.stream_ptr()))

    def close(self):
        h, self.handle = self.handle, None
        if h:
            self.lib.irs_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_layout(cfg: EngineConfig, rank, world, ghost_max=0, margin=0):
    """planes a rank owns / holds (pure host arithmetic, no GPU): dict a, b, lo, hi, margin, ghost_max"""
    lib = L.load()
    c, sc, out = irs_config(cfg), L.IrsSlabConfig(ghost_max, margin), L.IrsSlabLayout()
    L.check(lib.irs_slab_plan_layout(C.byref(c), C.byref(sc), rank, world, C.byref(out)))
    return {n: getattr(out, n) for n, _ in out._fields_}


def plan_rounds(h, ghost_max=4, min_slab=1 << 30, n_buffers=3):
    """exchange rounds of the squaring steps for per-step ghost widths h (pure host arithmetic; csrc/slab.hip: plan_rounds);
    n_buffers: gradient fields the adjoint rotates through (3 in a context of several ranks, 2 otherwise)"""
    lib = L.load()
    n = len(h)
    arr = lambda: (C.c_int32 * 32)()
    hh, fr, fw, br, bw = (C.c_int32 * n)(*h), arr(), arr(), arr(), arr()
    nf, nb = C.c_int32(), C.c_int32()
    L.check(lib.irs_slab_plan_rounds(hh, n, ghost_max, min_slab, n_buffers, fr, fw, C.byref(nf), br, bw, C.byref(nb)))
    return {'fwd_round': list(fr[:n]), 'fwd_width': list(fw[:nf.value]), 'bwd_round': list(br[:n]), 'bwd_width': list(bw[:nb.value])}


def trace(cfg: EngineConfig, rank, world, h, ghost_max=0, margin=0):
    """the operation list of one planned transition of `rank` (pure host arithmetic: what irs_slab_transition executes):
    list of dicts with the fields of irs_slab_op"""
    lib = L.load()
    c, sc = irs_config(cfg), L.IrsSlabConfig(ghost_max, margin)
    hh, n = (C.c_int32 * len(h))(*h), C.c_int32()
    L.check(lib.irs_slab_trace(C.byref(c), C.byref(sc), rank, world, hh, None, 0, C.byref(n)))
    ops = (L.IrsSlabOp * n.value)()
    L.check(lib.irs_slab_trace(C.byref(c), C.byref(sc), rank, world, hh, ops, n.value, C.byref(n)))
    return [{f: getattr(o, f) for f, _ in o._fields_} for o in ops]


class SlabEngine(TransitionEngine):
    """one rank's share of a chain: same surface as TransitionEngine, slab-local tensors (moving image whole; for SVFFD_3D the
    control-grid tensors v / sigma / eps / curr_state / grad_v whole too: `local_v`)"""

    def __init__(self, cfg: EngineConfig, device='cuda:0', comm: SlabComm = None, ghost_max=0, margin=0):
        self.comm = comm
        self._scfg = L.IrsSlabConfig(ghost_max, margin)
        super().__init__(cfg, device)
        lay = L.IrsSlabLayout()
        L.check(self.lib.irs_slab_get_layout(self._ctx, C.byref(lay)))
        self.rank, self.world = lay.rank, lay.world
        self.a, self.b, self.lo, self.hi = lay.a, lay.b, lay.lo, lay.hi
        self.margin, self.ghost_max = lay.margin, lay.ghost_max

    def _create(self, c):
        ctx = C.c_void_p()
        L.check(self.lib.irs_slab_create(C.byref(c), C.byref(self._scfg), self.comm.handle if self.comm else None, C.byref(ctx)))
        return ctx

    # ------------------------------------------------------------------ cutting / assembling
    def local(self, t):
        """the held planes [lo, hi) of a full (C, ch, D, H, W) tensor, contiguous, on the engine's device"""
        return t[:, :, self.lo:self.hi].to(self.device).contiguous()

    def local_v(self, t):
        """a velocity-grid tensor (v, sigma, eps, curr_state, grad_v) as the engine wants it: cut to the held planes for SVF_3D;
        WHOLE for SVFFD_3D, whose control grid is replicated on every rank"""
        return t.to(self.device).contiguous() if self.cfg.cps else self.local(t)

    def new_local(self, channels, dtype=torch.float32):
        return torch.zeros(self.cfg.no_chains, channels, self.hi - self.lo, *self.cfg.dims[1:], device=self.device, dtype=dtype)

    def owned(self, t_local):
        """view of the owned planes [a, b) of a slab-local tensor"""
        return t_local[:, :, self.a - self.lo:self.b - self.lo]

    def gather(self, t_local):
        """assemble the full (C, ch, D, H, W) tensor from every rank's owned planes (checks / logging); collective, on the host"""
        self.flush()  # transitions dropped by a failed ghost-width plan are re-run first (collective: same verdict on every rank)
        part = self.owned(t_local).cpu().contiguous()
        if self.world == 1:
            return part
        parts = [None] * self.world
        dist.all_gather_object(parts, part)
        return torch.cat(parts, dim=2)

    # ------------------------------------------------------------------ data
    @_on_device
    def prepare(self, fixed, moving):
        """fixed: dict of FULL or already slab-local `im` / `mask`; moving: dict with the FULL `im`"""
        D, nloc = self.cfg.dims[0], self.hi - self.lo
        cut = lambda t: t if t.shape[2] == nloc and nloc != D else self.local(self._base(t))
        fixed = {k: cut(v) for k, v in fixed.items() if k in ('im', 'mask')}
        moving = {'im': self._base(moving['im']).to(self.device).contiguous()}
        self._keep['fixed'], self._keep['moving'] = fixed, moving
        L.check(self.lib.irs_set_fixed(self._ctx, L.dev_ptr(fixed['im'], torch.float32), fixed['im'].shape[0], self._stream()))
        return fixed, moving

    @_on_device
    def gmm_init(self, fixed, moving, v_sample=None, warm_up=25):
        io = self._io(fixed, moving, None)
        L.check(self.lib.irs_slab_gmm_init(self._ctx, C.byref(io), L.dev_ptr(v_sample, torch.float32, allow_none=True), warm_up,
                                           self._stream()))

    @_on_device
    def transition(self, fixed, moving, v, sigma=None, eps=None, unif=None, outputs=None, timed=False):
        if timed:
            raise L.IrsError('per-stage timings come from the fused engine (TransitionEngine.transition(timed=True))')
        io = self._io(fixed, moving, v, sigma, eps, unif, outputs)
        L.check(self.lib.irs_slab_transition(self._ctx, C.byref(io), self._stream()))
        return None

    def timeline_arm(self, transitions=1):
        """the next `transitions` transitions record timing events around every exchange / all-reduce (the last one is kept)"""
        L.check(self.lib.irs_slab_timeline_arm(self._ctx, int(transitions)))

    @_on_device
    def timeline(self):
        """hand-overs of the last sampled transition, in schedule order: (entries, total_us).  Per entry: when the data was ready on
        the compute stream, how long the hand-over took on the communication stream (push, waiting for the peer, drain / reduce),
        when the compute stream reached the launch that needs it and how long it STALLED there (0: hidden behind interior work)"""
        n, tot = C.c_int32(0), C.c_float(0.0)
        buf = (L.IrsSlabTimelineEntry * 128)()
        L.check(self.lib.irs_slab_timeline_get(self._ctx, buf, 128, C.byref(n), C.byref(tot), self._stream()))
        names = {L.IRS_OP_EXCHANGE: 'exchange', L.IRS_OP_ALLREDUCE: 'all-reduce'}
        return [{'kind': names.get(e.kind, str(e.kind)), 'stage': e.stage, 'k': e.k, 'width': e.width, 'ready_us': round(e.ready_us, 1),
                 'handover_us': round(e.handover_us, 1), 'wait_at_us': round(e.wait_at_us, 1), 'stall_us': round(e.stall_us, 1)}
                for e in buf[:min(n.value, 128)]], float(tot.value)

    @_on_device
    def status(self):
        st = L.IrsSlabStatus()
        L.check(self.lib.irs_slab_status_get(self._ctx, C.byref(st), self._stream()))
        return {n: getattr(st, n) for n, _ in st._fields_}
