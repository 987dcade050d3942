"""z-slab decomposition of ONE chain over the ranks of a node (BASELINE.json config 4).

Every rank holds the full-size arrays (288 GB of HBM make that a non-issue) with global indexing, owns the planes
[a, b) of the volume and runs every stage of the transition on its slab through `irs_stage` (include/irsgmcmc.h).
Between stages the ranks exchange exactly the ghost planes the next stage reads:

  stencil halos      s + 1 planes of the perturbed velocity, 4 s planes of the warped image (LCC map + its adjoint)
  gather halos       before squaring step k: floor(max|d_k,z|) + 1 planes of d_k -- the width is EXACT, taken from the
                     displacement bound the forward kernels publish (all-reduced, MAX); the reference's "one-voxel halo"
                     holds for the early steps only
  adjoint halos      before adjoint step k: the same number of planes of the incoming gradient (the owner-computes gather
                     reads neighbouring sources instead of scattering into neighbouring slabs, so no reverse accumulate)
  scalars            three small all-reduces (SUM): regulariser energy, VD/GMM statistics (per chain), data term

Transport is torch.distributed point-to-point (`batch_isend_irecv`): backend "nccl" (= RCCL over xGMI) on a multi-GPU
node, "gloo" with host staging in the single-GPU rehearsal test.  Scalars end up identical on every rank (same reduced
inputs, deterministic scalar kernels), so the hyper-parameter state stays replicated without further traffic.
SVF_3D only.  The per-stage host orchestration costs a few synchronisations per squaring step (the halo width is read
back); folding the exchange into the library with RCCL calls on the compute stream is the next step.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import EngineConfig, TransitionEngine

(ST_BEGIN, ST_PERTURB, ST_SMOOTH, ST_ENERGY, ST_REG_SCALAR, ST_EXP_FWD, ST_OUTPUTS, ST_WARP, ST_RESIDUAL, ST_STATS,
 ST_CHAIN_SCALAR, ST_DATA_BWD, ST_WARP_BWD, ST_EXP_BWD, ST_UPDATE, ST_FINALIZE) = range(16)
(BUF_NOISY, BUF_STEP, BUF_GRAD_A, BUF_GRAD_B, BUF_SIGMA_M, BUF_DMAX, BUF_STAT_SUM, BUF_ENERGY_SUM, BUF_NLL_SUM) = range(9)


class _RawDevice:
    """__cuda_array_interface__ shim: lets torch view a device pointer owned by the C library"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


def slab_bounds(D, world, rank):
    return (rank * D) // world, ((rank + 1) * D) // world


class SlabEngine(TransitionEngine):
    def __init__(self, cfg: EngineConfig, device='cuda:0'):
        if cfg.cps:
            raise L.IrsError('the slab decomposition supports SVF_3D only')
        super().__init__(cfg, device)
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.host_staging = self.on and dist.get_backend() != 'nccl'
        D = cfg.dims[0]
        self.a, self.b = slab_bounds(D, self.world, self.rank)
        self.min_slab = min(slab_bounds(D, self.world, r)[1] - slab_bounds(D, self.world, r)[0] for r in range(self.world))
        C_, dims = cfg.no_chains, tuple(cfg.dims)
        self._field = lambda which, idx=0: self._view(which, idx, (C_, 3, *dims), '<f4')
        self._field_aos = lambda which, idx=0: self._view(which, idx, (C_, *dims, 3), '<f4')
        self.noisy = self._field(BUF_NOISY)
        # fields that only the squaring-step kernels touch are interleaved (C,D,H,W,3) (include/irsgmcmc.h: irs_layout): a
        # run of ghost planes is then one contiguous block per chain.  (tensor view, index of the z axis) per buffer:
        lay = lambda what, k: int(self.lib.irs_layout(self._ctx, what, k))
        self.steps = [(self._field_aos(BUF_STEP, k), 1) if lay(0, k) == 1 else (self._field(BUF_STEP, k), 2) for k in range(cfg.no_steps)]
        self._grad_planar = {BUF_GRAD_A: self._field(BUF_GRAD_A), BUF_GRAD_B: self._field(BUF_GRAD_B)}
        self._grad_aos = {BUF_GRAD_A: self._field_aos(BUF_GRAD_A), BUF_GRAD_B: self._field_aos(BUF_GRAD_B)}
        self._grad_in_aos = [lay(1, k) == 1 for k in range(cfg.no_steps)]
        self._grad_in_buf = []
        for k in range(cfg.no_steps):
            ib, ob = C.c_int(), C.c_int()
            L.check(self.lib.irs_grad_buffers(self._ctx, k, C.byref(ib), C.byref(ob)))
            self._grad_in_buf.append(ib.value)
        self.dmax = [self._view(BUF_DMAX, k, (C_, 4), '<f4') for k in range(cfg.no_steps + 1)]
        self.stat_sum = self._view(BUF_STAT_SUM, 0, (21,), '<f8')
        self.energy_sum = self._view(BUF_ENERGY_SUM, 0, (C_,), '<f8')
        self.nll_sum = self._view(BUF_NLL_SUM, 0, (C_,), '<f8')
        self.exchanged_planes = 0  # bookkeeping for tests / reports

    # ------------------------------------------------------------------ plumbing
    def _view(self, which, index, shape, typestr):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        L.check(self.lib.irs_buffer(self._ctx, which, index, C.byref(ptr), C.byref(nbytes)))
        return torch.as_tensor(_RawDevice(ptr.value, shape, typestr), device=self.device)

    def _stage(self, io, stage, k=0, lo=None, hi=None):
        lo = self.a if lo is None else lo
        hi = self.b if hi is None else hi
        # ~60 stage calls per transition: the stream handle and the io reference are looked up once per transition
        if self._irs_stage(self._ctx, self._io_ref, stage, k, int(lo), int(hi), self._stream):
            L.check(1)

    def _allreduce(self, t, op):
        if not self.on or self.world == 1:
            return
        if self.host_staging:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def _halo(self, t, h, zdim=2):
        """make planes [a-h, a) and [b, b+h) of `t` valid by receiving them from the neighbouring ranks; `zdim` is the index
        of the z axis: 2 for planar (C, ch, D, H, W) tensors, 1 for interleaved (C, D, H, W, ch) ones.
        A run of planes that is one contiguous block (interleaved fields of a single chain) is sent from / received into the
        field itself; otherwise it goes through a packed copy.  gloo (rehearsal on one GPU) stages through the host."""
        if not self.on or self.world == 1 or h <= 0:
            return
        if h > self.min_slab:
            raise L.IrsError(f'ghost zone of {h} planes exceeds the smallest slab ({self.min_slab} planes): '
                             f'use fewer ranks for this displacement / volume')
        a, b, D = self.a, self.b, t.shape[zdim]
        ops, unpack = [], []

        def send(lo, hi, peer):
            x = t.narrow(zdim, lo, hi - lo)
            x = x if x.is_contiguous() else x.contiguous()
            ops.append(dist.P2POp(dist.isend, x.cpu() if self.host_staging else x, peer))

        def recv(lo, hi, peer):
            if hi <= lo:
                return
            x = t.narrow(zdim, lo, hi - lo)
            if x.is_contiguous() and not self.host_staging:
                ops.append(dist.P2POp(dist.irecv, x, peer))  # straight into the field
                return
            buf = torch.empty(x.shape, dtype=x.dtype, device='cpu' if self.host_staging else x.device)
            ops.append(dist.P2POp(dist.irecv, buf, peer))
            unpack.append((buf, x))

        if self.rank + 1 < self.world:  # upper neighbour owns [b, ...)
            send(b - h, b, self.rank + 1)
            recv(b, min(b + h, D), self.rank + 1)
        if self.rank > 0:
            send(a, a + h, self.rank - 1)
            recv(max(a - h, 0), a, self.rank - 1)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for buf, x in unpack:
            x.copy_(buf)
        self.exchanged_planes += 2 * h * (t.shape[1] if zdim == 2 else t.shape[-1])

    def _bound_z(self, k):
        """max |d_k| along z in voxels over all chains (after the MAX all-reduce) -- one small device read"""
        return float(self.dmax[k][:, 2].max().item())

    # ------------------------------------------------------------------ the transition
    def transition(self, fixed, moving, v, sigma=None, eps=None, unif=None, outputs=None, timed=False):
        cfg, a, b = self.cfg, self.a, self.b
        s, n = cfg.sobolev_s or 0, cfg.no_steps
        D = cfg.dims[0]
        outputs = dict(outputs or {})
        # the residual / warped image / smoothed state must be addressable for the exchanges
        for key, ch in (('curr_state', 3), ('im_moving_warped', 1), ('residuals', 1)):
            if outputs.get(key) is None:
                outputs[key] = self._own(key, (cfg.no_chains, ch, *cfg.dims))
        io = self._io(fixed, moving, v, sigma, eps, unif, outputs)
        self._io_ref, self._stream, self._irs_stage = C.byref(io), L.stream_ptr(), self.lib.irs_stage
        gmm = cfg.data_loss == 'GMM'
        ls = cfg.lcc_s if gmm else 0

        self._stage(io, ST_BEGIN)
        self._stage(io, ST_PERTURB)
        if s > 0:
            self._halo(self.noisy, s + 1)
        self._stage(io, ST_SMOOTH, 0, max(a - 1, 0), min(b + 1, D))
        self._allreduce(self.dmax[0], dist.ReduceOp.MAX)
        if s == 0:
            self._halo(outputs['curr_state'], 1)
        self._stage(io, ST_ENERGY)
        self._allreduce(self.energy_sum, dist.ReduceOp.SUM)
        self._stage(io, ST_REG_SCALAR)

        halo = self._forward_steps(io, n)
        self._stage(io, ST_OUTPUTS)
        self._stage(io, ST_WARP)
        if gmm:
            self._halo(outputs['im_moving_warped'], 4 * ls)
            self._stage(io, ST_RESIDUAL, 0, max(a - 2 * ls, 0), min(b + 2 * ls, D))
        else:
            self._halo(outputs['im_moving_warped'], 1)
            self._stage(io, ST_RESIDUAL, 0, a, min(b + 1, D))
        for ch in range(cfg.no_chains):
            self._stage(io, ST_STATS, ch)
            self._allreduce(self.stat_sum, dist.ReduceOp.SUM)
            self._stage(io, ST_CHAIN_SCALAR, ch)
            self._stage(io, ST_DATA_BWD, ch)
        self._allreduce(self.nll_sum, dist.ReduceOp.SUM)
        self._stage(io, ST_WARP_BWD)
        for k in range(n - 1, -1, -1):
            ib = self._grad_in_buf[k]
            if self._grad_in_aos[k]:
                self._halo(self._grad_aos[ib], halo[k], 1)
            else:
                self._halo(self._grad_planar[ib], halo[k])
            self._stage(io, ST_EXP_BWD, k)
        self._stage(io, ST_UPDATE)
        self._stage(io, ST_FINALIZE)
        return None

    def _forward_steps(self, io, n):
        """the n squaring steps with ghost-plane exchange; returns the ghost width held for every d_k.

        First transition (or after a misprediction): exact mode -- the bound of d_k is MAX-all-reduced and read back
        before step k (one host sync per step).  Afterwards: predicted mode -- widths come from the previous transition's
        bounds plus one spare plane, nothing is read back during the loop; the bounds of all steps are all-reduced in ONE
        operation afterwards (the adjoint's variant selection needs the global bound anyway) and checked; a misprediction
        re-runs the loop in exact mode."""
        dmax_all = self._view(BUF_DMAX, 0, (n + 1, self.cfg.no_chains, 4), '<f4')
        pred = getattr(self, '_halo_pred', None)
        if pred is not None and self.world > 1:
            used = [1] + [min(p + 1, self.min_slab) for p in pred[1:]]
            for k in range(n):
                if k > 0:
                    self._halo(self.steps[k - 1][0], used[k], self.steps[k - 1][1])
                self._stage(io, ST_EXP_FWD, k)
            self._allreduce(dmax_all, dist.ReduceOp.MAX)
            need = [int(math.floor(x)) + 1 for x in dmax_all[:n, :, 2].max(dim=1).values.tolist()]
            if all(nd <= u for nd, u in zip(need, used)):
                self._halo_pred = need
                self.mispredictions = getattr(self, 'mispredictions', 0)
                return used
            self.mispredictions = getattr(self, 'mispredictions', 0) + 1
            # fall through: redo with exact widths (the bounds of steps 1.. must be rebuilt from a clean slate)
            dmax_all[1:].zero_()
        halo = [0] * n
        for k in range(n):
            h = int(math.floor(self._bound_z(k))) + 1  # taps and gather sources of a voxel lie within floor(max|d|) + 1 planes
            halo[k] = h
            if k == 0:
                if h > 1:
                    raise L.IrsError('|d_0| >= 1 voxel: velocity field too large for 12 squaring steps')
            else:
                self._halo(self.steps[k - 1][0], h, self.steps[k - 1][1])
            self._stage(io, ST_EXP_FWD, k)
            self._allreduce(self.dmax[k + 1], dist.ReduceOp.MAX)
        self._halo_pred = halo
        return halo

    def _own(self, key, shape):
        t = self._keep.get(('own', key))
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(shape, device=self.device, dtype=torch.float32)
            self._keep[('own', key)] = t
        return t

    def gather_slabs(self, t):
        """assemble a full (C, ch, D, H, W) tensor from every rank's own planes (for checks / logging); collective"""
        if not self.on or self.world == 1:
            return t
        D = t.shape[2]
        full = t.clone()
        for r in range(self.world):
            lo, hi = slab_bounds(D, self.world, r)
            part = full[:, :, lo:hi].contiguous()
            if self.host_staging:
                h = part.cpu()
                dist.broadcast(h, src=r)
                part = h.to(t.device)
            else:
                dist.broadcast(part, src=r)
            full[:, :, lo:hi].copy_(part)
        return full
