"""Rank bookkeeping of the multi-process runs (one process per GPU over torch.distributed; backend "nccl" == RCCL on
ROCm, "gloo" in the CPU tests): max-over-ranks timing, whole-job rates, per-rank Philox keys, and the two exchanges the
CHAIN decomposition has -- the per-chain scalars that get logged and the pooled posterior moments of the displacement.
(Chains only share 2K + 2 hyper-parameter scalars in the reference, trainer.py:316-327, so ranks can sample their own
chains with no data-path collective: `bench.py --decomp chains`.)  The z-slab decomposition of ONE chain, the default of
`bench.py --gpus N`, lives in the library: ir_sgmcmc_amd/slab.py, csrc/slab.hip.  Used by bench.py.
"""
import torch
import torch.distributed as dist


class ChainParallel:
    def __init__(self):
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1

    def _device(self):
        return torch.device('cuda', torch.cuda.current_device()) if self.on and dist.get_backend() == 'nccl' else torch.device('cpu')

    def chain_seed(self, base_seed):
        """distinct Philox key per rank"""
        return int(base_seed) + self.rank

    def barrier(self):
        if self.on:
            dist.barrier()

    def max_over_ranks(self, x):
        if not self.on:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def job_rate(self, steps, elapsed):
        """whole-job transitions/s: every rank ran `steps` transitions, the job took the slowest rank's time"""
        return self.world * steps / self.max_over_ranks(elapsed)

    def gather_chain_scalars(self, scalars):
        """{name: [per local chain]} -> {name: [per global chain, rank-major]}"""
        if not self.on:
            return {k: list(v) for k, v in scalars.items()}
        out = {}
        for k in sorted(scalars):
            t = torch.tensor(scalars[k], dtype=torch.float64, device=self._device())
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t)
            out[k] = [float(x) for p in parts for x in p.cpu()]
        return out

    def merge_moments(self, n, mean, m2):
        """pool (count, mean, sum of squared deviations) across ranks (Chan et al.): one all_reduce of [n, n*mean, m2']"""
        if not self.on:
            return n, mean, m2
        dev = self._device()
        mean, m2 = mean.to(dev).double(), m2.to(dev).double()
        buf = torch.cat([torch.tensor([float(n)], dtype=torch.float64, device=dev), (mean * n).flatten(), m2.flatten(),
                         (mean * mean * n).flatten()])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        k = mean.numel()
        N = buf[0]
        gmean = buf[1:1 + k] / N
        # sum_r [m2_r + n_r mean_r^2] - N gmean^2
        gm2 = buf[1 + k:1 + 2 * k] + buf[1 + 2 * k:1 + 3 * k] - N * gmean * gmean
        return int(N.item()), gmean.reshape(mean.shape).float().cpu() if dev.type == 'cpu' else gmean.reshape(mean.shape).float(), \
            gm2.reshape(mean.shape).float().cpu() if dev.type == 'cpu' else gm2.reshape(mean.shape).float()
