"""The N > 1 launch path of bench.py / the chain-parallel driver on CPU: world_size 2, gloo backend.

Ranks run independent chains of the same pair (no data-path collective); what crosses ranks is the timing
(max over ranks) and, in the chain-parallel driver, the per-chain scalars that the reference logs.  The test
exercises exactly that reduction logic with the CPU oracle standing in for the device step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ir_sgmcmc_amd.parallel import ChainParallel
    cp = ChainParallel()
    assert cp.world == world and cp.rank == rank
    # each rank owns chains [rank*c, (rank+1)*c); scalars are gathered in rank order
    per_rank = {'data_term': [10.0 * rank + 1.0, 10.0 * rank + 2.0], 'alpha': [0.5 + rank, 0.25 + rank]}
    allv = cp.gather_chain_scalars(per_rank)
    assert allv['data_term'] == [1.0, 2.0, 11.0, 12.0] and allv['alpha'] == [0.5, 0.25, 1.5, 1.25]
    # whole-job throughput = all ranks' transitions / slowest rank's time
    assert cp.max_over_ranks(1.0 + rank) == pytest.approx(float(world))
    assert cp.job_rate(steps=10, elapsed=1.0 + rank) == pytest.approx(world * 10 / float(world))
    # seeds differ per rank, deterministically
    assert cp.chain_seed(1234) == 1234 + rank
    # pooled posterior moments over ranks (mean / M2 merge, used for the displacement statistics)
    torch.manual_seed(0)
    data = torch.randn(8, 5)
    mine = data[rank * 4:(rank + 1) * 4]
    n, mean, m2 = cp.merge_moments(4, mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0))
    assert n == 8 and torch.allclose(mean, data.mean(0), atol=1e-6)
    assert torch.allclose(m2 / (n - 1), data.var(0), atol=1e-5)
    cp.barrier()
    if rank == 0:
        out.put('ok')
    dist.destroy_process_group()


def test_chain_parallel_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 'ok'
