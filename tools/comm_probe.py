"""What ONE hand-over of a slab transport costs when nothing else runs: `world` ranks on cuda:0 (children started before anything here
touches the GPU), neighbour exchanges of several sizes and the three small all-reduces of a transition, back to back, host-timed
(`irs_comm_probe`).  The software part of an exchange -- push / flag / wait / drain kernels and their launches; on a node the wire time
of xGMI comes on top, and the ranks do not share one device's queues.

    python tools/comm_probe.py [--worlds 2,4] [--transport ipc|rehearsal]
"""
import argparse
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q, transport):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ir_sgmcmc_amd.slab import SlabComm
    torch.cuda.set_device(0)
    comm = SlabComm.create(transport, 'cuda:0')
    comm.selftest()
    res = {'transport': comm.describe()}
    # one plane of a 256^2 three-channel field is 786 432 bytes: 1, 3 and 8 planes (the widths of a round), and an empty hand-over
    for nbytes in (16, 786432, 3 * 786432, 8 * 786432):
        dist.barrier()
        ex, ar = comm.probe(nbytes, 21, 200)
        res[f'exchange_{nbytes}_bytes_us'] = ex
        res['allreduce_21_doubles_us'] = ar
    dist.barrier()
    if rank == 0:
        q.put(res)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument('--worlds', default='2,4')
    ap.add_argument('--transport', default='ipc')
    a = ap.parse_args()
    out = {}
    for world in [int(w) for w in a.worlds.split(',')]:
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, port, q, a.transport)) for r in range(world)]
        [p.start() for p in ps]
        [p.join(200) for p in ps]
        [p.kill() for p in ps if p.is_alive()]
        out[f'ranks_{world}'] = q.get(timeout=5) if all(p.exitcode == 0 for p in ps) else {'error': [p.exitcode for p in ps]}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
