"""world_size-2 gloo test of the N>1 path used by bench.py / multi-GPU inference (CPU only)."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sipmask_amd.dist_shard import gather_counts, shard_range, timed_steps


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 100):
        for ws in (1, 2, 3, 8):
            parts = [shard_range(n, r, ws) for r in range(ws)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    lo, hi = shard_range(9, rank, ws)
    done = []

    def step():
        time.sleep(0.02 * (rank + 1))      # rank 1 is the slow one
        done.append(1)

    el = timed_steps(step, 3)
    counts = gather_counts([10 * i for i in range(lo, hi)])
    out[rank] = (lo, hi, len(done), el, counts.tolist())
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_gather():
    ws = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(ws, port, out), nprocs=ws, join=True)
        r0, r1 = out[0], out[1]
    assert (r0[0], r0[1], r1[0], r1[1]) == (0, 5, 5, 9)          # disjoint, contiguous slices
    assert r0[2] == r1[2] == 3                                    # every rank ran exactly K steps
    assert abs(r0[3] - r1[3]) < 1e-9 and r0[3] >= 3 * 0.04 - 1e-3   # MAX over ranks, same on all ranks
    assert r0[4] == r1[4] == [10 * i for i in range(9)]           # gathered in global image order


def _bucket_worker(rank, world, port, out):
    import torch.nn as nn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sipmask_amd.dist_train import GradBucketer
    torch.manual_seed(0)                       # same parameters on every rank
    net = nn.Sequential(nn.Linear(37, 50), nn.ReLU(), nn.Linear(50, 20), nn.ReLU(), nn.Linear(20, 3))
    net[2].bias.requires_grad_(False)          # a frozen parameter is simply not bucketed
    b = GradBucketer(net.parameters(), bucket_bytes=4096)       # several buckets
    g = torch.Generator().manual_seed(100 + rank)               # different data per rank
    res = []
    for step in range(2):
        x = torch.randn(8, 37, generator=g)
        net.zero_grad()
        net(x).square().sum().backward()
        local = [p.grad.clone().numpy().tolist() for p in net.parameters() if p.requires_grad]
        b.finish()
        res.append((local, [p.grad.clone().numpy().tolist() for p in net.parameters() if p.requires_grad]))
    out[rank] = (len(b.buckets), res)
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_gloo_world2():
    """a17: GradBucketer averages gradients across 2 ranks (gloo), bucket by bucket, from backward hooks."""
    import numpy as np
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
        r0, r1 = out[0], out[1]
    assert r0[0] == r1[0] >= 2
    for step in range(2):
        l0, m0 = r0[1][step]
        l1, m1 = r1[1][step]
        for a, b, x0, x1 in zip(l0, l1, m0, m1):
            mean = (np.asarray(a) + np.asarray(b)) / 2
            np.testing.assert_allclose(np.asarray(x0), mean, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(np.asarray(x1), mean, rtol=1e-6, atol=1e-7)
