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
    import copy
    twin = copy.deepcopy(net)                  # same weights, no bucketer: this rank's LOCAL gradients
    b = GradBucketer(net.parameters(), bucket_bytes=4096)       # several buckets
    # the gradients live in the buckets: .grad IS a view of the flat buffer, for good
    assert all(p.grad.data_ptr() == v.data_ptr() for bk in b.buckets for p, v in zip(bk["params"], bk["views"]))
    g = torch.Generator().manual_seed(100 + rank)               # different data per rank
    res = []
    for step in range(2):
        x = torch.randn(8, 37, generator=g)
        twin.zero_grad()
        twin(x).square().sum().backward()
        local = [p.grad.clone().numpy().tolist() for p in twin.parameters() if p.requires_grad]
        if step == 1:
            net.zero_grad()                    # a caller that resets .grad to None: the hook folds the fresh tensor back in
        b.zero_grad()
        net(x).square().sum().backward()       # hooks launch each bucket's in-place all-reduce as it fills
        b.finish()
        assert all(p.grad.data_ptr() == v.data_ptr() for bk in b.buckets for p, v in zip(bk["params"], bk["views"]))
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


def _uneven_worker(rank, world, port, out):
    """rank 1 has a branch that receives NO gradient (the 'image batch without positives' case: loss_mask has no
    graph on that rank, so sip_cof / sip_mask_lat get no .grad there while the other rank's do)"""
    import torch.nn as nn
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sipmask_amd.dist_train import GradBucketer
    torch.manual_seed(0)
    trunk = nn.Linear(16, 32)
    head_a = nn.Linear(32, 40)            # always used
    head_b = nn.Linear(32, 24)            # the "mask branch": used on rank 0 only
    params = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
    import copy
    twins = [copy.deepcopy(m) for m in (trunk, head_a, head_b)]          # this rank's LOCAL gradients, no bucketer
    tparams = [p for m in twins for p in m.parameters()]
    b = GradBucketer(params, bucket_bytes=1024)                 # 1 KB buckets: several, of different sizes
    sizes = [bk["flat"].numel() for bk in b.buckets]
    g = torch.Generator().manual_seed(7 + rank)
    res = []
    for step in range(2):
        x = torch.randn(4, 16, generator=g)
        for p in tparams:
            p.grad = None
        h = torch.relu(twins[0](x))
        tl = twins[1](h).square().sum()
        if rank == 0:
            tl = tl + twins[2](h).square().sum()
        tl.backward()
        local = [(None if p.grad is None else p.grad.clone()) for p in tparams]
        b.zero_grad()
        h = torch.relu(trunk(x))
        loss = head_a(h).square().sum()
        if rank == 0:
            loss = loss + head_b(h).square().sum()
        loss.backward()
        b.finish()
        res.append(([None if t is None else t.numpy().tolist() for t in local],
                    [p.grad.clone().numpy().tolist() for p in params]))
    out[rank] = (sizes, res)
    dist.destroy_process_group()


def test_bucket_order_is_rank_independent_when_a_rank_misses_gradients():
    """with per-bucket launches straight from the hooks, a rank without gradients for one bucket would issue
    its all-reduces in a different order than its peers (mismatched sizes -> hang / wrong sums).  Buckets are now
    launched strictly in index order; missing gradients count as zeros (mean over ranks)."""
    import numpy as np
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_uneven_worker, args=(2, port, out), nprocs=2, join=True)
        (s0, r0), (s1, r1) = out[0], out[1]
    assert s0 == s1 and len(s0) >= 3 and len(set(s0)) > 1        # several buckets of different sizes
    for step in range(2):
        l0, m0 = r0[step]
        l1, m1 = r1[step]
        for a, c, x0, x1 in zip(l0, l1, m0, m1):
            za = np.zeros_like(np.asarray(x0)) if a is None else np.asarray(a)
            zc = np.zeros_like(np.asarray(x0)) if c is None else np.asarray(c)
            mean = (za + zc) / 2
            np.testing.assert_allclose(np.asarray(x0), mean, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(np.asarray(x1), mean, rtol=1e-6, atol=1e-7)
        assert any(c is None for c in l1) and all(a is not None for a in l0)


def _video_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sipmask_amd.dist_shard import run_videos
    videos = [list(range(n)) for n in (8, 3, 5, 9, 2)]           # 5 videos of different lengths
    state = {"prev": None, "resets": 0}

    def reset():
        state["prev"], state["resets"] = None, state["resets"] + 1

    def frame(vi, fi, fr):
        # a stand-in tracker: needs the previous frame of THE SAME video (sequential state), reset at frame 0
        assert (fi == 0) == (state["prev"] is None) and (fi == 0 or state["prev"] == (vi, fi - 1))
        state["prev"] = (vi, fi)
        return vi * 100 + fr

    res = run_videos(videos, frame, reset)
    frames = sum(len(v) for v in res.values())
    all_frames = gather_counts([frames])
    out[rank] = ({k: v for k, v in res.items()}, state["resets"], all_frames.tolist())
    dist.destroy_process_group()


def test_vis_clips_are_sharded_by_video_world2():
    """BASELINE config #5 / SURVEY 8(e): VIS shards whole videos (tracker state is sequential inside a clip,
    V/mmdet/models/anchor_heads/sipmask_head.py:169-171,620-667): every video on exactly one rank, frames in order,
    tracker reset per video, frame counts balanced."""
    from sipmask_amd.dist_shard import shard_videos
    assert shard_videos([8, 3, 5, 9, 2], 2) == [[1, 3, 4], [0, 2]]          # 14 vs 13 frames
    assert shard_videos([4, 4, 4], 1) == [[0, 1, 2]]
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_video_worker, args=(2, port, out), nprocs=2, join=True)
        (r0, n0, f0), (r1, n1, f1) = out[0], out[1]
    assert sorted(list(r0) + list(r1)) == [0, 1, 2, 3, 4] and not (set(r0) & set(r1))
    assert n0 == len(r0) and n1 == len(r1)
    for res in (r0, r1):
        for vi, frames in res.items():
            assert frames == [vi * 100 + f for f in range(len(frames))]
    assert f0 == f1 == [14, 13]


def test_bench_self_launches_its_ranks_cpu_stub():
    """`python bench.py --gpus 2` with no launcher must become 2 ranks (it silently measured one GPU).
    SIPMASK_BENCH_STUB=1 swaps the GPU work for a sleeping step and RCCL for gloo; everything else -- re-exec under
    torch.distributed.run, rendezvous on 127.0.0.1, barrier-fenced timing, rank-0 JSON line -- is the real path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIPMASK_BENCH_STUB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == [0, 1] and d["steps"] == 3
    assert d["ms_per_step"] >= 20.0 - 1.0          # MAX over ranks: rank 1 sleeps 20 ms per step
    # and a mismatching launcher is an error, not a silent 1-GPU run
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env1, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


# ---- cross-rank result merge (M/mmdet/apis/test.py:75-147) -------------------------------------------------------------
def _collect_worker(rank, ws, port, out):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from sipmask_amd.dist_shard import collect_results
    n = 7                                                # odd: the ranks own 4 and 3 images
    mk = lambda i: (np.full((i % 3, 5), float(i), np.float32), [dict(size=[8, 9], counts=bytes([65 + i] * (i + 1)))] * (i % 3))
    lo, hi = shard_range(n, rank, ws)
    a = collect_results([mk(i) for i in range(lo, hi)], size=n)                     # contiguous shard, rank 0 only
    b = collect_results([mk(i) for i in range(lo, hi)], size=n, dst=None)           # ... on every rank
    # DistributedSampler(shuffle=False) order: rank r owns r, r + ws, ...; the sampler pads the last round by repeating
    idx = list(range(n)) + list(range((-n) % ws))
    c = collect_results([mk(i) for i in idx[rank::ws]], size=n, order="interleaved", dst=None)
    ok = lambda res: res is not None and len(res) == n and all(
        float(r[0].sum()) == float(mk(i)[0].sum()) and r[0].shape == mk(i)[0].shape and r[1] == mk(i)[1] for i, r in enumerate(res))
    out[rank] = (a is None, a is not None and ok(a), ok(b), ok(c))
    dist.destroy_process_group()


def test_collect_results_merges_in_dataset_order_world2():
    ws = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_collect_worker, args=(ws, port, out), nprocs=ws, join=True)
        r0, r1 = out[0], out[1]
    assert r0 == (False, True, True, True)            # rank 0 holds the merged list (boxes + RLE dicts, dataset order)
    assert r1 == (True, False, True, True)            # dst=0: the other rank gets None; dst=None: every rank gets it


def test_collect_results_single_process_passthrough():
    from sipmask_amd.dist_shard import collect_results
    assert collect_results([1, 2, 3], size=2) == [1, 2]


def test_rank_cpu_slices_are_disjoint_and_numa_local(tmp_path):
    from sipmask_amd.dist_shard import local_cpus_of_gpu, rank_cpu_slice, _parse_cpulist
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    d = tmp_path / "0000:05:00.0"
    d.mkdir()
    (d / "local_cpulist").write_text("0-15,64-79\n")
    assert local_cpus_of_gpu("0000:05:00.0", str(tmp_path)) == list(range(16)) + list(range(64, 80))
    assert local_cpus_of_gpu("0000:06:00.0", str(tmp_path)) is None
    # 8 ranks, GPUs 0-3 on socket 0 (CPUs 0-31), 4-7 on socket 1 (32-63): every rank gets 8 CPUs of ITS socket, all disjoint
    s0, s1 = list(range(32)), list(range(32, 64))
    by_rank = [s0] * 4 + [s1] * 4
    sl = [rank_cpu_slice(r, 8, by_rank, range(64)) for r in range(8)]
    assert all(len(s) == 8 for s in sl) and len(set(c for s in sl for c in s)) == 64
    assert all(set(sl[r]) <= set(by_rank[r]) for r in range(8))
    # no topology information: an even contiguous split
    sl = [rank_cpu_slice(r, 4, None, range(10)) for r in range(4)]
    assert [len(s) for s in sl] == [2, 2, 2, 4] and sorted(c for s in sl for c in s) == list(range(10))
    # a cgroup that hides a GPU's local CPUs: fall back to what the process may use
    assert rank_cpu_slice(0, 2, [[100, 101], [100, 101]], range(4)) == [0, 1]
