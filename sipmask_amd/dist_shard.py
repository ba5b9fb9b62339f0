"""Batch sharding for multi-GPU inference: one process per GPU, each rank owns a disjoint slice of the images,
NO data-path collective (the reference does the same with DistributedSampler(shuffle=False) + per-GPU forward,
M/tools/test.py:120-149).  The only collectives are a barrier + a MAX all-reduce around a timed region and an
all-gather of small host-side results (reference: M/mmdet/apis/test.py:75-147 collects pickles).
Backend-agnostic (RCCL via "nccl" on GPUs; "gloo" in the CPU tests).
"""
import time

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world_size):
    """Contiguous slice [lo, hi) of n_items owned by ``rank``; sizes differ by at most one and
    lower ranks get the extra items (the order DistributedSampler(shuffle=False) would give after
    a sort by rank)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _collective(ws):
    """collectives run for world size > 1 -- and, with SIPMASK_FORCE_DIST=1 under an initialised process group, also for a
    single rank (a 1-GPU box then executes the same RCCL calls the N-GPU job issues)"""
    import os
    return ws > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("SIPMASK_FORCE_DIST") == "1")


def timed_steps(step_fn, steps, sync_fn=None, device=None):
    """Run ``step_fn`` ``steps`` times between two (barrier + device sync) fences and return the MAX
    elapsed seconds over ranks -- the bench.py contract."""
    rank, ws = world()

    def fence():
        if sync_fn is not None:
            sync_fn()
        if _collective(ws):
            dist.barrier()
            if sync_fn is not None:
                sync_fn()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence()
    elapsed = time.perf_counter() - t0
    if _collective(ws):
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_counts(local_counts, device=None):
    """All-gather a small 1-D int tensor per rank (e.g. detections per image) onto every rank,
    concatenated in rank order = global image order of shard_range."""
    rank, ws = world()
    t = torch.as_tensor(local_counts, dtype=torch.int64, device=device or "cpu").reshape(-1)
    if not _collective(ws):
        return t
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(mx, dtype=torch.int64, device=t.device)
    pad[:t.numel()] = t
    bufs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:int(s.item())] for b, s in zip(bufs, sizes)])


# ---- SipMask-VIS: shard by VIDEO, never by frame ---------------------------------------------------------------
# The tracker state (prev_roi_feats / prev_bboxes / prev_det_labels, V/mmdet/models/anchor_heads/sipmask_head.py:
# 169-171) is sequential inside a video and reset at its first frame (`is_first`, :620-667), so the unit of work
# that can move between GPUs is a whole video (V/tools/test_video.py runs the frames of a video in order).

def shard_videos(frame_counts, world_size):
    """Assign whole videos to ranks, balancing the number of FRAMES: longest-processing-time greedy (videos sorted by
    length, each to the least loaded rank; ties -> lower rank), deterministic on every rank without communication.
    Returns world_size lists of video indices, each in ascending (dataset) order."""
    loads = [0] * world_size
    owned = [[] for _ in range(world_size)]
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        owned[r].append(i)
        loads[r] += int(frame_counts[i])
    return [sorted(v) for v in owned]


def run_videos(videos, frame_fn, reset_fn, rank=None, world_size=None):
    """Run this rank's share of ``videos`` (list of frame lists): ``reset_fn()`` at every video start (is_first),
    then ``frame_fn(video_index, frame_index, frame)`` strictly in frame order.  Returns {video index: [results]}.
    No collective: results are merged on the host afterwards (gather_counts / the reference's pickle collection)."""
    if rank is None:
        rank, world_size = world()
    mine = shard_videos([len(v) for v in videos], world_size)[rank]
    out = {}
    for vi in mine:
        reset_fn()
        out[vi] = [frame_fn(vi, fi, fr) for fi, fr in enumerate(videos[vi])]
    return out
