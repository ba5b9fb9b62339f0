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


# ---- cross-rank result merge (evaluation) ------------------------------------------------------------------------
def collect_results(local_results, size=None, order="contiguous", device=None, dst=0):
    """Merge per-rank result lists into ONE list in dataset order on rank `dst` (None on the other ranks; dst=None: on
    every rank) -- the reference's multi_gpu_test / collect_results (M/mmdet/apis/test.py:75-147: every rank pickles its
    part to a shared tmpdir, rank 0 loads, zips and truncates to len(dataset)).  Here the pickles travel through ONE
    all_gather of byte tensors (RCCL on GPUs: `device` = this rank's GPU; gloo: CPU), no filesystem.
    order: "contiguous" = rank r owns shard_range(size, r, world) (this package's batch shard);
           "interleaved" = rank r owns items r, r + world, ... (DistributedSampler(shuffle=False), M/tools/test.py:120-149:
           the sampler pads the last round by repeating items, which `size` truncates, test.py:108-110).
    `local_results` are arbitrary picklable per-image results, e.g. (bbox_results, segm_results) with RLE dicts."""
    import pickle
    rank, ws = world()
    local_results = list(local_results)
    if not _collective(ws):
        return local_results if size is None else local_results[:size]
    blob = torch.frombuffer(bytearray(pickle.dumps(local_results, protocol=pickle.HIGHEST_PROTOCOL)), dtype=torch.uint8)
    dev = device or "cpu"
    n = torch.tensor([blob.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = torch.zeros(max(sizes), dtype=torch.uint8, device=dev)
    pad[:blob.numel()] = blob.to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    if dst is not None and rank != dst:
        return None
    parts = [pickle.loads(b[:s].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes)]
    if order == "contiguous":
        merged = [x for p in parts for x in p]
    elif order == "interleaved":
        merged = []
        for i in range(max(len(p) for p in parts)):
            merged.extend(p[i] for p in parts if i < len(p))
    else:
        raise ValueError("order must be 'contiguous' or 'interleaved', got %r" % (order,))
    return merged if size is None else merged[:size]


# ---- CPU affinity of a rank (one process per GPU) -----------------------------------------------------------------
def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def local_cpus_of_gpu(pci_bus_id, sysfs="/sys/bus/pci/devices"):
    """CPUs of the NUMA node the GPU hangs off (`local_cpulist` of its PCI function), or None if sysfs does not say"""
    import os
    path = os.path.join(sysfs, pci_bus_id.lower(), "local_cpulist")
    try:
        with open(path) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def rank_cpu_slice(local_rank, local_world, gpu_cpus_by_rank, all_cpus):
    """The CPUs rank `local_rank` should run on: its GPU's NUMA-local CPUs, divided evenly among the ranks whose GPUs share
    that CPU set (so two ranks never pin themselves onto the same cores); without topology information an even
    contiguous split of `all_cpus`.  Pure function (tested on the CPU)."""
    mine = gpu_cpus_by_rank[local_rank] if gpu_cpus_by_rank else None
    if not mine:
        cpus, peers = sorted(all_cpus), list(range(local_world))
    else:
        allowed = set(all_cpus)
        cpus = sorted(c for c in mine if c in allowed) or sorted(all_cpus)
        peers = [r for r in range(local_world) if gpu_cpus_by_rank[r] and set(gpu_cpus_by_rank[r]) == set(mine)]
    i, n = peers.index(local_rank), len(peers)
    per = max(1, len(cpus) // n)
    sl = cpus[i * per:(i + 1) * per] if i < n - 1 else cpus[i * per:]
    return sl or cpus


def pin_rank(local_rank, local_world):
    """Pin this process (and the threads it spawns later) to CPUs next to its GPU.  The host side of a step is launch
    enqueue + small pinned copies: on a two-socket box a rank scheduled on the far socket pays a cross-socket hop per
    launch.  The reference leaves this to the launcher's environment (M/tools/dist_test.sh); here bench.py /
    relaunch_with_ranks call it per rank.  Returns the CPU list (None if affinity is unsupported / SIPMASK_PIN_CPUS=0)."""
    import os
    if os.environ.get("SIPMASK_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world <= 1:
        return None
    try:
        all_cpus = sorted(os.sched_getaffinity(0))
        by_rank = None
        if torch.cuda.is_available():
            by_rank = []
            for r in range(local_world):
                p = torch.cuda.get_device_properties(r) if r < torch.cuda.device_count() else None
                bdf = None if p is None else "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0),
                                                                   getattr(p, "pci_device_id", 0))
                by_rank.append(local_cpus_of_gpu(bdf) if bdf else None)
        cpus = rank_cpu_slice(local_rank, local_world, by_rank, all_cpus)
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, RuntimeError, AttributeError):
        return None
