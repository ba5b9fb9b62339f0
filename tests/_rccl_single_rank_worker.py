"""Worker of tests/test_gpu_api.py::test_collective_path_runs_through_rccl_on_one_rank: ONE rank under an initialised "nccl"
(= RCCL) process group with SIPMASK_FORCE_DIST=1, so the collectives an N-GPU job issues -- the timing fence's barrier + MAX
all-reduce, the all_gather of counts and of pickled results, the bucketed in-place gradient all-reduce overlapped with
backward -- execute on the 1-GPU test box although the world size is 1.  Prints RCCL_PATH_OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert os.environ.get("SIPMASK_FORCE_DIST") == "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from oracle import model as OM              # test infrastructure: reference-init weights
    from sipmask_amd import hip_ops as H
    from sipmask_amd import sipmask_head  # noqa: F401
    from sipmask_amd.dist_shard import collect_results, gather_counts, timed_steps
    from sipmask_amd.dist_train import GradBucketer, HipSGD, head_train_step
    from sipmask_amd.registry import build_head
    # ---- inference-side collectives
    calls = []
    el = timed_steps(lambda: calls.append(torch.ones(8, device=dev).sum()), 3, sync_fn=torch.cuda.synchronize, device=dev)
    assert len(calls) == 3 and el > 0
    assert gather_counts([3, 1, 4], device=dev).cpu().tolist() == [3, 1, 4]
    res = [(np.arange(5, dtype=np.float32), [dict(size=[4, 4], counts=b"04")]), (np.zeros(0, np.float32), [])]
    got = collect_results(res, size=2, device=dev)
    assert len(got) == 2 and got[0][1] == res[0][1] and np.array_equal(got[0][0], res[0][0])
    # ---- training-side: bucketed all-reduce through RCCL, gradients living in the buckets
    head = build_head(dict(type='SipMaskHead', num_classes=81, in_channels=256, stacked_convs=4, feat_channels=256,
                           strides=[8, 16, 32, 64, 128], center_sampling=True, center_sample_radius=1.5)).cuda()
    sd = {k[len("bbox_head."):]: v for k, v in OM.init_state_dict(50, seed=17, calibrate=True).items()
          if k.startswith("bbox_head.")}
    sd["fcos_cls.bias"].fill_(-3.0)
    head.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(5)
    B = 2
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    feats = [torch.randn(B, 256, h, w, generator=g).cuda() for h, w in sizes]
    rng = np.random.RandomState(0)
    gtb, gtl, gtm = [], [], []
    yy, xx = np.mgrid[:128, :160]
    for _ in range(B):
        xy = rng.rand(4, 2) * np.array([90.0, 70.0])
        wh = rng.rand(4, 2) * np.array([60.0, 50.0]) + 12
        b = np.concatenate([xy, np.minimum(xy + wh, [159, 127])], 1).astype(np.float32)
        gtb.append(torch.from_numpy(b).cuda())
        gtl.append(torch.from_numpy(rng.randint(1, 81, 4).astype(np.int64)).cuda())
        gtm.append(np.stack([((xx >= bb[0]) & (xx <= bb[2]) & (yy >= bb[1]) & (yy <= bb[3])).astype(np.uint8) for bb in b]))
    metas = [dict(img_shape=(128, 160, 3), pad_shape=(128, 160, 3), scale_factor=1.0) for _ in range(B)]
    opt = HipSGD(head.named_parameters(), lr=0.0, momentum=0.0, weight_decay=0.0)
    plain_loss = head_train_step(head, feats, gtb, gtl, gtm, metas, opt)
    plain = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in head.named_parameters()}
    opt.zero_grad()
    bucket = GradBucketer([p for p in head.parameters() if p.requires_grad], bucket_bytes=4 << 20, force=True)
    assert bucket.force and len(bucket.buckets) >= 2
    launched = []
    real = dist.all_reduce

    def spy(t, *a, **kw):
        launched.append(int(t.numel()))
        return real(t, *a, **kw)

    dist.all_reduce = spy
    try:
        for step in range(3):
            loss = head_train_step(head, feats, gtb, gtl, gtm, metas, opt, bucketer=bucket)
            for k in plain_loss:
                assert abs(loss[k] - plain_loss[k]) <= 1e-4 * max(1.0, abs(plain_loss[k])), (step, k)
            for n, p in head.named_parameters():
                if not p.requires_grad or plain[n] is None:
                    continue
                err = float((p.grad - plain[n]).norm() / (plain[n].norm() + 1e-20))
                assert err < 5e-3, (step, n, err)
    finally:
        dist.all_reduce = real
        bucket.remove()
    assert len(launched) == 3 * len(bucket.buckets), (launched, len(bucket.buckets))     # every bucket, every step, via RCCL
    assert sorted(launched[:len(bucket.buckets)]) == sorted(b["flat"].numel() for b in bucket.buckets)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_PATH_OK buckets=%d" % len(bucket.buckets))


if __name__ == "__main__":
    main()
