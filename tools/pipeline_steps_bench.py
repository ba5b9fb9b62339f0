#!/usr/bin/env python
"""EXPERIMENT: throughput of the R50 B=4 step with TWO STEPS IN FLIGHT -- two full plans (own buffers, own hipGraph) replayed
alternately on two streams, step k+1 enqueued while step k runs -- against the shipped structure (one step at a time, its batch
cut into two concurrent B=2 chains).  Variants: plan = one B=4 chain (with / without internal side lanes) or two B=2 chains."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd.engine import SipMaskEngine, SubBatchPlan
from sipmask_amd.synthetic import build_synthetic_detector, calibrate_cls_bias
dev = torch.device("cuda:0")
H_, W_, B, K = 800, 1344, 4, 60
det = build_synthetic_detector(50, seed=0)
g = torch.Generator().manual_seed(1234)
imgs = [torch.randn(B, 3, H_, W_, generator=g).to(dev) for _ in range(3)]
shape = (H_, 1333, 3)
eng = det.prepare(B, (H_, W_), shape, lanes=1)
calibrate_cls_bias(det, eng, imgs[0].clone(), target_per_img=1000)
del eng
torch.cuda.empty_cache()
sd = det.state_dict()


def single(multi_stream):
    e = SipMaskEngine(sd, B, (H_, W_), 50, det.test_cfg, 81, img_shape=shape)
    e.multi_stream = multi_stream
    return e


def sub():
    return SubBatchPlan([SipMaskEngine(sd, B // 2, (H_, W_), 50, det.test_cfg, 81, img_shape=shape, sub_plan=True) for _ in range(2)])


def capture(plan):
    static = imgs[0].clone()
    plan.run(static)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        plan.run(static)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        plan.run(static)
    return static, gr


def bench(name, mk, nslot):
    slots = [capture(mk()) for _ in range(nslot)]
    streams = [torch.cuda.Stream() for _ in range(nslot)]
    def loop(n):
        for k in range(n):
            st, (static, gr) = streams[k % nslot], slots[k % nslot]
            with torch.cuda.stream(st):
                static.copy_(imgs[k % 3], non_blocking=True)
                gr.replay()
    loop(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-58s %7.1f img/s  %.3f ms/step" % (name, B * K / dt, dt / K * 1e3), flush=True)
    del slots
    torch.cuda.empty_cache()


for rep in range(2):
    bench("one step in flight, two B=2 chains (shipped structure)", sub, 1)
    bench("two steps in flight, each two B=2 chains", sub, 2)
    bench("two steps in flight, each one B=4 chain, no side lanes", lambda: single(False), 2)
    bench("two steps in flight, each one B=4 chain with side lanes", lambda: single(True), 2)
    bench("three steps in flight, each one B=4 chain, no side lanes", lambda: single(False), 3)
