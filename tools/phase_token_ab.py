#!/usr/bin/env python
"""EXPERIMENT (round 6): does ordering the PHASES of the steps in flight help?  The shipped PipelinedPlan replays one monolithic
hipGraph per step on the slot's stream and lets the hardware interleave the three streams.  Variants here cut a slot's plan into
phase graphs (A = stem .. FPN, bandwidth-shaped; B = head + post-processing, MFMA- / latency-shaped) and pass a token per phase
from step to step (A of step k+1 starts only after A of step k has finished: the backbones never run beside each other).
    python tools/phase_token_ab.py [--precision bf16|head_x3] [--depth 3]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd.synthetic import build_synthetic_detector, calibrate_cls_bias
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--depth", type=int, default=3)
ap.add_argument("--steps", type=int, default=400)
args = ap.parse_args()
dev = torch.device("cuda:0")
H_, W_, B = 800, 1344, 4
det = build_synthetic_detector(50, seed=0)
g = torch.Generator().manual_seed(1234)
imgs = [torch.randn(B, 3, H_, W_, generator=g).to(dev) for _ in range(3)]
shape = (H_, 1333, 3)
eng = det.prepare(B, (H_, W_), shape, lanes=1)
calibrate_cls_bias(det, eng, imgs[0].clone(), target_per_img=1000)
del eng
torch.cuda.empty_cache()
plan = det.prepare(B, (H_, W_), shape, precision=args.precision, lanes="auto", in_flight=args.depth)
plan.capture(imgs[0])
torch.cuda.synchronize()
engs = plan.plans
labels = [l for l, _ in engs[0].steps]
cut = next(i for i, l in enumerate(labels) if "head" in l or l.startswith("split:") or "tower" in l)
cut2 = next((i for i, l in enumerate(labels) if l.startswith("det") or "select" in l), len(labels))
print("steps %d; phase A = [0, %d) ... %s | phase B starts with %s | post starts at %d with %s" % (
    len(labels), cut, labels[cut - 1], labels[cut], cut2, labels[cut2] if cut2 < len(labels) else None), flush=True)


def cap(e, steps):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _, fn in steps:
            fn()
    return gr


def phase_graphs(cuts):
    out = []
    for e in engs:
        assert not e.multi_stream
        bounds = [0] + list(cuts) + [len(e.steps)]
        out.append([cap(e, e.steps[a:b]) for a, b in zip(bounds[:-1], bounds[1:])])
    torch.cuda.synchronize()
    return out


streams = plan.streams


def loop_mono(n):
    for k in range(n):
        plan.submit(imgs[k % 3])


def make_loop(graphs, tokens):
    nph = len(graphs[0])
    def loop(n):
        prev = [None] * nph
        for k in range(n):
            s = k % len(engs)
            st = streams[s]
            with torch.cuda.stream(st):
                plan.static[s].copy_(imgs[k % 3], non_blocking=True)
                for p in range(nph):
                    if tokens[p] and prev[p] is not None:
                        st.wait_event(prev[p])
                    graphs[s][p].replay()
                    if tokens[p]:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        prev[p] = ev
    return loop


two = phase_graphs([cut])
three = phase_graphs([cut, cut2]) if cut2 < len(labels) else None
variants = {"monolithic (shipped)": loop_mono,
            "A|B graphs, no tokens": make_loop(two, [False, False]),
            "A|B, token on A": make_loop(two, [True, False]),
            "A|B, token on B": make_loop(two, [False, True]),
            "A|B, tokens on both": make_loop(two, [True, True])}
if three is not None:
    variants["A|B|post, tokens on A and B"] = make_loop(three, [True, True, False])
    variants["A|B|post, token on A"] = make_loop(three, [True, False, False])
res = {k: [] for k in variants}
for rnd in range(3):
    for name, loop in variants.items():
        loop(12)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rnd:
            res[name].append(B * args.steps / dt)
for name, v in res.items():
    print("%-32s %s img/s" % (name, " / ".join("%.1f" % x for x in v)), flush=True)
