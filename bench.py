#!/usr/bin/env python
"""Benchmarks of the SipMask hot path on N MI355X of one node (one process per GPU, no data-path collective in
inference; RCCL gradient all-reduce in training).  BASELINE.json configs:

  --config r50    (default, configs[1]) SipMask-R50 800x1333 (padded 800x1344) inference, batch 4 per GPU, images sharded
                  by batch; step = one batch: NCHW image -> ResNet-50 -> FPN -> SipMaskHead -> top-k / NMS -> fused mask assembly.
                  --in-flight N (default 3): N steps in flight (engine.PipelinedPlan: N complete plans used round-robin, step
                  k+1 enqueued while step k runs; every timed step still is one full batch and all K steps finish inside the
                  timed region); --in-flight 1: one step at a time, its batch cut into two concurrent half-batch chains
  --config r101   (configs[2]) the same with the R101 backbone (sipmask_r101_caffe_fpn_gn_ms_4x.py)
  --config train  (configs[3]) SipMask-R50 training step: forward_train + loss + backward + bucketed gradient
                  all-reduce (RCCL) + SGD, 4 images per GPU
  --config vis    (configs[4]) SipMask-VIS R50 on YouTube-VIS-shaped clips (8 frames of 3x384x640 = 640x360 padded),
                  sharded BY VIDEO (the tracker is sequential inside a clip); step = 8 clips per GPU, pipelined
  --precision f32 the parity plan (exact-f32 MFMA convs) instead of the bf16 throughput plan (r50 / r101)
  --precision head_x3  bf16 backbone + FPN, split-precision head (the reference head's fp32 arithmetic to ~1e-4)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config ...]
`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N
ranks (one per GPU); the driver's own `python -m torch.distributed.run ... bench.py --gpus N` works unchanged.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel) and `cpu_baseline`
(the CPU oracle on the host cores on a bounded sample; rank 0, N=1 only).  The default r50 line also carries (round 4;
`--no-extras` / `--extras-budget` bound them):
  timed_region_s          seconds of the contract's K timed steps (`value` is computed from them)
  steady_state            the same step over a window of >= 2 s (every rank; same fences)
  with_results            ... with the results returned to the host per batch: device RLE + async D2H behind every step, the
                          RLE dicts built on the host inside the timed window (PipelinedPlan.submit(pack=True) / fetch)
  parity                  the TIMED plan against the fp32 CPU oracle from the same images
  parity_plan             the plan that meets north_star's tolerance (`--precision head_x3`), timed with the same structure over
                          >= 2 s in this run, its parity measured on identical head inputs (and from the image)
  mask_assemble_worst_case  the timed plan's mask assembly on 100 image-sized boxes per image
  other_configs           BASELINE configs[2]-[4] (r101 / train / vis) as child runs under a time budget, each with its own
                          cpu_baseline (and parity for r101)
  parity_pairs            (top level) one throughput <-> one parity per plan, both measured in this run ON IDENTICAL FPN FEATURES
  post_processing         mask assembly / RLE cost of the timed detections (`--det-boxes coco`: the calibrated detections' boxes
                          replaced by a fixed-seed draw from COCO's area mix) beside the raw synthetic ones (`tiny`)
  --config eval_shapes    an evaluation loop over the six commonest keep_ratio canvases: plan build + capture seconds per
                          shape, steady img/s while the shapes alternate, HBM held by the cached plans
"""
import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32 (= the f32 vector rate), same guide
IMG_H, IMG_W = 800, 1344           # 800x1333 padded to a multiple of 32 (cfg Pad size_divisor=32)
SSD_HW = (544, 544)                # M/configs/sipmask/sipmask_r50_caffe_fpn_ssd_6x.py:98-109 (img_scale, keep_ratio=False)
VIS_H, VIS_W, VIS_T = 384, 640, 8  # 640x360 frames padded to 32 (V/ config size_divisor=32), frames per clip
VIS_CLIPS = int(os.environ.get("SIPMASK_VIS_CLIPS", "8"))   # clips per GPU per step of --config vis (pipelined: SipMaskVIS.clip_test_many)
STUB = os.environ.get("SIPMASK_BENCH_STUB", "0") == "1"   # CPU test hook: gloo + a sleeping step, no GPU work


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--in-flight", type=int, default=3, dest="in_flight",
                    help="inference configs: steps in flight (engine.PipelinedPlan); 1 = one step at a time, its batch cut into "
                         "two concurrent half-batch chains (engine.SubBatchPlan)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=("r50", "r101", "train", "vis", "eval_shapes", "ssd"), default="r50",
                    help="r50 / r101 / train / vis: BASELINE configs[1]-[4]; eval_shapes: an evaluation loop over keep_ratio canvases; "
                         "ssd: the 544 x 544 SSD-style config (sipmask_r50_caffe_fpn_ssd_6x.py: two-conv towers without GroupNorm, "
                         "fast_nms, 8 images per GPU)")
    ap.add_argument("--precision", choices=("bf16", "f32", "head_x3"), default="bf16",
                    help="bf16 = the throughput plan (BASELINE configs[1] names bf16); head_x3 = bf16 backbone + FPN with the "
                         "split-precision head (binary16 halves, three MFMA terms, f32 activations: mask logits within 1e-3 "
                         "of the fp32 reference head on identical features); f32 = the all-f32 parity plan")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (BASELINE configs[1]: 4; --config ssd: 8)")
    ap.add_argument("--depth", type=int, default=None, help="backbone depth (overrides the config's)")
    ap.add_argument("--lanes", type=int, default=0, help="run the batch as this many independent sub-batch plans on "
                    "concurrent HIP streams (engine.SubBatchPlan); 0 = the product default (2 for even batches >= 4)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--free-run", type=int, default=0, choices=[0, 1, 2],
                    help="with --sub-graphs: the sub-plan chains replay on their own streams without a per-step join "
                         "(1 = started in phase, 2 = started half a pass apart)")
    ap.add_argument("--sub-graphs", type=int, default=0, choices=[0, 1, 2],
                    help="with 2 sub-plans: one hipGraph per sub-plan on concurrent streams (1 = sub-plans keep their "
                         "internal side lanes, 2 = single-lane sub-plans) instead of one graph around both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=0.0, dest="cpu_budget",
                    help="seconds the CPU oracle's timed sample may take (0 = the config's default: 30 / 40 / 20 s)")
    ap.add_argument("--det-boxes", choices=("coco", "tiny"), default="coco", dest="det_boxes",
                    help="inference configs: boxes handed to mask assembly / RLE inside the timed step.  coco (default): the "
                         "calibrated detections keep their scores, labels and coefficients, their boxes are replaced by a "
                         "fixed-seed draw from COCO's published area mix (41 %% small, 34 %% medium, 24 %% large) at the network "
                         "input scale; tiny: the raw synthetic detections (random-weight boxes: 0.4 %% of the image on average)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the contract line (value / roofline [/ cpu_baseline, parity]): skip steady_state, with_results, "
                         "parity_plan, mask_assemble_worst_case and other_configs")
    ap.add_argument("--extras-budget", type=float, default=240.0, dest="extras_budget",
                    help="wall-clock seconds the extra blocks of the default line may take together (an extra that does not "
                         "fit is reported as skipped)")
    ap.add_argument("--breakdown", type=str, default="", help="write the per-step HIP-event breakdown to this file")
    ap.add_argument("--tower-only", type=int, default=0, metavar="N",
                    help="profiling aid: launch only the dominant kernel (first tower launch of the plan, GN "
                         "statistics fused) N times and exit, so a rocprofv3 --stats / --pmc run sees that kernel alone")
    a = ap.parse_args(argv)
    if a.steps is None:
        a.steps = {"r50": 50, "r101": 50, "train": 10, "vis": 10, "eval_shapes": 64, "ssd": 50}[a.config]      # >= 50 iterations after 10 warm-ups
    if a.warmup is None:
        a.warmup = {"r50": 10, "r101": 10, "train": 3, "vis": 2, "eval_shapes": 12, "ssd": 10}[a.config]
    if a.batch is None:
        a.batch = 8 if a.config == "ssd" else 4
    if a.depth is None:
        a.depth = 101 if a.config == "r101" else 50
    if a.config == "eval_shapes" and a.in_flight < 2:
        a.in_flight = 3
    return a


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_with_ranks(args):
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run (one per GPU)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline_inference(det, depth, seed=0, budget_s=0.0, hw=None, img_shape=None, ssd=False):
    """The CPU oracle (kind "port": the reference has no CPU path) on ONE 800x1344 image: 1 warm-up +
    up to 5 timed forwards of extract_feat -> head -> get_masks (no RLE) inside a ~30 s budget, MEDIAN reported."""
    import torch
    from oracle import model as OM
    # the GPU box has hundreds of host cores; torch-CPU convs at this size stop scaling (and oversubscribe badly)
    # beyond a few dozen threads, so the port runs on at most 32 of them
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    hw = hw or (IMG_H, IMG_W)
    img_shape = img_shape or (IMG_H, 1333, 3)
    cfg = dict(OM.DEFAULT_TEST_CFG, score_thr=float(det.test_cfg["score_thr"]))
    img = torch.randn(1, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(seed))
    times = []
    budget_s = budget_s or 30.0
    budget = time.perf_counter() + budget_s
    for it in range(6):
        t0 = time.perf_counter()
        with torch.no_grad():
            cls, bb, ctr, cof, fm = OM.detector_forward(sd, img, depth)
            OM.get_masks_single([c[0] for c in cls], [c[0] for c in bb], [c[0] for c in ctr], [c[0] for c in cof],
                                fm[0], img_shape, cfg, scale_factor=[1.0, 1.0, 1.0, 1.0] if ssd else 1.0, ssd_flag=ssd)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget and len(times) >= 2:
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    t = timed[len(timed) // 2]
    return dict(value=round(1.0 / t, 4), unit="img/s", cores=cores, kind="port",
                sample="1 image 3x%dx%d fp32 per forward, torch-CPU oracle (oneDNN convs + restated deform/NMS/"
                       "mask ops), 1 warm-up + %d timed forward(s) in a %.0f s budget, median %.2f s" % (hw[0], hw[1], len(timed), budget_s, t))


def cpu_baseline_train(det, depth, seed=0, budget_s=0.0):
    """The CPU oracle's TRAINING step on ONE 800x1344 image (kind "port"): backbone -> FPN -> head -> loss (oracle/model.py,
    oracle/loss.py; torch-CPU autograd through the restated ops) -> backward, 1 warm-up + up to 3 timed steps in a ~40 s
    budget, median.  No optimizer step (negligible next to the convolutions)."""
    import numpy as np
    import torch
    from oracle import loss as OL
    from oracle import model as OM
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    pn = set(n for n, p in det.named_parameters() if p.requires_grad)
    sd = {k: (v.detach().float().cpu().clone().requires_grad_(k in pn)) for k, v in det.state_dict().items()}
    img = torch.randn(1, 3, IMG_H, IMG_W, generator=torch.Generator().manual_seed(seed))
    gtb, gtl, gtm = synthetic_gt(seed, 1, IMG_H, IMG_W, torch.device("cpu"))
    times = []
    budget_s = budget_s or 40.0
    budget = time.perf_counter() + budget_s
    for it in range(4):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = OM.head_forward(sd, OM.fpn_forward(sd, OM.backbone_forward(sd, img, depth)))
        losses, _ = OL.head_loss(out[0], out[1], out[2], out[3], out[4], gtb, gtl, gtm)
        sum(losses.values()).backward()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget and len(times) >= 2:
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    t = timed[len(timed) // 2]
    return dict(value=round(1.0 / t, 4), unit="img/s", cores=cores, kind="port",
                sample="1 image 3x800x1344 fp32 per step (forward + loss + backward, no optimizer), torch-CPU oracle, 1 warm-up + "
                       "%d timed step(s) in a %.0f s budget, median %.2f s" % (len(timed), budget_s, t))


def cpu_baseline_vis(det, seed=0, budget_s=0.0):
    """The CPU oracle on ONE 384x640 frame (kind "port"): backbone -> FPN -> head + track branch -> VIS post-processing
    (fast_nms, mask assembly, embedding gather); 1 warm-up + up to 5 timed frames in a ~20 s budget, median."""
    import torch
    from oracle import model as OM
    from oracle import vis as OV
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    img = torch.randn(1, 3, VIS_H, VIS_W, generator=torch.Generator().manual_seed(seed))
    from sipmask_amd.synthetic import VIS_TEST_CFG
    times = []
    budget_s = budget_s or 20.0
    budget = time.perf_counter() + budget_s
    for it in range(6):
        t0 = time.perf_counter()
        with torch.no_grad():
            pyr = OM.fpn_forward(sd, OM.backbone_forward(sd, img, 50))
            cls, bb, ctr, cof, fm = OM.head_forward(sd, pyr)
            tf = OV.track_forward(sd, pyr)
            r = OV.get_masks_single_vis([c[0] for c in cls], [c[0] for c in bb], [c[0] for c in ctr], [c[0] for c in cof], fm[0],
                                        (VIS_H - 24, VIS_W, 3), dict(VIS_TEST_CFG))
            if r["det_bboxes"].shape[0]:
                OV.extract_box_feature_center(tf[0], torch.as_tensor(r["det_bboxes"])[:, :4])
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget and len(times) >= 2:
            break
    timed = sorted(times[1:]) if len(times) > 1 else times
    t = timed[len(timed) // 2]
    return dict(value=round(1.0 / t, 4), unit="frames/s", cores=cores, kind="port",
                sample="1 frame 3x384x640 fp32 per forward (backbone, FPN, head, track branch, fast_nms, masks), torch-CPU "
                       "oracle, 1 warm-up + %d timed frame(s) in a %.0f s budget, median %.2f s" % (len(timed), budget_s, t))


def oracle_record(det, img, depth):
    """the CPU oracle's forward of `img` with the detector's weights as they are (checker leg, like the CPU baseline)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import parity_baseline as PB
    sd = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    return PB.oracle_forward(sd, img.detach().float().cpu(), depth)


def parity_of_timed_plan(plan, ora, batch):
    """A step of the TIMED plan against the CPU oracle on the same images and weights: mask-logit max-abs error at the
    oracle's detections (sipmask_head.py:609-620, north_star's quantity) and the detections in common, per image.  The
    plan's buffers hold that step's tensors (the caller ran it on the oracle's images and synchronised)."""
    import torch
    import parity_baseline as PB
    torch.cuda.synchronize()
    stages, dets = PB.compare_plan(plan, plan.results(), ora, batch, True, with_masks=False)
    p = PB.parity_summary(stages, dets)
    p["mask_logit_rel_fro"] = round(stages["mask_logits"]["rel_fro"], 5)
    p["setting"] = "from the same IMAGES (the bf16 backbone's rounding is part of this number for bf16 and head_x3)"
    p["oracle_seconds"] = round(ora["seconds"], 1)
    return p


def parity_on_identical_features(det, ora, batch, shape, precision):
    """north_star's setting -- "outputs match the reference head on identical inputs": the oracle's fp32 FPN outputs are
    fed to a head-only plan of the same precision built like a slot of the timed pipeline (SipMaskEngine.for_head(...,
    pipelined=True): same kernels, tiles and launch shapes as the timed head), its head outputs, detections and mask
    logits compared with the oracle's (tools/parity_baseline.py, tests/test_gpu_baseline_shape.py)."""
    import torch
    import parity_baseline as PB
    from sipmask_amd.engine import SipMaskEngine
    sizes = [tuple(p.shape[-2:]) for p in ora["pyr"]]
    hsd = {k: v.detach() for k, v in det.state_dict().items() if k.startswith("bbox_head.")}
    heng = SipMaskEngine.for_head(hsd, batch, sizes, img_shape=shape, precision=precision, pipelined=True)
    heng.load_pyramid([p.cuda() for p in ora["pyr"]])
    heng.run_head(with_post=True)
    torch.cuda.synchronize()
    stages = PB.compare_engine(heng, ora, batch, False)
    dets = PB.compare_detections(heng, heng.results(), ora, batch, with_masks=False)
    p = PB.parity_summary(stages, dets)
    p["mask_logit_rel_fro"] = round(stages["mask_logits"]["rel_fro"], 8)
    p["head_outputs_rel_fro"] = {k: float("%.3g" % stages[k]["rel_fro"]) for k in ("cls_logits", "bbox_pred", "centerness", "cof", "basis")}
    p["setting"] = "identical head inputs: the oracle's fp32 FPN features fed to a head-only plan (north_star's tolerance 1e-3)"
    del heng
    torch.cuda.empty_cache()
    return p


# ------------------------------------------------------------------------------------------------ post-processing workload
COCO_AREA_MIX = (("small", 0.41, 8.0, 32.0), ("medium", 0.34, 32.0, 96.0), ("large", 0.24, 96.0, 420.0))


def coco_boxes(nsets, batch, max_num, img_h, img_w, seed=7, scale_xy=None):
    """`nsets` x [batch, max_num, 4] boxes (x1, y1, x2, y2, network-input pixels) drawn with a fixed seed from COCO's published
    object-size mix (cocodataset.org detection evaluation: ~41 % of the objects small (area < 32^2 px), 34 % medium, 24 % large
    (> 96^2) in the ORIGINAL image, typically 640 x 480): sqrt(area) log-uniform inside its class, aspect ratio log-uniform in
    [1/2, 2], scaled by the keep_ratio resize of a 640 x 480 image to the 1333 x 800 test scale (x 1.667), centre uniform,
    clipped to the image.  SipMask's random-weight detections are 0.4 % of the image on average and make mask assembly /
    RLE almost free; this is the load an evaluation run puts on rows a9 / a12 (sipmask_head.py:609-662)."""
    import numpy as np
    import torch
    rng = np.random.RandomState(seed)
    scale = 800.0 / 480.0
    probs = np.array([m[1] for m in COCO_AREA_MIX])
    probs = probs / probs.sum()
    out = np.zeros((nsets, batch, max_num, 4), np.float32)
    cls = rng.choice(len(COCO_AREA_MIX), size=(nsets, batch, max_num), p=probs)
    for k, (_, _, lo, hi) in enumerate(COCO_AREA_MIX):
        m = cls == k
        n = int(m.sum())
        side = np.exp(rng.uniform(np.log(lo), np.log(hi), n)) * scale          # sqrt(area) at the network input scale
        ar = np.exp(rng.uniform(np.log(0.5), np.log(2.0), n))
        sxy = scale_xy or (scale, scale)           # (keep_ratio=False configs stretch the two axes differently)
        w, h = np.minimum(side / scale * sxy[0] * np.sqrt(ar), img_w - 2.0), np.minimum(side / scale * sxy[1] / np.sqrt(ar), img_h - 2.0)
        cx, cy = rng.uniform(w / 2, img_w - 1 - w / 2), rng.uniform(h / 2, img_h - 1 - h / 2)
        out[m] = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    frac = {name: float((cls == k).mean()) for k, (name, _, _, _) in enumerate(COCO_AREA_MIX)}
    area = (out[..., 2] - out[..., 0]) * (out[..., 3] - out[..., 1])
    return torch.from_numpy(out), dict(mix=frac, mean_box_share_of_image=float(area.mean() / (img_h * img_w)))


def install_det_boxes(plan, boxes, flag):
    """Benchmark workload hook: one launch-plan step behind "nms" in every engine of `plan` -- ONE launch of
    sm_det_boxes_override -- that, while `flag` (a device bool) is set, overwrites the x1, y1, x2, y2 of the plan's detections
    with the next of `boxes` ([nsets, B, max_num, 4] on the device; a per-engine device counter cycles through the sets, so
    consecutive steps of one slot see different rectangles).  Scores, labels, kept indices and coefficients stay the
    detector's.  Capture-safe; idempotent; remove_det_boxes() takes it out again (the plans live in the detector's cache)."""
    import torch
    from sipmask_amd import hip_ops as H
    plans = getattr(plan, "plans", None) or [plan]
    for p in plans:
        b0 = 0
        for e in (getattr(p, "engines", None) or [p]):
            if not any(lbl == "det_boxes" for lbl, _ in e.steps):
                det = e.nms_out["det"]
                assert det.is_contiguous()
                sel = boxes[:, b0:b0 + e.batch, :det.shape[1]].contiguous()
                cnt = torch.zeros(1, dtype=torch.int32, device=det.device)
                i = [k for k, (lbl, _) in enumerate(e.steps) if lbl == "nms"][0] + 1
                e.steps.insert(i, ("det_boxes", (lambda det=det, sel=sel, cnt=cnt: H.det_boxes_override(det, sel, cnt, flag))))
                e.lanes.insert(i, 0)
            b0 += e.batch


def remove_det_boxes(plan):
    """takes the workload hook out of every engine of `plan` (graphs captured with it keep replaying it: call this when the
    timing is over and the plan objects stay in the detector's plan cache)"""
    plans = getattr(plan, "plans", None) or [plan]
    for p in plans:
        for e in (getattr(p, "engines", None) or [p]):
            for i in reversed([k for k, (lbl, _) in enumerate(e.steps) if lbl == "det_boxes"]):
                del e.steps[i]
                del e.lanes[i]


def post_processing_block(eng, img, flag, shape, info):
    """mask assembly + device RLE of ONE chain of the timed plan, eager, HIP events on the launch stream, for both box
    workloads: ms per launch, RLE bytes and runs per step (the masks of random weights are noise inside their boxes: many
    more runs per mask than an object's silhouette -- an upper bound on the RLE work of a real checkpoint)."""
    import torch
    res = {}
    post = {lbl: fn for lbl, fn in eng.steps if lbl in ("det_boxes", "mask_assemble")}
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for mode in ("coco", "tiny"):
        flag.fill_(mode == "coco")
        eng.run(img[:eng.batch].contiguous())
        rle_ms, ma_ms, nbytes, nruns, need = [], [], 0, 0, 8192
        for it in range(7):
            e0.record()
            post["det_boxes"]()
            post["mask_assemble"]()
            e1.record()
            try:
                r = eng.encode_rle(shape[:2], fetch=False, max_runs=need)
                e2.record()
                torch.cuda.synchronize()
                mn = int(r["nruns"].min())
                if mn < 0:                                   # a mask needs more runs than the workspace holds: grow once
                    need = 1 << int(math.ceil(math.log2(-mn + 1)))
                    continue
            except RuntimeError as ex:
                res[mode] = dict(error=str(ex)[:200])
                break
            if it >= 2:
                ma_ms.append(e0.elapsed_time(e1))
                rle_ms.append(e1.elapsed_time(e2))
            nd = eng.nms_out["ndet"].cpu().tolist()
            nbytes = int(r["offsets"][-1])
            nruns = int(sum(int(r["nruns"][b * eng.max_num:b * eng.max_num + nd[b]].sum()) for b in range(eng.batch)))
        if ma_ms:
            det = eng.nms_out["det"]
            nd = eng.nms_out["ndet"].cpu().tolist()
            area = sum(float(((det[b, :nd[b], 2] - det[b, :nd[b], 0]).clamp_min(0) * (det[b, :nd[b], 3] - det[b, :nd[b], 1]).clamp_min(0)).sum())
                       for b in range(eng.batch))
            res[mode] = dict(mask_assemble_ms=round(sorted(ma_ms)[len(ma_ms) // 2], 4), rle_ms=round(sorted(rle_ms)[len(rle_ms) // 2], 4),
                             rle_bytes_per_step=nbytes, rle_runs_per_step=nruns, rle_max_runs=need, detections=int(sum(nd)),
                             box_pixels_per_step=int(area), mean_box_share_of_image=round(area / max(1, sum(nd)) / (shape[0] * shape[1]), 5))
    flag.fill_(info["timed"] == "coco")
    res["note"] = ("one chain of the timed plan (%d images), eager; the timed steps run the `%s` workload.  coco = %s" %
                   (eng.batch, info["timed"], info["coco"]))
    return res


# ------------------------------------------------------------------------------------------------ inference configs
def run_inference(args, rank, world, dev):
    import torch
    from sipmask_amd.dist_shard import gather_counts, timed_steps
    from sipmask_amd.synthetic import build_synthetic_detector, calibrate_cls_bias
    B = args.batch
    ssd = args.config == "ssd"
    det = build_synthetic_detector(args.depth, seed=0, ssd=ssd)
    # the 544 x 544 SSD-style config resizes without keep_ratio (sipmask_r50_caffe_fpn_ssd_6x.py:98-109): no padding
    IMG_H, IMG_W = (SSD_HW if ssd else (globals()["IMG_H"], globals()["IMG_W"]))
    thr = float(det.test_cfg["score_thr"])
    sfkw = dict(scale_factor=[1.0, 1.0, 1.0, 1.0]) if ssd else {}     # keep_ratio=False pipelines carry [w, h, w, h] (sipmask_head.py:629-630)
    g = torch.Generator().manual_seed(1234 + rank)
    # NSETS synthetic batches resident in HBM; every step copies the next one into the plan's static input (device to
    # device, inside the timed region) so no step sees the images -- and the mask rectangles -- of the step before
    NSETS = 3
    imgs = [torch.randn(B, 3, IMG_H, IMG_W, generator=g).to(dev) for _ in range(NSETS)]
    img = imgs[0].clone()
    shape = (IMG_H, IMG_W if ssd else 1333, 3)
    eng = det.prepare(B, (IMG_H, IMG_W), shape, precision=args.precision, lanes=1, **sfkw)
    if ssd:      # no norm layer in these towers: bring the sampling offsets back to ~1 px (conv_offset rescaled -> plan rebuilt)
        from sipmask_amd.synthetic import calibrate_offset_scale
        calibrate_offset_scale(det, eng, img, target_std=1.0)
        del eng
        torch.cuda.empty_cache()
        eng = det.prepare(B, (IMG_H, IMG_W), shape, precision=args.precision, lanes=1, **sfkw)
    calibrate_cls_bias(det, eng, img, target_per_img=1000, score_thr=thr)   # updates fcos_cls.bias in place -> plan rebuilt
    del eng
    torch.cuda.empty_cache()
    eng = det.prepare(B, (IMG_H, IMG_W), shape, precision=args.precision, lanes=1, **sfkw)

    # ---- the timed plan: det.prepare's default runs an even batch >= 4 as two concurrent half-batch chains
    # (engine.SubBatchPlan); --lanes 1 forces the single plan
    del eng
    torch.cuda.empty_cache()
    pipelined = args.in_flight > 1 and not args.no_graph and not args.sub_graphs and not args.tower_only
    plan = det.prepare(B, (IMG_H, IMG_W), shape, precision=args.precision, lanes=args.lanes or "auto",
                       in_flight=args.in_flight if pipelined else 1, **sfkw)
    if args.tower_only and args.in_flight > 1:       # the profiling aid times the kernel of the plan the pipelined default runs
        plan = det.prepare(B, (IMG_H, IMG_W), shape, precision=args.precision, lanes=args.lanes or 1, slot=1, pipelined=True, **sfkw)
        plan.multi_stream = False
    # ---- the post-processing workload of the timed steps (--det-boxes): see coco_boxes / install_det_boxes
    first = plan.plans[0] if hasattr(plan, "plans") else plan
    first = first.engines[0] if hasattr(first, "engines") else first
    det_flag = torch.zeros((), dtype=torch.bool, device=dev)
    box_sets, box_info = coco_boxes(NSETS, B, first.max_num, shape[0], shape[1],
                                    scale_xy=(IMG_W / 640.0, IMG_H / 480.0) if ssd else None)
    box_sets = box_sets.to(dev)
    hooked = not args.tower_only and not first.benchmark
    if hooked:
        install_det_boxes(plan, box_sets, det_flag)
        det_flag.fill_(args.det_boxes == "coco")
    subplans = getattr(plan, "engines", None)
    eng = plan.plans[0] if pipelined else (subplans[0] if subplans else plan)   # the plan whose launches the breakdown / roofline time
    if pipelined:
        subplans = getattr(eng, "engines", None)
        eng = subplans[0] if subplans else eng
    eng_img = img[:eng.batch].contiguous()
    run_all = lambda: plan.run(img)
    if args.tower_only:      # the dominant kernel of the SAME plan the breakdown below times, alone and back to back
        eng.run(eng_img)
        towers_ = [c for c in eng.convs if c.name.startswith("head.cls_convs") or c.name.startswith("head.tower")]
        # (head_x3: the SECOND tower launch -- three half products on paired operands; the first reads the bf16 pyramid on two)
        tower = towers_[1] if (args.precision == "head_x3" and len(towers_) > 1) else towers_[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            tower()
        e0.record()
        for _ in range(args.tower_only):
            tower()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.tower_only
        print(json.dumps({"kernel": tower.name, "plan_batch": eng.batch, "patch_kernel": bool(getattr(tower, "patch", False)),
                          "launches": args.tower_only, "ms_per_launch": round(ms, 4),
                          "tflops": round(tower.flops / ms / 1e9, 1), "gflop": round(tower.flops / 1e9, 2),
                          "mfma_tflops": round(getattr(tower, "mfma_flops", tower.flops) / ms / 1e9, 1), "mode": tower.mode,
                          "algorithmic_mb": round(tower.bytes / 1e6, 1)}))
        return None

    # ---- warm-up (eager), then optional graph capture
    if pipelined:
        plan.capture(img)
    else:
        for _ in range(max(1, min(args.warmup, 2))):
            run_all()
    torch.cuda.synchronize()
    graph = None
    sub_graphs = False
    if pipelined:
        pass
    elif args.sub_graphs and not args.no_graph and hasattr(plan, "capture"):
        plan.capture(img, multi_stream=(args.sub_graphs == 1))
        sub_graphs = True
    elif not args.no_graph:
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run_all()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run_all()
        except Exception as e:  # capture problems must not invalidate the measurement: fall back to eager
            print("[bench] graph capture failed (%s); running eagerly" % str(e)[:200], file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    free_run = sub_graphs and args.free_run
    if free_run == 2:
        plan.offset_chains()

    nstep = [0]

    def step():
        if pipelined:           # enqueue and return: the slot copies the batch into its own input on its own stream
            plan.submit(imgs[nstep[0] % NSETS])
            nstep[0] += 1
            return
        if not free_run:        # (free-running chains still read the previous batch when the next step is enqueued)
            img.copy_(imgs[nstep[0] % NSETS])
        nstep[0] += 1
        if free_run:
            plan.replay(join=False)
        elif sub_graphs:
            plan.replay()
        elif graph is not None:
            graph.replay()
        else:
            run_all()

    for _ in range(args.warmup):
        step()
    # barrier + torch.cuda.synchronize() on both sides, MAX over ranks (tested with gloo in tests/test_dist_shard.py)
    elapsed = timed_steps(step, args.steps, sync_fn=torch.cuda.synchronize, device=dev)
    if free_run or pipelined:
        plan.join()
    ndet = gather_counts(plan.results()["ndet"].to(torch.int64), device=dev).cpu().tolist()

    # ---- extra timed windows (every rank runs them: their fences are collective)
    extras = {}
    t_extras = time.perf_counter()
    left = lambda: args.extras_budget - (time.perf_counter() - t_extras)
    do_extras = not args.no_extras and not args.no_graph and not free_run
    if do_extras:
        # (a) the same step over a window of >= 2 s: the K steps of the contract are a 65-160 ms burst, during which the
        # shader clock is still settling (profiles/r03_mfma_peak_clock.txt); this is the steady-state figure
        n_ss = max(args.steps, int(math.ceil(2.2 / (elapsed / args.steps))))
        e_ss = timed_steps(step, n_ss, sync_fn=torch.cuda.synchronize, device=dev)
        if pipelined:
            plan.join()
        extras["steady_state"] = dict(value=round(B * n_ss * world / e_ss, 3), unit="img/s", steps=n_ss,
                                      timed_region_s=round(e_ss, 3), ms_per_step=round(e_ss / n_ss * 1e3, 3),
                                      note="the timed step repeated for >= 2 s after the contract's K steps (same plan, same "
                                           "fences); `value` above is the contract's K-step figure")
        if hooked and args.det_boxes == "coco":
            # the workload of rounds 1-4 (the raw synthetic detections: 0.4 % of the image per box) on the SAME plan, so that the
            # trajectory of the driver's lines stays comparable (VERDICT r5 #9 / ADVICE r5): K contract steps, then >= 2 s
            det_flag.fill_(False)
            for _ in range(args.warmup):
                step()
            e_t = timed_steps(step, args.steps, sync_fn=torch.cuda.synchronize, device=dev)
            e_ts = timed_steps(step, n_ss, sync_fn=torch.cuda.synchronize, device=dev)
            if pipelined:
                plan.join()
            det_flag.fill_(True)
            extras["value_tiny_boxes"] = round(B * args.steps * world / e_t, 3)
            extras["steady_state"]["value_tiny_boxes"] = round(B * n_ss * world / e_ts, 3)
            extras["value_tiny_boxes_note"] = ("the same K timed steps (and the >= 2 s window, under steady_state) with the raw "
                                               "synthetic detections' own boxes -- the workload of the round 1-4 lines; `value` "
                                               "carries COCO-sized boxes since round 5 (config.det_boxes)")
    if do_extras and pipelined:
        # (b) the step WITH its results returned to the host, as the reference's evaluation loop does per batch
        # (M/mmdet/apis/test.py:12-72, sipmask_head.py:645-662): behind every step, on the slot's stream, sm_mask_rects +
        # sm_rle_encode and asynchronous copies of boxes / labels / counts / RLE strings into pinned buffers; the host
        # consumes them (RLE dicts per detection) before the slot is reused, i.e. `in_flight` steps later
        queue = []                                      # slots with unread results, oldest first
        stat = dict(dets=0, rle_bytes=0, batches=0, host_s=0.0, wait_s=0.0, pack_s=0.0)
        wr_runs = [8192]

        def consume(k):
            t0 = time.perf_counter()
            for boxes, labels, rles in plan.fetch(k):
                stat["dets"] += len(rles)
                stat["rle_bytes"] += sum(len(r["counts"]) for r in rles)
            stat["batches"] += 1
            stat["host_s"] += time.perf_counter() - t0
            stat["wait_s"] += plan.last_fetch["wait_s"]
            stat["pack_s"] += plan.last_fetch["pack_s"]

        n_wr = max(args.steps, int(math.ceil(1.5 / (elapsed / args.steps))))
        calls = [0]

        def step_results():
            # submit FIRST, then read the results of the step submitted `depth` steps ago (its slot is the one just resubmitted:
            # every slot double-buffers its pinned result sets), so `depth` steps stay in flight while the host builds the dicts
            queue.append(plan.submit(imgs[nstep[0] % NSETS], pack=True, canvas_hw=shape[:2], max_runs=wr_runs[0]))
            nstep[0] += 1
            if len(queue) > plan.depth:
                consume(queue.pop(0))
            calls[0] += 1
            if calls[0] == n_wr:                       # the last step of the window: drain, every result is consumed inside it
                while queue:
                    consume(queue.pop(0))

        # the RLE workspace is sized for the longest mask of the workload (noise masks in COCO-sized boxes need far more runs
        # than a silhouette): one eager probe on the first slot
        torch.cuda.synchronize()
        try:
            while True:
                r = plan.plans[0].encode_rle(shape[:2], fetch=False, max_runs=wr_runs[0])
                mn = int(r["nruns"].min())
                if mn >= 0:
                    break
                wr_runs[0] = 1 << int(math.ceil(math.log2(-mn + 1)) + 1)
            plan.plans[0]._rle = None
        except Exception:
            pass
        for _ in range(plan.depth + 3):                # warm-up: pinned buffers, RLE workspaces
            queue.append(plan.submit(imgs[nstep[0] % NSETS], pack=True, canvas_hw=shape[:2], max_runs=wr_runs[0]))
            nstep[0] += 1
            if len(queue) > plan.depth:
                consume(queue.pop(0))
        while queue:
            consume(queue.pop(0))
        stat.update(dets=0, rle_bytes=0, batches=0, host_s=0.0, wait_s=0.0, pack_s=0.0)
        try:
            e_wr = timed_steps(step_results, n_wr, sync_fn=torch.cuda.synchronize, device=dev)
        except RuntimeError as ex:                     # (a mask with more runs than the probed workspace: reported, not fatal)
            e_wr = None
            extras["with_results"] = dict(error=str(ex)[:300])
            queue.clear()
            torch.cuda.synchronize()
        if e_wr is not None:
            extras["with_results"] = dict(
                value=round(B * n_wr * world / e_wr, 3), unit="img/s", steps=n_wr, timed_region_s=round(e_wr, 3),
                ms_per_step=round(e_wr / n_wr * 1e3, 3), batches_consumed_this_rank=stat["batches"],
                detections_per_step=round(stat["dets"] / max(1, stat["batches"]), 1),
                rle_bytes_per_step=int(stat["rle_bytes"] / max(1, stat["batches"])),
                host_ms_per_step_reading_results=dict(
                    wait_ms=round(stat["wait_s"] / max(1, stat["batches"]) * 1e3, 3),
                    pack_ms=round(stat["pack_s"] / max(1, stat["batches"]) * 1e3, 3),
                    total_ms=round(stat["host_s"] / max(1, stat["batches"]) * 1e3, 3),
                    note="wait = the host blocked on the step's event (GPU still running: not host work); pack = building the "
                         "per-detection RLE dicts from the pinned buffers"),
                rle_max_runs=wr_runs[0], det_boxes=args.det_boxes if hooked else "tiny",
                per_step="device-side RLE of the step's masks (sm_mask_rects + sm_rle_encode) on the slot's stream + async D2H of "
                         "det_bboxes / det_labels / ndet / run counts / offsets / RLE strings into pinned buffers; the host builds "
                         "the per-detection RLE dicts of step k after submitting step k + in_flight (double-buffered pinned result "
                         "sets per slot: PipelinedPlan.submit(pack=True) / fetch)")
        plan.join()
    if do_extras and hooked and rank == 0 and hasattr(eng, "encode_rle"):
        extras["post_processing"] = post_processing_block(eng, img, det_flag, shape, dict(timed=args.det_boxes, coco=json.dumps(box_info)))

    # ---- per-step HIP-event breakdown (eager, on the launch stream) -> roofline of the dominant kernel
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in eng.steps]
    reps = 3
    acc = [0.0] * len(eng.steps)
    eng.img = img[:eng.batch].contiguous()      # the last timed step's images: the chain recomputes the same bits
    for r in range(reps):
        for (label, fn), (e0, e1) in zip(eng.steps, ev):
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(ev):
            acc[i] += e0.elapsed_time(e1) / reps
    conv_ms = {c.name: None for c in eng.convs + eng.fused}
    for (label, _), ms in zip(eng.steps, acc):
        if label.startswith("conv:"):
            conv_ms[label[5:]] = ms
    towers = [c for c in eng.convs if c.name.startswith("head.cls_convs") or c.name.startswith("head.reg_convs")]
    grouped = [c for c in eng.convs if c.name.startswith("head.tower")]      # cls+reg tower convs of a depth per launch
    if grouped:
        towers = grouped
    tower_ms_bracketed = sum(conv_ms[c.name] for c in towers) / len(towers)    # one event pair around ONE launch: + ~7 us of event / launch latency
    # the dominant kernel's average launch duration: every tower launch 20 x back to back between one event pair (the events'
    # own latency and the gap behind the preceding step amortised), mean of 3 repetitions -- the figure rocprofv3's per-kernel
    # average (profiles/) has to agree with.  The launches rewrite their own outputs with the same bits.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b2b, NB2B = [], 20
    for c in towers:
        c()
        for _ in range(3):
            e0.record()
            for _ in range(NB2B):
                c()
            e1.record()
            torch.cuda.synchronize()
            b2b.append(e0.elapsed_time(e1) / NB2B)
    tower_ms = sum(b2b) / len(b2b)
    rows_m = int(round(towers[0].flops / (2.0 * 256 * 2304) / (2 if grouped else 1)))     # positions of one tower conv (22 400 per 800 x 1344 image)
    x3 = args.precision == "head_x3"
    # x3: three binary16 half products per element product -- the MFMA pipe does 3x the algorithmic FLOPs, and THAT is what
    # the roofline fraction prices (the algorithmic figure is reported beside it)
    # (mean over the tower launches, like tower_ms: the x3 plan's first launch reads the bf16 pyramid on two terms, the others on three)
    tower_flops = sum(getattr(c, "mfma_flops", c.flops) for c in towers) / len(towers)
    all_conv_ms = sum(v for v in conv_ms.values())
    all_conv_ms += sum(ms for (label, _), ms in zip(eng.steps, acc) if label == "stem_fused")   # conv1 + pool in one launch
    all_conv_flops = eng.total_conv_flops()
    fpn = [c for c in eng.convs if c.name.startswith("fpn.")]
    fpn_tf = sum(c.flops for c in fpn) / (sum(conv_ms[c.name] for c in fpn) * 1e-3) / 1e12
    achieved = tower_flops / (tower_ms * 1e-3) / 1e12
    f32 = args.precision == "f32"
    peak = MFMA_F32_PEAK_TFLOPS if f32 else MFMA_BF16_PEAK_TFLOPS
    # HBM traffic of the dominant kernel: PMC passes cannot run inside this process; the number comes from the
    # committed rocprofv3 summary of the same kernel/shape (profiles/), per launch
    traffic, traffic_src = None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[0-9]_pmc_tower_conv%s.json" % ("_head_x3" if x3 else ""))))     # the latest round's passes
    pmc_file = cands[-1] if cands else ""
    pmc = json.load(open(pmc_file)) if pmc_file else None
    if pmc and pmc.get("plan_batch") == eng.batch and pmc.get("launch") in [c.name for c in towers] and args.precision in ("bf16", "head_x3"):
        # (bf16: the PMC passes ran on the plan's first tower launch; head_x3: on the SECOND -- paired operands, three half
        # products -- whose operand bytes differ from the first's two-term launch: the figure is that launch's)
        traffic = round(pmc["hbm_bytes_per_launch"] / 1e6, 1)
        traffic_src = "profiles/%s (launch %s; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), MB per launch" % (
            os.path.basename(pmc_file), pmc.get("launch"))
    if args.breakdown and rank == 0:
        with open(args.breakdown, "w") as f:
            f.write("# per-step HIP event times (ms), eager launches, plan batch %d, mean of %d\n" % (eng.batch, reps))
            cinfo = {"conv:" + c.name: c for c in eng.convs + eng.fused}
            for (label, _), ms in zip(eng.steps, acc):
                c = cinfo.get(label)
                if c is not None:
                    f.write("%-44s %9.4f ms %8.2f GFLOP %8.1f MB %7.1f TFLOP/s %7.0f GB/s\n" %
                            (label, ms, c.flops / 1e9, c.bytes / 1e6, c.flops / ms / 1e9, c.bytes / ms / 1e6))
                elif label == "stem_fused":     # algorithmic bytes: the f32 image in, the pooled bf16 rows out
                    h2, w2 = ((IMG_H - 1) // 2) // 2 + 1, ((IMG_W - 1) // 2) // 2 + 1      # conv 7x7/2 pad 3, pool 3x3/2 pad 1
                    sb = eng.img.numel() * 4 + eng.batch * h2 * w2 * 128
                    f.write("%-44s %9.4f ms %8.2f GFLOP %8.1f MB %7.1f TFLOP/s %7.0f GB/s\n" %
                            (label, ms, eng.stem_flops / 1e9, sb / 1e6, eng.stem_flops / ms / 1e9, sb / ms / 1e6))
                else:
                    f.write("%-44s %9.4f ms\n" % (label, ms))
            f.write("# sum %.3f ms; convs %.3f ms = %.1f TFLOP/s over %.1f GFLOP\n" %
                    (sum(acc), all_conv_ms, all_conv_flops / all_conv_ms / 1e9, all_conv_flops / 1e9))
    if f32:
        kernel = ("conv_f32_kernel<2,2,2,2> (v_mfma_f32_32x32x2_f32, 128x128 tile, 16-wide K steps, register-staged "
                  "loader) = tower 3x3 256->256 over 5 FPN levels (M=%d,N=256,K=2304)" % (rows_m))
    elif x3:
        kernel = ("conv3x3_patch_kernel, binary16 operands (v_mfma_f32_32x32x16_f16): cls+reg tower 3x3 256->256 of one depth "
                  "over 5 FPN levels as ONE grouped launch on split operands [hi|lo|hi] x [hi|hi|lo] (2 x (M=%d,N=256,"
                  "K=3*2304)), f32 output, fixed-point GroupNorm statistics fused; achieved = MFMA FLOPs issued, mean over the "
                  "plan's %d tower launches (3 x %.1f algorithmic GFLOP each; the first reads the bf16 FPN outputs as [hi|hi] x "
                  "[hi|lo]: 2 x)" % (rows_m, len(towers), towers[0].flops / 1e9))
    elif getattr(towers[0], "patch", False):
        kernel = ("conv3x3_patch_kernel (input patch + 2 taps of weights resident in LDS via LDS-DMA, 256x256 tile on 8 "
                  "waves, GroupNorm statistics fused)%s = tower 3x3 256->256 over 5 FPN levels (%sM=%d,N=256,K=2304)"
                  % (", cls+reg towers of one depth as ONE grouped launch" if grouped else "", "2 x " if grouped else "",
                     rows_m))
    elif grouped:
        kernel = ("conv_igemm_kernel<2,4,4,2,false,true,0,7> (LDS-DMA, 256x256 tile on 8 waves, 64-wide K steps, "
                  "hand-placed DMA issue, GroupNorm statistics fused), cls+reg tower 3x3 256->256 of one depth as ONE "
                  "grouped launch over 5 FPN levels (2 x (M=%d,N=256,K=2304))" % (rows_m))
    else:
        kernel = ("conv_igemm_kernel<2,2,2,2,false,true,0,3> (LDS-DMA, 128x128 tile, 64-wide K steps, flat loader + "
                  "pipelined fragment reads, GroupNorm statistics fused) = tower 3x3 256->256 over 5 FPN levels "
                  "(M=%d,N=256,K=2304)" % (rows_m))
    out = {
        "metric": ("img/s SipMask-R%d 544x544 SSD-style inference (ResNet%d+FPN+SipMaskHead(ssd_flag, 2-conv towers, no GN)+fast_nms+"
                   "mask assembly)" % (args.depth, args.depth)) if ssd else
                  "img/s SipMask-R%d 800x1333 inference (ResNet%d+FPN+SipMaskHead+NMS+mask assembly)" % (args.depth, args.depth),
        "value": round(B * args.steps * world / elapsed, 3),
        "unit": "img/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "timed_region_s": round(elapsed, 4),
        "dtype": "f32" if f32 else ("bf16 (backbone, FPN) + 3 x f16 split products / f32 activations (head)" if x3 else "bf16"),
        "data": "synthetic (randn images, %d batches resident in HBM rotated through the plan's input every step; "
                "reference-init random weights + SURVEY 8d calibration overrides)" % NSETS,
        "config": {"workload": "SipMask-R%d FPN inference, batch=%d/GPU, %s, %s, score_thr %s, "
                               "%s, max_per_img 100" % (args.depth, B, "3x544x544 (sipmask_r50_caffe_fpn_ssd_6x.py: keep_ratio=False)" if ssd
                                                            else "3x800x1344 (800x1333 padded)",
                                                            "f32 storage + exact-f32 MFMA (parity plan)" if f32 else
                                                            ("bf16 backbone + FPN, split-precision (x3) head" if x3 else
                                                             "bf16 storage + f32 accumulate"), ("%.2f" % thr).lstrip("0"),
                                                            "fast_nms .5 (top 200 per class)" if ssd else "nms .5"),
                   "global_batch": B * world, "parallelism": "dp%d (batch shard, no collective)" % world,
                   "launch": (("hipGraph replay, %d steps in flight: %d complete plans (own buffers, graph and stream) used round-robin, "
                               "step k+1 is enqueued while step k runs (engine.PipelinedPlan); every step is one batch of %d images"
                               % (plan.depth, plan.depth, B)) if pipelined else
                              ("hipGraph replay" if graph is not None else "eager")) +
                             ("" if not subplans else ", %d sub-batch plans of %d images on concurrent streams"
                              % (len(subplans), eng.batch)),
                   "steps_in_flight": plan.depth if pipelined else 1,
                   "detections_per_image": ndet,
                   # what mask assembly / RLE see inside the timed step (post_processing has the cost of both workloads)
                   "det_boxes": (args.det_boxes if hooked else "tiny"),
                   "det_boxes_note": ("coco = the detections' boxes replaced by a fixed-seed draw from COCO's area mix "
                                      "(%s); tiny = the raw random-weight detections" % json.dumps(box_info)),
                   # the tolerance each plan is held to (tests/test_gpu_baseline_shape.py; parity_pairs has this run's numbers)
                   "stated_fp_tolerance": {
                       "bf16": "mask logits rel-Frobenius <= 2 % on identical FPN features (<= 5 % from the image), NMS keep "
                               "indices bit-exact given the plan's own head outputs",
                       "head_x3": "mask logits max-abs <= 1e-3 on identical FPN features (north_star), detections in the "
                                  "oracle's order"},
                   # FeatureAlign's kernel, chosen by a one-off measurement on the first eager run (engine._tune_deform)
                   "deform_kernel": getattr(eng, "deform_choice", None)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_mb_per_launch": round(towers[0].bytes / 1e6, 1), "kernel": kernel,
                     "gflop_per_launch": round(tower_flops / 1e9, 2), "ms_per_launch": round(tower_ms, 4),
                     "ms_per_launch_method": "HIP events on the launch stream, %d launches back to back per event pair, mean over the "
                                             "plan's %d tower launches x 3 repetitions" % (NB2B, len(towers)),
                     "ms_per_launch_one_event_pair_per_launch": round(tower_ms_bracketed, 4),
                     "all_convs_tflops": round(all_conv_flops / (all_conv_ms * 1e-3) / 1e12, 2),
                     "fpn_convs_tflops": round(fpn_tf, 2),
                     "conv_gflop_per_step": round(plan.total_conv_flops() / 1e9, 1)},
    }
    out.update(extras)
    if world == 1 and rank == 0 and do_extras and getattr(eng, "fused_masks", False):
        out["mask_assemble_worst_case"] = mask_assemble_worst_case(eng)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_inference(det, args.depth, budget_s=args.cpu_budget, hw=(IMG_H, IMG_W), img_shape=shape, ssd=ssd)
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not ssd:    # (ssd: parity lives in tests/test_gpu_api.py, 160 x 160)
        # parity of the TIMED plan object: one more step of it on the images the oracle gets (its own boxes: the workload
        # hook is off while anything is compared)
        det_flag.fill_(False)
        ora = oracle_record(det, imgs[0], args.depth)
        if pipelined:
            plan.run(imgs[0])
            target = plan.plans[plan.last_slot]
        else:
            img.copy_(imgs[0])
            (plan.replay() if sub_graphs else (graph.replay() if graph is not None else run_all()))
            target = plan
        out["parity"] = parity_of_timed_plan(target, ora, B)
        feat_own = parity_on_identical_features(det, ora, B, shape, args.precision)
        out["parity"]["identical_features"] = feat_own
        det_flag.fill_(hooked and args.det_boxes == "coco")
        if do_extras and pipelined and args.precision == "bf16" and left() > 45:
            out["parity_plan"] = parity_plan_block(det, args, imgs, ora, shape, dev, (box_sets, det_flag) if hooked else None)
        elif do_extras and args.precision == "bf16":
            out["parity_plan"] = dict(skipped="extras budget" if pipelined else "needs the pipelined plan")
        # one throughput <-> one parity per plan, measured in THIS run on identical FPN features (north_star's setting)
        pair = lambda name, v, ms, f: dict(plan=name, img_s=v, ms_per_step=ms,
                                           mask_logit_max_abs_features=f["mask_logit_max_abs"],
                                           mask_logit_rel_fro_features=f["mask_logit_rel_fro"],
                                           mask_logit_ref_max_abs=f["mask_logit_ref_max_abs"],
                                           keep_same_order=f["same_order"], common_dets=f["common_dets"])
        ss = extras.get("steady_state", {})
        out["parity_pairs"] = [pair(args.precision, ss.get("value", out["value"]), ss.get("ms_per_step", out["ms_per_step"]), feat_own)]
        pp = out.get("parity_plan") or {}
        if "identical_features" in pp:
            out["parity_pairs"].append(pair("head_x3", pp["value"], pp["ms_per_step"], pp["identical_features"]))
        out["parity_pairs_note"] = ("img_s = the >= 2 s window of each plan (same pipeline structure, same --det-boxes workload); "
                                    "parity = the oracle's fp32 FPN features fed to a head-only plan of that precision built "
                                    "like a slot of the timed pipeline")
    if hooked:                 # the plan objects stay in the detector's cache: leave them as prepare() built them
        det_flag.fill_(False)
        remove_det_boxes(plan)
    if world == 1 and rank == 0 and do_extras and args.config == "r50" and args.precision == "bf16":
        out["other_configs"] = other_configs(min(left(), 200.0))
    return out


def parity_plan_block(det, args, imgs, ora, shape, dev, workload=None):
    """The plan that MEETS north_star's tolerance, timed in the same run as the bf16 line: `precision="head_x3"` (bf16
    backbone + FPN, the head in split precision: f32 activations, every product as three binary16 half products, f32
    accumulation).  Same structure as the default line (steps in flight, one B-image chain per step, inputs rotated);
    its parity block is measured here too -- on identical head inputs (north_star's setting) and from the image."""
    import torch
    from sipmask_amd.dist_shard import timed_steps
    B = args.batch
    plan3 = det.prepare(B, (IMG_H, IMG_W), shape, precision="head_x3", lanes=args.lanes or "auto", in_flight=args.in_flight)
    if workload is not None:                        # the timed steps of this plan carry the same post-processing workload
        install_det_boxes(plan3, workload[0], workload[1])
    plan3.capture(imgs[0])
    n = [0]

    def step3():
        plan3.submit(imgs[n[0] % len(imgs)])
        n[0] += 1

    for _ in range(6):
        step3()
    e = timed_steps(step3, 20, sync_fn=torch.cuda.synchronize, device=dev)
    steps = max(20, int(math.ceil(2.0 / (e / 20))))
    e = timed_steps(step3, steps, sync_fn=torch.cuda.synchronize, device=dev)
    plan3.join()
    if workload is not None:
        was = bool(workload[1])
        workload[1].fill_(False)
    plan3.run(imgs[0])
    from_image = parity_of_timed_plan(plan3.plans[plan3.last_slot], ora, B)
    if workload is not None:
        workload[1].fill_(was)
    del plan3
    torch.cuda.empty_cache()
    feat = parity_on_identical_features(det, ora, B, shape, "head_x3")
    ok = feat["mask_logit_max_abs"] <= 1e-3 and all(feat["same_order"])
    return dict(precision="head_x3", value=round(B * steps / e, 3), unit="img/s", ms_per_step=round(e / steps * 1e3, 3),
                steps=steps, timed_region_s=round(e, 3), steps_in_flight=args.in_flight,
                mask_logit_max_abs=feat["mask_logit_max_abs"], mask_logit_ref_max_abs=feat["mask_logit_ref_max_abs"],
                same_order=feat["same_order"], common_dets=feat["common_dets"], meets_north_star_tolerance=bool(ok),
                identical_features=feat, from_image=from_image,
                dtype="bf16 (backbone, FPN) + 3 x f16 split products / f32 activations (head)")


def mask_assemble_worst_case(eng):
    """sm_mask_assemble_lo with the worst geometry a batch can bring: max_per_img detections per image, every box the
    whole image (each step rewrites every mask plane completely: B x 100 x 800 x 1344 u8) -- the timed steps' cost follows
    the synthetic detections' (small) rectangles.  HIP events over 10 launches on the launch stream."""
    import torch
    from sipmask_amd import hip_ops as H
    B, n = eng.batch, eng.max_num
    buf = H.mask_assemble_lo_alloc(B, n, eng.ho, eng.wo, eng.device)
    det = torch.zeros(B, n, 5, device=eng.device)
    det[..., 2], det[..., 3], det[..., 4] = float(eng.W), float(eng.H), 0.9
    keep = torch.arange(n, dtype=eng.nms_out["keep"].dtype, device=eng.device).repeat(B, 1).contiguous()
    ndet = torch.full((B,), n, dtype=eng.nms_out["ndet"].dtype, device=eng.device)
    h0, w0 = eng._basis_h0w0
    run = lambda: H.mask_assemble_lo(eng.basis_lo, h0, w0, 4, eng.sel["cofs"], keep, det, ndet, eng.ho, eng.wo, eng.box_mul,
                                     2.0, eng.up, eng.mask_thr, buf, per_image=eng.geom_tab)
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out_mb = B * n * eng.ho * eng.wo / 1e6
    covered = float((buf["masks"][..., :eng.wo] != 0).float().mean())
    del buf
    return dict(ms_per_step=round(ms, 4), detections=B * n, box="whole image", mask_mb_written=round(out_mb, 1),
                achieved_gb_s=round(out_mb / ms, 1), hbm_peak_gb_s=8000.0, frac=round(out_mb / ms / 8000.0, 4),
                mask_pixels_set=round(covered, 4),
                note="timed plan's launch (sm_mask_assemble_lo) on 100 image-sized boxes per image; the u8 mask canvas written "
                     "once per step is the algorithmic traffic (SURVEY 8d: 107.5 MB per image)")


def other_configs(budget_s):
    """BASELINE configs[2]-[4] in the SAME driver run: each as `bench.py --config ... --no-extras --no-cpu-baseline` in a
    child process on the same GPU (one at a time, a time limit each), reduced to value / ms_per_step / roofline.frac."""
    import subprocess
    res = {}
    t0 = time.perf_counter()
    # every child also times the CPU oracle of ITS config on the host cores (bounded sample) and, for r101, measures the timed
    # plan's parity -- VERDICT r4 weak #12: config #3 had neither in the driver line
    for name, extra in (("r101", ["--config", "r101", "--steps", "100", "--warmup", "10", "--cpu-budget", "10"]),
                        ("train", ["--config", "train", "--steps", "10", "--warmup", "3", "--cpu-budget", "20"]),
                        ("vis", ["--config", "vis", "--steps", "8", "--warmup", "2", "--cpu-budget", "8"]),
                        ("ssd", ["--config", "ssd", "--steps", "100", "--warmup", "10", "--cpu-budget", "6"])):
        remaining = budget_s - (time.perf_counter() - t0)
        if remaining < 25:
            res[name] = dict(skipped="extras budget (%.0f s left)" % max(0.0, remaining))
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extras"] + extra
        if remaining < 60:                          # not enough left for the CPU sample: the GPU figure alone
            cmd.append("--no-cpu-baseline")
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SIPMASK_FORCE_DIST"):
            env.pop(k, None)
        try:
            t1 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=min(remaining, 100.0), env=env, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                res[name] = dict(error="rc %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
                continue
            d = json.loads(line[-1])
            rf = d.get("roofline") or {}
            res[name] = dict(metric=d["metric"], value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], steps=d["steps"],
                             timed_region_s=round(d["ms_per_step"] * d["steps"] / 1e3, 3), dtype=d["dtype"],
                             roofline_frac=rf.get("frac"), roofline_achieved=rf.get("achieved"), roofline_unit=rf.get("unit"),
                             roofline_kernel=(rf.get("kernel") or "")[:120], wall_s=round(time.perf_counter() - t1, 1))
            cb = d.get("cpu_baseline")
            if cb:
                res[name]["cpu_baseline"] = dict(value=cb["value"], unit=cb["unit"], cores=cb["cores"], kind=cb["kind"],
                                                 sample=cb["sample"][:160])
            pr = d.get("parity")
            if pr:
                f = pr.get("identical_features") or {}
                res[name]["parity"] = dict(mask_logit_max_abs_image=pr.get("mask_logit_max_abs"), common_dets_image=pr.get("common_dets"),
                                           mask_logit_max_abs_features=f.get("mask_logit_max_abs"),
                                           mask_logit_rel_fro_features=f.get("mask_logit_rel_fro"),
                                           keep_same_order_features=f.get("same_order"))
        except subprocess.TimeoutExpired:
            res[name] = dict(skipped="time limit")
    return res


# ------------------------------------------------------------------------------------------------ evaluation over many shapes
EVAL_CANVASES = ((800, 1344), (800, 1216), (800, 1088), (1344, 800), (1216, 800), (800, 1120))


def run_eval_shapes(args, rank, world, dev):
    """What an evaluation run does that the fixed-canvas line does not (M/configs/sipmask/sipmask_r50_caffe_fpn_gn_1x.py:72-87:
    Resize(1333 x 800, keep_ratio) + Pad(32) gives every batch its own padded canvas, portrait or landscape, and every image its
    own img_shape / scale_factor; M/mmdet/apis/test.py:12-72 consumes the results batch by batch): batches cycle through the six
    commonest canvases in a fixed-seed order, each with per-image img_metas, results fetched as RLE dicts a few batches behind.
    Reported: first-touch seconds per shape (launch-plan build + hipGraph capture of every slot), steady img/s while the shapes
    alternate, host ms per batch, HBM held by the cached plans."""
    import numpy as np
    import torch
    from sipmask_amd.dist_shard import timed_steps
    from sipmask_amd.synthetic import build_synthetic_detector, calibrate_cls_bias
    B = args.batch
    det = build_synthetic_detector(args.depth, seed=0)
    eng = det.prepare(B, (IMG_H, IMG_W), (IMG_H, 1333, 3), lanes=1)
    calibrate_cls_bias(det, eng, torch.randn(B, 3, IMG_H, IMG_W, generator=torch.Generator().manual_seed(1234)).to(dev), target_per_img=1000)
    del eng
    det._engines.clear()
    det._engines.capacity = len(EVAL_CANVASES) + 2       # the default LRU of 4 would rebuild a plan on most batches
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    rng = np.random.RandomState(11 + rank)
    g = torch.Generator().manual_seed(77 + rank)
    # COCO originals (w, h) behind each canvas: scale = min(1333 / long side, 800 / short side)
    origs = {True: ((640, 480), (640, 427), (500, 375), (640, 426), (612, 612)), False: ((480, 640), (427, 640), (375, 500), (426, 640))}
    batches = {}
    for hw in EVAL_CANVASES:
        metas = []
        for _ in range(B):
            land = hw[1] >= hw[0]
            ow, oh = origs[land][rng.randint(len(origs[land]))]
            sf = min(1333.0 / max(ow, oh), 800.0 / min(ow, oh))
            ih, iw = min(hw[0], int(oh * sf + 0.5)), min(hw[1], int(ow * sf + 0.5))
            metas.append(dict(img_shape=(ih, iw, 3), ori_shape=(oh, ow, 3), pad_shape=(hw[0], hw[1], 3), scale_factor=float(sf)))
        batches[hw] = (torch.randn(B, 3, hw[0], hw[1], generator=g).to(dev), metas)
    base_mem = torch.cuda.memory_allocated()
    first = {}
    plans = {}
    for hw in EVAL_CANVASES:
        img, metas = batches[hw]
        t0 = time.perf_counter()
        plan = det.plan_for_metas(B, hw, metas, in_flight=args.in_flight)
        t1 = time.perf_counter()
        plan.capture(img)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        first["%dx%d" % hw] = dict(build_s=round(t1 - t0, 3), eager_run_and_capture_s=round(t2 - t1, 3))
        plans[hw] = plan
    held = torch.cuda.memory_allocated() - base_mem
    order = [EVAL_CANVASES[i] for i in rng.randint(len(EVAL_CANVASES), size=4096)]
    fifo, n = [], [0]
    stat = dict(dets=0, batches=0, lookup_s=0.0)

    def step():
        hw = order[n[0] % len(order)]
        n[0] += 1
        img, metas = batches[hw]
        t0 = time.perf_counter()
        plan = det.plan_for_metas(B, hw, metas, in_flight=args.in_flight)      # the cache lookup an evaluation loop pays per batch
        stat["lookup_s"] += time.perf_counter() - t0
        fifo.append((plan, plan.submit(img, metas, pack=True)))      # canvases: every image's own img_shape
        while len(fifo) > args.in_flight:
            p, k = fifo.pop(0)
            stat["dets"] += sum(len(r) for _, _, r in p.fetch(k))
            stat["batches"] += 1

    def drain():
        while fifo:
            p, k = fifo.pop(0)
            stat["dets"] += sum(len(r) for _, _, r in p.fetch(k))
            stat["batches"] += 1

    for _ in range(args.warmup):
        step()
    drain()
    stat.update(dets=0, batches=0, lookup_s=0.0)
    t_host = time.perf_counter()
    elapsed = timed_steps(step, args.steps, sync_fn=lambda: (drain(), torch.cuda.synchronize()), device=dev)
    assert len(det._engines) == len(EVAL_CANVASES), "a plan was evicted and rebuilt inside the loop"
    return {
        "metric": "img/s SipMask-R%d evaluation loop over %d keep_ratio canvases (per-image img_metas, RLE results per batch)" % (args.depth, len(EVAL_CANVASES)),
        "value": round(B * args.steps * world / elapsed, 3), "unit": "img/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "dtype": "bf16",
        "data": "synthetic (randn images, one resident batch per canvas; reference-init random weights + calibration overrides)",
        "config": {"workload": "batches of %d images cycling through the canvases %s in a fixed-seed order; every image its own img_shape / "
                               "scale_factor; %d steps in flight per canvas plan; results (boxes, labels, RLE dicts) fetched %d batches "
                               "behind" % (B, list(EVAL_CANVASES), args.in_flight, args.in_flight),
                   "global_batch": B * world, "parallelism": "dp%d (batch shard, no collective)" % world,
                   "first_touch": first,
                   "first_touch_note": "build = launch-plan construction of every slot (weights folded / re-laid out, buffers "
                                       "allocated); eager_run_and_capture = each slot's first eager run + hipGraph capture",
                   "plans_cached": len(det._engines), "hbm_held_by_plans_gb": round(held / 2 ** 30, 2),
                   "host_ms_per_batch_plan_lookup": round(stat["lookup_s"] / max(1, args.steps) * 1e3, 3),
                   "detections_per_batch": round(stat["dets"] / max(1, stat["batches"]), 1)},
        "roofline": None, "cpu_baseline": None,
    }


# ------------------------------------------------------------------------------------------------ training step
def synthetic_gt(rank, B, Hh, Ww, dev, n=6):
    """per image n boxes with elliptical masks (seeded per rank)"""
    import numpy as np
    import torch
    rng = np.random.RandomState(rank)
    gtb, gtl, gtm = [], [], []
    yy, xx = np.mgrid[:Hh, :Ww]
    for _ in range(B):
        xy = rng.rand(n, 2) * np.array([Ww * 0.6, Hh * 0.6])
        wh = rng.rand(n, 2) * np.array([Ww * 0.35, Hh * 0.35]) + 24
        b = np.concatenate([xy, np.minimum(xy + wh, [Ww - 1, Hh - 1])], 1).astype(np.float32)
        m = np.zeros((n, Hh, Ww), np.uint8)
        for k in range(n):
            cx, cy, rx, ry = (b[k, 0] + b[k, 2]) / 2, (b[k, 1] + b[k, 3]) / 2, (b[k, 2] - b[k, 0]) / 2, (b[k, 3] - b[k, 1]) / 2
            m[k] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
        gtb.append(torch.from_numpy(b).to(dev))
        gtl.append(torch.from_numpy(rng.randint(1, 81, n).astype(np.int64)).to(dev))
        gtm.append(m)
    return gtb, gtl, gtm


def run_train(args, rank, world, dev):
    import torch
    from sipmask_amd.dist_shard import timed_steps
    from sipmask_amd.dist_train import GradBucketer, HipSGD, detector_train_step
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(args.depth, seed=0).to(dev)
    det.train()
    B = args.batch
    # two synthetic batches (images + ground truth) resident on the device, alternated step by step
    NSETS = 2
    gi = torch.Generator().manual_seed(100 + rank)
    sets = [(torch.randn(B, 3, IMG_H, IMG_W, generator=gi).to(dev),) + synthetic_gt(rank * NSETS + k, B, IMG_H, IMG_W, dev)
            for k in range(NSETS)]
    metas = [dict(img_shape=(IMG_H, IMG_W, 3), pad_shape=(IMG_H, IMG_W, 3), scale_factor=1.0) for _ in range(B)]
    opt = HipSGD(det.named_parameters(), lr=0.0005, momentum=0.9, weight_decay=1e-4)
    # SIPMASK_FORCE_DIST=1 (one rank under a launcher): run the collective path anyway -- a 1-GPU box can then execute the
    # RCCL all-reduce of the gradient buckets (world size 1) that the N-GPU job uses
    import torch.distributed as dist
    forced = dist.is_available() and dist.is_initialized()
    bucket = GradBucketer([p for p in det.parameters() if p.requires_grad], force=forced) if (world > 1 or forced) else None
    losses = {}
    nstep = [0]

    def step():
        # loss values stay on the device during the timed steps (read once after them): no per-step host sync
        img, gtb, gtl, gtm = sets[nstep[0] % NSETS]
        nstep[0] += 1
        losses.update(detector_train_step(det, img, metas, gtb, gtl, gtm, opt, bucket, sync=False))

    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(step, args.steps, sync_fn=torch.cuda.synchronize, device=dev)
    # whole-step MFMA fraction: forward + data-gradient + weight-gradient GEMMs of every trained conv ~ 3x the
    # inference conv FLOPs of the trained part (stem + layer1 run without a graph: 1x)
    fwd_gflop = 450.9 * B                           # conv FLOPs per image
    train_gflop = fwd_gflop * 3.0 - 2.0 * (2.5 + 14.3) * 2 * B
    ms = elapsed / args.steps * 1e3
    step_tflops = train_gflop / ms                  # GFLOP per ms = TFLOP/s (analytic FLOPs over wall time: context only)
    # ---- dominant kernel of the step, timed live with HIP events on the launch stream: the direct weight-gradient kernel
    # (csrc/wgrad_direct.hip) on its largest launch -- dW of a tower conv (3x3, 256 -> 256) over the whole pyramid of the
    # batch: [256 x 2304] += gout^T [256 x rows] . im2col(x) [rows x 2304], rows = B * 22 400
    from sipmask_amd import hip_ops as H
    LEV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    lvw = H.Levels(B, LEV)
    xw = (torch.randn(lvw.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    gw_in = (torch.randn(lvw.rows, 256, device=dev) * 0.1).to(torch.bfloat16)
    gw_out = torch.empty(9 * 256, 256, device=dev)
    dw = H.make_conv_desc(B, LEV, LEV, lvw.row0, lvw.row0, 256, 256, 256, 3, 1, 1, 256, 256)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(6):
        e0.record()
        for _ in range(4):
            H.conv2d_bwd(dw, xw, None, None, gw_in, None, gw_out, None)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 4)
    wg_ms = sorted(ts)[len(ts) // 2]
    wg_gflop = 2.0 * lvw.rows * 256 * 2304 / 1e9
    wg_mb = (lvw.rows * 256 * 2 * 2 + 2304 * 256 * 4) / 1e6          # x + gout read once (bf16), dW written (f32)
    achieved = wg_gflop / wg_ms
    out = {
        "metric": "img/s SipMask-R%d training step (forward_train + loss + backward + gradient all-reduce + SGD)" % args.depth,
        "value": round(B * args.steps * world / elapsed, 3),
        "unit": "img/s",
        "ms_per_step": round(ms, 3),
        "dtype": "bf16",
        "data": "synthetic (randn images, 6 boxes + elliptical masks per image; %d batches resident on the device, alternated "
                "step by step; reference-init random weights)" % NSETS,
        "config": {"workload": "SipMask-R%d training step, %d img/GPU, 3x800x1344, bf16 MFMA operands + f32 accumulate / "
                               "f32 master weights, SGD momentum .9 wd 1e-4" % (args.depth, B),
                   "global_batch": B * world,
                   "parallelism": "dp%d (gradient all-reduce over %s, 64 MB buckets overlapped with backward)" %
                                  (world, "RCCL" if bucket is not None else "no collective at 1 GPU"),
                   "losses": {k: round(float(v), 4) for k, v in losses.items()},
                   "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel": "wgrad_direct_kernel (dW straight from NHWC rows: LDS-DMA sub-tiles + ds_read_b64_tr_b16 "
                               "fragments, split-K with float atomics) on its largest launch: dW of a tower conv 3x3 256->256 "
                               "over the batch's pyramid (M=256, N=2304, K=%d positions)" % lvw.rows,
                     "gflop_per_launch": round(wg_gflop, 2), "ms_per_launch": round(wg_ms, 4),
                     "algorithmic_mb_per_launch": round(wg_mb, 1),
                     "whole_step_tflops": round(step_tflops, 2),
                     "whole_step_note": "analytic %.0f GFLOP (forward + dgrad + wgrad of every trained conv) over the step's "
                                        "wall time" % train_gflop},
        "cpu_baseline": None,
    }
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_train(det, args.depth, budget_s=args.cpu_budget)
    return out


# ------------------------------------------------------------------------------------------------ VIS clips
def run_vis(args, rank, world, dev):
    import torch
    from sipmask_amd.dist_shard import run_videos, shard_videos, timed_steps
    from sipmask_amd.synthetic import build_synthetic_vis_detector, calibrate_cls_bias
    det = build_synthetic_vis_detector(seed=0)
    shape = (VIS_H - 24, VIS_W, 3)                  # 360x640 frames padded to 384x640
    # every step: VIS_CLIPS clips per GPU (videos shard over the GPUs); a clip's frames run as ONE batch through the plan, its
    # identity matching in frame order behind it; the clips of a step are pipelined (SipMaskVIS.clip_test_many: clip i+1 is
    # enqueued before the results of clip i are fetched)
    clips_per_step = world * VIS_CLIPS
    NSETS = 2                                        # two sets of clips resident on the device, alternated step by step
    g = torch.Generator().manual_seed(4321)
    clips = [torch.randn(VIS_T, 3, VIS_H, VIS_W, generator=g) for _ in range(clips_per_step * NSETS)]
    eng = det.prepare(1, (VIS_H, VIS_W), shape)
    calibrate_cls_bias(det, eng, clips[0][:1].to(dev), target_per_img=300, score_thr=0.03)
    del eng
    mine = shard_videos([VIS_T] * clips_per_step, world)[rank]
    clips_dev = {(k, vi): clips[k * clips_per_step + vi].to(dev) for k in range(NSETS) for vi in mine}
    use_graph = not args.no_graph
    metas = [dict(img_shape=shape, ori_shape=shape, pad_shape=(VIS_H, VIS_W, 3), scale_factor=1.0, is_first=(t == 0))
             for t in range(VIS_T)]
    counts = []
    nstep = [0]

    def step():
        counts.clear()
        k = nstep[0] % NSETS
        nstep[0] += 1
        # whole videos, in order; tracker reset by is_first of frame 0
        for res in det.clip_test_many([clips_dev[(k, vi)] for vi in mine], [metas] * len(mine), encode=False, graph=use_graph,
                                      slots=2):
            counts.append(sum(len(b) for b, _ in res))

    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(step, args.steps, sync_fn=torch.cuda.synchronize, device=dev)
    frames = clips_per_step * VIS_T * args.steps
    plan = det.prepare(VIS_T, (VIS_H, VIS_W), shape, 1.0, False, lanes=1)   # slot 0 of the timed path
    flops = plan.total_conv_flops() / VIS_T
    ms_frame = elapsed / (VIS_CLIPS * VIS_T * args.steps) * 1e3
    # ---- dominant kernel, timed live with HIP events (eager launches of one chain of the timed plan): the grouped
    # tower launch (cls + reg 3x3 256 -> 256 of one depth over the 5 levels of the chain's frames)
    eng = plan.engines[0] if hasattr(plan, "engines") else plan
    eng.img = clips_dev[(0, mine[0])][:eng.batch].contiguous()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in eng.steps]
    acc = [0.0] * len(eng.steps)
    for r in range(3):
        for (label, fn), (e0, e1) in zip(eng.steps, ev):
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(ev):
            acc[i] += e0.elapsed_time(e1) / 3
    tms = {label[5:]: ms for (label, _), ms in zip(eng.steps, acc) if label.startswith("conv:")}
    towers = [c for c in eng.convs if c.name.startswith("head.tower")] or \
        [c for c in eng.convs if c.name.startswith("head.reg_convs") or c.name.startswith("head.cls_convs")]
    tower_ms = sum(tms[c.name] for c in towers) / len(towers)
    tower_tf = towers[0].flops / tower_ms / 1e9
    out = {
        "metric": "frames/s SipMask-VIS R50 on 640x360 clips (backbone+FPN+head+track head, fast_nms, mask assembly, "
                  "identity matching)",
        "value": round(frames / elapsed, 3),
        "unit": "frames/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "dtype": "bf16",
        "data": "synthetic (randn frames, %d clip sets resident on the device alternated step by step; reference-init random "
                "weights + calibration overrides)" % NSETS,
        "config": {"workload": "SipMask-VIS R50, clips of %d frames 3x%dx%d (640x360 padded), %d clips per GPU per step; the "
                               "frames of a clip run as one batch (one 8-frame launch chain, two clips in flight), identity matching in frame "
                               "order by one kernel per clip (device tracker state); the clips of a step are pipelined (clip "
                               "i+1 is enqueued before the results of clip i are fetched)" % (VIS_T, VIS_H, VIS_W, VIS_CLIPS),
                   "global_batch": clips_per_step, "parallelism": "dp%d (sharded by video, no collective)" % world,
                   "launch": ("hipGraph replay of the whole clip" if use_graph else "eager") +
                             " + one host wait per clip (ids, boxes, labels and counts land in pinned buffers)",
                   "tracked_objects_last_step_this_rank": list(counts)},
        "roofline": {"bound": "mfma", "achieved": round(tower_tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(tower_tf / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel": "%s = %s, cls+reg tower 3x3 256->256 of one depth over 5 levels of %d frames (one chain)" %
                               (towers[0].name, "conv3x3_patch_kernel" if getattr(towers[0], "patch", False) else "conv_igemm_kernel",
                                eng.batch),
                     "gflop_per_launch": round(towers[0].flops / 1e9, 2), "ms_per_launch": round(tower_ms, 4),
                     "algorithmic_mb_per_launch": round(towers[0].bytes / 1e6, 1),
                     "whole_frame_tflops": round(flops / ms_frame / 1e9, 2),
                     "whole_frame_note": "all conv launches of one 384x640 frame (%.1f GFLOP) over the frame time incl. the "
                                         "host-side matching" % (flops / 1e9)},
        "cpu_baseline": None,
    }
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_vis(det, budget_s=args.cpu_budget)
    return out


# ------------------------------------------------------------------------------------------------ driver
def run_stub(args, rank, world):
    """CPU hook for tests/test_dist_shard.py: the launch / rendezvous / timing / JSON path without a GPU"""
    from sipmask_amd.dist_shard import gather_counts, timed_steps
    elapsed = timed_steps(lambda: time.sleep(0.01 * (rank + 1)), args.steps)
    seen = gather_counts([rank]).tolist()
    return {"metric": "stub", "value": round(args.batch * args.steps * world / elapsed, 3), "unit": "img/s",
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "dtype": "none", "data": "none",
            "config": {"workload": "stub", "ranks_seen": seen}, "roofline": None, "cpu_baseline": None}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_with_ranks(args)                   # does not return
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if args.tower_only:      # profiling aid: keep every dispatch sequential so that rocprofv3's per-kernel average is
        os.environ["SIPMASK_MULTI_STREAM"] = "0"     # the isolated kernel, not two towers sharing the chip
    dev = None
    if not STUB:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:            # one process per GPU: keep each rank's launch thread on the CPUs next to its GPU
            from sipmask_amd.dist_shard import pin_rank
            pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1 or os.environ.get("SIPMASK_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:     # bind the communicator to this rank's GPU up front (no "guessing device ID" in the first barrier)
            # RCCL prints a five-line version banner on STDOUT when the first communicator is created; stdout carries the
            # one JSON line of the contract and nothing else, so fd 1 points at stderr while that happens
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                try:       # the banner is printf'ed into libc's buffer: flush it while fd 1 still points at stderr
                    import ctypes
                    ctypes.CDLL(None).fflush(None)
                except Exception:
                    pass
                os.dup2(saved, 1)
                os.close(saved)
        assert dist.get_world_size() == world
    if STUB:
        out = run_stub(args, rank, world)
    elif args.config in ("r50", "r101", "ssd"):
        out = run_inference(args, rank, world, dev)
    elif args.config == "eval_shapes":
        out = run_eval_shapes(args, rank, world, dev)
    elif args.config == "train":
        out = run_train(args, rank, world, dev)
    else:
        out = run_vis(args, rank, world, dev)
    if rank == 0 and out is not None:
        line = {"metric": out.pop("metric"), "value": out.pop("value"), "unit": out.pop("unit"), "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": out.pop("ms_per_step"),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None}
        line.update(out)
        line.setdefault("cpu_baseline", None)
        if os.environ.get("SIPMASK_DIAG_SKIP"):
            # tools/marginal_cost.sh: launches were left out of the captured graphs -- this is NOT a benchmark result
            line["diag_skip"] = os.environ["SIPMASK_DIAG_SKIP"]
            line["value_with_launches_skipped"], line["value"] = line["value"], None
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1 or os.environ.get("SIPMASK_FORCE_DIST") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
