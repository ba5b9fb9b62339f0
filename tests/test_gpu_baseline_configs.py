"""Parity at the shapes of the BASELINE configs that tests/test_gpu_baseline_shape.py does not cover (VERDICT r5 "missing" #1):

  configs[2]  SipMask-R101, bf16 plan, batch 4, 3x800x1344 -- a slot of the pipelined plan `bench.py --config r101` times
  configs[3]  SipMask-R50 training step, 4 images of 3x800x1344 -- `SipMask.forward_train` + backward (bench.py --config train)
  configs[4]  SipMask-VIS R50, clips of 8 frames of 3x384x640 -- `SipMaskVIS.clip_test_many` (bench.py --config vis)

The oracle side (oracle/, torch-CPU fp32) takes 10-40 s per test on the GPU box's host cores.  Every bound below is 1.5 x the
value measured on MI355X (profiles/r06_baseline_config_parity.txt has the measured tables), not a guess."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

DUMP = os.environ.get("SIPMASK_TEST_DUMP")       # path prefix: the measured tables are written there


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _dump(name, lines):
    if DUMP:
        with open(DUMP + "." + name, "w") as f:
            f.write("\n".join(lines) + "\n")


def test_r101_bf16_pipelined_plan_at_baseline_shape():
    """BASELINE configs[2] (sipmask_r101_caffe_fpn_gn_ms_4x.py): the deeper backbone (layer3 = 23 blocks) on the plan object
    `bench.py --config r101` times -- a slot of det.prepare(4, ..., in_flight=3) -- against the fp32 oracle from the same
    images, stage by stage.  Bounds = 1.5 x measured (backbone / FPN stages <= 1.0 % relative Frobenius, head outputs <= 1.6 %,
    mask logits 2.2 %); 100 detections per image on both sides, 67-79 of them in common -- this untrained net's ranking keys
    are dense in near-ties, and 33 bf16 bottlenecks move more of them than R50's 16 (90-93 there); a second pass reproduces
    every bit."""
    _need_gpu()
    import parity_baseline as PB
    rep = PB.run(101, 4, "bf16", features_too=False, verbose=False, plan="pipelined")
    assert rep["steps_in_flight"] == 3 and rep["rerun_bit_identical"]
    img = rep["image"]
    _dump("r101_b4", ["%-12s rel_fro %.5f max_abs %.5f" % (k, v["rel_fro"], v["max_abs"]) for k, v in img.items()] +
          [str(d) for d in rep["detections"]])
    for k in ("C2", "C3", "C4", "C5", "P3", "P4", "P5", "P6", "P7"):
        assert img[k]["rel_fro"] < 0.015, (k, img[k])
    for k in ("cls_logits", "bbox_pred", "centerness", "cof", "basis"):
        assert img[k]["rel_fro"] < 0.024, (k, img[k])
    assert img["mask_logits"]["rel_fro"] < 0.034, img["mask_logits"]
    for d in rep["detections"]:
        assert d["ndet_engine"] == d["ndet_oracle"] == 100 and d["common"] >= 55, d


def test_vis_clip_at_baseline_shape():
    """BASELINE configs[4]: two consecutive 8-frame clips of ONE video, 3x384x640 (640x360 padded), through
    SipMaskVIS.clip_test_many with hipGraph replay and two slots -- the call `bench.py --config vis` times -- at the VIS test
    config (fast_nms, score_thr 0.03, max_per_img 10, mask threshold 0.5; V/mmdet/models/anchor_heads/sipmask_head.py:565-667).
      (1) forward: FPN levels and head outputs of every frame against the fp32 oracle (bf16 stage bounds);
      (2) post-processing on the ENGINE's own head outputs (identical f32 inputs on both sides): labels exact, boxes 1e-5,
          every mask equal to the oracle's away from the 0.5 threshold, the track embeddings at the box centres exact;
      (3) identities: the ids of all 16 frames equal the oracle Tracker's walked over the engine's detections in frame
          order, the second clip continuing the memory of the first; results keyed by object id, RLE masks on the canvas."""
    _need_gpu()
    from oracle import model as OM
    from oracle import ops as O
    from oracle import vis as OV
    from sipmask_amd.synthetic import build_synthetic_vis_detector, calibrate_cls_bias
    dev = torch.device("cuda")
    T, Hh, Ww = 8, 384, 640
    shape = (Hh - 24, Ww, 3)
    det = build_synthetic_vis_detector(seed=0).to(dev)
    g = torch.Generator().manual_seed(77)
    base = torch.randn(1, 3, Hh, Ww, generator=g)
    clips = [torch.cat([base + torch.randn(base.shape, generator=g) * 0.02 for _ in range(T)]) for _ in range(2)]
    eng1 = det.prepare(1, (Hh, Ww), shape)
    calibrate_cls_bias(det, eng1, clips[0][:1].to(dev), target_per_img=300, score_thr=0.03)
    del eng1
    metas = [[dict(img_shape=shape, ori_shape=shape, pad_shape=(Hh, Ww, 3), scale_factor=1.0, is_first=(t == 0 and ci == 0))
              for t in range(T)] for ci in range(2)]
    det.bbox_head.reset_tracker()
    many = det.clip_test_many([c.to(dev) for c in clips], metas, rescale=True, encode=True, graph=True, slots=2)
    torch.cuda.synchronize()
    assert len(many) == 2 and all(len(c) == T for c in many)
    sd = {k: v.detach().float().cpu() for k, v in det.state_dict().items()}
    cfg = dict(det.test_cfg)
    tracker = OV.Tracker()
    lines, total = [], 0
    for ci in range(2):
        eng = det.prepare(T, (Hh, Ww), shape, 1.0, True, lanes=1, slot=ci)      # the slot that ran clip ci; its buffers hold it
        assert eng.vis and eng.batch == T and eng.mask_thr == 0.5
        # (1) forward against the oracle, frame by frame
        with torch.no_grad():
            pyr = OM.fpn_forward(sd, OM.backbone_forward(sd, clips[ci], 50))
            ocls, obb, octr, ocof, ofm = OM.head_forward(sd, pyr)[:5]
            otf = OV.track_forward(sd, pyr)
        lv = eng.lv
        rel = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-30))
        for l, (h, w) in enumerate(lv.sizes):
            got = eng.pyr[lv.row0[l]:lv.row0[l] + T * h * w].float().view(T, h, w, 256).permute(0, 3, 1, 2)
            r = rel(got, pyr[l])
            lines.append("clip %d P%d rel %.5f" % (ci, l + 3, r))
            assert r < 0.015, ("P%d" % (l + 3), r)            # measured <= 0.0095
        cls, bb, ctr, cof, fm = eng.head_outputs()
        b0 = float(sd["bbox_head.fcos_cls.bias"][0])
        cat = lambda ts: torch.cat([t.float().cpu().reshape(T, t.shape[1], -1) for t in ts], 2)
        h0, w0 = lv.sizes[0]
        tf = eng.track_feats.view(T, h0, w0, 512).permute(0, 3, 1, 2).float().cpu()
        # bounds = 1.5 x measured (cls 0.0136, bbox 0.0060, ctr 0.0097, cof 0.0143, basis 0.0125, track 0.0116)
        for name, a, b, bound in (("cls", cat(cls) - b0, cat(ocls) - b0, 0.021), ("bbox", cat(bb), cat(obb), 0.0095),
                                  ("ctr", cat(ctr), cat(octr), 0.015), ("cof", cat(cof), cat(ocof), 0.022),
                                  ("basis", fm.float().cpu(), ofm, 0.019), ("track", tf, otf, 0.018)):
            r = rel(a, b)
            lines.append("clip %d %-6s rel %.5f" % (ci, name, r))
            assert r < bound, (name, r)
        # (2) + (3): the oracle's post-processing and tracker on the engine's own head outputs
        cls, bb, ctr, cof = ([t.float().cpu() for t in x] for x in (cls, bb, ctr, cof))
        fmc = fm.float().cpu()
        nd = eng.nms_out["ndet"].cpu().tolist()
        for t in range(T):
            r = OV.get_masks_single_vis([c[t] for c in cls], [x[t] for x in bb], [c[t] for c in ctr], [c[t] for c in cof],
                                        fmc[t], shape, cfg, 1.0, True)
            n = int(nd[t])
            total += n
            assert n == r["det_bboxes"].shape[0], (ci, t, n, r["det_bboxes"].shape)
            d = eng.nms_out["det"][t, :n].cpu().numpy()
            lab = eng.nms_out["labels"][t, :n].cpu().numpy()
            np.testing.assert_array_equal(lab, r["det_labels"])
            np.testing.assert_allclose(d, r["det_bboxes"], rtol=1e-6, atol=1e-5)
            feats = OV.extract_box_feature_center(tf[t], torch.from_numpy(r["det_bboxes"][:, :4]))
            np.testing.assert_array_equal(eng.det_feats[t, :n].cpu().numpy(), feats.numpy())
            rid = tracker.step(r["det_bboxes"], r["det_labels"], feats, metas[ci][t]["is_first"])
            bres, sres = many[ci][t]
            assert set(bres) == set(int(i) for i in rid if i >= 0) == set(sres), (ci, t)
            for i, oid in enumerate(rid):
                if oid < 0 or list(rid).count(oid) != 1:
                    continue
                np.testing.assert_array_equal(bres[int(oid)]["bbox"], d[i])
                assert int(bres[int(oid)]["label"]) == int(lab[i])
                dec = O.rle_decode(O.rle_from_string(sres[int(oid)]["counts"]), shape[0], shape[1])
                m = r["masks"][i].numpy()
                hh, ww = min(shape[0], m.shape[0]), min(shape[1], m.shape[1])
                diff = dec[:hh, :ww] != m[:hh, :ww]
                assert int(diff.sum()) <= 4, (ci, t, i, int(diff.sum()))
                assert bool(((r["up"][i][:hh, :ww] - 0.5).abs()[torch.from_numpy(diff)] < 1e-4).all())
    _dump("vis_clip8", lines + ["detections over 16 frames: %d" % total])
    assert total >= 32, total                      # the calibration produced work: >= 2 objects per frame on average
    first = set(many[0][-1][0]) & set(many[1][0][0])
    assert first, "the second clip continues identities of the first (tracker memory carried across clip_test_many slots)"


def _synthetic_gt(g, B, Hh, Ww, n):
    """n boxes per image (x1, y1, x2, y2), labels 1..80 and rectangular-blob masks, as tests/test_gpu_api.py builds them"""
    gtb, gtl, gtm = [], [], []
    for b in range(B):
        wh = torch.rand(n, 2, generator=g) * torch.tensor([Ww * 0.45, Hh * 0.45]) + 24
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([float(Ww), float(Hh)]) - wh - 2) + 1
        boxes = torch.cat([xy, xy + wh], 1)
        masks = np.zeros((n, Hh, Ww), np.uint8)
        for i in range(n):
            x1, y1, x2, y2 = [int(v) for v in boxes[i]]
            cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
            yy, xx = np.mgrid[y1:y2 + 1, x1:x2 + 1]
            masks[i, y1:y2 + 1, x1:x2 + 1] = (((xx - cx) / max(1, (x2 - x1) / 2)) ** 2 + ((yy - cy) / max(1, (y2 - y1) / 2)) ** 2) <= 1.0
        gtb.append(boxes)
        gtl.append(torch.randint(1, 81, (n,), generator=g))
        gtm.append(masks)
    return gtb, gtl, gtm


def test_training_step_at_baseline_shape():
    """BASELINE configs[3]: SipMask-R50 `forward_train` + backward on 4 images of 3x800x1344 (sipmask_r50_caffe_fpn_gn_1x.py:
    imgs_per_gpu=4; M/mmdet/models/anchor_heads/sipmask_head.py:289-498) -- the graph `bench.py --config train` times --
    against torch-CPU autograd through the fp32 oracle (same weights, same ground truth).  Bounds = 1.5 x measured on MI355X
    (profiles/r06_baseline_config_parity.txt):
      * loss_cls / loss_bbox / loss_centerness within 0.3 % (measured 0.09 / 0.09 / 0.01 %), loss_mask within 2.7 % (measured
        1.8 %: the BCE over the box crops sees the bf16 mask logits, 2-3 % relative Frobenius in the forward tests);
      * EVERY trainable tensor -- trunk, neck and head, 100 tensors -- cosine >= 0.9985 with the oracle's gradient and relative
        error <= 6.8 % (measured worst: 0.9990 / 4.5 %, reg_convs.0.gn.bias).  At this shape a gradient sums over 89 600 head
        positions per image and the rounding noise that dominates the 128 x 160 composite test averages out -- no cosine > 0.9
        wiring bound is needed here;
      * the frozen parts (stem, stage 1, every BatchNorm) have no gradient on either side."""
    _need_gpu()
    from oracle import loss as OL
    from oracle import model as OM
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, seed=3).cuda()
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-3.0)
    det.train()
    g = torch.Generator().manual_seed(11)
    B, Hh, Ww = 4, 800, 1344
    img = torch.randn(B, 3, Hh, Ww, generator=g)
    gtb, gtl, gtm = _synthetic_gt(g, B, Hh, Ww, 6)
    pnames = set(n for n, _ in det.named_parameters())
    osd = {k: (v.detach().cpu().clone().requires_grad_(True) if k in pnames else v.detach().cpu().clone())
           for k, v in det.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    oout = OM.detector_forward(osd, img, 50)
    oloss, _ = OL.head_loss(oout[0], oout[1], oout[2], oout[3], oout[4], gtb, gtl, gtm)
    sum(oloss.values()).backward()
    del oout
    metas = [dict(img_shape=(Hh, 1333, 3), pad_shape=(Hh, Ww, 3), scale_factor=1.0) for _ in range(B)]
    loss = det.forward_train(img.cuda(), metas, [b.cuda() for b in gtb], [l.cuda() for l in gtl], gt_masks=gtm)
    lines = []
    for k in sorted(loss):
        a, b = float(loss[k].detach()), float(oloss[k].detach())
        lines.append("%-12s hip %.6f oracle %.6f rel %.5f" % (k, a, b, abs(a - b) / max(1e-9, abs(b))))
    sum(loss.values()).backward()
    torch.cuda.synchronize()
    table = []
    for name, p in det.named_parameters():
        ref = osd[name].grad
        if p.grad is None:
            assert ref is None or float(ref.abs().max()) == 0.0 or not p.requires_grad, name
            continue
        assert ref is not None, name
        got = p.grad.detach().float().cpu()
        assert bool(torch.isfinite(got).all()), name
        if float(ref.norm()) == 0.0 and float(got.norm()) == 0.0:      # e.g. the Scale of a level without positives
            continue
        err = float((got - ref).norm() / (ref.norm() + 1e-30))
        cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-30))
        table.append((name, err, cos))
    lines += ["%-60s err %.4f cos %.5f" % w for w in table]
    _dump("train_b4", lines)
    for k in loss:
        a, b = float(loss[k].detach()), float(oloss[k].detach())
        tol = 2.7e-2 if k == "loss_mask" else 3e-3
        assert abs(a - b) <= tol * max(1.0, abs(b)), (k, a, b)
    params = dict(det.named_parameters())
    assert params["backbone.conv1.weight"].grad is None and params["backbone.layer1.0.conv1.weight"].grad is None
    bad = [(n, round(e, 4), round(c, 5)) for n, e, c in table if c < 0.9985 or e > 0.068]
    assert len(table) >= 95, len(table)
    assert not bad, bad[:10]
