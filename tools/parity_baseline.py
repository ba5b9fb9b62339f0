#!/usr/bin/env python
"""Parity of the HIP engine against the fp32 CPU oracle AT THE BASELINE SHAPE (BASELINE.json configs[1]/[2]:
R50 batch 4 / R101 batch 1, 3x800x1344), stage by stage, ending at the quantity north_star names: the mask
logits (sipmask_head.py:609-620, `feat_mask . cof_q` before the sigmoid).

Two comparisons per precision mode (`--precision bf16|f32`):
  * "image": same image on both sides -- every stage of the engine vs the oracle;
  * "features": the oracle's fp32 FPN outputs are fed to a head-only plan (SipMaskEngine.for_head), i.e. the head
    sees IDENTICAL inputs (sipmask_head.py:241-287 + :609-633), which is the setting north_star's 1e-3 is stated for.
Mask logits are compared at the ORACLE's detections (its kept candidates' (level, position) pick the engine's
coefficient rows), so the number does not depend on which side's NMS kept what.

The oracle outputs are cached in /tmp (about 2 s per image and forward on 32 cores).  Used by
tests/test_gpu_baseline_shape.py; as a script it writes the JSON report (copied to profiles/ by hand).
Test infrastructure only: imports oracle/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

IMG_H, IMG_W = 800, 1344


def _err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    return dict(rel_fro=float((got - ref).norm() / (ref.norm() + 1e-30)), max_abs=float(d.max()),
                mean_abs=float(d.mean()), ref_max_abs=float(ref.abs().max()), ref_rms=float(ref.pow(2).mean().sqrt()))


def build_case(depth, batch, seed=0, hw=(IMG_H, IMG_W)):
    """synthetic detector of BASELINE.json + images; fcos_cls.bias calibrated ON THE ORACLE's logits of image 0 so both
    sides share one state_dict"""
    from oracle import model as OM
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(depth, seed=seed)
    sd = {k: v.detach().float().cpu().clone() for k, v in det.state_dict().items()}
    img = torch.randn(batch, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(1234))
    return det, sd, img


def oracle_outputs(sd, img, depth, cache_tag):
    """feats, pyr, head outputs (+aux) and per-image post-processing of the oracle; cached in /tmp"""
    from oracle import model as OM
    path = "/tmp/sipmask_parity_oracle_%s.pt" % cache_tag
    if os.path.exists(path):
        return torch.load(path)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    with torch.no_grad():
        feats = OM.backbone_forward(sd, img, depth)
        pyr = OM.fpn_forward(sd, feats)
        # calibration: ~1000 scores per image above score_thr, from image 0's bias-free logits
        out = OM.head_forward(sd, pyr, return_aux=True)
        aux = out[5]
        b_old = float(sd["bbox_head.fcos_cls.bias"][0])
        allc = torch.cat([c[0].reshape(-1) for c in out[0]]) - b_old
        bias = OM.calibrate_cls_bias(sd, allc, target=1000)       # writes sd["bbox_head.fcos_cls.bias"]
        # fcos_cls re-evaluated with the calibrated bias (not "logit - old + new": that rounds differently)
        cls = [torch.nn.functional.conv2d(y, sd["bbox_head.fcos_cls.weight"], sd["bbox_head.fcos_cls.bias"], 1, 1)
               for y in aux["aligned"]]
        out = (cls,) + tuple(out[1:5])
        H, W = img.shape[-2:]
        post = []
        for b in range(img.shape[0]):
            r = OM.get_masks_single([c[b] for c in out[0]], [c[b] for c in out[1]], [c[b] for c in out[2]],
                                    [c[b] for c in out[3]], out[4][b], (H, W - 11 if W == IMG_W else W, 3),
                                    OM.DEFAULT_TEST_CFG)
            post.append({k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in r.items()
                         if k in ("det_bboxes", "det_labels", "idxs_keep", "cand_level", "cand_pos", "det_cofs")})
    res = dict(feats=feats, pyr=pyr, out=out, post=post, cls_bias=bias, seconds=time.time() - t0)
    torch.save(res, path)
    return res


def oracle_forward(sd, img, depth):
    """the oracle record of `oracle_outputs` for a state_dict that is used AS IS (no calibration, no cache): what
    bench.py compares its timed plan with"""
    from oracle import model as OM
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    t0 = time.time()
    with torch.no_grad():
        feats = OM.backbone_forward(sd, img, depth)
        pyr = OM.fpn_forward(sd, feats)
        out = tuple(OM.head_forward(sd, pyr)[:5])
        H, W = img.shape[-2:]
        post = []
        for b in range(img.shape[0]):
            r = OM.get_masks_single([c[b] for c in out[0]], [c[b] for c in out[1]], [c[b] for c in out[2]],
                                    [c[b] for c in out[3]], out[4][b], (H, W - 11 if W == IMG_W else W, 3),
                                    OM.DEFAULT_TEST_CFG)
            post.append({k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in r.items()
                         if k in ("det_bboxes", "det_labels", "idxs_keep", "cand_level", "cand_pos", "det_cofs")})
    return dict(feats=feats, pyr=pyr, out=out, post=post, cls_bias=float(sd["bbox_head.fcos_cls.bias"][0]),
                seconds=time.time() - t0)


def mask_logit_errors(eng, ora, b):
    """|engine logit - oracle logit| over the 4 quadrant logit maps of the ORACLE's detections of image b.
    engine logits = engine basis . engine coefficient rows at the oracle's kept (level, position)."""
    p = ora["post"][b]
    if "det_cofs" not in p:
        return None
    keep = p["idxs_keep"].long()
    lev, pos = p["cand_level"][keep], p["cand_pos"][keep]
    lv = eng.lv
    rows = torch.tensor([lv.row0[int(l)] + b * lv.sizes[int(l)][0] * lv.sizes[int(l)][1] + int(q)
                         for l, q in zip(lev, pos)], dtype=torch.long)
    cof = eng.cls_cof[rows.to(eng.cls_cof.device)][:, eng.ncls:].float().cpu()          # [N,128]
    basis = eng.basis.view(eng.batch, eng.hm * eng.wm, 32)[b].float().cpu()                # [Hm*Wm,32]
    ocof = p["det_cofs"].float()
    obasis = ora["out"][4][b].float().permute(1, 2, 0).reshape(-1, 32)                     # feat_masks [32,Hm,Wm]
    # the quadrant logits exactly as oracle.ops.mask_assemble forms them (img @ cof_q^T), [4,Hm*Wm,N]
    ref = torch.stack([obasis @ ocof[:, 32 * q:32 * (q + 1)].t() for q in range(4)], 0)
    got = torch.stack([basis @ cof[:, 32 * q:32 * (q + 1)].t() for q in range(4)], 0)
    e = _err(got, ref)
    e["ndet"] = int(keep.numel())
    e["cof_max_abs"] = float((cof - p["det_cofs"].float()).abs().max())
    return e


def compare_engine(eng, ora, B, from_image=True):
    rep = {}
    if from_image:
        for i, (buf, h, w, c) in enumerate(eng.backbone_feats):
            rep["C%d" % (i + 2)] = _err(buf.float().view(B, h, w, c).permute(0, 3, 1, 2), ora["feats"][i])
        lv = eng.lv
        for l, (h, w) in enumerate(lv.sizes):
            got = eng.pyr[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, 256).permute(0, 3, 1, 2)
            rep["P%d" % (l + 3)] = _err(got, ora["pyr"][l])
    cls, bb, ctr, cof, fm = eng.head_outputs()
    ocls, obb, octr, ocof, ofm = ora["out"][:5]
    b0 = float(ora["cls_bias"])
    cat = lambda ts: torch.cat([t.float().cpu().reshape(t.shape[0], t.shape[1], -1) for t in ts], 2)
    rep["cls_logits"] = _err(cat(cls) - b0, cat(ocls) - b0)
    rep["bbox_pred"] = _err(cat(bb), cat(obb))
    rep["centerness"] = _err(cat(ctr), cat(octr))
    rep["cof"] = _err(cat(cof), cat(ocof))
    rep["basis"] = _err(fm, ofm)
    ml = [mask_logit_errors(eng, ora, b) for b in range(B)]
    ml = [m for m in ml if m]
    rep["mask_logits"] = dict(max_abs=max(m["max_abs"] for m in ml), mean_abs=float(np.mean([m["mean_abs"] for m in ml])),
                              rel_fro=float(np.mean([m["rel_fro"] for m in ml])),
                              ref_max_abs=max(m["ref_max_abs"] for m in ml), ref_rms=float(np.mean([m["ref_rms"] for m in ml])),
                              ndet=[m["ndet"] for m in ml], cof_max_abs=max(m["cof_max_abs"] for m in ml))
    return rep


def engine_det_keys(eng, res, b):
    """(level, position, label) of the engine's detections of image b (keep = candidate slot; level l owns the slots
    [cand0_l, cand0_l + min(nms_pre, h_l*w_l)); cand_pos = position inside the level)"""
    n = int(res["ndet"][b])
    keep = res["idxs_keep"][b, :n].cpu().long()
    pos = eng.sel["cand_pos"][b].cpu().long()[keep]
    bounds = np.cumsum([0] + [min(eng.cfg["nms_pre"], h * w) for h, w in eng.lv.sizes])
    lev = np.searchsorted(bounds, keep.numpy(), side="right") - 1
    lab = res["det_labels"][b, :n].cpu().numpy()
    return [(int(l), int(q), int(c)) for l, q, c in zip(lev, pos.numpy(), lab)]


def compare_detections(eng, res, ora, B, with_masks=True):
    """engine detections vs the oracle's as sets of (level, position, label) -- logits that differ by rounding can swap
    two near-equal ranking keys, so positions in the sorted lists are not compared -- and, for the common
    detections, boxes and the final masks (differences counted by their distance to the 0.4 threshold)."""
    from oracle import ops as O
    out = []
    for b in range(B):
        p = ora["post"][b]
        n = int(res["ndet"][b])
        keep = p["idxs_keep"].long()
        okeys = [(int(l), int(q), int(c)) for l, q, c in zip(p["cand_level"][keep], p["cand_pos"][keep], p["det_labels"])]
        gkeys = engine_det_keys(eng, res, b)
        gi = {k: i for i, k in enumerate(gkeys)}
        pairs = [(i, gi[k]) for i, k in enumerate(okeys) if k in gi]
        d = dict(ndet_engine=n, ndet_oracle=len(okeys), common=len(pairs),
                 same_order=bool(okeys == gkeys))
        if pairs:
            oi = torch.tensor([a for a, _ in pairs])
            gj = torch.tensor([c for _, c in pairs])
            d["common_box_max_abs"] = float((res["det_bboxes"][b].cpu()[gj] - p["det_bboxes"].float()[oi]).abs().max())
            if with_masks:
                m = O.mask_assemble(ora["out"][4][b], p["det_cofs"][oi], p["det_bboxes"][oi])   # oracle masks (not cached: 1 GB)
                gm = res["masks"][b].cpu()[gj]
                diff = gm != m["masks"]
                dist = (m["up"] - 0.4).abs()[diff]
                d["common_mask_pixels"] = int(diff.numel())
                d["common_mask_pixels_differ"] = int(diff.sum())
                d["common_mask_pixels_beyond_1e-3_of_thr"] = int((dist > 1e-3).sum())
        out.append(d)
    return out


def sub_view(ora, b0, b1):
    """the oracle record restricted to images [b0, b1) (one chain of a SubBatchPlan)"""
    return dict(feats=[f[b0:b1] for f in ora["feats"]], pyr=[p[b0:b1] for p in ora["pyr"]],
                out=tuple([t[b0:b1] for t in o] if isinstance(o, (list, tuple)) else o[b0:b1] for o in ora["out"][:5]),
                post=ora["post"][b0:b1], cls_bias=ora["cls_bias"], seconds=ora["seconds"])


def _merge(reports):
    """stage reports of the chains of a SubBatchPlan -> one report: worst case per stage (rel_fro: the chains' maximum)"""
    out = {}
    for k in reports[0]:
        vs = [r[k] for r in reports]
        m = {}
        for f in vs[0]:
            if f == "ndet":
                m[f] = sum((v[f] for v in vs), [])
            elif f in ("mean_abs", "ref_rms"):
                m[f] = float(np.mean([v[f] for v in vs]))
            else:
                m[f] = max(v[f] for v in vs)
        out[k] = m
    return out


def compare_plan(plan, res, ora, B, from_image=True, with_masks=True):
    """stage report + detection report of a launch plan -- a SipMaskEngine or a SubBatchPlan (every chain against its
    slice of the oracle record; `res` = the plan's result dict over the whole batch)"""
    chains = getattr(plan, "engines", None)
    if not chains:
        return compare_engine(plan, ora, B, from_image), compare_detections(plan, res, ora, B, with_masks)
    reps, dets, b0 = [], [], 0
    for e in chains:
        o = sub_view(ora, b0, b0 + e.batch)
        reps.append(compare_engine(e, o, e.batch, from_image))
        dets += compare_detections(e, {k: v[b0:b0 + e.batch] for k, v in res.items()}, o, e.batch, with_masks)
        b0 += e.batch
    return _merge(reps), dets


def parity_summary(stage_report, det_report):
    """the two numbers north_star names, for the bench line: mask-logit max-abs error and detections in common"""
    return dict(mask_logit_max_abs=round(stage_report["mask_logits"]["max_abs"], 6),
                mask_logit_ref_max_abs=round(stage_report["mask_logits"]["ref_max_abs"], 3),
                common_dets=[d["common"] for d in det_report], oracle_dets=[d["ndet_oracle"] for d in det_report],
                engine_dets=[d["ndet_engine"] for d in det_report], same_order=[d["same_order"] for d in det_report])


def run(depth=50, batch=4, precision="bf16", features_too=True, verbose=True, plan="single"):
    """plan: "single" = one SipMaskEngine over the batch; "subbatch" = what bench.py --in-flight 1 times, i.e.
    det.prepare(batch, ..., lanes="auto") (engine.SubBatchPlan: two half-batch chains, no split-K, uniform patch tiles);
    "pipelined" = what bench.py times by default (a slot of engine.PipelinedPlan)."""
    from sipmask_amd.engine import SipMaskEngine
    dev = torch.device("cuda")
    det, sd, img = build_case(depth, batch)
    ora = oracle_outputs(sd, img, depth, "r%d_b%d" % (depth, batch))
    sd["bbox_head.fcos_cls.bias"].fill_(ora["cls_bias"])
    kw = {} if precision == "bf16" else dict(precision=precision)
    report = dict(depth=depth, batch=batch, hw=[IMG_H, IMG_W], precision=precision, plan=plan, oracle_seconds=ora["seconds"])
    if plan == "subbatch":
        with torch.no_grad():
            det.bbox_head.fcos_cls.bias.fill_(ora["cls_bias"])
        eng = det.prepare(batch, (IMG_H, IMG_W), (IMG_H, 1333, 3), precision=precision, lanes="auto")
        report["chains"] = [e.batch for e in getattr(eng, "engines", [eng])]
    elif plan == "pipelined":
        # what bench.py times by default: one slot of det.prepare(batch, ..., in_flight=3) (engine.PipelinedPlan: complete
        # single-chain plans built for CU time -- big tiles, no split-K, no side lanes); the slots are identical plans and
        # a step is one of them run on one batch, so the comparison runs slot 0 through the pipeline's own submit / results
        with torch.no_grad():
            det.bbox_head.fcos_cls.bias.fill_(ora["cls_bias"])
        pipe = det.prepare(batch, (IMG_H, IMG_W), (IMG_H, 1333, 3), precision=precision, in_flight=3)
        report["steps_in_flight"] = pipe.depth
        pipe.run(img.to(dev))                                  # slot 0 (captures its graph)
        eng = pipe.plans[0]
    else:
        eng = SipMaskEngine(sd, batch, (IMG_H, IMG_W), depth, img_shape=(IMG_H, 1333, 3), **kw)
    res = eng.run(img.to(dev))
    torch.cuda.synchronize()
    first = {k: v.clone() for k, v in res.items() if k != "masks"}
    report["image"], report["detections"] = compare_plan(eng, res, ora, batch, True, with_masks=(precision != "bf16"))
    report["parity"] = parity_summary(report["image"], report["detections"])
    # run to run: a second pass of the same plan over the same images must reproduce every integer output bit for bit
    res2 = eng.run(img.to(dev))
    torch.cuda.synchronize()
    report["rerun_bit_identical"] = bool(all(torch.equal(first[k], res2[k]) for k in first))
    del eng
    torch.cuda.empty_cache()
    if features_too:
        sizes = [tuple(p.shape[-2:]) for p in ora["pyr"]]
        hsd = {k: v for k, v in sd.items() if k.startswith("bbox_head.")}
        # (plan="pipelined": the head-only plan is built like a slot of the pipeline -- same kernels, tiles and launch shapes
        # as the timed head; bench.py's parity_on_identical_features does the same)
        heng = SipMaskEngine.for_head(hsd, batch, sizes, img_shape=(IMG_H, 1333, 3), pipelined=(plan == "pipelined"), **kw)
        heng.load_pyramid([p.to(dev) for p in ora["pyr"]])
        heng.run_head(with_post=True)
        torch.cuda.synchronize()
        report["features"] = compare_engine(heng, ora, batch, False)
        report["features_detections"] = compare_detections(heng, heng.results(), ora, batch, with_masks=False)
        report["features_parity"] = parity_summary(report["features"], report["features_detections"])
    if verbose:
        for sec in ("image", "features"):
            if sec in report:
                for k, v in report[sec].items():
                    print("%-9s %-12s rel %.3e  max_abs %.3e  (ref max %.3g rms %.3g)" %
                          (sec, k, v["rel_fro"], v["max_abs"], v["ref_max_abs"], v["ref_rms"]))
        print("detections:", json.dumps(report["detections"]))
        print("parity:", json.dumps(report["parity"]), "rerun_bit_identical:", report["rerun_bit_identical"])
        if "features_parity" in report:
            print("features parity:", json.dumps(report["features_parity"]))
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--plan", default="single", choices=("single", "subbatch", "pipelined"))
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rep = run(a.depth, a.batch, a.precision, plan=a.plan)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)
