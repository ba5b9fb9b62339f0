"""End-to-end GPU parity: the bf16 HIP engine against the fp32 CPU oracle on the same seeded
weights and image, stage by stage.  bf16 storage of ~60 stacked conv layers cannot meet 1e-3 on
logits end to end (SURVEY section 7 "hard parts"); the stated bound here is a relative Frobenius
error per stage, and the tight (bit-exact / 1e-5) parity lives in test_gpu_kernels.py where
both sides see identical inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as OM  # noqa: E402


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.engine import SipMaskEngine
    torch.manual_seed(0)
    B, Hh, Ww = 2, 192, 256
    sd = OM.init_state_dict(50, 0)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(B, 3, Hh, Ww, generator=g)
    feats = OM.backbone_forward(sd, img)
    pyr = OM.fpn_forward(sd, feats)
    # calibrate fcos_cls.bias so that a few hundred scores pass score_thr
    out = OM.head_forward(sd, pyr)
    allc = torch.cat([c[0].reshape(-1) for c in out[0]]) - sd["bbox_head.fcos_cls.bias"][0]
    OM.calibrate_cls_bias(sd, allc, target=400)
    out = OM.head_forward(sd, pyr, return_aux=True)
    eng = SipMaskEngine(sd, B, (Hh, Ww), 50)
    res = eng.run(img.cuda())
    torch.cuda.synchronize()
    return dict(sd=sd, img=img, feats=feats, pyr=pyr, out=out, eng=eng, res=res, B=B, hw=(Hh, Ww))


def test_backbone_fpn_features(setup):
    eng, B = setup["eng"], setup["B"]
    for i, (buf, h, w, c) in enumerate(eng.backbone_feats):
        got = buf.float().view(B, h, w, c).permute(0, 3, 1, 2)
        r = _rel(got, setup["feats"][i])
        assert r < 0.02, ("C%d" % (i + 2), r)
    lv = eng.lv
    for l, (h, w) in enumerate(lv.sizes):
        got = eng.pyr[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, 256).permute(0, 3, 1, 2)
        r = _rel(got, setup["pyr"][l])
        assert r < 0.03, ("P%d" % (l + 3), r)


def test_head_outputs(setup):
    eng = setup["eng"]
    cls, bb, ctr, cof, fm = eng.head_outputs()
    ocls, obb, octr, ocof, ofm, aux = setup["out"]
    for l in range(5):
        assert cls[l].shape == ocls[l].shape and bb[l].shape == obb[l].shape
        assert ctr[l].shape == octr[l].shape and cof[l].shape == ocof[l].shape
        assert _rel(bb[l], obb[l]) < 0.05, ("bbox", l, _rel(bb[l], obb[l]))
        assert _rel(cof[l], ocof[l]) < 0.15, ("cof", l, _rel(cof[l], ocof[l]))
        # cls logits have a large constant bias; compare after removing it
        b0 = float(setup["sd"]["bbox_head.fcos_cls.bias"][0])
        assert _rel(cls[l] - b0, ocls[l] - b0) < 0.15, ("cls", l, _rel(cls[l] - b0, ocls[l] - b0))
    assert fm.shape == ofm.shape
    assert _rel(fm, ofm) < 0.08, _rel(fm, ofm)


def test_postprocess_matches_oracle_on_engine_head_outputs(setup):
    """Feed the ENGINE's own head outputs to the oracle post-processing: from there on both sides see
    identical f32 inputs, so keep indices / labels must be bit-exact and masks identical away from
    the 0.4 threshold."""
    eng, res, B = setup["eng"], setup["res"], setup["B"]
    cls, bb, ctr, cof, fm = [[t.cpu().float() for t in x] if isinstance(x, list) else x.cpu().float()
                             for x in eng.head_outputs()]
    Hh, Ww = setup["hw"]
    total = 0
    for b in range(B):
        r = OM.get_masks_single([c[b] for c in cls], [x[b] for x in bb], [c[b] for c in ctr], [c[b] for c in cof],
                                fm[b], (Hh, Ww, 3), OM.DEFAULT_TEST_CFG)
        n = int(res["ndet"][b])
        total += n
        assert n == r["det_bboxes"].shape[0]
        np.testing.assert_array_equal(res["idxs_keep"][b, :n].cpu().numpy(), r["idxs_keep"])
        np.testing.assert_array_equal(res["det_labels"][b, :n].cpu().numpy(), r["det_labels"])
        np.testing.assert_allclose(res["det_bboxes"][b, :n].cpu().numpy(), r["det_bboxes"], rtol=1e-6, atol=1e-6)
        if n:
            gm = res["masks"][b, :n].cpu()
            diff = gm != r["masks"]
            assert bool(((r["up"] - 0.4).abs()[diff] < 1e-4).all())
            assert int(diff.sum()) <= max(5, int(1e-5 * diff.numel()))
    assert total > 0, "calibration failed: no detections, NMS/mask path not exercised"


def test_r101_backbone_features():
    """BASELINE config #3 (SipMask-R101): same engine, depth 101 (layer3 has 23 blocks)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.engine import SipMaskEngine
    sd = OM.init_state_dict(101, 0)
    img = torch.randn(1, 3, 128, 160, generator=torch.Generator().manual_seed(3))
    eng = SipMaskEngine(sd, 1, (128, 160), 101)
    eng.run(img.cuda())
    torch.cuda.synchronize()
    feats = OM.backbone_forward(sd, img, 101)
    pyr = OM.fpn_forward(sd, feats)
    # 23 blocks: 3 convs each + the shortcut conv (SIPMASK_PAIR_1X1=1, an A/B: conv3 of block i + conv1 of block i + 1 per launch)
    npair = len([t for t in eng.fused if t.name.startswith("backbone.layer3.") and t.name.endswith(".conv3+")])
    assert len([c for c in eng.convs if c.name.startswith("backbone.layer3.")]) == 23 * 3 + 1 - 2 * npair
    for i, (buf, h, w, c) in enumerate(eng.backbone_feats):
        got = buf.float().view(1, h, w, c).permute(0, 3, 1, 2)
        assert _rel(got, feats[i]) < 0.03, (i, _rel(got, feats[i]))
    lv = eng.lv
    h, w = lv.sizes[0]
    got = eng.pyr[:h * w].float().view(1, h, w, 256).permute(0, 3, 1, 2)
    assert _rel(got, pyr[0]) < 0.04


def test_sipmask_pp_dcn_backbone_and_rescoring():
    """SipMask++ (configs/sipmask/sipmask++_r101_caffe_fpn_ssd_6x.py, at R50 depth for test time): DeformConvPack
    in block 0 and every 3rd block of stages 2-4, SSD head layout, mask rescoring branch.  Stage parity of the
    backbone, and the rescoring chain on the engine's own cropped masks / detections against the oracle
    (bf16 convs vs f32: 3 % of the largest score)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sipmask_amd.engine import SipMaskEngine
    dcn = (False, True, True, True)
    sd = OM.init_state_dict(50, 2, stacked_convs=2, norm=False, stage_with_dcn=dcn, rescoring=True)
    sd["bbox_head.fcos_cls.bias"].fill_(-3.0)
    img = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    cfg = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.1, nms=dict(type='nms', iou_thr=0.5), max_per_img=100)
    eng = SipMaskEngine(sd, 1, (256, 256), 50, cfg, ssd_flag=True)
    r = eng.run(img.cuda())
    torch.cuda.synchronize()
    ndcn = len([c for c in eng.convs if c.name.endswith("conv2.conv_offset")])
    assert ndcn == 2 + 2 + 1 and eng.rescorer is not None
    feats = OM.backbone_forward(sd, img, 50)
    for i, (buf, h, w, c) in enumerate(eng.backbone_feats):
        got = buf.float().view(1, h, w, c).permute(0, 3, 1, 2)
        assert _rel(got, feats[i]) < 0.03, (i, _rel(got, feats[i]))
    n = int(r["ndet"][0])
    assert n > 3
    pos = eng.rescorer.pos_masks[0, :n].cpu()
    lab = r["det_labels"][0, :n].cpu()
    sc = r["det_bboxes"][0, :n, 4].cpu()
    ref = OM.mask_rescoring(sd, pos, lab, sc)
    got = r["mask_scores"][0, :n].cpu()
    assert float(ref.max()) > 0
    assert float((got - ref).abs().max()) < 0.03 * float(ref.max()), (got, ref)
    assert float(r["mask_scores"][0, n:].abs().max()) == 0.0 if n < eng.max_num else True


def _head_bits(e):
    """the tensors every integer decision of get_bboxes hangs off, as raw bits"""
    return [t.clone() for t in (e.cls_cof, e.reg_out, e.basis_lo)]


def test_sub_batch_plan_matches_single_plan(monkeypatch):
    """SipMask.prepare(lanes=2) (engine.SubBatchPlan: two concurrent half-batch launch chains writing slices of one
    set of outputs) against the single plan of the same batch: the same kernels on the same images.  GroupNorm
    statistics are fixed-point integer sums of per-position partials (csrc/common.h: gn_fix), i.e. independent of
    arrival order AND of how a plan cuts its tensors into tiles, so the two plans agree BIT FOR BIT: head outputs,
    keep indices, labels, boxes, masks (VERDICT r2 #1b; the reference's GroupNorm is deterministic, norm.py:12-55)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sipmask_amd.engine as E
    from sipmask_amd.engine import SubBatchPlan
    from sipmask_amd.synthetic import build_synthetic_detector
    # sub-plans run without split-K (engine.py: sub_plan); the single plan of this comparison must sum in the same order
    monkeypatch.setattr(E, "_SPLIT_K", False)
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.0)
    img = torch.randn(4, 3, 192, 256, generator=torch.Generator().manual_seed(5)).cuda()
    one = det.prepare(4, (192, 256), (192, 256, 3), lanes=1)
    r1 = {k: v.clone() for k, v in one.run(img).items()}
    cc1, rg1 = one.cls_cof.clone(), one.reg_out.clone()
    two = det.prepare(4, (192, 256), (192, 256, 3))                      # "auto": 2 lanes for an even batch >= 4
    assert isinstance(two, SubBatchPlan) and len(two.engines) == 2
    r2 = two.run(img)
    torch.cuda.synchronize()
    assert int(r1["ndet"].sum()) > 0
    lv = one.lv
    for i, e in enumerate(two.engines):
        for l, (h, w) in enumerate(lv.sizes):
            sl1 = slice(lv.row0[l] + 2 * i * h * w, lv.row0[l] + (2 * i + 2) * h * w)
            sl2 = slice(e.lv.row0[l], e.lv.row0[l] + 2 * h * w)
            assert torch.equal(e.cls_cof[sl2], cc1[sl1]), ("cls_cof", i, l)
            assert torch.equal(e.reg_out[sl2], rg1[sl1]), ("reg_out", i, l)
    for k in ("ndet", "idxs_keep", "det_labels", "det_bboxes", "masks"):
        assert torch.equal(r1[k], r2[k]), k
    # a second run with other images reuses the plans and the shared output tensors
    r3 = two.run(torch.flip(img, dims=[0]))
    torch.cuda.synchronize()
    assert r3["ndet"].data_ptr() == r2["ndet"].data_ptr()
    # run to run: two eager runs and hipGraph replays (one graph per sub-plan, each with its internal side lanes, on
    # concurrent streams) of the SAME plan on the same static images give the same bits
    static = img.clone()
    ref = {k: v.clone() for k, v in two.run(static).items()}
    bits = [_head_bits(e) for e in two.engines]
    torch.cuda.synchronize()
    again = two.run(static)
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(ref[k], again[k]), ("eager rerun", k)
    two.capture(static, multi_stream=True)
    for k in two.out:
        two.out[k].zero_()
    for _ in range(3):
        r4 = two.replay()
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(ref[k], r4[k]), ("graph replay", k)
        for e, b in zip(two.engines, bits):
            for t, t0 in zip(_head_bits(e), b):
                assert torch.equal(t, t0)
    for e in two.engines:
        e.multi_stream = False


@pytest.mark.parametrize("depth", [2, 3])
def test_pipelined_plan_equals_one_step_at_a_time(depth, monkeypatch):
    """engine.PipelinedPlan (SipMask.prepare(in_flight=N): N complete plans, each with its own buffers / hipGraph / stream,
    steps submitted back to back) against the single plan run one step at a time: 24 steps over 5 different batches, every
    step's detections, keep indices and masks equal bit for bit -- the steps in flight share weights and nothing else."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sipmask_amd.engine as E
    from sipmask_amd.engine import PipelinedPlan
    from sipmask_amd.synthetic import build_synthetic_detector
    # slots run without split-K (engine.py: pipelined); the single plan of this comparison must sum in the same order
    monkeypatch.setattr(E, "_SPLIT_K", False)
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.0)
    g = torch.Generator().manual_seed(17)
    batches = [torch.randn(4, 3, 192, 256, generator=g).cuda() for _ in range(5)]
    one = det.prepare(4, (192, 256), (192, 256, 3), lanes=1)
    keys = ("ndet", "idxs_keep", "det_labels", "det_bboxes", "masks")
    want = []
    for b in batches:
        r = one.run(b)
        torch.cuda.synchronize()
        want.append({k: r[k].clone() for k in keys})
    assert sum(int(w["ndet"].sum()) for w in want) > 0
    assert any(not torch.equal(want[0]["masks"], w["masks"]) for w in want[1:])     # the batches do differ
    pipe = det.prepare(4, (192, 256), (192, 256, 3), in_flight=depth)
    assert isinstance(pipe, PipelinedPlan) and pipe.depth == depth and len({id(p) for p in pipe.plans}) == depth
    assert all(p.extra_conv_flags and not p.split_k and not p.multi_stream for p in pipe.plans)
    assert det.prepare(4, (192, 256), (192, 256, 3), in_flight=depth) is pipe        # cached like every plan
    order = [(3 * i + i // 5) % 5 for i in range(24)]
    got, pending = [], []
    for step, bi in enumerate(order):
        slot = pipe.submit(batches[bi])
        pending.append((slot, bi))
        if len(pending) == depth:               # read a slot's results before it is submitted again
            s0, b0 = pending.pop(0)
            r = pipe.results(s0)
            got.append((b0, {k: r[k].clone() for k in keys}))
    for s0, b0 in pending:
        r = pipe.results(s0)
        got.append((b0, {k: r[k].clone() for k in keys}))
    torch.cuda.synchronize()
    assert len(got) == 24
    for i, (bi, r) in enumerate(got):
        for k in keys:
            assert torch.equal(r[k], want[bi][k]), (i, bi, k)
    # the plain interface: one step, start to finish
    r = pipe.run(batches[2])
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(r[k], want[2][k]), k


def test_fused_bottleneck_plan_is_bit_identical(monkeypatch):
    """The launch plan with fused bottleneck tails in layer1 / layer2 (conv2+conv3, and conv2+conv3+next conv1) gives the
    same bits as the plan of separate conv launches: C2..C5 features, head outputs and the detections."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sipmask_amd.engine as E
    sd = OM.init_state_dict(50, 0)
    sd["bbox_head.fcos_cls.bias"].fill_(-2.5)
    img = torch.randn(2, 3, 160, 224, generator=torch.Generator().manual_seed(9)).cuda()
    outs = {}
    # (layer1.0's fused SHORTCUT conv -- round 4 -- keeps the shortcut in f32 where the separate launch rounds it to bf16:
    # not bit-identical by design, switched off here and covered by the next test)
    monkeypatch.setattr(E, "_FUSE_SHORTCUT", 0)
    for mode in (0, 1, 2, 3, 4):                 # 3 = tails everywhere, chained conv1 in layer1 only; 4 = + layer3's 1x1 pairs
        monkeypatch.setattr(E, "_FUSE_BOTTLENECK", min(mode, 2) if mode < 3 else 1)
        monkeypatch.setattr(E, "_CHAIN_CONV1", 1 if mode >= 3 else 0)
        monkeypatch.setattr(E, "_PAIR_1X1", mode == 4)
        eng = E.SipMaskEngine(sd, 2, (160, 224), 50)
        tails = [t for t in eng.fused if hasattr(t, "w2")]
        assert len(tails) == (0 if mode == 0 else 7) and len(eng.fused) - len(tails) == (5 if mode == 4 else 0)
        assert sum(t.w1n is not None for t in tails) == {0: 0, 1: 0, 2: 5, 3: 2, 4: 2}[mode]
        r = eng.run(img)
        torch.cuda.synchronize()
        outs[mode] = ([f[0].clone() for f in eng.backbone_feats], _head_bits(eng),
                      [r[k].clone() for k in ("ndet", "idxs_keep", "det_labels", "det_bboxes", "masks")])
    for mode in (1, 2, 3, 4):
        for a, b in zip(outs[0][0], outs[mode][0]):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), mode
        # the head behind the (bit-identical) features is reproducible too: its GroupNorm statistics are integer sums
        for a, b in zip(outs[0][1] + outs[0][2], outs[mode][1] + outs[mode][2]):
            assert torch.equal(a, b), mode
    assert int(outs[0][2][0].sum()) > 0


def test_fused_shortcut_conv_plan_matches_separate_launch(monkeypatch):
    """round 4: layer1's first block with its 1x1 shortcut conv inside the fused tail (csrc/bottleneck.hip CDS; the 256-channel
    shortcut tensor is never written) against the plan that runs the shortcut as its own launch: C2..C5 features within the
    one bf16 rounding the separate launch adds (the fused path keeps the shortcut in f32 until the block output)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import sipmask_amd.engine as E
    sd = OM.init_state_dict(50, 0)
    img = torch.randn(2, 3, 160, 224, generator=torch.Generator().manual_seed(9)).cuda()
    feats = {}
    for on in (False, True):
        monkeypatch.setattr(E, "_FUSE_SHORTCUT", 2 if on else 0)
        eng = E.SipMaskEngine(sd, 2, (160, 224), 50)
        assert sum(t.x_block is not None for t in eng.fused) == (2 if on else 0)        # layer1's and layer2's first blocks
        assert any(c.name == "backbone.layer1.0.downsample" for c in eng.convs) == (not on)
        assert any(c.name == "backbone.layer2.0.downsample" for c in eng.convs) == (not on)
        assert any(c.name == "backbone.layer3.0.downsample" for c in eng.convs)
        eng.run(img)
        torch.cuda.synchronize()
        feats[on] = [f[0].float().clone() for f in eng.backbone_feats]
    # a bf16 ulp is 2^-8 relative and a fraction of layer1.0's outputs move by one; the 13 blocks behind C2 carry that on
    # (two bf16 pipelines of this random-weight trunk agree to ~1 %: test_backbone_fpn_features holds each to 2 % of the oracle)
    rels = [float((a - b).norm() / a.norm()) for a, b in zip(feats[False], feats[True])]
    assert rels[0] < 5e-3 and max(rels) < 2e-2, rels


@pytest.mark.parametrize("rescale", [False, True])
def test_mixed_size_batch_reads_img_shape_and_scale_factor_per_image(rescale):
    """VERDICT r2 #6 / ADVICE r1: a keep_ratio pipeline gives every image of a batch its own img_shape and scale_factor,
    and the reference post-processes image i with img_metas[i] (sipmask_head.py:517-541: box clamp :579, rescale
    :587-588, crop boxes * scale_factor / 2 :623, mask upsampling by 2 / scale_factor :632).  Three images of different
    sizes through input_pipeline.prepare_batch -> SipMask.get_masks: against the oracle's get_masks_single fed with the
    plan's OWN head outputs and image i's metas -- keep indices, labels bit-exact, boxes 1e-6, masks (each on its own
    Ho x Wo) equal away from the threshold; and the RLE strings, each on its own canvas."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import ops as O
    from sipmask_amd.input_pipeline import prepare_batch
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.2)
    g = torch.Generator().manual_seed(11)
    shapes = [(150, 200), (120, 260), (210, 170)]                       # original sizes: three different scale factors
    imgs = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8).cuda() for h, w in shapes]
    batch, metas = prepare_batch(imgs, img_scale=(320, 224))
    sfs = [float(m["scale_factor"]) for m in metas]
    assert len(set(sfs)) == 3 and len(set(m["img_shape"] for m in metas)) == 3
    plan = det.plan_for_metas(3, tuple(batch.shape[-2:]), metas, rescale=rescale)
    again = det.plan_for_metas(3, tuple(batch.shape[-2:]), list(reversed(metas)), rescale=rescale)
    assert again is plan                                                 # same canvas bounds -> the same cached plan
    r = det.get_masks(batch, metas, rescale=rescale)
    torch.cuda.synchronize()
    assert r["out_hw"] == plan.out_hw and len(set(r["out_hw"])) == 3
    cls, bb, ctr, cof, fm = [[t.cpu().float() for t in x] if isinstance(x, list) else x.cpu().float()
                             for x in plan.head_outputs()]
    canv = [tuple((m["ori_shape"] if rescale else m["img_shape"])[:2]) for m in metas]
    rles = plan.encode_rle(canv)
    total = 0
    for b in range(3):
        ref = OM.get_masks_single([c[b] for c in cls], [x[b] for x in bb], [c[b] for c in ctr], [c[b] for c in cof], fm[b],
                                  metas[b]["img_shape"], OM.DEFAULT_TEST_CFG, metas[b]["scale_factor"], rescale)
        n = int(r["ndet"][b])
        total += n
        assert n == ref["det_bboxes"].shape[0]
        np.testing.assert_array_equal(r["idxs_keep"][b, :n].cpu().numpy(), ref["idxs_keep"])
        np.testing.assert_array_equal(r["det_labels"][b, :n].cpu().numpy(), ref["det_labels"])
        np.testing.assert_allclose(r["det_bboxes"][b, :n].cpu().numpy(), ref["det_bboxes"], rtol=1e-6, atol=1e-6)
        if n == 0:
            continue
        ho, wo = r["out_hw"][b]
        assert tuple(ref["masks"].shape[-2:]) == (ho, wo)
        gm = r["masks"][b, :n, :ho, :wo].cpu()
        diff = gm != ref["masks"]
        assert bool(((ref["up"] - 0.4).abs()[diff] < 1e-4).all()) and int(diff.sum()) <= max(5, int(1e-5 * diff.numel()))
        assert len(rles[b]) == n
        for i in range(n):
            assert rles[b][i]["size"] == list(canv[b])
            assert rles[b][i]["counts"] == O.paste_and_encode(gm[i].numpy(), canv[b])["counts"]
    assert total > 10
    # a plan whose canvas is too small for a batch says so instead of writing outside its masks
    small = det.prepare(3, tuple(batch.shape[-2:]), None, max(sfs) * 1.5, rescale)
    with pytest.raises(ValueError):
        small.set_image_metas(metas)


@pytest.mark.parametrize("env,cycles", [({}, 2000), ({"AMD_SERIALIZE_KERNEL": "3"}, 500), ({"HSA_ENABLE_SDMA": "0"}, 500),
                                        ({"SIPMASK_STRESS_POISON": "1"}, 500),
                                        ({"SIPMASK_STRESS_DEPTH": "6", "SIPMASK_STRESS_SHAPE": "128,160,1"}, 4000),
                                        ({"SIPMASK_STRESS_SHAPE": "192,256,4", "SIPMASK_STRESS_BURST": "6000"}, 24)])
def test_pipelined_plan_stress(env, cycles):
    """VERDICT r5 #3 (an unexplained SIGABRT inside torch.cuda.synchronize() of a PipelinedPlan test on one box in round 5) --
    ROOT-CAUSED in round 6 with this test as the reproducer.  Thousands of submit(pack=True) / fetch cycles over the slots of a
    PipelinedPlan with hipGraph replay, batches and img_metas varying from step to step, every result held to the single
    plan's, the slot's stream queried after every submit -- in its own process (tests/_pipeline_stress_worker.py): as shipped,
    with every kernel serialised by the runtime (AMD_SERIALIZE_KERNEL=3), with the SDMA engines off, with every
    uninitialised buffer of the plans poisoned (0x7f bytes: results must not depend on what an allocation held before), in the
    configuration in which the fault was first bisected (six slots of one 128 x 160 image) and in the one that reproduces it
    in every run (four 192 x 256 images per step, 6 000 steps submitted back to back).
    The cause ("Memory access fault by GPU node ... Reason: Unknown", then SIGABRT): a barrier race in
    nms_class_kernel's bitonic sort -- its keys are reached through a generic pointer, hipcc emits flat_store and leaves the
    next step's s_barrier without a wait, so beside another step's LDS traffic a padding key could end up among the valid
    ones and its candidate index 0xffffffff sent a load 64 GB past the boxes: 8 of 8 workers of the last variant died within
    3 000 steps, 0 of 6 in 10 000 with sm_syncthreads_flat() (profiles/r06_pipeline_nms_sort_race.txt).  (A first bisection
    had blamed the six hipMemcpyAsync device -> pinned-host copies behind every step -- 27 of 199 workers of the six-slot
    variant died with them, 0 of 130 without; with the sort fixed the same copies give 0 of 24: they had widened the
    race's window.  They are one sm_copy_segments launch now for the host calls it saves.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pipeline_stress_worker.py")
    r = subprocess.run([sys.executable, worker, str(cycles)], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and ("PIPELINE_STRESS_OK %d" % cycles) in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_pipelined_pack_prefix_adapts_and_long_batches_fall_back(monkeypatch):
    """PipelinedPlan.fetch with RLE strings longer than the prefix that travelled with the step (round 6: the prefix starts
    small and adapts): the rest comes from the slot's double-buffered device strings by a second copy, the slot's next packs
    carry twice the batch seen, and every RLE dict equals the single plan's."""
    import sipmask_amd.engine as E
    from sipmask_amd.synthetic import build_synthetic_detector
    monkeypatch.setattr(E, "_SPLIT_K", False)
    monkeypatch.setattr(E.PipelinedPlan, "PACK_PREFIX_MIN", 64)          # bytes: every batch of this test is longer
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.0)
    g = torch.Generator().manual_seed(29)
    H_, W_ = 192, 256
    batches = [torch.randn(2, 3, H_, W_, generator=g).cuda() for _ in range(3)]
    one = det.prepare(2, (H_, W_), (H_, W_, 3), lanes=1)
    want = []
    for b in batches:
        r = one.run(b)
        torch.cuda.synchronize()
        want.append(one.encode_rle((H_, W_)))
    assert sum(len(x["counts"]) for w in want for img in w for x in img) > 3 * 64
    pipe = det.prepare(2, (H_, W_), (H_, W_, 3), in_flight=2)
    seen = []
    for rep in range(3):
        for bi, b in enumerate(batches):
            k = pipe.submit(b, pack=True, canvas_hw=(H_, W_))
            res = pipe.fetch(k)
            for img in range(2):
                assert res[img][2] == want[bi][img], (rep, bi, img)
            seen.append(dict(pipe._pack_prefix))
    assert seen[0] and all(v >= 128 and (v & (v - 1)) == 0 for v in seen[-1].values())      # grown, a power of two
    assert seen[-1] == seen[-4]                                                            # and settled


def test_pipelined_submit_packs_results_and_keeps_metas_per_slot(monkeypatch):
    """PipelinedPlan.submit(img, img_metas, pack=True) / fetch (round 4):
    (1) result packing behind every step on the slot's stream (sm_mask_rects + sm_rle_encode + asynchronous copies into
        pinned buffers) returns per image exactly what the single plan's results() + encode_rle() give
        (sipmask_head.py:645-662: boxes, labels, one RLE dict per detection);
    (2) img_metas travel with the SUBMIT: consecutive batches with different img_shape / scale_factor are in flight at the
        same time and every step is post-processed with its own metas (set_image_metas used to rewrite the
        tables of every slot at once, under steps still running)."""
    import sipmask_amd.engine as E
    from sipmask_amd.synthetic import build_synthetic_detector
    # slots run without split-K (engine.py: pipelined); the single plan of this comparison must sum in the same order
    monkeypatch.setattr(E, "_SPLIT_K", False)
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.0)
    g = torch.Generator().manual_seed(23)
    H_, W_ = 192, 256
    batches = [torch.randn(2, 3, H_, W_, generator=g).cuda() for _ in range(4)]
    # two meta sets: full-size images, and images whose valid area is smaller (boxes clamp to img_shape)
    metas = [[dict(img_shape=(H_, W_, 3), scale_factor=1.0)] * 2,
             [dict(img_shape=(150, 200, 3), scale_factor=1.0), dict(img_shape=(176, 230, 3), scale_factor=1.0)]]
    one = det.prepare(2, (H_, W_), (H_, W_, 3), lanes=1)
    want = []
    for i, b in enumerate(batches):
        one.set_image_metas(metas[i % 2])
        r = one.run(b)
        torch.cuda.synchronize()
        rle = one.encode_rle((H_, W_))
        nd = r["ndet"].cpu().tolist()
        want.append([(r["det_bboxes"][k, :nd[k]].cpu().numpy().copy(), r["det_labels"][k, :nd[k]].cpu().numpy().copy(), rle[k])
                     for k in range(2)])
    assert sum(len(w[2]) for ws in want for w in ws) > 0
    # the two meta sets do change the outcome (otherwise the test would not see a mix-up)
    one.set_image_metas(metas[0])
    r = one.run(batches[1])
    torch.cuda.synchronize()
    assert not np.array_equal(r["det_bboxes"][0, :int(r["ndet"][0])].cpu().numpy(), want[1][0][0])
    pipe = det.prepare(2, (H_, W_), (H_, W_, 3), in_flight=3)
    got, pending = [], []
    order = [0, 1, 2, 3, 1, 0, 3, 2, 0, 1]
    for bi in order:
        # submit first, read the step submitted `depth` steps ago afterwards: its slot is the one just resubmitted, so that
        # slot momentarily holds TWO unread result sets (double-buffered pinned buffers) and three steps stay in flight
        pending.append((pipe.submit(batches[bi], img_metas=metas[bi % 2], pack=True, canvas_hw=(H_, W_)), bi))
        if len(pending) > pipe.depth:
            s0, b0 = pending.pop(0)
            assert s0 == pending[-1][0]
            got.append((b0, pipe.fetch(s0)))
    for s0, b0 in pending:
        got.append((b0, pipe.fetch(s0)))
    with pytest.raises(RuntimeError):
        pipe.fetch(0)                                    # nothing unread left
    for _ in range(2 * pipe.depth):
        pipe.submit(batches[0], pack=True, canvas_hw=(H_, W_))
    with pytest.raises(RuntimeError):                    # a third unread set on a slot would overwrite the first
        pipe.submit(batches[0], pack=True, canvas_hw=(H_, W_))
    assert len(got) == len(order)
    for bi, res in got:
        for k in range(2):
            np.testing.assert_array_equal(res[k][0], want[bi][k][0])
            np.testing.assert_array_equal(res[k][1], want[bi][k][1])
            assert res[k][2] == want[bi][k][2], (bi, k)
    assert pipe.out_hw == pipe.plans[pipe.last_slot].out_hw


def test_pipelined_pack_with_own_scale_factor_and_canvas_per_image(monkeypatch):
    """A keep_ratio batch in the pipeline (round 5): every image its own scale_factor -- hence its own mask size
    floor(Hm * 2 / scale_factor) inside the batch's planes (sipmask_head.py:621-633) -- and its own RLE canvas
    (img_shape, :645-653).  submit(img, img_metas, pack=True) packs them in ONE launch with a per-image table
    (sm_rle_encode_images) and returns exactly what the single plan + per-image oracle-checked encode_rle give."""
    import sipmask_amd.engine as E
    from sipmask_amd.synthetic import build_synthetic_detector
    monkeypatch.setattr(E, "_SPLIT_K", False)
    det = build_synthetic_detector(50, seed=0)
    with torch.no_grad():
        det.bbox_head.fcos_cls.bias.fill_(-2.0)
    g = torch.Generator().manual_seed(31)
    H_, W_ = 192, 256
    batches = [torch.randn(2, 3, H_, W_, generator=g).cuda() for _ in range(3)]
    metas = [[dict(img_shape=(180, 250, 3), ori_shape=(108, 150, 3), scale_factor=1.6667),
              dict(img_shape=(192, 240, 3), ori_shape=(96, 120, 3), scale_factor=2.0)],
             [dict(img_shape=(190, 256, 3), ori_shape=(101, 136, 3), scale_factor=1.875),
              dict(img_shape=(176, 230, 3), ori_shape=(110, 144, 3), scale_factor=1.6)]]
    # (a plan whose canvas / window bounds cover all four images: smallest and largest scale_factor of the run)
    one = det.prepare(2, (H_, W_), None, 1.5625, False, "bf16", 1, scale_factor_max=2.0)
    want = []
    for i, b in enumerate(batches):
        one.set_image_metas(metas[i % 2])
        r = one.run(b)
        torch.cuda.synchronize()
        assert len(set(one.out_hw)) == 2                            # two different mask sizes inside one batch
        rle = one.encode_rle()                                      # canvases from the metas: img_shape of every image
        nd = r["ndet"].cpu().tolist()
        for k in range(2):
            assert all(d["size"] == list(metas[i % 2][k]["img_shape"][:2]) for d in rle[k])
            # the strings decode to the masks the plan holds (cropped / padded to the canvas as the reference does)
            from oracle import ops as O
            ho, wo = one.out_hw[k]
            ch, cw = metas[i % 2][k]["img_shape"][:2]
            for j in range(min(nd[k], 5)):
                m = r["masks"][k, j, :ho, :wo].cpu().numpy()
                assert rle[k][j] == O.paste_and_encode(m, (ch, cw)), (i, k, j)
        want.append([(r["det_bboxes"][k, :nd[k]].cpu().numpy().copy(), r["det_labels"][k, :nd[k]].cpu().numpy().copy(), rle[k])
                     for k in range(2)])
    assert sum(len(w[2]) for ws in want for w in ws) > 0
    pipe = det.prepare(2, (H_, W_), None, 1.5625, False, "bf16", 1, scale_factor_max=2.0, in_flight=3)
    pending, got = [], []
    for bi in [0, 1, 2, 1, 0, 2, 2]:
        pending.append((pipe.submit(batches[bi], img_metas=metas[bi % 2], pack=True), bi))
        if len(pending) > pipe.depth:
            s0, b0 = pending.pop(0)
            got.append((b0, pipe.fetch(s0)))
    for s0, b0 in pending:
        got.append((b0, pipe.fetch(s0)))
    for bi, res in got:
        for k in range(2):
            np.testing.assert_array_equal(res[k][0], want[bi][k][0])
            np.testing.assert_array_equal(res[k][1], want[bi][k][1])
            assert res[k][2] == want[bi][k][2], (bi, k)


def test_forward_dummy_returns_the_head_outputs(setup):
    """SingleStageDetector.forward_dummy (single_stage.py:52-59): extract_feat + bbox_head without post-processing, the five
    output lists of SipMaskHead.forward -- the same tensors the full plan computes on the same image."""
    from sipmask_amd.synthetic import build_synthetic_detector
    det = build_synthetic_detector(50, seed=0)
    img = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(5)).cuda()
    outs = det.forward_dummy(img)
    torch.cuda.synchronize()
    assert len(outs) == 5 and [len(o) for o in outs[:4]] == [5] * 4
    assert tuple(outs[0][0].shape) == (2, 80, 16, 20) and tuple(outs[3][0].shape) == (2, 128, 16, 20)
    assert tuple(outs[4].shape) == (2, 32, 64, 80)
    got = [[t.clone() for t in o] for o in outs[:4]] + [outs[4].clone()]
    eng = det.prepare(2, (128, 160), lanes=1)
    eng.run(img)
    torch.cuda.synchronize()
    ref = eng.head_outputs()
    for a, b in zip(got[:4], ref[:4]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(got[4], ref[4])


def test_fpn_output_convs_as_one_launch_with_per_level_weights(setup, monkeypatch):
    """fpn_convs[0..2] on the merged laterals (fpn.py:154-157) as ONE launch of the patch-resident kernel with per-level
    weights and biases (sm_conv_desc.w_level_stride, engine._LevelConv; round 4): P3-P7 against the oracle at the bound of the
    separate launches, and against the plan of separate launches within bf16 accumulation-order rounding; P6 / P7 hang off
    the grouped launch's P5."""
    import sipmask_amd.engine as E
    from sipmask_amd.engine import SipMaskEngine
    B, (Hh, Ww) = setup["B"], setup["hw"]
    base = setup["eng"]
    assert not base.fpn_grouped                      # 24 tiles at this size: the planner keeps three launches
    monkeypatch.setattr(E, "_FPN_GROUPED", "1")
    eng = SipMaskEngine(setup["sd"], B, (Hh, Ww), 50)
    assert eng.fpn_grouped and [c.name for c in eng.convs if c.name.startswith("fpn.out")] == ["fpn.outs"]
    g = [c for c in eng.convs if c.name == "fpn.outs"][0]
    assert g.desc.w_level_stride == g.w[0].numel() and g.desc.bias_level_stride == 256 and g.desc.nlev == 3
    eng.run(setup["img"].cuda())
    torch.cuda.synchronize()
    lv = eng.lv
    for l, (h, w) in enumerate(lv.sizes):
        got = eng.pyr[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, 256).permute(0, 3, 1, 2)
        assert _rel(got, setup["pyr"][l]) < 0.03, ("P%d" % (l + 3))
        ref = base.pyr[lv.row0[l]:lv.row0[l] + B * h * w].float().view(B, h, w, 256).permute(0, 3, 1, 2)
        assert _rel(got, ref) < 6e-3, ("P%d vs separate launches" % (l + 3))
    # the weights of the levels really differ: level 1 through level 0's matrix would be far off
    w0 = setup["sd"]["neck.fpn_convs.0.conv.weight"]
    w1 = setup["sd"]["neck.fpn_convs.1.conv.weight"]
    assert float((w0 - w1).abs().max()) > 1e-3
