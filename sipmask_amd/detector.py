"""SipMask single-stage detector -- drop-in for ``@DETECTORS`` ``SipMask``
(M/mmdet/models/detectors/sipmask.py:5-16, single_stage.py:9-96, base.py:97-149).

extract_feat + bbox_head + get_bboxes run as ONE static HIP launch plan (sipmask_amd/engine.py).
"""
import torch
import torch.nn as nn

from . import benchmark_train, modules, sipmask_head  # noqa: F401  (register ResNet / FPN / SipMaskHead / FCOSSipMaskHead)
from .fp16 import auto_fp16
from .plan_cache import PlanCache, module_tensors
from .registry import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module
class SipMask(nn.Module):

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.fp16_enabled = False              # base.py:26; fp16.wrap_fp16_model switches it on
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._engines = PlanCache().attach_invalidation(self)
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return self.neck is not None

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str) and pretrained.startswith(("open-mmlab://", "http://", "https://", "modelzoo://")):
            # every reference sipmask config names a model-zoo URL (pretrained='open-mmlab://resnet50_caffe'); there is no
            # downloader here, so say loudly that the backbone starts from random weights instead of doing it silently
            import warnings
            warnings.warn("SipMask.init_weights: pretrained=%r cannot be resolved (no model-zoo access): the backbone is "
                          "RANDOMLY initialised -- pass a local checkpoint path or load_state_dict() afterwards" % pretrained,
                          RuntimeWarning, stacklevel=2)
            pretrained = None
        self.backbone.init_weights(pretrained=pretrained if isinstance(pretrained, str) else None)
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()
        self._engines.clear()

    def invalidate_plans(self):
        """Drop every cached launch plan.  Needed only after writing weights in a way the cache cannot see (through
        `.data` or a raw-pointer kernel: plan_cache.py); load_state_dict, optimizer steps and in-place tensor ops are seen."""
        self._engines.invalidate()
        head = getattr(self.bbox_head, "_engines", None)
        if head is not None:
            head.invalidate()

    def prepare(self, batch, img_hw, img_shape=None, scale_factor=1.0, rescale=False, precision="bf16", lanes="auto",
                scale_factor_max=None, in_flight=1, slot=0, pipelined=False):
        """Build (or fetch) the static launch plan for this input geometry; weights are snapshotted,
        BN folded, re-laid out as bf16 GEMM operands.  Call again after load_state_dict.
        img_shape / scale_factor are the DEFAULT metas of every image; plan.set_image_metas(img_metas) gives each image
        its own before a run (scale_factor then sizes the mask canvas: the batch's smallest; scale_factor_max: its largest).
        scale_factor / rescale: img_meta['scale_factor'] and the rescale flag of simple_test (boxes and masks
        in original-image coordinates, sipmask_head.py:587-588,621-632).  precision: "bf16" = the throughput plan,
        "f32" = the parity plan (exact-f32 MFMA convs, every tensor f32: held to the fp32 reference within
        accumulation-order rounding).  lanes: the batch as this many concurrent sub-batch plans (engine.SubBatchPlan);
        "auto" = 2 for even batches >= 4, else 1.  in_flight > 1: an engine.PipelinedPlan of that many complete plans
        (steps submitted back to back overlap: the throughput structure; `lanes` then describes each of them, "auto" = 1).
        slot: distinguishes otherwise identical plans in the cache; pipelined: the plan is a slot of a PipelinedPlan
        (SipMaskEngine(pipelined=True): no split-K, big tiles, no side lanes)."""
        import numpy as np
        from .engine import SipMaskEngine
        sf_key = tuple(np.asarray(scale_factor, np.float64).reshape(-1))
        sfm_key = None if scale_factor_max is None else tuple(np.asarray(scale_factor_max, np.float64).reshape(-1))

        def build_plan(lanes, pipelined):
            """one complete launch plan (uncached): a SipMaskEngine, or a SubBatchPlan of `lanes` half-batch chains"""
            if lanes == "auto":    # SubBatchPlan: two concurrent half-batch chains pay off from 2 images per chain on
                lanes = 2 if (batch >= 4 and batch % 2 == 0 and not getattr(self.bbox_head, "rescoring_flag", False)) else 1
            assert batch % lanes == 0
            sd = self.state_dict()
            mk = lambda b: SipMaskEngine(sd, b, img_hw, self.backbone.depth, self.test_cfg, self.bbox_head.num_classes,
                                         strides=self.bbox_head.strides, img_shape=img_shape,
                                         ssd_flag=self.bbox_head.ssd_flag, scale_factor=scale_factor, rescale=rescale,
                                         precision=precision, sub_plan=lanes > 1, scale_factor_max=scale_factor_max,
                                         pipelined=bool(pipelined) and lanes == 1)
            if lanes == 1:
                return mk(batch)
            from .engine import SubBatchPlan
            return SubBatchPlan([mk(batch // lanes) for _ in range(lanes)])

        if in_flight > 1:
            # ONE cache entry for the whole pipeline: its slots live inside the PipelinedPlan only (as separate entries they
            # took in_flight + 1 places of the LRU and evicted each other -- ADVICE r3); scale_factor_max is part of the key
            from .engine import PipelinedPlan
            ln = 1 if lanes == "auto" else lanes
            key = ("pipelined", batch, tuple(img_hw), tuple(img_shape or ()), sf_key, rescale, precision, ln, in_flight, sfm_key)
            return self._engines.get(key, module_tensors(self), lambda: PipelinedPlan(
                [build_plan(ln, True) for _ in range(in_flight)]))
        # (slot and pipelined are ALWAYS part of the key: a pipelined plan -- big tiles, no split-K, no side lanes -- must not
        # share an entry with the latency-shaped plan of the same shape)
        key = (batch, tuple(img_hw), tuple(img_shape or ()), sf_key, rescale, precision, lanes, sfm_key, int(slot), bool(pipelined))
        # plans are valid for the weights they were built from: PlanCache drops them when any parameter / buffer has
        # been updated in place since (optimizer.step, load_state_dict, mmcv load_checkpoint)
        return self._engines.get(key, module_tensors(self), lambda: build_plan(lanes, pipelined))

    def plan_for_metas(self, batch, img_hw, img_metas, rescale=False, precision="bf16", lanes="auto", in_flight=1):
        """The launch plan for a batch whose images carry their OWN img_shape / scale_factor (a keep_ratio pipeline:
        every image of a batch is resized by a different factor; the reference reads img_metas[img_id],
        sipmask_head.py:517-541).  The plan depends on the batch only through the mask canvas (the smallest scale_factor)
        and the kernels' source-window bound (the largest), both rounded outward to 1/16 so that consecutive batches of
        an evaluation run share plans; the per-image values go to the device tables (set_image_metas).
        in_flight > 1: the engine.PipelinedPlan of that geometry is returned as it is -- its metas travel with every
        submit(img, img_metas), each slot keeps the tables of the batch it was submitted with (the evaluation loop of
        M/mmdet/apis/test.py:12-72 over keep_ratio batches: bench.py --config eval_shapes)."""
        import numpy as np
        sfl = [np.asarray(m.get('scale_factor', 1.0), np.float64).reshape(-1) for m in img_metas]
        if any(a.size not in (1, 4) for a in sfl):
            raise ValueError("scale_factor is a scalar (keep_ratio) or [w, h, w, h]")
        scalar = all(a.size == 1 for a in sfl)
        sfs = np.stack([np.broadcast_to(a, (4,)) for a in sfl])
        lo = np.maximum(np.floor(sfs.min(0) * 16.0) / 16.0, 1.0 / 16.0)
        hi = np.ceil(sfs.max(0) * 16.0) / 16.0
        if scalar:
            lo, hi = float(lo[0]), float(hi[0])
        else:
            lo, hi = lo.astype(np.float32), hi.astype(np.float32)
        plan = self.prepare(batch, img_hw, None, lo, rescale, precision, lanes, scale_factor_max=hi, in_flight=in_flight)
        return plan if in_flight > 1 else plan.set_image_metas(img_metas)

    def get_masks(self, img, img_metas=None, rescale=False):
        """Batch-capable tensor-only inference (compare before RLE): dict of device tensors
        det_bboxes [B,max,5], det_labels [B,max], idxs_keep [B,max], ndet [B], masks u8 [B,max,Hc,Wc] on the batch's
        canvas, and "out_hw": every image's own (Ho, Wo) -- its masks are masks[b, :ndet[b], :Ho, :Wo]."""
        if not img_metas:
            eng = self.prepare(img.shape[0], tuple(img.shape[-2:]))
        else:
            eng = self.plan_for_metas(img.shape[0], tuple(img.shape[-2:]), img_metas, rescale)
        r = dict(eng.run(img))
        r["out_hw"] = list(eng.out_hw)
        return r

    def simple_test(self, img, img_meta, rescale=False):
        """single_stage.py:75-96: returns (bbox_results, segm_results) of image 0; segm_results[label] is the
        list of COCO RLE dicts of that class (sipmask_head.py:655-657), encoded on device."""
        meta = img_meta[0]
        shape = tuple(meta['img_shape'])
        eng = self.prepare(img.shape[0], tuple(img.shape[-2:]), shape, meta.get('scale_factor', 1.0), bool(rescale))
        r = eng.run(img)
        canvas = tuple(meta['ori_shape'])[:2] if rescale else shape[:2]              # sipmask_head.py:648-653
        rle = eng.encode_rle(canvas)[0]
        n = int(r["ndet"][0])
        d, l = r["det_bboxes"][0, :n].cpu().numpy(), r["det_labels"][0, :n].cpu().numpy()
        ncls = self.bbox_head.num_classes - 1
        bbox_results = [d[l == i, :] for i in range(ncls)]                       # bbox2result, transforms.py:181-199
        segm_results = [[rle[j] for j in range(n) if l[j] == i] for i in range(ncls)]
        if "mask_scores" in r:                   # SipMask++: (cls_segms, mask_scores), sipmask_head.py:659-660
            ms = r["mask_scores"][0, :n].cpu().numpy()
            segm_results = (segm_results, [ms[l == i] for i in range(ncls)])
        return bbox_results, segm_results

    def forward_dummy(self, img):
        """single_stage.py:52-59 (`tools/get_flops.py`): extract_feat + bbox_head, no post-processing.  Returns the head's
        outputs as SipMaskHead.forward does (cls_scores, bbox_preds, centernesses, cof_preds, feat_masks: NCHW views of
        the launch plan's buffers) -- the conv-only entry SURVEY 5 recommends for a roofline run."""
        eng = self.prepare(img.shape[0], tuple(img.shape[-2:]), lanes=1)
        return eng.run_convs(img)

    def forward_test(self, imgs, img_metas, **kwargs):
        assert len(imgs) == 1, "aug test is not on the SipMask path"
        assert imgs[0].size(0) == 1                      # base.py:118-119
        return self.simple_test(imgs[0], img_metas[0], **kwargs)

    def extract_feat_rows(self, img):
        """extract_feat (single_stage.py:41-47) on row tensors: (pyramid rows bf16 [sum_l B*h_l*w_l, C], Levels) -- the
        head's input layout, all levels in one matrix (ops_rows.py)."""
        from . import hip_ops as H
        outs = self.backbone.forward_rows(img)
        if self.with_neck:
            outs = self.neck.forward_rows(outs)
        rows = outs[0][0] if len(outs) == 1 else torch.cat([r for r, _ in outs])
        return rows, H.Levels(img.shape[0], [lv.sizes[0] for _, lv in outs])

    def extract_feat_train(self, img):
        """extract_feat (single_stage.py:41-47) as a differentiable graph of HIP autograd ops; NCHW tensors (views of the
        row tensors the graph runs on)."""
        from .modules import _train_rows_enabled
        if _train_rows_enabled(img) and self.with_neck:
            from .ops_rows import rows_to_nchw
            rows, lv = self.extract_feat_rows(img)
            return tuple(rows_to_nchw(rows[lv.row0[l]:lv.row0[l] + lv.batch * h * w], lv.batch, h, w)
                         for l, (h, w) in enumerate(lv.sizes))
        x = self.backbone.forward_train(img)
        return self.neck.forward_train(x) if self.with_neck else x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None):
        """single_stage.py:49-73: losses of one batch.  Backbone (BN frozen, stages <= frozen_stages without a
        graph), FPN and head run layer by layer on the HIP forward/backward ops; `.backward()` on the summed losses
        fills every trainable parameter's .grad."""
        self.bbox_head.train()
        from .modules import _train_rows_enabled
        kw = {}
        if _train_rows_enabled(img) and self.with_neck and hasattr(self.bbox_head, "forward_rows") and \
                self.bbox_head.rows_path_ok():
            if gt_masks is not None and hasattr(self.bbox_head, "prepare_targets"):
                # the ground-truth half of the loss first: its nonzero() syncs then wait on an idle device instead of
                # draining the forward's launch queue in the middle of the step (the loss checks the geometry it was built for)
                sizes = self._head_sizes(tuple(img.shape[-2:]))
                kw["_targets"] = self.bbox_head.prepare_targets(sizes, (4 * sizes[0][0], 4 * sizes[0][1]), gt_bboxes, gt_labels,
                                                                gt_masks, img.device)
            outs = self.bbox_head.forward_rows(*self.extract_feat_rows(img))
        else:
            outs = self.bbox_head(self.extract_feat_train(img))
        return self.bbox_head.loss(*outs, gt_bboxes, gt_labels, img_metas, self.train_cfg,
                                   gt_bboxes_ignore=gt_bboxes_ignore, gt_masks_list=gt_masks, **kw)

    def _head_sizes(self, hw):
        """(h, w) of the five head levels for an input of size hw: the conv arithmetic of the caffe-style ResNet (7x7 s2 p3,
        3x3 max-pool s2 p1, one stride-2 1x1 per stage) and of the FPN's two extra 3x3 s2 p1 convs (fpn.py:120-135)"""
        out = []
        for n in hw:
            n = (n + 6 - 7) // 2 + 1
            n = (n + 2 - 3) // 2 + 1
            lv = []
            for _ in range(3):
                n = (n - 1) // 2 + 1
                lv.append(n)
            for _ in range(2):
                n = (n + 2 - 3) // 2 + 1
                lv.append(n)
            out.append(lv)
        return list(zip(*out))

    @auto_fp16(apply_to=('img', ))             # M/mmdet/models/detectors/base.py:136
    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)
