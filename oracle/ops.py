"""CPU oracle: op-level restatement of the SipMask hot path (TEST INFRASTRUCTURE ONLY).

This file is a from-scratch CPU restatement (numpy / torch-CPU fp32) of the
arithmetic the reference performs in its CUDA ops and ATen chains.  It is the
parity checker: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``sipmask_amd``) never imports ``oracle``.

Parity pinning:
  * NMS is pinned against the reference's own golden keep-index vectors
    (``tests/golden/nms_kat.json`` generated from
    ``/root/reference/SipMask-benchmark/tests/test_nms.py:16-58,60-221``) and the
    doctest vectors of ``SipMask-mmdetection/mmdet/ops/nms/nms_wrapper.py:25-34``
    and ``mmdet/core/bbox/geometry.py:22-44``.
  * The reference has no test for anything else, but most of it is plain PyTorch that runs here: outputs of the
    reference's OWN code (SipMaskHead.forward / get_bboxes / loss / fcos_target / fast_nms of M/, the VIS head with
    its tracker, B/'s head and SipMaskPostProcessor, mmdet's distance2bbox / bbox_overlaps / multiclass_nms_idx / loss
    modules) are committed as ``tests/golden/ref_vectors.npz`` (made by ``tests/golden/make_reference_vectors.py``;
    ``tests/golden/ref_loader.py`` lists the stand-ins for the compiled / absent pieces) and
    ``tests/test_reference_vectors.py`` holds this package to them: distance2bbox, bbox_overlaps, multiclass_nms_idx,
    fast_nms, mask_assemble, the python-formula focal loss and the whole graphs in model.py / loss.py / vis.py /
    fcos_core.py are pinned that way.
  * Still **parity unpinned** (compiled CUDA extension or absent third-party code on the reference side, nothing to
    run or compare with): deform_conv / deform_conv_backward, the CUDA crop_split / crop_split_gt kernels (their
    restatement agrees with the reference's python ``crop_split`` except on the pixel row/column next to a box centre,
    see the test), the CUDA focal-loss kernel (== the python formula), COCO RLE, the input pipeline.  Those are
    validated by internal cross-checks in tests/test_oracle_*.py (deform(offset=0) == conv2d, integer-offset ==
    shifted conv, crop_split(c=1) == crop_split_gt, CUDA focal formula == python focal formula, fp64 gradcheck).

Path prefixes used in citations: M/ = SipMask-mmdetection/, B/ = SipMask-benchmark/,
V/ = SipMask-VIS/ under /root/reference.
"""

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# ordering helpers
# ----------------------------------------------------------------------------


def sort_desc_stable(values):
    """Total order used everywhere the reference calls sort/topk on a GPU
    (tie order unspecified there): value descending, index
    ascending.  Returns int64 indices."""
    v = np.asarray(values)
    # stable argsort on negated keys gives (value desc, index asc); -0.0/0.0 tie
    return np.argsort(-v, kind="stable").astype(np.int64)


def topk_desc(values, k):
    """torch.topk(sorted=True) restated with the oracle tie order."""
    order = sort_desc_stable(values)
    return order[:k]


def sigmoid_ref(x):
    """The ranking sigmoid of get_bboxes_single (M/mmdet/models/anchor_heads/sipmask_head.py:566-567,
    `cls_score...sigmoid()`, `centerness...sigmoid()`), evaluated in float64 and rounded ONCE to float32.
    The reference calls torch's f32 sigmoid, whose expf is not bit-defined across CPU / CUDA / ROCm libraries
    (1 ulp apart), and every integer decision downstream (top-k order, score > score_thr, NMS order) hangs off
    these values.  A float64 evaluation rounded to float32 is reproducible between any two < 1-ulp double exp()
    implementations (except within ~2^-52 of an f32 rounding boundary), so the HIP kernels (common.h:
    sigmoid_rank) and this oracle produce the same bits; it differs from torch.sigmoid by at most 1 f32 ulp
    (tests/test_reference_vectors.py holds both against the reference's own outputs)."""
    t = torch.as_tensor(x)
    v = t.detach().to(torch.float64)
    return (1.0 / (1.0 + torch.exp(-v))).to(torch.float32)


# ----------------------------------------------------------------------------
# boxes
# ----------------------------------------------------------------------------


def distance2bbox(points, distance, max_shape=None):
    """M/mmdet/core/bbox/transforms.py:202-224."""
    x1 = points[:, 0] - distance[:, 0]
    y1 = points[:, 1] - distance[:, 1]
    x2 = points[:, 0] + distance[:, 2]
    y2 = points[:, 1] + distance[:, 3]
    if max_shape is not None:
        x1 = x1.clamp(min=0, max=max_shape[1] - 1)
        y1 = y1.clamp(min=0, max=max_shape[0] - 1)
        x2 = x2.clamp(min=0, max=max_shape[1] - 1)
        y2 = y2.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], -1)


def bbox_overlaps(b1, b2, is_aligned=False):
    """IoU with the +1 width convention, M/mmdet/core/bbox/geometry.py:47-88."""
    rows, cols = b1.size(0), b2.size(0)
    if rows * cols == 0:
        return b1.new_zeros(rows, 1) if is_aligned else b1.new_zeros(rows, cols)
    if is_aligned:
        lt = torch.max(b1[:, :2], b2[:, :2])
        rb = torch.min(b1[:, 2:], b2[:, 2:])
        wh = (rb - lt + 1).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
        a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
        return overlap / (a1 + a2 - overlap)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[:, :, 0] * wh[:, :, 1]
    a1 = (b1[:, 2] - b1[:, 0] + 1) * (b1[:, 3] - b1[:, 1] + 1)
    a2 = (b2[:, 2] - b2[:, 0] + 1) * (b2[:, 3] - b2[:, 1] + 1)
    return overlap / (a1[:, None] + a2 - overlap)


# ----------------------------------------------------------------------------
# NMS
# ----------------------------------------------------------------------------


def _iou_plus1_f32(a, bs):
    """devIoU, M/mmdet/ops/nms/src/nms_kernel.cu:14-22, evaluated in float32 with
    the same operation order (one box ``a`` against an array ``bs``)."""
    f = np.float32
    left = np.maximum(a[0], bs[:, 0])
    right = np.minimum(a[2], bs[:, 2])
    top = np.maximum(a[1], bs[:, 1])
    bottom = np.minimum(a[3], bs[:, 3])
    width = np.maximum((right - left + f(1)).astype(f), f(0))
    height = np.maximum((bottom - top + f(1)).astype(f), f(0))
    inter = (width * height).astype(f)
    sa = f(f(a[2] - a[0] + f(1)) * f(a[3] - a[1] + f(1)))
    sb = ((bs[:, 2] - bs[:, 0] + f(1)).astype(f) * (bs[:, 3] - bs[:, 1] + f(1)).astype(f)).astype(f)
    return (inter / ((sa + sb).astype(f) - inter).astype(f)).astype(f)


def nms(dets, iou_thr, mode="gpu"):
    """Greedy NMS on ``dets[n,5]`` (x1,y1,x2,y2,score), float32.

    mode="gpu": the reference GPU behaviour, suppress on IoU >  thr
                (M/mmdet/ops/nms/src/nms_kernel.cu:61, host scan :113-138).
    mode="cpu": M/mmdet/ops/nms/src/nms_cpu.cpp:5-60, suppress on IoU >= thr.
    Returns kept ORIGINAL indices sorted ascending (nms_kernel.cu:135-138 /
    nonzero(suppressed==0) in nms_cpu.cpp), int64.
    Sort order on ties: score desc, index asc (oracle-defined).
    """
    dets = np.ascontiguousarray(np.asarray(dets, dtype=np.float32))
    n = dets.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    order = sort_desc_stable(dets[:, 4])
    boxes = dets[order, :4]
    suppressed = np.zeros(n, dtype=bool)
    thr = np.float32(iou_thr)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            iou = _iou_plus1_f32(boxes[i], boxes[i + 1:])
            if mode == "gpu":
                suppressed[i + 1:] |= iou > thr
            else:
                suppressed[i + 1:] |= iou >= thr
    kept = np.sort(order[np.asarray(keep, dtype=np.int64)])
    return kept.astype(np.int64)


def multiclass_nms_idx(multi_bboxes, multi_scores, score_thr, iou_thr, max_num=-1,
                       score_factors=None, mode="gpu"):
    """M/mmdet/core/post_processing/bbox_nms.py:79-146 restated on numpy float32.

    multi_bboxes [K,4], multi_scores [K,C+1] (column 0 = background),
    score_factors [K] (centerness).  The class test is on the raw class score
    (bbox_nms.py:111) and the centerness multiply happens afterwards (:121-122).
    Returns (det_bboxes [N,5] f32, det_labels [N] i64, idxs_keep [N] i64).
    Within a class the kept rows come back in ascending candidate order
    (nms_kernel.cu:135-138); classes are concatenated ascending; if more than
    max_num, top max_num by score (desc, position asc) (bbox_nms.py:135-140).
    """
    mb = np.asarray(multi_bboxes, dtype=np.float32)
    ms = np.asarray(multi_scores, dtype=np.float32)
    sf = None if score_factors is None else np.asarray(score_factors, dtype=np.float32)
    num_classes = ms.shape[1]
    bboxes, labels, idxs = [], [], []
    all_idx = np.arange(ms.shape[0], dtype=np.int64)
    for c in range(1, num_classes):
        sel = ms[:, c] > np.float32(score_thr)
        if not sel.any():
            continue
        _b = mb[sel]
        _s = ms[sel, c]
        if sf is not None:
            _s = (_s * sf[sel]).astype(np.float32)
        dets = np.concatenate([_b, _s[:, None]], axis=1)
        ki = nms(dets, iou_thr, mode=mode)
        bboxes.append(dets[ki])
        labels.append(np.full((ki.shape[0],), c - 1, dtype=np.int64))
        idxs.append(all_idx[sel][ki])
    if bboxes:
        bboxes = np.concatenate(bboxes)
        labels = np.concatenate(labels)
        idxs = np.concatenate(idxs)
        if max_num >= 0 and bboxes.shape[0] > max_num:
            inds = sort_desc_stable(bboxes[:, 4])[:max_num]
            bboxes, labels, idxs = bboxes[inds], labels[inds], idxs[inds]
    else:
        bboxes = np.zeros((0, 5), np.float32)
        labels = np.zeros((0,), np.int64)
        idxs = np.zeros((0,), np.int64)
    return bboxes, labels, idxs


def fast_nms(boxes, scores, cofs, iou_threshold=0.5, top_k=200, score_thr=0.1, max_out=100):
    """M/mmdet/models/anchor_heads/sipmask_head.py:868-910 (+ jaccard :912-938,
    intersect :941-960): IoU WITHOUT +1.  boxes [K,4], scores [C,K], cofs [K,D]."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    cofs = np.asarray(cofs, np.float32)
    C, K = scores.shape
    k = min(top_k, K)
    idx = np.stack([sort_desc_stable(scores[c])[:k] for c in range(C)])      # [C,k]
    sc = np.take_along_axis(scores, idx, axis=1)                               # [C,k]
    bx = boxes[idx.reshape(-1)].reshape(C, k, 4)
    mk = cofs[idx.reshape(-1)].reshape(C, k, -1)
    max_xy = np.minimum(bx[:, :, None, 2:], bx[:, None, :, 2:])
    min_xy = np.maximum(bx[:, :, None, :2], bx[:, None, :, :2])
    wh = np.clip(max_xy - min_xy, 0, None).astype(np.float32)
    inter = (wh[..., 0] * wh[..., 1]).astype(np.float32)
    area = ((bx[:, :, 2] - bx[:, :, 0]) * (bx[:, :, 3] - bx[:, :, 1])).astype(np.float32)
    union = (area[:, :, None] + area[:, None, :]).astype(np.float32) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = (inter / union).astype(np.float32)
    iou = np.triu(iou, k=1)
    # torch.max propagates NaN (0/0 for degenerate boxes); np.max does too
    iou_max = iou.max(axis=1)
    keep = (iou_max <= np.float32(iou_threshold)) & (sc > np.float32(score_thr))
    classes = np.broadcast_to(np.arange(C)[:, None], keep.shape)[keep]
    b = bx[keep]
    m = mk[keep]
    s = sc[keep]
    order = sort_desc_stable(s)[:max_out]
    return (np.concatenate([b[order], s[order][:, None]], axis=1).astype(np.float32),
            classes[order].astype(np.int64), m[order])


# ----------------------------------------------------------------------------
# deformable convolution v1
# ----------------------------------------------------------------------------


def _pair2(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def deform_conv_grouped(x, offset, weight, stride=1, padding=1, dilation=1, groups=1, deformable_groups=1):
    """The op with conv groups (deform_conv_cuda.cpp:176-236): the columns are sampled for ALL channels with the
    deformable-group offsets (channel c -> deformable group c // (C / deformable_groups), kernel.cu:206), then conv
    group g contracts its C/groups channels with its Co/groups filters.  Restated as: full-channel columns via the
    groups=1 path with an identity-like trick is wasteful, so sample per conv group with per-channel offsets instead."""
    B, C, H, W = x.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci * groups == C and Co % groups == 0
    cg, cog, cpd = C // groups, Co // groups, C // deformable_groups
    kk2 = 2 * kh * kw
    outs = []
    for g in range(groups):
        # per-channel deformable group of the slice -> run the groups=1 restatement once per distinct deformable group
        # present in the slice and with only that group's channels, then add (the contraction is linear in the channels)
        acc = None
        for d in sorted(set(c // cpd for c in range(g * cg, (g + 1) * cg))):
            ch = [c for c in range(g * cg, (g + 1) * cg) if c // cpd == d]
            part = deform_conv(x[:, ch], offset[:, d * kk2:(d + 1) * kk2], weight[g * cog:(g + 1) * cog][:, [c - g * cg for c in ch]],
                               stride, padding, dilation, 1)
            acc = part if acc is None else acc + part
        outs.append(acc)
    return torch.cat(outs, 1)


def deform_conv(x, offset, weight, stride=1, padding=1, dilation=1, deformable_groups=1, col_round=None):
    """Deformable conv v1 forward, groups=1.

    col_round (test aid, not in the reference): applied to the sampled columns before the contraction -- the bf16 plan
    blends the four corners in f32 and rounds the SAMPLE once to bf16 (the MFMA operand type); with
    col_round=lambda t: t.bfloat16().float() this restatement follows that arithmetic, so the kernel can be held to
    it tightly (accumulation order + rare rounding flips) instead of to a bound that has to absorb operand rounding.

    Sampling: M/mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:191-243 (offset
    channel layout [g, 2*(i*kw+j)+{0:h,1:w}] :216,222-223; sample taken iff
    -1 < h_im < H and -1 < w_im < W :229); bilinear with per-corner zero padding
    :85-115; contraction with weight.flatten(1): deform_conv_cuda.cpp:231-236.
    x [B,C,H,W], offset [B, G*2*kh*kw, Ho, Wo], weight [Co,C,kh,kw].
    Works for float32 or float64.
    """
    B, C, H, W = x.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci == C
    G = deformable_groups
    cpg = C // G
    (sh, sw), (ph, pw), (dh_, dw_) = _pair2(stride), _pair2(padding), _pair2(dilation)   # deform_conv.py:32-34 (_pair)
    Ho = (H + 2 * ph - (dh_ * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw_ * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (B, G * 2 * kh * kw, Ho, Wo), (offset.shape, (B, G * 2 * kh * kw, Ho, Wo))
    dt = x.dtype
    ho = torch.arange(Ho, dtype=dt).view(1, Ho, 1) * sh - ph
    wo = torch.arange(Wo, dtype=dt).view(1, 1, Wo) * sw - pw
    xf = x.reshape(B, G, cpg, H * W)
    cols = x.new_zeros(B, G, cpg, kh * kw, Ho, Wo)
    off = offset.view(B, G, kh * kw, 2, Ho, Wo)
    for g in range(G):
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                h_im = ho + i * dh_ + off[:, g, t, 0]      # [B,Ho,Wo]
                w_im = wo + j * dw_ + off[:, g, t, 1]
                valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
                h_low = torch.floor(h_im)
                w_low = torch.floor(w_im)
                lh = h_im - h_low
                lw = w_im - w_low
                hh, hw = 1 - lh, 1 - lw
                h_low = h_low.long()
                w_low = w_low.long()
                h_high = h_low + 1
                w_high = w_low + 1

                def corner(hi, wi, ok):
                    ok = ok & valid
                    lin = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo)
                    v = torch.gather(xf[:, g], 2, lin.expand(B, cpg, Ho * Wo)).view(B, cpg, Ho, Wo)
                    return v * ok.view(B, 1, Ho, Wo).to(dt)

                v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
                v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
                v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
                v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
                w1 = (hh * hw).unsqueeze(1)
                w2 = (hh * lw).unsqueeze(1)
                w3 = (lh * hw).unsqueeze(1)
                w4 = (lh * lw).unsqueeze(1)
                cols[:, g, :, t] = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    if col_round is not None:
        cols = col_round(cols)
    cols = cols.view(B, C * kh * kw, Ho * Wo)
    out = torch.matmul(weight.reshape(Co, C * kh * kw), cols)
    return out.view(B, Co, Ho, Wo)


def deform_conv_backward(x, offset, weight, grad_out, stride=1, padding=1, dilation=1, deformable_groups=1):
    """DeformConvFunction.backward (M/mmdet/ops/dcn/deform_conv.py:60-96) -> (grad_input, grad_offset, grad_weight).

    * columns = weight^T @ grad_output (deform_conv_cuda.cpp:262-374);
    * grad_input: every column element is scattered to the in-bounds bilinear corners of its sampling point with
      get_gradient_weight (deform_conv_cuda_kernel.cu:117-142, scatter loop :279-343);
    * grad_offset[b, g, 2t+{0,1}] = sum_{c in group g} col * get_coordinate_weight (:144-188, :375-433; a sample
      outside (-1,H)x(-1,W) contributes 0, :423-426);
    * grad_weight = grad_output @ im2col(x, offset)^T, scale 1 (deform_conv_cuda.cpp:376-490, deform_conv.py:92).
    Works for float32 or float64."""
    B, C, H, W = x.shape
    Co, Ci, kh, kw = weight.shape
    G = deformable_groups
    cpg = C // G
    (sh, sw), (ph, pw), (dh_, dw_) = _pair2(stride), _pair2(padding), _pair2(dilation)
    Ho = (H + 2 * ph - (dh_ * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw_ * (kw - 1) + 1)) // sw + 1
    dt = x.dtype
    gcol = torch.einsum("ocij,bohw->bcijhw", weight, grad_out).reshape(B, G, cpg, kh * kw, Ho, Wo)
    ho = torch.arange(Ho, dtype=dt).view(1, Ho, 1) * sh - ph
    wo = torch.arange(Wo, dtype=dt).view(1, 1, Wo) * sw - pw
    xf = x.reshape(B, G, cpg, H * W)
    off = offset.view(B, G, kh * kw, 2, Ho, Wo)
    gx = x.new_zeros(B, G, cpg, H * W)
    goff = offset.new_zeros(B, G, kh * kw, 2, Ho, Wo)
    cols = x.new_zeros(B, G, cpg, kh * kw, Ho, Wo)
    for g in range(G):
        for t in range(kh * kw):
            i, j = t // kw, t % kw
            h_im = ho + i * dh_ + off[:, g, t, 0]
            w_im = wo + j * dw_ + off[:, g, t, 1]
            valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            hl, wl = torch.floor(h_im), torch.floor(w_im)
            lh, lw = h_im - hl, w_im - wl
            hl, wl = hl.long(), wl.long()
            hh_, wh_ = hl + 1, wl + 1
            top = gcol[:, g, :, t]                                            # [B,cpg,Ho,Wo]
            corners = ((hl, wl, (hl >= 0) & (wl >= 0), (1 - lh) * (1 - lw), -(1 - lw), -(1 - lh)),
                       (hl, wh_, (hl >= 0) & (wh_ <= W - 1), (1 - lh) * lw, -lw, (1 - lh)),
                       (hh_, wl, (hh_ <= H - 1) & (wl >= 0), lh * (1 - lw), (1 - lw), -lh),
                       (hh_, wh_, (hh_ <= H - 1) & (wh_ <= W - 1), lh * lw, lw, lh))
            dh = x.new_zeros(B, cpg, Ho, Wo)
            dw = x.new_zeros(B, cpg, Ho, Wo)
            for hi, wi, ok, wgt, ch, cw in corners:
                ok = (ok & valid).view(B, 1, Ho, Wo).to(dt)
                lin = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, cpg, Ho * Wo)
                v = torch.gather(xf[:, g], 2, lin).view(B, cpg, Ho, Wo) * ok
                gx[:, g].scatter_add_(2, lin, (top * wgt.unsqueeze(1) * ok).reshape(B, cpg, Ho * Wo))
                cols[:, g, :, t] += wgt.unsqueeze(1) * v
                dh += ch.unsqueeze(1) * v
                dw += cw.unsqueeze(1) * v
            goff[:, g, t, 0] = (top * dh).sum(1)
            goff[:, g, t, 1] = (top * dw).sum(1)
    gw = torch.einsum("bohw,bcthw->oct", grad_out, cols.reshape(B, C, kh * kw, Ho, Wo)).reshape(Co, C, kh, kw)
    return gx.view(B, C, H, W), goff.view(B, G * 2 * kh * kw, Ho, Wo), gw


# ----------------------------------------------------------------------------
# crop_split / crop_split_gt  (CUDA-kernel semantics, NOT the python fallback)
# ----------------------------------------------------------------------------


def _crop_cells(H, W, rois, c):
    """Shared index math of M/mmdet/ops/crop/src/crop_split_cuda_kernel.cu:34-52.

    Returns (inside [H,W,N] bool, cell [H,W,N] int) with the kernel's exact
    float32/float64 mix: roi_w = float32((double)(x2-x1) + 0.1) / c), and
    idx_w = (int)((float)(pw - x1) / roi_w)."""
    rois = np.asarray(rois, dtype=np.float32)
    x1, y1, x2, y2 = rois[:, 0], rois[:, 1], rois[:, 2], rois[:, 3]
    pw = np.arange(W, dtype=np.float32).reshape(1, W, 1)
    ph = np.arange(H, dtype=np.float32).reshape(H, 1, 1)
    inside = (pw >= x1) & (ph >= y1) & (pw < x2) & (ph < y2)
    roi_w = (((x2 - x1).astype(np.float32).astype(np.float64) + 0.1) / c).astype(np.float32)
    roi_h = (((y2 - y1).astype(np.float32).astype(np.float64) + 0.1) / c).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        idx_w = np.trunc(((pw - x1).astype(np.float32) / roi_w).astype(np.float32))
        idx_h = np.trunc(((ph - y1).astype(np.float32) / roi_h).astype(np.float32))
    idx_w = np.where(inside, idx_w, 0).astype(np.int64)
    idx_h = np.where(inside, idx_h, 0).astype(np.int64)
    return inside, idx_h * c + idx_w


def crop_split(data, rois, c=2):
    """CropSplitKernelForward, M/mmdet/ops/crop/src/crop_split_cuda_kernel.cu:19-59.
    data [c*c,H,W,N] -> out [H,W,N]; zero outside the (unclamped, float) box."""
    data = np.asarray(data)
    cc, H, W, N = data.shape
    assert cc == c * c
    inside, cell = _crop_cells(H, W, rois, c)
    picked = np.take_along_axis(data, cell[None], axis=0)[0]
    return np.where(inside, picked, 0).astype(data.dtype)


def crop_split_backward(grad_out, rois, c=2):
    """CropSplitKernelBack :90-127 (1:1 scatter)."""
    grad_out = np.asarray(grad_out)
    H, W, N = grad_out.shape
    inside, cell = _crop_cells(H, W, rois, c)
    gin = np.zeros((c * c, H, W, N), dtype=grad_out.dtype)
    np.put_along_axis(gin, cell[None], np.where(inside, grad_out, 0)[None], axis=0)
    return gin


def crop_split_gt(data, rois):
    """CropSplitGtKernelForward, M/mmdet/ops/crop/src/crop_split_gt_cuda_kernel.cu:19-49."""
    data = np.asarray(data)
    H, W, N = data.shape
    inside, _ = _crop_cells(H, W, rois, 1)
    return np.where(inside, data, 0).astype(data.dtype)


# ----------------------------------------------------------------------------
# sigmoid focal loss (CUDA formula)
# ----------------------------------------------------------------------------


def sigmoid_focal_loss_forward(logits, targets, gamma=2.0, alpha=0.25):
    """SigmoidFocalLossForward, M/mmdet/ops/sigmoid_focal_loss/src/
    sigmoid_focal_loss_cuda.cu:24-59.  logits [N,C] float, targets [N] int64
    (0 = background, c+1 = class c).  Returns losses [N,C]."""
    x = logits
    N, C = x.shape
    d = torch.arange(C).view(1, C)
    t = targets.view(N, 1)
    c1 = (t == d + 1).to(x.dtype)
    c2 = ((t >= 0) & (t != d + 1)).to(x.dtype)
    p = torch.sigmoid(x)
    flt_min = torch.finfo(torch.float32).tiny
    term1 = (1 - p) ** gamma * torch.log(p.clamp(min=flt_min))
    ge = (x >= 0).to(x.dtype)
    term2 = p ** gamma * (-x * ge - torch.log1p(torch.exp(x - 2 * x * ge)))
    return -c1 * term1 * alpha - c2 * term2 * (1 - alpha)


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma=2.0, alpha=0.25):
    """SigmoidFocalLossBackward :62-97."""
    x = logits
    N, C = x.shape
    d = torch.arange(C).view(1, C)
    t = targets.view(N, 1)
    c1 = (t == d + 1).to(x.dtype)
    c2 = ((t >= 0) & (t != d + 1)).to(x.dtype)
    p = torch.sigmoid(x)
    flt_min = torch.finfo(torch.float32).tiny
    term1 = (1 - p) ** gamma * (1 - p - p * gamma * torch.log(p.clamp(min=flt_min)))
    ge = (x >= 0).to(x.dtype)
    log1mp = -x * ge - torch.log1p(torch.exp(x - 2 * x * ge))
    term2 = p ** gamma * (log1mp * (1 - p) * gamma - p)
    return (-c1 * term1 * alpha - c2 * term2 * (1 - alpha)) * d_losses


def py_sigmoid_focal_loss(pred, target, gamma=2.0, alpha=0.25):
    """py_sigmoid_focal_loss, M/mmdet/models/losses/focal_loss.py:10-25 (no
    weight/reduction) -- the independent formula used to cross-check the CUDA one."""
    N, C = pred.shape
    onehot = F.one_hot(target, C + 1)[:, 1:].to(pred.dtype)
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    fw = (alpha * onehot + (1 - alpha) * (1 - onehot)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(pred, onehot, reduction="none") * fw


# ----------------------------------------------------------------------------
# mask assembly
# ----------------------------------------------------------------------------


def mask_assemble(feat_mask, det_cofs, det_boxes, scale_factor=1.0, rescale=None,
                  mask_thr=0.4, up_scale=2, ssd_flag=False):
    """M/mmdet/models/anchor_heads/sipmask_head.py:609-633.  ssd_flag: scale_factor is the [w,h,w,h] array of a
    keep_ratio=False pipeline and the upsampling is per axis, `scale / scale_factor[3:1:-1]` (:629-630).

    feat_mask [32,Hm,Wm] f32, det_cofs [N,128], det_boxes [N,>=4].
    Returns dict(pos_masks [N,Hm,Wm] f32 after sigmoid+crop,
                 logits [4,Hm,Wm,N] (pre-sigmoid quadrant logits),
                 up [N,Ho,Wo] f32 bilinear-upsampled, masks [N,Ho,Wo] uint8)."""
    feat_mask = torch.as_tensor(feat_mask, dtype=torch.float32)
    det_cofs = torch.as_tensor(det_cofs, dtype=torch.float32)
    det_boxes = torch.as_tensor(det_boxes, dtype=torch.float32)
    img = feat_mask.permute(1, 2, 0)
    logits = torch.stack([img @ det_cofs[:, 32 * q:32 * (q + 1)].t() for q in range(4)], 0)
    probs = torch.sigmoid(logits)
    sf = np.asarray(scale_factor, np.float32).reshape(-1)
    if rescale is None:                       # sipmask_head.py:621-622
        sf = sf * 0 + 1.0
    rois = det_boxes[:, :4] * torch.from_numpy(sf) / up_scale             # scalar or per coordinate (:623)
    pos = torch.from_numpy(crop_split(probs.numpy(), rois.numpy(), 2)).permute(2, 0, 1).contiguous()
    if ssd_flag:
        ups = (float(up_scale / sf[3]), float(up_scale / sf[2]))           # scale / scale_factor[3:1:-1]
    else:
        assert sf.size == 1, "the non-SSD path divides by a scalar scale_factor (:632)"
        # a python float in img_meta (keep_ratio=True): the division happens in double
        ups = up_scale / (1.0 if rescale is None else float(np.asarray(scale_factor, np.float64).reshape(-1)[0]))
    up = F.interpolate(pos.unsqueeze(0), scale_factor=ups, mode="bilinear", align_corners=False).squeeze(0)
    masks = (up > mask_thr).to(torch.uint8)
    return dict(pos_masks=pos, logits=logits, up=up, masks=masks, rois=rois)


# ----------------------------------------------------------------------------
# COCO run-length encoding (result packing, sipmask_head.py:645-657)
# ----------------------------------------------------------------------------
# The reference calls pycocotools.mask.encode (sipmask_head.py:655), a third-party dependency that is NOT under
# /root/reference and is not pinned by it (requirements name no version; the READMEs install cocoapi from git).
# The functions below restate the published algorithm of cocoapi `common/maskApi.c` (rleEncode, rleToString,
# rleFrString, rleDecode).  PARITY UNPINNED: there is no pycocotools in this image to generate golden strings;
# the tests pin the restatement through encode -> string -> parse -> decode round trips only.


def rle_counts(mask):
    """maskApi.c rleEncode: column-major scan, alternating run lengths starting with the zeros run
    (which is 0 when the mask starts with a 1).  mask [H,W] of 0/1 -> list of ints."""
    m = np.asarray(mask, np.uint8)
    v = m.T.reshape(-1)                                  # column-major (Fortran order, sipmask_head.py:656)
    n = v.size
    if n == 0:
        return [0]
    change = np.flatnonzero(v[1:] != v[:-1]) + 1          # positions where the value flips
    lead = [0] if v[0] != 0 else []                       # rleEncode starts with p=0: a mask that starts with 1
    edges = np.concatenate([[0], lead, change, [n]])      # gets a leading zero-length run of zeros
    cnts = np.diff(edges)
    return [int(c) for c in cnts]


def rle_to_string(cnts):
    """maskApi.c rleToString: counts i>2 are stored as the difference to counts[i-2]; each value is emitted as
    5-bit groups, least significant first, bit 0x20 = 'more', sign carried by bit 0x10 of the last group; +48."""
    out = bytearray()
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5                                        # arithmetic shift (Python ints: floor), as `long`
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    """maskApi.c rleFrString (the inverse of rle_to_string)."""
    cnts = []
    p = 0
    s = bytes(s)
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def rle_decode(cnts, h, w):
    """maskApi.c rleDecode -> [H,W] uint8."""
    v = np.zeros(h * w, np.uint8)
    p, val = 0, 0
    for c in cnts:
        v[p:p + c] = val
        p += c
        val ^= 1
    return v.reshape(w, h).T.copy()


def rle_encode(mask):
    """pycocotools.mask.encode for one [H,W] mask -> {'size': [H,W], 'counts': bytes}."""
    m = np.asarray(mask, np.uint8)
    return dict(size=[int(m.shape[0]), int(m.shape[1])], counts=rle_to_string(rle_counts(m)))


def paste_and_encode(mask, canvas_hw):
    """sipmask_head.py:648-656: paste the top-left min(mask, canvas) window onto a zero canvas, encode."""
    m = np.asarray(mask, np.uint8)
    H, W = int(canvas_hw[0]), int(canvas_hw[1])
    im = np.zeros((H, W), np.uint8)
    h, w = min(m.shape[0], H), min(m.shape[1], W)
    im[:h, :w] = m[:h, :w]
    return rle_encode(im)
