"""The reference's CUDA-only ops, compiled as host C++ from where they lie (oracle/build_ref.py + oracle/ref_shim:
their device functions and host algebra run on the CPU), against the oracle's restatements -- the EXECUTION pin that
round 1 lacked for deform conv forward/backward, CropSplit forward/backward, CropSplitGt and the sigmoid focal
loss.  The calls below restate the argument order of the reference's Python wrappers
(M/mmdet/ops/dcn/deform_conv.py:16-96, M/mmdet/ops/crop/crop_split.py:9-39, crop_split_gt.py:9-27,
M/mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py:11-32).  CPU only; skipped when oracle/_ref was not built
(the build needs /root/reference: `python oracle/build_ref.py`; the GPU box gets the prebuilt modules)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ops as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import build_ref  # noqa: E402


def _mod(name):
    m = build_ref.load_host_shim(name)
    if m is None:
        pytest.skip("oracle/_ref/%s.so not built" % name)
    return m


def _ref_deform_forward(m, x, off, w, stride, pad, dil, G, step=64):
    B = x.shape[0]
    kh, kw = w.shape[2:]
    Ho = (x.shape[2] + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (x.shape[3] + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    out = x.new_empty(B, w.shape[0], Ho, Wo)
    bufs = [x.new_empty(0), x.new_empty(0)]
    m.deform_conv_forward_cuda(x, w, off, out, bufs[0], bufs[1], kw, kh, stride, stride, pad, pad, dil, dil, 1, G,
                               min(step, B))
    return out, bufs


@pytest.mark.parametrize("cfg", [
    # B, C, H, W, Cout, k, stride, pad, dil, G
    (2, 16, 9, 11, 8, 3, 1, 1, 1, 4),      # FeatureAlign geometry (deformable_groups=4)
    (1, 8, 10, 7, 6, 3, 1, 1, 1, 1),       # DeformConvPack geometry (SipMask++ backbone)
    (2, 8, 12, 9, 4, 3, 2, 1, 1, 2),       # strided
    (1, 4, 9, 9, 4, 3, 1, 2, 2, 1),        # dilated
])
def test_deform_conv_forward_backward_vs_reference_execution(cfg):
    m = _mod("ref_deform_conv")
    B, C, H, W, Co, k, s, p, d, G = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    off = torch.randn(B, G * 2 * k * k, Ho, Wo, generator=g) * 2.0
    off[0, :, 0, 0] = 0.0                     # integer sample positions
    off[0, 0::2, -1, -1] = -float(H)          # far outside -> zeros
    ref, bufs = _ref_deform_forward(m, x, off, w, s, p, d, G)
    got = O.deform_conv(x, off, w, s, p, d, G)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
    # backward: deform_conv_backward_input_cuda / _parameters_cuda as DeformConvFunction.backward calls them
    go = torch.randn(ref.shape, generator=g)
    gi, goff, gw = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
    m.deform_conv_backward_input_cuda(x, off, go, gi, goff, w, bufs[0], k, k, s, s, p, p, d, d, 1, G, min(64, B))
    # im2col_step 1 here: with a larger step the host code views `at::zeros_like(gradOutput.transpose_(1, 2))`, which
    # PyTorch 1.1 returned contiguous and 2.x returns with the transposed strides (deform_conv_cuda.cpp:423-432); the
    # step only groups images per GEMM, the sum over images is the same
    m.deform_conv_backward_parameters_cuda(x, off, go, gw, bufs[0], bufs[1], k, k, s, s, p, p, d, d, 1, G, 1, 1)
    ogi, ogoff, ogw = O.deform_conv_backward(x, off, w, go, s, p, d, G)
    torch.testing.assert_close(torch.as_tensor(ogi), gi, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(torch.as_tensor(ogoff), goff, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(torch.as_tensor(ogw), gw, rtol=1e-4, atol=1e-5)


def _rois(rng, n, H, W):
    xy = rng.rand(n, 2).astype(np.float32) * np.array([W * 0.8, H * 0.8], np.float32) - 3
    r = np.concatenate([xy, xy + rng.rand(n, 2).astype(np.float32) * np.array([W * 0.7, H * 0.7], np.float32) + 0.05], 1)
    r[0] = [2.0, 3.0, 2.0 + 7.0, 3.0 + 5.0]          # integer corners: the >= / < edges of crop_split_cuda_kernel.cu:45
    r[1] = [-5.5, -2.25, W + 4.0, H + 1.5]           # larger than the map
    return r.astype(np.float32)


@pytest.mark.parametrize("c", [1, 2, 3])
def test_crop_split_forward_backward_vs_reference_execution(c):
    m = _mod("ref_crop_split")
    rng = np.random.RandomState(10 + c)
    H, W, N = 23, 31, 9
    data = rng.rand(c * c, H, W, N).astype(np.float32)
    rois = _rois(rng, N, H, W)
    out = torch.zeros(H, W, N)
    m.crop_split_cuda_forward(torch.from_numpy(data), torch.from_numpy(rois), out, H, W, c, N)
    np.testing.assert_array_equal(O.crop_split(data, rois, c), out.numpy())
    go = rng.rand(H, W, N).astype(np.float32)
    gin = torch.zeros(c * c, H, W, N)
    m.crop_split_cuda_backward(torch.from_numpy(go), torch.from_numpy(rois), gin, H, W, c, N)
    np.testing.assert_array_equal(O.crop_split_backward(go, rois, c), gin.numpy())


def test_crop_split_gt_vs_reference_execution():
    m = _mod("ref_crop_split_gt")
    rng = np.random.RandomState(3)
    H, W, N = 19, 27, 7
    data = rng.rand(H, W, N).astype(np.float32)
    rois = _rois(rng, N, H, W)
    out = torch.zeros(H, W, N)
    m.crop_split_gt_cuda_forward(torch.from_numpy(data), torch.from_numpy(rois), out, H, W, 2, N)
    np.testing.assert_array_equal(O.crop_split_gt(data, rois), out.numpy())


def test_sigmoid_focal_loss_vs_reference_execution():
    m = _mod("ref_focal_loss")
    g = torch.Generator().manual_seed(4)
    n, C = 257, 80
    logits = torch.randn(n, C, generator=g) * 4
    logits[0, :4] = torch.tensor([0.0, 30.0, -30.0, 88.0])
    targets = torch.randint(0, C + 1, (n,), generator=g)
    targets[:5] = torch.tensor([0, 1, C, 2, 0])              # 0 = background, 1..C = classes (sigmoid_focal_loss_cuda.cu:31-35)
    ref = m.forward(logits, targets, C, 2.0, 0.25)
    got = O.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25)
    torch.testing.assert_close(torch.as_tensor(got), ref, rtol=2e-6, atol=1e-7)
    d = torch.rand(n, C, generator=g)
    refb = m.backward(logits, targets, d, C, 2.0, 0.25)
    gotb = O.sigmoid_focal_loss_backward(logits, targets, d, 2.0, 0.25)
    torch.testing.assert_close(torch.as_tensor(gotb), refb, rtol=2e-5, atol=1e-6)
