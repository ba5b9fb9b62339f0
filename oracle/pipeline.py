"""CPU oracle of the test-time input pipeline.  TEST INFRASTRUCTURE ONLY.

Restates M/mmdet/datasets/pipelines/transforms.py Resize (:24-175, mmcv.imrescale -> cv2.resize INTER_LINEAR),
Normalize (:362-403, mmcv.imnormalize) and Pad (:405-455) in numpy.  mmcv and OpenCV are third-party and absent
here: the resize follows cv2's documented INTER_LINEAR geometry (half-pixel centres, border clamp) in float
arithmetic and rounds to uint8; cv2's 8-bit fast path uses 11-bit fixed-point coefficients and may differ from this by
one grey level on a small fraction of pixels (resize_bilinear_u8_fixedpoint restates that path from OpenCV's source and
tests/test_oracle_ops.py measures the distance: <= 1 grey level).  PARITY UNPINNED.
"""
import numpy as np


def rescale_size(h, w, scale):
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5), f


def resize_bilinear_u8(img, nh, nw):
    img = np.asarray(img, np.uint8)
    h0, w0 = img.shape[:2]
    sy, sx = np.float32(np.float64(h0) / nh), np.float32(np.float64(w0) / nw)
    fy = (np.arange(nh, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5)
    fx = (np.arange(nw, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5)

    def split(f, n):
        i0 = np.floor(f).astype(np.int64)
        l = (f - i0.astype(np.float32)).astype(np.float32)
        l[i0 < 0] = 0
        i0 = np.maximum(i0, 0)
        l[i0 >= n - 1] = 0
        i0 = np.minimum(i0, n - 1)
        return i0, np.minimum(i0 + 1, n - 1), l
    y0, y1, ly = split(fy, h0)
    x0, x1, lx = split(fx, w0)
    im = img.astype(np.float32)
    lx_ = lx[None, :, None]
    top = im[y0][:, x0] + lx_ * (im[y0][:, x1] - im[y0][:, x0])
    bot = im[y1][:, x0] + lx_ * (im[y1][:, x1] - im[y1][:, x0])
    out = top + ly[:, None, None] * (bot - top)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_bilinear_u8_fixedpoint(img, nh, nw):
    """OpenCV's 8-bit INTER_LINEAR as its source states it (modules/imgproc/src/resize.cpp, resizeGeneric_ with
    HResizeLinear<uchar,int,short,2048> and VResizeLinear<uchar,int,short,...>; restated from the published algorithm --
    cv2 itself is absent, so this is NOT a pin): the two interpolation weights of an axis are rounded to 11-bit fixed point
    (cvRound(w * 2048), round-half-even), the horizontal pass keeps 32-bit sums S = p0 * a0 + p1 * a1, and the vertical pass
    is  ((b0 * (S0 >> 4) >> 16) + (b1 * (S1 >> 4) >> 16) + 2) >> 2.  tests/test_oracle_ops.py measures how far the float
    restatement above is from this: never more than ONE grey level."""
    img = np.asarray(img, np.uint8)
    h0, w0 = img.shape[:2]

    def coeffs(n_out, n_in):
        scale = np.float64(n_in) / n_out
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
        f = f.astype(np.float32)                               # cv2 computes fx in float
        i0 = np.floor(f).astype(np.int64)
        l = (f - i0.astype(np.float32)).astype(np.float32)
        l[i0 < 0] = 0
        i0 = np.maximum(i0, 0)
        l[i0 >= n_in - 1] = 0
        i0 = np.minimum(i0, n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        a0 = np.rint((np.float32(1) - l) * np.float32(2048)).astype(np.int64)      # np.rint: half to even = cvRound
        a1 = np.rint(l * np.float32(2048)).astype(np.int64)
        return i0, i1, a0, a1
    y0, y1, b0, b1 = coeffs(nh, h0)
    x0, x1, a0, a1 = coeffs(nw, w0)
    im = img.astype(np.int64)
    rows = im[:, x0] * a0[None, :, None] + im[:, x1] * a1[None, :, None]          # [h0, nw, C] horizontal pass
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def prepare(img, img_scale=(1333, 800), mean=(102.9801, 115.9465, 122.7717), std=(1.0, 1.0, 1.0), to_rgb=False,
            size_divisor=32):
    """-> (float32 [3,Hp,Wp], meta)"""
    h, w = img.shape[:2]
    nh, nw, f = rescale_size(h, w, img_scale)
    r = resize_bilinear_u8(img, nh, nw).astype(np.float32)
    if to_rgb:
        r = r[..., ::-1]
    r = (r - np.asarray(mean, np.float32)) * (np.float32(1) / np.asarray(std, np.float32))
    hp, wp = -(-nh // size_divisor) * size_divisor, -(-nw // size_divisor) * size_divisor
    out = np.zeros((3, hp, wp), np.float32)
    out[:, :nh, :nw] = r.transpose(2, 0, 1)
    return out, dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), pad_shape=(hp, wp, 3), scale_factor=f)
