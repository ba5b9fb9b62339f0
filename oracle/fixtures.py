"""Platform-independent test inputs.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference-run fixtures (tests/golden/ref_vectors.npz, made by tests/golden/make_reference_vectors.py) hold only
the reference's OUTPUTS; the inputs are re-created bit-for-bit wherever the tests run.  numpy's legacy RandomState
integer stream is frozen across versions and platforms, and the values below are small integers times a power of
two, so every input is exactly representable in float32 and identical everywhere.
"""
import numpy as np
import torch

PYRAMID = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]       # level sizes of a 96x128 image, strides 8..128
IMG_HW = (96, 128)
STRIDES = (8, 16, 32, 64, 128)


def exact(seed, shape, lo=-2 ** 12, hi=2 ** 12, scale=2.0 ** -10):
    """integers in [lo, hi) times `scale` as float32 (exact)"""
    rs = np.random.RandomState(seed)
    return (rs.randint(lo, hi, size=shape).astype(np.float32) * np.float32(scale)).astype(np.float32)


def exact_unique(seed, shape, bits=12):
    """distinct multiples of 2**-bits in [0, 1) (no ties: sort orders are then implementation independent)"""
    n = int(np.prod(shape))
    assert n <= 2 ** bits
    rs = np.random.RandomState(seed)
    return (rs.permutation(2 ** bits)[:n].astype(np.float32) * np.float32(2.0 ** -bits)).reshape(shape)


def texact(seed, shape, *a, **kw):
    return torch.from_numpy(exact(seed, shape, *a, **kw))


def head_state_dict(template, seed=100):
    """every tensor of an oracle head state dict re-filled with exact values of an init-like magnitude
    (conv weights ~ +-0.03, GN weight ~ 1, biases small; fcos_cls.bias at the prior of 0.01)"""
    out = {}
    for i, (k, v) in enumerate(sorted(template.items())):
        shp = tuple(v.shape)
        if k.endswith("gn.weight") or k.endswith("norm.weight"):
            t = 1.0 + exact(seed + i, shp, -64, 64, 2.0 ** -10)
        elif k.endswith("scale"):
            t = 1.0 + exact(seed + i, shp, 0, 32, 2.0 ** -7)
        elif k.endswith("bias"):
            t = exact(seed + i, shp, -64, 64, 2.0 ** -10)
            if k.endswith("fcos_cls.bias"):
                t = t - np.float32(4.59375)
        else:
            t = exact(seed + i, shp, -512, 512, 2.0 ** -14)
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32).reshape(shp))
    return out


def trunk_state_dict(template, seed=700):
    """backbone / neck tensors re-filled with exact values: conv weights are integers in [-512, 512) times the power of
    two that brings their spread closest to sqrt(2 / fan_in); BN statistics positive; DCN offset convs small"""
    out = {}
    for i, (k, v) in enumerate(sorted(template.items())):
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shp, dtype=v.dtype)
            continue
        if k.endswith("running_var"):
            t = 0.5 + exact(seed + i, shp, 0, 256, 2.0 ** -8)
        elif k.endswith("running_mean"):
            t = exact(seed + i, shp, -32, 32, 2.0 ** -8)
        elif v.dim() == 1 and k.endswith("weight"):
            t = 1.0 + exact(seed + i, shp, -16, 16, 2.0 ** -8)
        elif v.dim() == 1:
            t = exact(seed + i, shp, -16, 16, 2.0 ** -8)
        elif "conv_offset" in k:
            t = exact(seed + i, shp, -64, 64, 2.0 ** -14)
        else:
            fan_in = int(np.prod(shp[1:]))
            e = int(np.round(np.log2(np.sqrt(2.0 / fan_in) / 295.0)))
            t = exact(seed + i, shp, -512, 512, 2.0 ** e)
        out[k] = torch.from_numpy(np.asarray(t, dtype=np.float32).reshape(shp))
    return out


def pyramid_feats(seed, batch, channels=256, sizes=PYRAMID):
    return [texact(seed + l, (batch, channels, h, w)) for l, (h, w) in enumerate(sizes)]


def head_outputs(seed, batch, num_fg, sizes=PYRAMID, strides=STRIDES):
    """synthetic SipMaskHead outputs in the reference's layout: cls logits with a sprinkle of confident positions,
    positive l/t/r/b distances of a few strides, centerness logits, 128 coefficients, 32 basis maps at stride 4"""
    cls, box, ctr, cof = [], [], [], []
    for l, (h, w) in enumerate(sizes):
        c = exact(seed + 10 * l, (batch, num_fg, h, w), -2 ** 12, 2 ** 11, 2.0 ** -9) - np.float32(2.0)    # [-10, 2)
        cls.append(torch.from_numpy(c))
        box.append(torch.from_numpy(exact(seed + 10 * l + 1, (batch, 4, h, w), 64, 1024, 2.0 ** -8) * np.float32(strides[l])))
        ctr.append(texact(seed + 10 * l + 2, (batch, 1, h, w), -2 ** 11, 2 ** 12, 2.0 ** -10))
        cof.append(texact(seed + 10 * l + 3, (batch, 128, h, w), -2 ** 10, 2 ** 10, 2.0 ** -10))
    fm = texact(seed + 77, (batch, 32, sizes[0][0] * 4, sizes[0][1] * 4), 0, 2 ** 11, 2.0 ** -10)          # post-ReLU: >= 0
    return cls, box, ctr, cof, fm


def ground_truth(seed, batch, num_fg, img_hw=IMG_HW, max_gt=5):
    """per image: gt boxes [G,4] (x1,y1,x2,y2 inside the image), labels [G] in 1..num_fg, uint8 masks [G,H,W]
    (an ellipse inside each box)"""
    rs = np.random.RandomState(seed)
    H, W = img_hw
    boxes, labels, masks = [], [], []
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(batch):
        g = int(rs.randint(1, max_gt + 1))
        x1 = rs.randint(0, W - 24, size=g)
        y1 = rs.randint(0, H - 24, size=g)
        bw = rs.randint(16, 72, size=g)
        bh = rs.randint(16, 72, size=g)
        x2 = np.minimum(x1 + bw, W - 1)
        y2 = np.minimum(y1 + bh, H - 1)
        bx = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
        m = np.zeros((g, H, W), np.uint8)
        for i in range(g):
            cx, cy = (x1[i] + x2[i]) / 2.0, (y1[i] + y2[i]) / 2.0
            rx, ry = max((x2[i] - x1[i]) / 2.0, 1.0), max((y2[i] - y1[i]) / 2.0, 1.0)
            m[i] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).astype(np.uint8)
        boxes.append(torch.from_numpy(bx))
        labels.append(torch.from_numpy(rs.randint(1, num_fg + 1, size=g).astype(np.int64)))
        masks.append(m)
    return boxes, labels, masks
