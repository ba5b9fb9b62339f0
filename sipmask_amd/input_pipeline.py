"""Test-time input pipeline on the GPU: MultiScaleFlipAug(img_scale) / Resize(keep_ratio) / Normalize /
Pad(size_divisor) / ImageToTensor / collate of the reference config
(M/configs/sipmask/sipmask_r50_caffe_fpn_gn_1x.py:72-87, M/mmdet/datasets/pipelines/transforms.py) as ONE HIP launch
per image (sm_preprocess_u8): uint8 HWC images already on the device in, a padded float NCHW batch + img_metas out.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

CAFFE_NORM = dict(mean=[102.9801, 115.9465, 122.7717], std=[1.0, 1.0, 1.0], to_rgb=False)      # cfg :60-61


def rescale_size(h, w, scale):
    """mmcv.imrescale size rule: scale = (long edge, short edge) bound, new size rounded half up."""
    long_e, short_e = max(scale), min(scale)
    f = min(long_e / max(h, w), short_e / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5), f


def prepare_batch(images, img_scale=(1333, 800), keep_ratio=True, size_divisor=32, norm=CAFFE_NORM):
    """images: list of uint8 [h,w,3] tensors on the device (BGR as loaded by the reference).  Returns
    (img float32 [B,3,Hp,Wp], img_metas) with the reference's meta keys; the batch is padded to the largest image
    (collate) rounded up to size_divisor."""
    lib = _lib.load()
    _lib.require_cuda(*images)
    metas, sizes = [], []
    for im in images:
        h, w = int(im.shape[0]), int(im.shape[1])
        if keep_ratio:
            nh, nw, f = rescale_size(h, w, img_scale)
            sf = f
        else:                                   # mmcv.imresize: scale = (w, h)
            nw, nh = int(img_scale[0]), int(img_scale[1])
            sf = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
        sizes.append((h, w, nh, nw))
        metas.append(dict(ori_shape=(h, w, 3), img_shape=(nh, nw, 3), scale_factor=sf, flip=False))
    hp = max(-(-s[2] // size_divisor) * size_divisor for s in sizes)
    wp = max(-(-s[3] // size_divisor) * size_divisor for s in sizes)
    out = torch.empty(len(images), 3, hp, wp, dtype=torch.float32, device=images[0].device)
    mean = (C.c_float * 3)(*[float(v) for v in norm["mean"]])
    std = (C.c_float * 3)(*[float(v) for v in norm["std"]])
    for i, (im, (h, w, nh, nw)) in enumerate(zip(images, sizes)):
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("images must be uint8 [h, w, 3]")
        _lib.check(lib.sm_preprocess_u8(_lib.ptr(im.contiguous()), h, w, nh, nw, hp, wp, mean, std,
                                        int(bool(norm.get("to_rgb", False))), _lib.ptr(out[i]), _lib.stream_ptr()),
                   "sm_preprocess_u8")
        metas[i]["pad_shape"] = (hp, wp, 3)
    return out, metas
