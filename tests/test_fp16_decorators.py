"""auto_fp16 / force_fp32 / cast_tensor_type / wrap_fp16_model against the behaviour of M/mmdet/core/fp16/{decorators,utils,
hooks}.py (CPU; the decorators are pure host logic)."""
import collections

import numpy as np
import pytest
import torch
import torch.nn as nn

from sipmask_amd import fp16 as F16


class _M(nn.Module):
    def __init__(self):
        super().__init__()
        self.fp16_enabled = False
        self.gn = nn.GroupNorm(2, 4)
        self.conv = nn.Conv2d(4, 4, 1)

    @F16.auto_fp16()
    def every(self, x, y, tag="t"):
        return x, y, tag

    @F16.auto_fp16(apply_to=("pred",), out_fp32=True)
    def some(self, pred, others):
        return pred * 1, others

    @F16.force_fp32(apply_to=("cls_scores", "bbox_preds"))
    def loss(self, cls_scores, bbox_preds, cof_preds, extra=None):
        return cls_scores, bbox_preds, cof_preds, extra

    @F16.force_fp32(apply_to=("feats",), out_fp16=True)
    def roi(self, feats):
        return [f + 1 for f in feats]


@pytest.fixture(autouse=True)
def _default_dtype():
    F16.set_reduced_dtype(torch.bfloat16)
    yield
    F16.set_reduced_dtype(torch.bfloat16)


def test_disabled_module_is_untouched():
    m = _M()
    x = torch.ones(2)
    a, b, t = m.every(x, x.double())
    assert a.dtype == torch.float32 and b.dtype == torch.float64 and t == "t"
    assert m.loss(x.half(), x, x)[0].dtype == torch.half


@pytest.mark.parametrize("low", [torch.bfloat16, torch.half])
def test_auto_fp16_casts_named_arguments_only(low):
    F16.set_reduced_dtype(low)
    m = _M()
    m.fp16_enabled = True
    x = torch.ones(2)
    a, b, t = m.every(x, y=[x, {"k": x}, "s", np.ones(2)])
    assert a.dtype == low and b[0].dtype == low and b[1]["k"].dtype == low and b[2] == "s" and isinstance(b[3], np.ndarray)
    assert isinstance(b, list) and t == "t"
    p, o = m.some(x, x)                       # pred -> low -> (out_fp32) back to float; others never touched
    assert p.dtype == torch.float32 and o.dtype == torch.float32
    p, o = m.some(pred=x, others=x.to(low))
    assert p.dtype == torch.float32 and o.dtype == torch.float32     # out_fp32 casts every tensor of the output
    # the reference's cast ignores the source type: an integer tensor in a named argument is cast as well
    assert m.every(torch.ones(2, dtype=torch.int64), x)[0].dtype == low


def test_force_fp32_and_out_fp16():
    m = _M()
    m.fp16_enabled = True
    lo = torch.ones(3, dtype=torch.bfloat16)
    c, b, cof, extra = m.loss([lo, lo], (lo,), lo, extra=lo)
    assert c[0].dtype == torch.float32 and isinstance(b, tuple) and b[0].dtype == torch.float32
    assert cof.dtype == torch.bfloat16 and extra.dtype == torch.bfloat16      # cof_preds stay reduced (sipmask_head.py:289)
    out = m.roi([lo, lo])
    assert out[0].dtype == torch.bfloat16 and float(out[0][0]) == 2.0


def test_mapping_types_and_namedtuple_like_containers_keep_their_type():
    od = collections.OrderedDict(a=torch.ones(1), b="x")
    r = F16.cast_tensor_type(od, torch.float, torch.half)
    assert isinstance(r, collections.OrderedDict) and r["a"].dtype == torch.half and r["b"] == "x"
    assert F16.cast_tensor_type(3, torch.float, torch.half) == 3
    assert F16.cast_tensor_type((torch.ones(1), 2), torch.float, torch.half)[0].dtype == torch.half


def test_only_module_methods_can_be_decorated():
    @F16.auto_fp16()
    def free(x):
        return x

    with pytest.raises(TypeError):
        free(torch.ones(1))

    class NotModule:
        @F16.force_fp32()
        def f(self, x):
            return x

    with pytest.raises(TypeError):
        NotModule().f(torch.ones(1))
    with pytest.raises(ValueError):
        F16.set_reduced_dtype(torch.float64)


def test_wrap_fp16_model_keeps_norms_in_fp32():
    m = F16.wrap_fp16_model(_M())
    assert m.fp16_enabled is True
    assert m.conv.weight.dtype == torch.bfloat16 and m.gn.weight.dtype == torch.float32
    y = m.gn(torch.randn(2, 4, 3, 3).bfloat16())          # computed in fp32, handed back in reduced precision
    assert y.dtype == torch.bfloat16
    ref = nn.GroupNorm(2, 4)
    x = torch.randn(2, 4, 3, 3).bfloat16()
    torch.testing.assert_close(m.gn(x).float(), ref(x.float()).bfloat16().float(), rtol=0, atol=0)


def test_registry_heads_declare_the_flag_and_decorate_like_the_reference():
    """sipmask_head.py:289,500: loss / get_bboxes force cls_scores, bbox_preds, centernesses to fp32 -- cof_preds and
    feat_masks stay as they are"""
    from sipmask_amd.sipmask_head import SipMaskHead
    h = SipMaskHead(num_classes=81, in_channels=256)
    assert h.fp16_enabled is False
    for fn in (SipMaskHead.loss, SipMaskHead.get_bboxes):
        assert hasattr(fn, "__wrapped__")
