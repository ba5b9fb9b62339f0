"""Mixed-precision decorators of the reference's module interface (M/mmdet/core/fp16/decorators.py:9-160, utils.py:7-23,
hooks.py:86-105): `auto_fp16`, `force_fp32`, `cast_tensor_type`, `wrap_fp16_model`.

Same names, arguments and behaviour:
  * a decorated method of an nn.Module casts the tensor arguments named in `apply_to` (None = every positional /
    keyword argument of the method's signature; defaults that are not passed are left alone) when -- and only when --
    the module has `fp16_enabled = True`; on anything that is not an nn.Module it raises TypeError;
  * `auto_fp16(out_fp32=True)` / `force_fp32(out_fp16=True)` cast the outputs back;
  * containers are walked like the reference does (mappings and iterables rebuilt with their own type, strings and numpy
    arrays passed through), and EVERY tensor in a named argument is cast, whatever its dtype -- the reference's
    `cast_tensor_type` ignores `src_type` (utils.py:8-9), so only name float tensors in `apply_to`.

One difference, by design: the reduced-precision type is a module-level setting.  The reference hard-codes torch.half
(its only mixed-precision mode, with a static loss scale of 512 -- hooks.py:31); this framework's reduced-precision storage
type is bfloat16 (f32 exponent range: no loss scaling, hence no Fp16OptimizerHook -- SURVEY section 2 row 13), so bfloat16
is the default and `set_reduced_dtype(torch.half)` gives the reference's exact casts.
"""
import functools
from collections import abc
from inspect import getfullargspec

import numpy as np
import torch
import torch.nn as nn

_REDUCED = [torch.bfloat16]


def set_reduced_dtype(dtype):
    """torch.bfloat16 (default) or torch.half: what `auto_fp16` casts to and `force_fp32(out_fp16=True)` casts back to"""
    if dtype not in (torch.bfloat16, torch.half):
        raise ValueError("reduced precision is torch.bfloat16 or torch.half, got %r" % (dtype,))
    _REDUCED[0] = dtype


def reduced_dtype():
    return _REDUCED[0]


def cast_tensor_type(inputs, src_type, dst_type):
    """utils.py:7-23 -- `src_type` is accepted and, as there, not consulted"""
    if isinstance(inputs, torch.Tensor):
        return inputs.to(dst_type)
    if isinstance(inputs, (str, np.ndarray)):
        return inputs
    if isinstance(inputs, abc.Mapping):
        return type(inputs)({k: cast_tensor_type(v, src_type, dst_type) for k, v in inputs.items()})
    if isinstance(inputs, abc.Iterable):
        return type(inputs)(cast_tensor_type(v, src_type, dst_type) for v in inputs)
    return inputs


def _casting_decorator(name, apply_to, to_dtype, back_dtype, cast_output):
    """the one wrapper behind both decorators: arguments -> to_dtype(), outputs -> back_dtype() if cast_output"""

    def wrapper(method):
        spec = getfullargspec(method)

        @functools.wraps(method)
        def wrapped(*args, **kwargs):
            if not args or not isinstance(args[0], nn.Module):
                raise TypeError("@%s can only be used to decorate the method of nn.Module" % name)
            if not getattr(args[0], "fp16_enabled", False):
                return method(*args, **kwargs)
            names = spec.args if apply_to is None else apply_to
            src, dst = back_dtype(), to_dtype()
            cast = lambda n, v: cast_tensor_type(v, src, dst) if n in names else v
            # positional arguments are matched to the signature by position (self included); extra *args pass through
            pos = [cast(spec.args[i], v) if i < len(spec.args) else v for i, v in enumerate(args)]
            kw = {k: cast(k, v) for k, v in kwargs.items()}
            out = method(*pos, **kw)
            return cast_tensor_type(out, dst, src) if cast_output else out

        return wrapped

    return wrapper


def auto_fp16(apply_to=None, out_fp32=False):
    """decorators.py:9-84: named float arguments -> reduced precision, optionally the outputs back to fp32"""
    return _casting_decorator("auto_fp16", apply_to, reduced_dtype, lambda: torch.float, out_fp32)


def force_fp32(apply_to=None, out_fp16=False):
    """decorators.py:87-160: named reduced-precision arguments -> fp32, optionally the outputs back to reduced precision"""
    return _casting_decorator("force_fp32", apply_to, lambda: torch.float, reduced_dtype, out_fp16)


def _norm_forward_in_fp32(forward):
    def fp32_forward(*args, **kwargs):
        out = forward(*cast_tensor_type(args, reduced_dtype(), torch.float),
                      **cast_tensor_type(kwargs, reduced_dtype(), torch.float))
        return cast_tensor_type(out, torch.float, reduced_dtype())
    return fp32_forward


def patch_norm_fp32(module):
    """hooks.py:94-103: normalisation layers keep fp32 parameters; GroupNorm also computes in fp32 and hands the result
    back in reduced precision (BatchNorm kernels take reduced-precision inputs with fp32 statistics natively)"""
    if isinstance(module, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):
        module.float()
        if isinstance(module, nn.GroupNorm):
            module.forward = _norm_forward_in_fp32(module.forward)
    for child in module.children():
        patch_norm_fp32(child)
    return module


def wrap_fp16_model(model):
    """hooks.py:86-92: parameters to reduced precision, norms back to fp32, `fp16_enabled` switched on wherever a module
    declares it.  (The launch plans of SipMask.prepare() read the f32 master weights and pick their own storage types --
    this is the switch for the torch-module path: training forward, `forward_train`, the decorated methods.)"""
    model.to(reduced_dtype())
    patch_norm_fp32(model)
    for m in model.modules():
        if hasattr(m, "fp16_enabled"):
            m.fp16_enabled = True
    return model
