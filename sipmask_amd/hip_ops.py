"""Thin tensor -> C-ABI adapters (plumbing only: shape checks, pointers, the current stream).

Everything numeric happens inside libsipmask_hip.so.  Shape / dtype / contiguity validation
lives here so the error messages mirror the reference's Python-side checks
(M/mmdet/ops/dcn/deform_conv.py:27-30,47,50-51; M/mmdet/ops/crop/src/crop_split_cuda.cpp:17).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, DetDesc, SM_MAX_LEVELS

BF16 = torch.bfloat16


def cout_tile(cout):
    return 32 if cout <= 32 else (64 if cout <= 64 else 128)


def prep_conv_weight(w, cin_pad=None):
    """[co,ci,kh,kw] float -> bf16 [cout_pad][Kp]; K order (kh,kw,ci), ci fastest; Kp % 64 == 0."""
    co, ci, kh, kw = w.shape
    cin_pad = cin_pad or ((ci + 7) // 8 * 8)
    tile = cout_tile(co)
    co_pad = (co + tile - 1) // tile * tile
    k = kh * kw * cin_pad
    kp = (k + 63) // 64 * 64
    out = torch.zeros(co_pad, kp, dtype=torch.float32, device=w.device)
    wp = torch.zeros(co, kh, kw, cin_pad, dtype=torch.float32, device=w.device)
    wp[..., :ci] = w.permute(0, 2, 3, 1).float()
    out[:co, :k] = wp.reshape(co, k)
    return out.to(BF16).contiguous(), co_pad


def patch_cout_pad(co):
    """weight rows of the patch-resident kernel: 32 for the convs with a handful of output channels (its 32-cout tile),
    else a multiple of 256"""
    return 32 if co <= 32 else (co + 255) // 256 * 256


def prep_conv_weight_patch(w, co_pad=None):
    """[co,ci,3,3] float -> bf16 [cout_pad][ci/32][9][32] for sm_conv3x3_patch (K order: 32-channel chunk, tap, channel);
    cout_pad a multiple of 256 (default) or 32 (co <= 32: the 32-cout tile), ci a multiple of 64."""
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and ci % 64 == 0
    co_pad = co_pad or (co + 255) // 256 * 256
    assert co_pad >= co and (co_pad % 128 == 0 or co_pad == 32)
    out = torch.zeros(co_pad, ci // 32, 9, 32, dtype=torch.float32, device=w.device)
    out[:co] = w.float().permute(0, 2, 3, 1).reshape(co, 9, ci // 32, 32).permute(0, 2, 1, 3)
    return out.reshape(co_pad, 9 * ci).to(BF16).contiguous(), co_pad


# ------------------------------------------------------------------------------- split-precision ("x3") operands
F16 = torch.float16


def x3_weight_scale(ws):
    """power of two s such that the largest |w| * s lies in [2^11, 2^12): the low half of a ~1e-2 weight (~5e-6) would sit
    in binary16's subnormals (quantum 6e-8) and lose the bits it exists for; scaled it is a normal number.  The conv
    undoes it exactly on the accumulator (sm_conv_desc.acc_scale = 1 / s).  One scale per LAUNCH (grouped convs share it)."""
    import math
    m = max(float(w.detach().abs().max()) for w in ws)
    return 2.0 ** math.floor(math.log2(4096.0 / m)) if m > 0 and math.isfinite(m) else 1.0


def _x3_halves(w, scale, terms=3):
    """w * scale as two binary16 halves along the input-channel axis in the order that pairs with activations laid out
    [hi | lo | hi]: [w_hi | w_hi | w_lo]  =>  x_hi*w_hi + x_lo*w_hi + x_hi*w_lo.  terms=2: [w_hi | w_lo] against
    activations [hi | hi] whose low half is zero (bf16 sources, split2_f16): the same sum without its zero term."""
    ws = w.float() * scale
    hi = ws.to(F16)
    lo = (ws - hi.float()).to(F16)
    return torch.cat([hi, hi, lo] if terms == 3 else [hi, lo], 1)      # [co, terms*ci, kh, kw] binary16 values


def _x3_pairs_channels(w, scale):
    """w * scale as binary16 halves PAIRED along the input-channel axis: [co, 2*ci, kh, kw] with, per 16 channels, the 16 hi
    halves followed by the 16 lo halves -- the weight side of sm_conv_desc.x3_pairs (activations: sm_split_pairs_f16)"""
    co, ci, kh, kw = w.shape
    assert ci % 16 == 0
    ws = w.float() * scale
    hi = ws.to(F16)
    lo = (ws - hi.float()).to(F16)
    g = lambda t: t.reshape(co, ci // 16, 1, 16, kh, kw)
    return torch.cat([g(hi), g(lo)], 2).reshape(co, 2 * ci, kh, kw)


def prep_conv_weight_x3p(w, scale):
    """[co,ci,kh,kw] float -> binary16 [cout_pad][Kp], K order (kh, kw, 2*ci paired) for sm_conv2d with SM_CONV_F16 and
    sm_conv_desc.x3_pairs (its 32-wide-K kernel: one K step = one [hi 16 | lo 16] group)"""
    w2 = _x3_pairs_channels(w, scale)
    co, ci2, kh, kw = w2.shape
    tile = cout_tile(co)
    co_pad = (co + tile - 1) // tile * tile
    k = kh * kw * ci2
    kp = (k + 63) // 64 * 64
    out = torch.zeros(co_pad, kp, dtype=F16, device=w.device)
    out[:co, :k] = w2.permute(0, 2, 3, 1).reshape(co, k)
    return out.contiguous(), co_pad


def prep_conv_weight_x3(w, scale, terms=3):
    """[co,ci,kh,kw] float -> binary16 [cout_pad][Kp], K order (kh,kw,terms*ci) for sm_conv2d with SM_CONV_F16"""
    w3 = _x3_halves(w, scale, terms)
    co, ci3, kh, kw = w3.shape
    assert ci3 % 8 == 0
    tile = cout_tile(co)
    co_pad = (co + tile - 1) // tile * tile
    k = kh * kw * ci3
    kp = (k + 63) // 64 * 64
    out = torch.zeros(co_pad, kp, dtype=F16, device=w.device)
    out[:co, :k] = w3.permute(0, 2, 3, 1).reshape(co, k)
    return out.contiguous(), co_pad


def prep_conv_weight_patch_x3(w, scale, co_pad=None, terms=3):
    """[co,ci,3,3] float -> binary16 [cout_pad][terms*ci/32][9][32] for sm_conv3x3_patch with SM_CONV_F16"""
    w3 = _x3_halves(w, scale, terms)
    co, ci3, kh, kw = w3.shape
    assert (kh, kw) == (3, 3) and ci3 % 64 == 0
    co_pad = co_pad or (co + 255) // 256 * 256
    assert co_pad >= co and (co_pad % 128 == 0 or co_pad == 32)
    out = torch.zeros(co_pad, ci3 // 32, 9, 32, dtype=F16, device=w.device)
    out[:co] = w3.permute(0, 2, 3, 1).reshape(co, 9, ci3 // 32, 32).permute(0, 2, 1, 3)
    return out.reshape(co_pad, 9 * ci3).contiguous(), co_pad


def prep_conv_weight_patch_x3p(w, scale, co_pad=None):
    """[co,ci,3,3] float -> binary16 [cout_pad][ci/16][9][hi 16 | lo 16] for sm_conv3x3_patch with SM_CONV_F16 and
    sm_conv_desc.x3_pairs (round 6): per cout row, 16-channel group and tap the halves hi = f16(w * scale), lo = f16(w * scale -
    hi) side by side -- 64 bytes, one fragment pair; the two taps of a weight stage stay one 128-byte line of the row."""
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and ci % 32 == 0
    ws = w.float() * scale
    hi = ws.to(F16)
    lo = (ws - hi.float()).to(F16)
    co_pad = co_pad or (co + 255) // 256 * 256
    assert co_pad >= co and co_pad % 256 == 0
    lay = lambda t: t.permute(0, 2, 3, 1).reshape(co, 9, ci // 16, 16).permute(0, 2, 1, 3)      # [co, ci/16, tap, 16]
    out = torch.zeros(co_pad, ci // 16, 9, 32, dtype=F16, device=w.device)
    out[:co, ..., :16] = lay(hi)
    out[:co, ..., 16:] = lay(lo)
    return out.reshape(co_pad, 18 * ci).contiguous(), co_pad


def split_pairs_f16(x, y, channels=None, ctot=None, coff=0):
    """x: f32 or bf16 rows [rows, >= channels] -> y binary16 [rows, 2*ctot] in the paired layout (per 16 channels [hi 16 | lo 16];
    sm_split_pairs_f16)"""
    _lib.require_cuda(x, y)
    channels = channels or x.shape[1]
    ctot = ctot or channels
    if x.dtype not in (torch.float32, BF16) or y.dtype != F16 or y.shape[1] != 2 * ctot or y.shape[0] != x.shape[0]:
        raise ValueError("split_pairs_f16: f32 / bf16 rows in, binary16 [rows, 2*ctot] out")
    lib = _lib.load()
    _lib.check(lib.sm_split_pairs_f16(_lib.ptr(x), int(x.dtype == torch.float32), x.shape[0], channels, x.stride(0),
                                      _lib.ptr(y), ctot, coff, _lib.stream_ptr()), "sm_split_pairs_f16")
    return y


def pairs_to_float(y, channels):
    """host-side inverse of the paired layout (tests): binary16 [rows, 2*C] -> (hi, lo) float [rows, C]"""
    v = y.float().view(y.shape[0], channels // 16, 2, 16)
    return v[:, :, 0].reshape(y.shape[0], channels), v[:, :, 1].reshape(y.shape[0], channels)


def prep_deform_weight_x3(w, scale, deform_groups):
    """[co, 64*G, 3, 3] float -> binary16 [cout_pad][G*2*9][hi 32 | lo 32] for sm_deform_conv2d_x3: per cout row and K step
    (group g, channel half c, tap t) the 32 channels g*64 + c*32 .. +31 of w * scale as hi = f16(v), lo = f16(v - hi);
    cout_pad a multiple of 256"""
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3) or ci != 64 * deform_groups:
        raise ValueError("sm_deform_conv2d_x3: 3x3 weights over 64 channels per deformable group")
    ws = w.float() * scale
    hi = ws.to(F16)
    lo = (ws - hi.float()).to(F16)
    co_pad = (co + 255) // 256 * 256
    lay = lambda t: t.reshape(co, deform_groups, 2, 32, 9).permute(0, 1, 2, 4, 3)      # [co, g, c, tap, 32]
    out = torch.zeros(co_pad, deform_groups, 2, 9, 64, dtype=F16, device=w.device)
    out[:co, ..., :32] = lay(hi)
    out[:co, ..., 32:] = lay(lo)
    return out.reshape(co_pad, deform_groups * 18 * 64).contiguous(), co_pad


def deform_conv2d_x3_supported(desc):
    return bool(_lib.load().sm_deform_conv2d_x3_supported(C.byref(desc)))


def deform_conv2d_x3_plan(desc):
    """host logic only: dict(blocks, row_tiles, col_tiles, window_pixels) of sm_deform_conv2d_x3, None when unsupported"""
    out = (C.c_int64 * 4)()
    rc = _lib.load().sm_deform_conv2d_x3_plan(C.byref(desc), out)
    if rc == -4:
        return None
    _lib.check(rc, "sm_deform_conv2d_x3_plan")
    return dict(blocks=int(out[0]), row_tiles=int(out[1]), col_tiles=int(out[2]), window_pixels=int(out[3]))


def deform_conv2d_x3(desc, x, offset, w_split, bias, y, gn_stats=None):
    """FeatureAlign's deformable conv in split precision on the LDS-window kernel: f32 rows in / out, binary16 [hi | lo]
    weights of prep_deform_weight_x3, optional fused GroupNorm statistics (sm_deform_conv2d_x3)"""
    _lib.require_cuda(x, offset, w_split, y)
    _check_gn_stats(gn_stats)
    if x.dtype != torch.float32 or y.dtype != torch.float32 or offset.dtype != torch.float32 or w_split.dtype != F16:
        raise ValueError("deform_conv2d_x3: f32 rows / offsets, binary16 split weights")
    _lib.check(_lib.load().sm_deform_conv2d_x3(C.byref(desc), _lib.ptr(x), _lib.ptr(offset), _lib.ptr(w_split), _lib.ptr(bias),
                                               _lib.ptr(y), _lib.ptr(gn_stats), _lib.stream_ptr()), "sm_deform_conv2d_x3")
    return y


def split3_f16(x, y, channels=None, ctot=None, coff=0):
    """x: f32 or bf16 rows [rows, >= channels] -> y binary16 [rows, 3*ctot]: [hi | lo | hi] of x's first `channels` channels at
    channel offset coff of each third (sm_split3_f16)"""
    _lib.require_cuda(x, y)
    channels = channels or x.shape[1]
    ctot = ctot or channels
    if x.dtype not in (torch.float32, BF16) or y.dtype != F16 or y.shape[1] != 3 * ctot or y.shape[0] != x.shape[0]:
        raise ValueError("split3_f16: f32 / bf16 rows in, binary16 [rows, 3*ctot] out")
    lib = _lib.load()
    _lib.check(lib.sm_split3_f16(_lib.ptr(x), int(x.dtype == torch.float32), x.shape[0], channels, x.stride(0), _lib.ptr(y),
                                 ctot, coff, _lib.stream_ptr()), "sm_split3_f16")
    return y


def split2_f16(x, y, channels=None, ctot=None, coff=0):
    """x: bf16 rows [rows, >= channels] -> y binary16 [rows, 2*ctot]: [hi | hi] (sm_split2_f16: the two-term operand of a source
    whose low half is zero)"""
    _lib.require_cuda(x, y)
    channels = channels or x.shape[1]
    ctot = ctot or channels
    if x.dtype != BF16 or y.dtype != F16 or y.shape[1] != 2 * ctot or y.shape[0] != x.shape[0]:
        raise ValueError("split2_f16: bf16 rows in, binary16 [rows, 2*ctot] out")
    lib = _lib.load()
    _lib.check(lib.sm_split2_f16(_lib.ptr(x), x.shape[0], channels, x.stride(0), _lib.ptr(y), ctot, coff, _lib.stream_ptr()),
               "sm_split2_f16")
    return y


def upsample_bilinear_x3(x, y, batch, h, w, c, factor, ctot, coff):
    """bilinear x factor of f32 rows -> the [hi | lo | hi] slice (coff) of a binary16 [rows, 3*ctot] tensor"""
    _lib.require_cuda(x, y)
    if x.dtype != torch.float32 or y.dtype != F16 or y.shape[1] != 3 * ctot:
        raise ValueError("upsample_bilinear_x3: f32 rows in, binary16 [rows, 3*ctot] out")
    _lib.check(_lib.load().sm_upsample_bilinear_x3(_lib.ptr(x), _lib.ptr(y), batch, h, w, c, factor, x.stride(0), ctot, coff,
                                                   _lib.stream_ptr()), "sm_upsample_bilinear_x3")
    return y


def upsample_sum2(a0, a1, a2, out, batch, h0, w0, c, relu=False):
    """out = [relu](a0 + up2(a1) + up4(a2)) on the [batch, h0, w0] grid (sm_upsample_sum2): bf16 a1 / a2 -> bf16 out (a0 None),
    or f32 a0 / a1 / a2 -> f32 rows or, when `out` is binary16 with 3*c channels, the [hi | lo | hi] split layout"""
    lib = _lib.load()
    _lib.require_cuda(a1, a2, out)
    is_f32 = a1.dtype == torch.float32
    # binary16 out: [hi | lo | hi] (3 * c columns) or the paired layout (2 * c columns, sm_split_pairs_f16)
    out_x3 = 0 if out.dtype != torch.float16 else (2 if out.shape[1] == 2 * c else 1)
    assert a1.dtype == a2.dtype and (a0 is None or a0.dtype == torch.float32)
    assert a1.shape == (batch * (h0 // 2) * (w0 // 2), c) and a2.shape == (batch * (h0 // 4) * (w0 // 4), c)
    assert out.shape == (batch * h0 * w0, (c, 3 * c, 2 * c)[out_x3]) and out.is_contiguous()
    _lib.check(lib.sm_upsample_sum2(_lib.ptr(a0), _lib.ptr(a1), _lib.ptr(a2), int(is_f32), batch, h0, w0, c, int(relu),
                                    int(out_x3), _lib.ptr(out), _lib.stream_ptr()), "sm_upsample_sum2")
    return out


def _lv_geometry(lv):
    nlev = len(lv)
    return nlev, (C.c_int32 * nlev)(*[h * w for h, w in lv.sizes]), (C.c_int64 * nlev)(*lv.row0)


def gn_stats_f32_fix(x, stats, lv, channels, groups=32):
    """fixed-point GroupNorm statistics of f32 pyramid rows (sm_gn_stats_f32_fix)"""
    _check_gn_stats(stats)
    nlev, hw, row0 = _lv_geometry(lv)
    _lib.check(_lib.load().sm_gn_stats_f32_fix(_lib.ptr(x), _lib.ptr(stats), lv.batch, nlev, hw, row0, channels, groups,
                                               _lib.stream_ptr()), "sm_gn_stats_f32_fix")
    return stats


def groupnorm_apply_x3(x, gamma, beta, stats, lv, channels, groups=32, eps=1e-5, relu=True, y_f32=None, y_split=None,
                       y_pairs=None):
    """normalise (+ReLU) f32 rows with fixed-point statistics -> f32 rows and / or [hi | lo | hi] binary16 rows; y_pairs:
    the paired layout instead (binary16 [rows, 2*C], sm_groupnorm_apply_x3p)"""
    _check_gn_stats(stats)
    if y_pairs is not None:
        if y_split is not None or x.dtype != torch.float32 or (y_f32 is not None and y_f32.dtype != torch.float32) or \
                y_pairs.dtype != F16 or y_pairs.shape[1] != 2 * channels:
            raise ValueError("groupnorm_apply_x3: f32 rows in; y_pairs binary16 [rows, 2*C] (and optionally f32 rows) out")
        nlev, hw, row0 = _lv_geometry(lv)
        _lib.check(_lib.load().sm_groupnorm_apply_x3p(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats), lv.batch,
                                                      nlev, hw, row0, channels, groups, eps, int(relu), _lib.ptr(y_f32),
                                                      _lib.ptr(y_pairs), _lib.stream_ptr()), "sm_groupnorm_apply_x3p")
        return
    if x.dtype != torch.float32 or (y_f32 is not None and y_f32.dtype != torch.float32) or \
            (y_split is not None and (y_split.dtype != F16 or y_split.shape[1] != 3 * channels)):
        raise ValueError("groupnorm_apply_x3: f32 rows in; f32 and / or binary16 [rows, 3*C] rows out")
    nlev, hw, row0 = _lv_geometry(lv)
    _lib.check(_lib.load().sm_groupnorm_apply_x3(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats), lv.batch, nlev,
                                                 hw, row0, channels, groups, eps, int(relu), _lib.ptr(y_f32),
                                                 _lib.ptr(y_split), _lib.stream_ptr()), "sm_groupnorm_apply_x3")


def copy_segments(pairs):
    """pairs: up to 8 (src, dst) tensor pairs of equal byte size, each contiguous; src on the device, dst on the device or
    in PINNED host memory -- ONE launch on the current stream copies them all (sm_copy_segments: the per-step results go to
    the host through it instead of through hipMemcpyAsync; see include/sipmask_hip.h)."""
    n = len(pairs)
    if not 1 <= n <= 8:
        raise ValueError("copy_segments: 1 to 8 segments")
    for s, d in pairs:
        if not s.is_cuda or not (d.is_cuda or d.is_pinned()) or not s.is_contiguous() or not d.is_contiguous() or \
                s.numel() * s.element_size() != d.numel() * d.element_size():
            raise ValueError("copy_segments: contiguous device source, device or pinned-host destination, equal byte sizes")
    src = (C.c_void_p * n)(*[s.data_ptr() for s, _ in pairs])
    dst = (C.c_void_p * n)(*[d.data_ptr() for _, d in pairs])
    nb = (C.c_int64 * n)(*[s.numel() * s.element_size() for s, _ in pairs])
    _lib.check(_lib.load().sm_copy_segments(n, src, dst, nb, _lib.stream_ptr()), "sm_copy_segments")


def det_boxes_override(det, sets, counter, flag):
    """det [..., n, >= 4] f32 (a contiguous detection table), sets [nsets, ..., n, 4] f32, counter int32 [1], flag bool / uint8
    scalar, all on the device: one launch of sm_det_boxes_override (evaluation-workload injection, bench.py --det-boxes)"""
    _lib.require_cuda(det, sets, counter, flag)
    n = det.numel() // det.shape[-1]
    if (det.dtype != torch.float32 or sets.dtype != torch.float32 or not det.is_contiguous() or not sets.is_contiguous()
            or sets.shape[-1] != 4 or sets.numel() != sets.shape[0] * n * 4 or counter.dtype != torch.int32
            or flag.element_size() != 1):
        raise ValueError("det_boxes_override: det [.., n, k >= 4] f32, sets [nsets, .., n, 4] f32, int32 counter, 1-byte flag")
    _lib.check(_lib.load().sm_det_boxes_override(_lib.ptr(det), det.shape[-1], _lib.ptr(sets), sets.shape[0], n,
                                                 _lib.ptr(counter), _lib.ptr(flag), _lib.stream_ptr()), "sm_det_boxes_override")


def conv3x3_patch_supported(desc):
    return bool(_lib.load().sm_conv3x3_patch_supported(C.byref(desc)))


def conv3x3_patch_tiles(desc):
    return int(_lib.load().sm_conv3x3_patch_tiles(C.byref(desc)))


def conv3x3_patch_plan(desc):
    """launch shape the library will use (host logic only): 256-position tiles + 128/192-position tiles, the estimated
    makespan in 256-position tile times, and `fill` = work / (256 CUs x makespan)"""
    out = (C.c_int64 * 4)()
    _lib.check(_lib.load().sm_conv3x3_patch_plan(C.byref(desc), out), "sm_conv3x3_patch_plan")
    big, small, spos, milli = (int(v) for v in out)
    ntn = desc.cout_pad // 128 if desc.patch_cout_tile == 128 else max(1, desc.cout_pad // 256)   # (cout_pad 32: one tile)
    groups = max(1, desc.ngroups)
    work = sum(desc.batch * desc.in_h[l] * (desc.in_w[l] + 2) for l in range(desc.nlev)) * ntn * groups / 256.0
    return dict(big=big, small=small, small_pos=spos, makespan=milli / 1000.0, work=work,
                fill=work / (256.0 * max(milli, 1) / 1000.0))


GN_FIX_SCALE = float(1 << 24)      # csrc/common.h: SM_GN_FIX_SHIFT


def gn_stats_alloc(n, device):
    """GroupNorm statistics workspace: n (sum, sum of squares) pairs as 64-bit fixed point (2^-24 units; integer
    accumulation makes the statistics independent of the order in which tiles arrive -- include/sipmask_hip.h)."""
    return torch.zeros(2 * n, dtype=torch.int64, device=device)


def gn_stats_to_float(stats):
    """fixed-point statistics -> float64 (tests, diagnostics)"""
    return stats.double() / GN_FIX_SCALE


def _check_gn_stats(stats):
    if stats is not None and stats.dtype != torch.int64:
        raise TypeError("GroupNorm statistics are int64 fixed point (hip_ops.gn_stats_alloc), got %s" % stats.dtype)


def conv3x3_patch(desc, x, w_patch, bias, y, gn_stats=None):
    _lib.require_cuda(x, w_patch, y)
    _check_gn_stats(gn_stats)
    lib = _lib.load()
    _lib.check(lib.sm_conv3x3_patch(C.byref(desc), _lib.ptr(x), _lib.ptr(w_patch), _lib.ptr(bias), _lib.ptr(y),
                                    _lib.ptr(gn_stats), _lib.stream_ptr()), "sm_conv3x3_patch")
    return y


def conv3x3_smallco_supported(desc):
    return bool(_lib.load().sm_conv3x3_smallco_supported(C.byref(desc)))


def conv3x3_smallco_tiles(desc):
    """blocks of the launch (host arithmetic of csrc/conv3x3_smallco.hip: one wave per 2 x 32-position tile)"""
    return sum(desc.batch * -(-desc.in_h[l] // 2) * -(-desc.in_w[l] // 32) for l in range(desc.nlev))


def prep_conv_weight_smallco(w, x3_scale=None, pairs=False):
    """[co <= 32, ci % 32 == 0, 3, 3] -> the A fragments of sm_conv3x3_smallco: bf16 [ci / 32][9][2][64][8] with
    lane = 32 * khalf + cout row (rows >= co zero) and channel = 32 * slice + 16 * half + 8 * khalf + e.
    x3_scale: the split-precision operand instead -- binary16 fragments of [w_hi | w_hi | w_lo] * scale over 3 * ci channels
    (SM_CONV_F16; pairs with activations laid out [hi | lo | hi])."""
    if x3_scale is not None:      # (pairs: 2 * ci channels, per 16 of them [hi 16 | lo 16] -- sm_conv_desc.x3_pairs)
        w = (_x3_pairs_channels(w, x3_scale) if pairs else _x3_halves(w, x3_scale)).float()
    co, ci, kh, kw = w.shape
    if kh != 3 or kw != 3 or co > 32 or ci % 32 != 0:
        raise ValueError("sm_conv3x3_smallco: 3x3, cout <= 32, cin % 32 == 0")
    w32 = torch.zeros(32, ci, 9, dtype=torch.float32, device=w.device)
    w32[:co] = w.float().reshape(co, ci, 9)
    # ci = slice*32 + half*16 + khalf*8 + e  ->  [m, slice, half, khalf, e, tap] -> [slice, tap, half, khalf, m, e]
    f = w32.view(32, ci // 32, 2, 2, 8, 9).permute(1, 5, 2, 3, 0, 4).contiguous()
    return f.view(ci // 32, 9, 2, 64, 8).to(F16 if x3_scale is not None else torch.bfloat16).contiguous()


def conv3x3_smallco(desc, x, w_frag, bias, y):
    _lib.require_cuda(x, w_frag, y)
    lib = _lib.load()
    _lib.check(lib.sm_conv3x3_smallco(C.byref(desc), _lib.ptr(x), _lib.ptr(w_frag), _lib.ptr(bias), _lib.ptr(y),
                                      _lib.stream_ptr()), "sm_conv3x3_smallco")
    return y


class Levels:
    """Row bookkeeping of a pyramid tensor: levels [(h,w)], batch -> row0 per level."""

    def __init__(self, batch, sizes):
        self.batch = batch
        self.sizes = [(int(h), int(w)) for h, w in sizes]
        self.row0 = []
        r = 0
        for h, w in self.sizes:
            self.row0.append(r)
            r += batch * h * w
        self.rows = r

    def __len__(self):
        return len(self.sizes)


def make_conv_desc(batch, in_sizes, out_sizes, in_row0, out_row0, cin, cout, cout_pad, k, stride, pad,
                   in_cstride, out_cstride, out_coff=0, flags=0, dil=1, res_cstride=0, res_sizes=None,
                   res_row0=None, scale_nch=0, level_scale=None, deform_groups=0, ngroups=1, x_group_rows=0, y_group_rows=0,
                   w_group_stride=0, bias_group_stride=0, gn_group_stride=0, acc_scale=0.0, patch_cout_tile=0, x3_pairs=0):
    d = ConvDesc()
    nlev = len(in_sizes)
    assert 1 <= nlev <= SM_MAX_LEVELS
    d.nlev, d.batch = nlev, batch
    for l in range(nlev):
        d.in_h[l], d.in_w[l] = in_sizes[l]
        d.out_h[l], d.out_w[l] = out_sizes[l]
        d.in_row0[l], d.out_row0[l] = in_row0[l], out_row0[l]
        if res_sizes is not None:
            d.res_h[l], d.res_w[l] = res_sizes[l]
            d.res_row0[l] = res_row0[l]
        d.level_scale[l] = 1.0 if level_scale is None else float(level_scale[l])
    d.cin, d.cout, d.cout_pad = cin, cout, cout_pad
    d.kh, d.kw = (k, k) if isinstance(k, int) else (int(k[0]), int(k[1]))      # k: int or (kh, kw)
    d.stride, d.pad, d.dil = stride, pad, dil
    d.in_cstride, d.out_cstride, d.out_coff = in_cstride, out_cstride, out_coff
    d.res_cstride = res_cstride
    d.flags = flags
    d.scale_nch = scale_nch
    d.deform_groups = deform_groups
    d.ngroups = ngroups
    d.x_group_rows, d.y_group_rows, d.w_group_stride = x_group_rows, y_group_rows, w_group_stride
    d.bias_group_stride, d.gn_group_stride = bias_group_stride, gn_group_stride
    d.acc_scale = float(acc_scale)
    d.patch_cout_tile = int(patch_cout_tile)
    d.x3_pairs = int(x3_pairs)
    return d


def conv_plan(desc, deformable=False, with_gn_stats=False):
    """What the library will launch for `desc` (sm_conv_plan_query: host logic only, works without a GPU)."""
    lib = _lib.load()
    p = _lib.ConvPlan()
    _lib.check(lib.sm_conv_plan_query(C.byref(desc), int(deformable), int(with_gn_stats), C.byref(p)), "sm_conv_plan_query")
    return {f: int(getattr(p, f)) for f, _ in _lib.ConvPlan._fields_}


def conv2d(desc, x, w, bias, residual, y):
    _lib.require_cuda(x, w, y)
    lib = _lib.load()
    _lib.check(lib.sm_conv2d(C.byref(desc), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(residual),
                             _lib.ptr(y), _lib.stream_ptr()), "sm_conv2d")
    return y


def conv2d_ws(desc, x, w, bias, residual, y, workspace):
    """sm_conv2d with a split-K workspace (uint8 tensor or None): see conv_plan(desc)["split_k"] / ["workspace_bytes"]"""
    _lib.require_cuda(x, w, y)
    lib = _lib.load()
    _lib.check(lib.sm_conv2d_ws(C.byref(desc), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(y),
                                _lib.ptr(workspace), 0 if workspace is None else workspace.numel(), _lib.stream_ptr()),
               "sm_conv2d_ws")
    return y


def deform_conv_window_plan(desc):
    """host logic only: dict(blocks, tile=(rows, cols), window_pixels) when the LDS-window kernel (csrc/deform_patch.hip)
    takes this deformable conv, None when the gather loader of conv_igemm.hip does"""
    out = (C.c_int64 * 4)()
    rc = _lib.load().sm_deform_conv_window_plan(C.byref(desc), out)
    if rc == -4:                       # SM_ERR_UNSUPPORTED: the gather loader's conv
        return None
    _lib.check(rc, "sm_deform_conv_window_plan")
    return dict(blocks=int(out[0]), tile=(int(out[1]), int(out[2])), window_pixels=int(out[3]))


def deform_conv2d(desc, x, offset, w, bias, y):
    _lib.require_cuda(x, offset, w, y)
    lib = _lib.load()
    _lib.check(lib.sm_deform_conv2d(C.byref(desc), _lib.ptr(x), _lib.ptr(offset), _lib.ptr(w), _lib.ptr(bias),
                                    _lib.ptr(y), _lib.stream_ptr()), "sm_deform_conv2d")
    return y


def conv2d_gn_stats(desc, x, offset, w, bias, residual, y, stats):
    """conv (deformable when offset is given) with the output's GroupNorm statistics fused in the epilogue"""
    _check_gn_stats(stats)
    lib = _lib.load()
    _lib.check(lib.sm_conv2d_gn_stats(C.byref(desc), _lib.ptr(x), _lib.ptr(offset), _lib.ptr(w), _lib.ptr(bias),
                                      _lib.ptr(residual), _lib.ptr(y), _lib.ptr(stats), _lib.stream_ptr()),
               "sm_conv2d_gn_stats")
    return y


def groupnorm_apply(x, y, gamma, beta, stats, lv, channels, groups=32, eps=1e-5, relu=True):
    _check_gn_stats(stats)
    lib = _lib.load()
    nlev = len(lv)
    hw = (C.c_int32 * nlev)(*[h * w for h, w in lv.sizes])
    row0 = (C.c_int64 * nlev)(*lv.row0)
    _lib.check(lib.sm_groupnorm_apply(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                      lv.batch, nlev, hw, row0, channels, groups, eps, int(relu), _lib.stream_ptr()),
               "sm_groupnorm_apply")
    return y


def offset_linear(reg, reg_cstride, w_off, lv, out, level_scale=None):
    lib = _lib.load()
    nlev = len(lv)
    row0 = (C.c_int64 * nlev)(*lv.row0)
    rows = (C.c_int32 * nlev)(*[lv.batch * h * w for h, w in lv.sizes])
    ls = None if level_scale is None else (C.c_float * nlev)(*[float(v) for v in level_scale])
    _lib.check(lib.sm_offset_linear(_lib.ptr(reg), reg_cstride, _lib.ptr(w_off), w_off.shape[0], row0, rows, ls,
                                    nlev, _lib.ptr(out), _lib.stream_ptr()), "sm_offset_linear")
    return out


def offset_linear_bwd(reg, reg_cstride, grad_out, out=None):
    """grad of FeatureAlign.conv_offset's weight: [nout, 4] = grad_out^T [nout, rows] . reg[:, :4] (sm_offset_linear_bwd;
    deterministic two-pass reduction on the device -- replaces the ATen matmul backward = a vendor GEMM)"""
    lib = _lib.load()
    _lib.require_cuda(reg, grad_out)
    rows, nout = grad_out.shape
    assert grad_out.dtype == torch.float32 and reg.dtype == torch.float32 and grad_out.is_contiguous()
    ws = torch.empty(int(lib.sm_offset_linear_bwd_workspace(rows, nout)) // 4, dtype=torch.float32, device=reg.device)
    if out is None:
        out = torch.empty(nout, 4, dtype=torch.float32, device=reg.device)
    _lib.check(lib.sm_offset_linear_bwd(_lib.ptr(reg), reg_cstride, _lib.ptr(grad_out), nout, rows, _lib.ptr(ws),
                                        _lib.ptr(out), _lib.stream_ptr()), "sm_offset_linear_bwd")
    return out


def bottleneck_tail(batch, h, w, channels, x, w2, b2, w3, b3, identity, y, w1_next=None, b1_next=None, t1_next=None):
    """conv2 + conv3 (+ the next block's conv1) of a ResNet bottleneck as one launch (resnet.py:167-200); weights in
    the prep_conv_weight layout ([cout][K], K = (kh, kw, cin)), bf16 rows, f32 biases"""
    lib = _lib.load()
    _lib.require_cuda(x, w2, b2, w3, b3, identity, y)
    C4 = 4 * channels
    assert tuple(w2.shape) == (channels, 9 * channels) and tuple(w3.shape) == (C4, channels), (w2.shape, w3.shape)
    assert x.shape[1] == channels and identity.shape[1] == C4 and y.shape[1] == C4
    assert min(x.shape[0], identity.shape[0], y.shape[0]) >= batch * h * w
    if w1_next is not None:
        assert tuple(w1_next.shape) == (channels, C4) and t1_next.shape[1] == channels and t1_next.shape[0] >= batch * h * w
    _lib.check(lib.sm_bottleneck_tail(batch, h, w, channels, _lib.ptr(x), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(w3),
                                      _lib.ptr(b3), _lib.ptr(identity), _lib.ptr(y),
                                      None if w1_next is None else _lib.ptr(w1_next),
                                      None if w1_next is None else _lib.ptr(b1_next),
                                      None if w1_next is None else _lib.ptr(t1_next), _lib.stream_ptr()),
               "sm_bottleneck_tail")
    return y


def bottleneck_tail_ds(batch, h, w, channels, x, w2, b2, w3_ds, b3_ds, x_block, y, w1_next=None, b1_next=None, t1_next=None,
                       ds_stride=1, ds_hw=None):
    """conv2 + conv3 + the block's 1x1 shortcut conv (resnet.py:453-469, stride 1) as one launch: w3_ds = [w3 | w_downsample]
    ([4C][C + Cds] bf16), b3_ds = b3 + b_downsample; x_block = the block input rows the shortcut conv reads."""
    lib = _lib.load()
    _lib.require_cuda(x, w2, b2, w3_ds, b3_ds, x_block, y)
    C4, cds = 4 * channels, x_block.shape[1]
    assert tuple(w2.shape) == (channels, 9 * channels) and tuple(w3_ds.shape) == (C4, channels + cds), (w2.shape, w3_ds.shape)
    dh, dw = ds_hw if ds_hw is not None else (h, w)
    assert x.shape[1] == channels and y.shape[1] == C4 and min(x.shape[0], y.shape[0]) >= batch * h * w
    assert x_block.shape[0] >= batch * dh * dw and x_block.is_contiguous()
    if w1_next is not None:
        assert tuple(w1_next.shape) == (channels, C4) and t1_next.shape[1] == channels and t1_next.shape[0] >= batch * h * w
    _lib.check(lib.sm_bottleneck_tail_ds(batch, h, w, channels, _lib.ptr(x), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(w3_ds),
                                         _lib.ptr(b3_ds), _lib.ptr(x_block), cds, ds_stride, dh, dw, _lib.ptr(y),
                                         None if w1_next is None else _lib.ptr(w1_next),
                                         None if w1_next is None else _lib.ptr(b1_next),
                                         None if w1_next is None else _lib.ptr(t1_next), _lib.stream_ptr()),
               "sm_bottleneck_tail_ds")
    return y


def conv1x1_pair(rows, channels, x, w3, b3, identity, y, w1_next, b1_next, t1_next):
    """conv3 (+ identity, ReLU) of one bottleneck + conv1 (ReLU) of the next as one launch (resnet.py:188-200, :175-178)"""
    lib = _lib.load()
    _lib.require_cuda(x, w3, b3, identity, y, w1_next, b1_next, t1_next)
    C4 = 4 * channels
    assert tuple(w3.shape) == (C4, channels) and tuple(w1_next.shape) == (channels, C4), (w3.shape, w1_next.shape)
    assert x.shape[1] == channels and identity.shape[1] == C4 and y.shape[1] == C4 and t1_next.shape[1] == channels
    assert min(x.shape[0], identity.shape[0], y.shape[0], t1_next.shape[0]) >= rows
    _lib.check(lib.sm_conv1x1_pair(rows, channels, _lib.ptr(x), _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(identity), _lib.ptr(y),
                                   _lib.ptr(w1_next), _lib.ptr(b1_next), _lib.ptr(t1_next), _lib.stream_ptr()),
               "sm_conv1x1_pair")
    return y


def relu_bf16(x, y):
    """y = relu(x), bf16, same shape (fpn.py:166-170: the ReLU in front of the P7 conv)"""
    lib = _lib.load()
    _lib.require_cuda(x, y)
    assert x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and x.numel() == y.numel()
    _lib.check(lib.sm_relu_bf16(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_ptr()), "sm_relu_bf16")
    return y


def deform_conv2d_bwd(desc, x, offset, w_t, gout, grad_x, grad_offset, grad_w_t):
    """grad_x f32 [rows][cin], grad_offset f32 like offset, grad_w_t f32 [K][cout]; None = skip."""
    lib = _lib.load()
    ws = torch.empty(int(lib.sm_deform_conv2d_bwd_workspace(C.byref(desc))), dtype=torch.uint8, device=x.device)
    _lib.check(lib.sm_deform_conv2d_bwd(C.byref(desc), _lib.ptr(x), _lib.ptr(offset), _lib.ptr(w_t), _lib.ptr(gout),
                                        _lib.ptr(grad_x), _lib.ptr(grad_offset), _lib.ptr(grad_w_t), _lib.ptr(ws),
                                        _lib.stream_ptr()), "sm_deform_conv2d_bwd")


def conv2d_bwd(desc, x, w_t, w_dgrad, gout, grad_x, grad_w_t, grad_bias):
    """Plain conv backward (see sm_conv2d_bwd).  Outputs may be None."""
    lib = _lib.load()
    ws = torch.empty(int(lib.sm_deform_conv2d_bwd_workspace(C.byref(desc))), dtype=torch.uint8, device=x.device)
    _lib.check(lib.sm_conv2d_bwd(C.byref(desc), _lib.ptr(x), _lib.ptr(w_t), _lib.ptr(w_dgrad), _lib.ptr(gout),
                                 _lib.ptr(grad_x), _lib.ptr(grad_w_t), _lib.ptr(grad_bias), _lib.ptr(ws),
                                 _lib.stream_ptr()), "sm_conv2d_bwd")


def groupnorm(x, y, gamma, beta, stats, lv, channels, groups=32, eps=1e-5, relu=True):
    _check_gn_stats(stats)
    lib = _lib.load()
    nlev = len(lv)
    hw = (C.c_int32 * nlev)(*[h * w for h, w in lv.sizes])
    row0 = (C.c_int64 * nlev)(*lv.row0)
    _lib.check(lib.sm_groupnorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                lv.batch, nlev, hw, row0, channels, groups, eps, int(relu), _lib.stream_ptr()),
               "sm_groupnorm")
    return y


def maxpool3x3s2(x, y, batch, h, w, c):
    lib = _lib.load()
    _lib.check(lib.sm_maxpool3x3s2(_lib.ptr(x), _lib.ptr(y), batch, h, w, c, _lib.stream_ptr()), "sm_maxpool3x3s2")
    return y


def prep_stem_weight(w):
    """[64, 3, 7, 7] f32 (BN folded) -> the operand of sm_stem_fused: bf16 [64][7][8][4] = (cout, kh, kw, cin), zeros in
    the kw = 7 and cin = 3 slots (two kw-adjacent pixels x 4 channels = one 16-byte MFMA fragment)."""
    co, ci, kh, kw = w.shape
    if (co, ci, kh, kw) != (64, 3, 7, 7):
        raise ValueError("sm_stem_fused is the 3 -> 64, 7x7 ResNet stem")
    wp = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device=w.device)
    wp[:, :, :7, :3] = w.float().permute(0, 2, 3, 1)
    return wp.to(torch.bfloat16).contiguous()


def stem_fused(img, w_stem, bias, y):
    """conv1 + folded bn1 + ReLU + maxpool of resnet.py:497-505 in one launch: img NCHW f32 -> y NHWC bf16 rows."""
    lib = _lib.load()
    _lib.require_cuda(img, w_stem, bias, y)
    if img.dtype != torch.float32 or not img.is_contiguous() or img.shape[1] != 3:
        raise ValueError("expected a contiguous float32 [B, 3, H, W] image")
    b, _, h, w = img.shape
    _lib.check(lib.sm_stem_fused(_lib.ptr(img), _lib.ptr(w_stem), _lib.ptr(bias), _lib.ptr(y), b, h, w,
                                 _lib.stream_ptr()), "sm_stem_fused")
    return y


def nchw_to_nhwc_bf16(x, y, cpad):
    lib = _lib.load()
    _lib.require_cuda(x, y)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("expected a contiguous float32 NCHW tensor")
    b, c, h, w = x.shape
    _lib.check(lib.sm_nchw_f32_to_nhwc_bf16(_lib.ptr(x), _lib.ptr(y), b, c, h, w, cpad, _lib.stream_ptr()),
               "sm_nchw_f32_to_nhwc_bf16")
    return y


def upsample_bilinear(x, y, batch, h, w, c, factor, in_cstride, out_cstride, out_coff, is_f32):
    lib = _lib.load()
    _lib.check(lib.sm_upsample_bilinear(_lib.ptr(x), _lib.ptr(y), batch, h, w, c, factor, in_cstride, out_cstride,
                                        out_coff, int(is_f32), _lib.stream_ptr()), "sm_upsample_bilinear")
    return y


# ---- exact-f32 plan (parity mode): f32 activations / weights, v_mfma_f32_32x32x2_f32 (csrc/conv_f32.hip)

def prep_conv_weight_f32(w, cin_pad=None):
    """[co,ci,kh,kw] float -> f32 [cout_pad][Kp]; K order (kh,kw,ci), ci fastest (padded to a multiple of 4); Kp % 16 == 0."""
    co, ci, kh, kw = w.shape
    cin_pad = cin_pad or ((ci + 3) // 4 * 4)
    tile = cout_tile(co)
    co_pad = (co + tile - 1) // tile * tile
    k = kh * kw * cin_pad
    kp = (k + 15) // 16 * 16
    out = torch.zeros(co_pad, kp, dtype=torch.float32, device=w.device)
    wp = torch.zeros(co, kh, kw, cin_pad, dtype=torch.float32, device=w.device)
    wp[..., :ci] = w.permute(0, 2, 3, 1).float()
    out[:co, :k] = wp.reshape(co, k)
    return out.contiguous(), co_pad


def conv2d_f32(desc, x, offset, w, bias, residual, y):
    """sm_conv2d_f32: offset given -> deformable conv; every tensor float32"""
    _lib.require_cuda(x, w, y)
    for t in (x, offset, w, bias, residual, y):
        if t is not None and t.dtype != torch.float32:
            raise ValueError("sm_conv2d_f32 takes float32 tensors, got %s" % t.dtype)
    lib = _lib.load()
    _lib.check(lib.sm_conv2d_f32(C.byref(desc), _lib.ptr(x), _lib.ptr(offset), _lib.ptr(w), _lib.ptr(bias),
                                 _lib.ptr(residual), _lib.ptr(y), _lib.stream_ptr()), "sm_conv2d_f32")
    return y


def groupnorm_f32(x, y, gamma, beta, stats, lv, channels, groups=32, eps=1e-5, relu=True):
    """stats: float64 workspace [batch*nlev*groups*2]"""
    lib = _lib.load()
    if stats.dtype != torch.float64:
        raise ValueError("sm_groupnorm_f32 needs a float64 statistics workspace")
    nlev = len(lv)
    hw = (C.c_int32 * nlev)(*[h * w for h, w in lv.sizes])
    row0 = (C.c_int64 * nlev)(*lv.row0)
    _lib.check(lib.sm_groupnorm_f32(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                    lv.batch, nlev, hw, row0, channels, groups, eps, int(relu), _lib.stream_ptr()),
               "sm_groupnorm_f32")
    return y


def maxpool3x3s2_f32(x, y, batch, h, w, c):
    lib = _lib.load()
    _lib.check(lib.sm_maxpool3x3s2_f32(_lib.ptr(x), _lib.ptr(y), batch, h, w, c, _lib.stream_ptr()), "sm_maxpool3x3s2_f32")
    return y


def nchw_to_nhwc_f32(x, y, cpad):
    lib = _lib.load()
    _lib.require_cuda(x, y)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("expected a contiguous float32 NCHW tensor")
    b, c, h, w = x.shape
    _lib.check(lib.sm_nchw_f32_to_nhwc_f32(_lib.ptr(x), _lib.ptr(y), b, c, h, w, cpad, _lib.stream_ptr()),
               "sm_nchw_f32_to_nhwc_f32")
    return y


def scale4(scale_factor):
    """img_meta['scale_factor'] -> [w, h, w, h] floats: a scalar (keep_ratio=True, transforms.py Resize) is
    repeated, a 4-array (keep_ratio=False) is taken as is."""
    import numpy as np
    a = np.asarray(scale_factor, dtype=np.float32).reshape(-1)
    if a.size == 1:
        a = np.repeat(a, 4)
    if a.size != 4:
        raise ValueError("scale_factor must be a scalar or 4 values, got %r" % (scale_factor,))
    return [float(v) for v in a]


def _pair(v):
    return (float(v[0]), float(v[1])) if isinstance(v, (tuple, list)) else (float(v), float(v))


def post_geometry(hm, wm, scale_factor, rescale, ssd_flag=False):
    """Crop / upsample geometry of get_bboxes_single (sipmask_head.py:621-632) ->
    (box_mul (x, y), up_scale (h, w), (ho, wo)).  Quirk preserved: crop boxes are multiplied by scale_factor unless
    rescale is None, while they were divided by it only when rescale is truthy (:587-588)."""
    import numpy as np
    a = np.asarray(scale_factor).reshape(-1)
    if rescale is None:
        mul, up = (1.0, 1.0), (2.0, 2.0)
    elif a.size == 1:
        v = float(a[0])                                   # python float in img_meta: double arithmetic
        mul, up = (v, v), (2.0 / v, 2.0 / v)
    else:
        if not ssd_flag:
            raise ValueError("a 4-element scale_factor needs ssd_flag=True (F.interpolate gets 2 factors, :629-630)")
        s4 = np.asarray(scale_factor, np.float32).reshape(4)
        if s4[0] != s4[2] or s4[1] != s4[3]:
            raise NotImplementedError("scale_factor must be [w, h, w, h]")
        mul = (float(s4[0]), float(s4[1]))
        up = (float(np.float32(2) / s4[3]), float(np.float32(2) / s4[2]))   # 2 / scale_factor[3:1:-1] in float32
    import math
    return mul, up, (int(math.floor(hm * up[0])), int(math.floor(wm * up[1])))


def image_geometry_tables(img_metas, hm, wm, rescale, ssd_flag=False):
    """Per-image post-processing geometry from the batch's img_metas (sipmask_head.py:517-541: get_bboxes_single runs with
    img_metas[img_id]['img_shape'] / ['scale_factor']): host tensors
      det  f32 [B, 6] = (img_h, img_w, scale_factor x 4)                       -> sm_det_desc.per_image
      geom f32 [B, 8] = (box_mul_x, box_mul_y, up_h, up_w, Ho, Wo, 1/up_h, 1/up_w) -> sm_mask_assemble*.per_image
    and the bounds a launch needs as scalars: canvas (max Ho, max Wo), the smallest up_scale (h, w)."""
    import numpy as np
    B = len(img_metas)
    det = torch.zeros(B, 6, dtype=torch.float32)
    geom = torch.zeros(B, 8, dtype=torch.float32)
    for b, m in enumerate(img_metas):
        sf = np.asarray(m.get('scale_factor', 1.0), np.float32).reshape(-1)
        sf4 = np.repeat(sf, 4) if sf.size == 1 else sf.reshape(4)
        det[b, 0], det[b, 1] = float(m['img_shape'][0]), float(m['img_shape'][1])
        det[b, 2:] = torch.from_numpy(sf4.astype(np.float32))
        mul, up, (ho, wo) = post_geometry(hm, wm, m.get('scale_factor', 1.0), rescale, ssd_flag)
        geom[b] = torch.tensor([mul[0], mul[1], up[0], up[1], ho, wo, float(np.float32(1.0 / up[0])), float(np.float32(1.0 / up[1]))])
    canvas = (int(geom[:, 4].max()), int(geom[:, 5].max()))
    up_min = (float(geom[:, 2].min()), float(geom[:, 3].min()))
    return det, geom, canvas, up_min


def make_det_desc(batch, sizes, strides, row0, num_classes, cls_cstride, cls_coff, cof_cstride, cof_coff,
                  reg_cstride, nms_pre, img_h, img_w, scale_factor=1.0, rescale=False, reg_prescaled=False, kmax=None):
    d = DetDesc()
    d.batch, d.nlev, d.num_classes = batch, len(sizes), num_classes
    ksum = 0
    for l, (h, w) in enumerate(sizes):
        d.h[l], d.w[l], d.stride[l], d.row0[l] = h, w, strides[l], row0[l]
        ksum += min(nms_pre, h * w) if nms_pre > 0 else h * w
    d.cls_cstride, d.cls_coff, d.cof_cstride, d.cof_coff = cls_cstride, cls_coff, cof_cstride, cof_coff
    d.reg_cstride, d.nms_pre, d.img_h, d.img_w, d.kmax = reg_cstride, nms_pre, img_h, img_w, ksum
    for i, v in enumerate(scale4(scale_factor)):
        d.scale_factor[i] = v
    d.rescale = int(bool(rescale))
    d.reg_prescaled = int(bool(reg_prescaled))
    if kmax is not None:          # pair selection (sm_pairs_select): sum_l min(nms_pre, h*w*C)
        d.kmax = int(kmax)
    return d


def det_select(desc, cls, reg, cof, out):
    """out: dict with preallocated boxes/scores/ctr/cofs/cand_pos/ncand/ws tensors."""
    lib = _lib.load()
    _lib.check(lib.sm_det_select(C.byref(desc), _lib.ptr(cls), _lib.ptr(reg), _lib.ptr(cof), _lib.ptr(out["boxes"]),
                                 _lib.ptr(out["scores"]), _lib.ptr(out["ctr"]), _lib.ptr(out["cofs"]),
                                 _lib.ptr(out["cand_pos"]), _lib.ptr(out["ncand"]), _lib.ptr(out["ws_sel"]),
                                 _lib.stream_ptr()), "sm_det_select")


def det_select_alloc(desc, device):
    lib = _lib.load()
    b, k, c = desc.batch, desc.kmax, desc.num_classes
    ws = lib.sm_det_select_workspace(C.byref(desc))
    if ws < 0:
        raise RuntimeError("sm_det_select_workspace: bad descriptor")
    f32, i32 = torch.float32, torch.int32
    return dict(boxes=torch.empty(b, k, 4, dtype=f32, device=device), scores=torch.empty(b, c, k, dtype=f32, device=device),
                ctr=torch.empty(b, k, dtype=f32, device=device), cofs=torch.empty(b, k, 128, dtype=f32, device=device),
                cand_pos=torch.empty(b, k, dtype=i32, device=device), ncand=torch.empty(b, dtype=i32, device=device),
                ws_sel=torch.empty(max(int(ws), 16), dtype=torch.uint8, device=device))


def multiclass_nms_alloc(batch, kmax, num_classes, max_num, device):
    lib = _lib.load()
    ws = lib.sm_multiclass_nms_workspace(batch, kmax, num_classes)
    return dict(det=torch.zeros(batch, max_num, 5, dtype=torch.float32, device=device),
                labels=torch.zeros(batch, max_num, dtype=torch.int64, device=device),
                keep=torch.zeros(batch, max_num, dtype=torch.int64, device=device),
                ndet=torch.zeros(batch, dtype=torch.int32, device=device),
                ws_nms=torch.empty(int(ws), dtype=torch.uint8, device=device))


def multiclass_nms(boxes, scores, ctr, ncand, score_thr, iou_thr, max_num, out):
    lib = _lib.load()
    b, c, k = scores.shape          # class-major [B][C][kmax]
    _lib.check(lib.sm_multiclass_nms(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(ctr), _lib.ptr(ncand), b, k, c,
                                     float(score_thr), float(iou_thr), int(max_num), _lib.ptr(out["det"]),
                                     _lib.ptr(out["labels"]), _lib.ptr(out["keep"]), _lib.ptr(out["ndet"]),
                                     _lib.ptr(out["ws_nms"]), _lib.stream_ptr()), "sm_multiclass_nms")


def pairs_select_alloc(desc, device):
    lib = _lib.load()
    b, k, c = desc.batch, desc.kmax, desc.num_classes
    ws = lib.sm_pairs_select_workspace(C.byref(desc))
    if ws < 0:
        raise ValueError("invalid pair-selection descriptor")
    return dict(boxes=torch.zeros(b, k, 4, dtype=torch.float32, device=device),
                scores=torch.zeros(b, c, k, dtype=torch.float32, device=device),
                ctr=torch.ones(b, k, dtype=torch.float32, device=device),
                cofs=torch.zeros(b, k, 128, dtype=torch.float32, device=device),
                lvl_cnt=torch.zeros(b, SM_MAX_LEVELS, dtype=torch.int32, device=device),
                ncand=torch.zeros(b, dtype=torch.int32, device=device),
                ws=torch.empty(int(ws), dtype=torch.uint8, device=device))


def pairs_select(desc, pre_nms_thresh, cls, reg, cof, out):
    lib = _lib.load()
    _lib.check(lib.sm_pairs_select(C.byref(desc), float(pre_nms_thresh), _lib.ptr(cls), _lib.ptr(reg), _lib.ptr(cof),
                                   _lib.ptr(out["boxes"]), _lib.ptr(out["scores"]), _lib.ptr(out["cofs"]),
                                   _lib.ptr(out["lvl_cnt"]), _lib.ptr(out["ncand"]), _lib.ptr(out["ws"]),
                                   _lib.stream_ptr()), "sm_pairs_select")


def fast_nms(boxes, scores, ctr, ncand, score_thr, iou_thr, top_k, max_num, out):
    lib = _lib.load()
    b, c, k = scores.shape
    _lib.check(lib.sm_fast_nms(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(ctr), _lib.ptr(ncand), b, k, c,
                               float(score_thr), float(iou_thr), int(top_k), int(max_num), _lib.ptr(out["det"]),
                               _lib.ptr(out["labels"]), _lib.ptr(out["keep"]), _lib.ptr(out["ndet"]),
                               _lib.ptr(out["ws_nms"]), _lib.stream_ptr()), "sm_fast_nms")


def mask_assemble(basis, basis_hwc, cofs, keep, det, ndet, hm, wm, ho, wo, box_mul, box_div, up_scale, thr,
                  masks, pos_masks=None, per_image=None):
    """box_mul: scalar or (x, y); up_scale: scalar or (h, w) -- per axis for keep_ratio=False pipelines.
    masks u8 [B, max_num, ho, pitch] with pitch % 4 == 0 and pitch >= wo (the logical width).
    per_image: device table f32 [B, 8] (image_geometry_tables) -- then ho / wo are the canvas, up_scale the batch's smallest."""
    lib = _lib.load()
    (mx, my), (uh, uw) = _pair(box_mul), _pair(up_scale)
    b, kmax = cofs.shape[0], cofs.shape[1]
    max_num = det.shape[1]
    _lib.check(lib.sm_mask_assemble(_lib.ptr(basis), int(basis_hwc), _lib.ptr(cofs), _lib.ptr(keep), _lib.ptr(det),
                                    _lib.ptr(ndet), b, kmax, max_num, hm, wm, ho, wo, int(masks.shape[-1]), mx, my,
                                    float(box_div),
                                    uh, uw, float(thr), _lib.ptr(masks), _lib.ptr(pos_masks), _lib.ptr(per_image),
                                    _lib.stream_ptr()), "sm_mask_assemble")
    return masks


def mask_assemble_lo_supported(batch, max_num, factor, up_scale):
    """does sm_mask_assemble_lo take this geometry (host logic; else the plan assembles from the upsampled basis)"""
    uh, uw = _pair(up_scale)
    return bool(_lib.load().sm_mask_assemble_lo_supported(int(batch), int(max_num), int(factor), float(uh), float(uw)))


def mask_assemble_lo_alloc(batch, max_num, ho, wo, device):
    """buffers owned by the plan for sm_mask_assemble_lo: the u8 masks (zeroed once), the per-slot tile-range state
    (zeroed with them) and the work-list workspace"""
    lib = _lib.load()
    pitch = (wo + 3) // 4 * 4
    return dict(masks=torch.zeros(batch, max_num, ho, pitch, dtype=torch.uint8, device=device),
                state=torch.zeros(batch * max_num, 4, dtype=torch.int32, device=device),
                ws=torch.empty(int(lib.sm_mask_assemble_lo_workspace(batch, max_num)), dtype=torch.uint8, device=device))


def mask_assemble_lo(basis_lo, lo_h, lo_w, factor, cofs, keep, det, ndet, ho, wo, box_mul, box_div, up_scale, thr, buf,
                     per_image=None):
    """fused-upsample, rectangle-tracked mask assembly (see sm_mask_assemble_lo); buf from mask_assemble_lo_alloc"""
    lib = _lib.load()
    (mx, my), (uh, uw) = _pair(box_mul), _pair(up_scale)
    b, kmax = cofs.shape[0], cofs.shape[1]
    max_num = det.shape[1]
    masks = buf["masks"]
    _lib.check(lib.sm_mask_assemble_lo(_lib.ptr(basis_lo), lo_h, lo_w, factor, _lib.ptr(cofs), _lib.ptr(keep), _lib.ptr(det),
                                       _lib.ptr(ndet), b, kmax, max_num, ho, wo, int(masks.shape[-1]), mx, my, float(box_div),
                                       uh, uw, float(thr), _lib.ptr(masks), _lib.ptr(buf["state"]), _lib.ptr(buf["ws"]),
                                       _lib.ptr(per_image), _lib.stream_ptr()), "sm_mask_assemble_lo")
    return masks


# ------------------------------------------------------------------------------- device RLE (result packing)
def rle_alloc(batch, max_num, canvas_w, device, max_runs=8192, packed_cap=None):
    lib = _lib.load()
    nd = batch * max_num
    if packed_cap is None:
        packed_cap = nd * max_runs              # ~1-2 characters per run in practice; checked after the launch
    ws = lib.sm_rle_workspace(batch, max_num, canvas_w, max_runs)
    return dict(counts=torch.empty(nd, max_runs, dtype=torch.int32, device=device),
                nruns=torch.zeros(nd, dtype=torch.int32, device=device),
                nchars=torch.zeros(nd, dtype=torch.int32, device=device),
                packed=torch.empty(int(packed_cap), dtype=torch.uint8, device=device),
                offsets=torch.zeros(nd + 1, dtype=torch.int64, device=device),
                rect=torch.zeros(nd, 4, dtype=torch.int32, device=device),
                ws=torch.empty(int(ws), dtype=torch.uint8, device=device), max_runs=max_runs, canvas_w=canvas_w)


def mask_rects(det, box_mul, box_div, up_scale, rect, per_image=None):
    lib = _lib.load()
    b, n = det.shape[0], det.shape[1]
    (mx, my), (uh, uw) = _pair(box_mul), _pair(up_scale)
    _lib.check(lib.sm_mask_rects(_lib.ptr(det), b, n, mx, my, float(box_div), uh, uw,
                                 _lib.ptr(rect), _lib.ptr(per_image), _lib.stream_ptr()), "sm_mask_rects")
    return rect


def rle_encode(masks, ndet, canvas_hw, out, rect=None, per_image=None):
    """masks u8 [B][max_num][ho][wo] -> run lengths + packed rleToString bytes, all on device (no sync).
    per_image: int32 device tensor [B, 4] = (mask_h, mask_w, canvas_h, canvas_w) of every image (sm_rle_encode_images);
    canvas_hw is then the largest canvas of the batch."""
    lib = _lib.load()
    b, n, ho, wo = masks.shape
    assert out["canvas_w"] >= int(canvas_hw[1])
    if per_image is not None and (per_image.dtype != torch.int32 or tuple(per_image.shape) != (b, 4) or not per_image.is_cuda):
        raise ValueError("rle_encode: per_image is an int32 device tensor [batch, 4]")
    _lib.check(lib.sm_rle_encode_images(_lib.ptr(masks), _lib.ptr(ndet), _lib.ptr(rect), _lib.ptr(per_image), b, n, ho, wo,
                                        int(canvas_hw[0]), int(canvas_hw[1]), int(out["max_runs"]), _lib.ptr(out["counts"]),
                                        _lib.ptr(out["nruns"]), _lib.ptr(out["nchars"]), _lib.ptr(out["packed"]),
                                        int(out["packed"].numel()), _lib.ptr(out["offsets"]), _lib.ptr(out["ws"]),
                                        _lib.stream_ptr()), "sm_rle_encode_images")


def rle_fetch(out, batch, max_num, ndet, canvas_hw):
    """Two D2H copies for the whole batch (offsets+run counts, then the packed strings) ->
    per image a list of {'size': [H, W], 'counts': bytes} (the pycocotools RLE dict, sipmask_head.py:655)."""
    nruns = out["nruns"].cpu().numpy()
    offs = out["offsets"].cpu().numpy()
    if (nruns < 0).any():
        raise RuntimeError("sm_rle_encode: max_runs=%d too small, a mask needs %d runs" % (out["max_runs"], -nruns.min()))
    total = int(offs[-1])
    if total > out["packed"].numel():
        raise RuntimeError("sm_rle_encode: packed capacity %d < %d" % (out["packed"].numel(), total))
    blob = out["packed"][:total].cpu().numpy().tobytes()
    per_img = isinstance(canvas_hw[0], (tuple, list))             # one canvas per image
    res = []
    for b in range(batch):
        size = [int(v) for v in (canvas_hw[b] if per_img else canvas_hw)[:2]]
        res.append([dict(size=list(size), counts=blob[offs[b * max_num + i]:offs[b * max_num + i + 1]])
                    for i in range(int(ndet[b]))])
    return res


# ------------------------------------------------------------------------------- fused mask loss (training)
def fcos_target(points, pstride, lo, hi, gt_boxes, gt_labels, ngt, center_sampling, radius):
    """sm_fcos_target: gt_boxes [B,gmax,4] f32, gt_labels [B,gmax] i64, ngt [B] i32 -> labels [B,S] i64, targets [B,S,4],
    gt_index [B,S] i32"""
    lib = _lib.load()
    _lib.require_cuda(points, gt_boxes)
    b, gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    s = points.shape[0]
    labels = torch.empty(b, s, dtype=torch.int64, device=points.device)
    targets = torch.empty(b, s, 4, dtype=torch.float32, device=points.device)
    gidx = torch.empty(b, s, dtype=torch.int32, device=points.device)
    _lib.check(lib.sm_fcos_target(_lib.ptr(points), _lib.ptr(pstride), _lib.ptr(lo), _lib.ptr(hi), _lib.ptr(gt_boxes),
                                  _lib.ptr(gt_labels), _lib.ptr(ngt), b, s, gmax, int(bool(center_sampling)), float(radius),
                                  _lib.ptr(labels), _lib.ptr(targets), _lib.ptr(gidx), _lib.stream_ptr()), "sm_fcos_target")
    return labels, targets, gidx


def mask_loss_fwd(basis, cof, rois, gt, idx_gt, out):
    lib = _lib.load()
    _lib.check(lib.sm_mask_loss_fwd(_lib.ptr(basis), 0, _lib.ptr(cof), _lib.ptr(rois), _lib.ptr(gt), _lib.ptr(idx_gt),
                                    cof.shape[0], basis.shape[1], basis.shape[2], _lib.ptr(out), _lib.stream_ptr()),
               "sm_mask_loss_fwd")


def mask_loss_bwd(basis, cof, rois, gt, idx_gt, grad_sum, grad_cof, grad_basis):
    lib = _lib.load()
    _lib.check(lib.sm_mask_loss_bwd(_lib.ptr(basis), 0, _lib.ptr(cof), _lib.ptr(rois), _lib.ptr(gt), _lib.ptr(idx_gt),
                                    cof.shape[0], basis.shape[1], basis.shape[2], _lib.ptr(grad_sum),
                                    _lib.ptr(grad_cof), _lib.ptr(grad_basis), _lib.stream_ptr()), "sm_mask_loss_bwd")


# ------------------------------------------------------------------------------- VIS tracking
def track_gather(track_feats, det, ndet, h, w, box_mul, out, stride=8.0):
    """track_feats f32 [B*h*w, C] (NHWC rows); det [B, max_num, 5]; out [B, max_num, C]."""
    lib = _lib.load()
    b, n, c = det.shape[0], det.shape[1], track_feats.shape[-1]
    _lib.check(lib.sm_track_gather(_lib.ptr(track_feats), _lib.ptr(det), _lib.ptr(ndet), b, n, h, w, c, float(box_mul),
                                   float(stride), _lib.ptr(out), _lib.stream_ptr()), "sm_track_gather")
    return out


def track_match(det_feats, prev_feats, det, det_labels, prev_boxes, prev_labels, coeff):
    """-> (comp [n, t+1], match_id int32 [n], match_score [n])"""
    lib = _lib.load()
    n, c = det_feats.shape
    t = prev_feats.shape[0]
    comp = torch.empty(n, t + 1, dtype=torch.float32, device=det_feats.device)
    mid = torch.zeros(n, dtype=torch.int32, device=det_feats.device)
    msc = torch.zeros(n, dtype=torch.float32, device=det_feats.device)
    _lib.check(lib.sm_track_match(_lib.ptr(det_feats), _lib.ptr(prev_feats), _lib.ptr(det), _lib.ptr(det_labels),
                                  _lib.ptr(prev_boxes), _lib.ptr(prev_labels), n, t, c, float(coeff[0]),
                                  float(coeff[1]), float(coeff[2]), _lib.ptr(comp), _lib.ptr(mid), _lib.ptr(msc),
                                  _lib.stream_ptr()), "sm_track_match")
    return comp, mid, msc


def track_state_alloc(channels, max_num, device, capacity=1024):
    """object memory of the device tracker (sm_track_clip): embeddings, boxes (x1,y1,x2,y2,score), labels, count"""
    return dict(feats=torch.zeros(capacity, channels, dtype=torch.float32, device=device),
                boxes=torch.zeros(capacity, 5, dtype=torch.float32, device=device),
                labels=torch.zeros(capacity, dtype=torch.int64, device=device),
                count=torch.zeros(1, dtype=torch.int32, device=device),
                comp=torch.empty(max_num, capacity + 1, dtype=torch.float32, device=device), capacity=capacity, max_num=max_num)


def track_clip(det_feats, det, det_labels, ndet, is_first, coeff, state, ids=None):
    """identity assignment of T frames in order on the device; det_feats [T,max,C], det [T,max,5], det_labels [T,max] i64,
    ndet / is_first i32 [T] device tensors -> ids i32 [T,max] (device; -1 beyond ndet / lost claims)"""
    _lib.require_cuda(det_feats, det, det_labels, ndet, is_first)
    T, mx, c = det_feats.shape
    if mx > state["max_num"]:
        raise ValueError("tracker state was allocated for %d detections per frame, got %d" % (state["max_num"], mx))
    if ids is None:
        ids = torch.empty(T, mx, dtype=torch.int32, device=det_feats.device)
    _lib.check(_lib.load().sm_track_clip(_lib.ptr(det_feats), _lib.ptr(det), _lib.ptr(det_labels), _lib.ptr(ndet),
                                         _lib.ptr(is_first), T, mx, c, float(coeff[0]), float(coeff[1]), float(coeff[2]),
                                         _lib.ptr(state["feats"]), _lib.ptr(state["boxes"]), _lib.ptr(state["labels"]),
                                         _lib.ptr(state["count"]), state["capacity"], _lib.ptr(state["comp"]), _lib.ptr(ids),
                                         _lib.stream_ptr()), "sm_track_clip")
    return ids


def mask_rescore(feat, labels, det, ndet, hw, out):
    """feat f32 [B*max_num*hw, C]; labels [B, max_num]; det [B, max_num, 5]; out f32 [B, max_num]."""
    lib = _lib.load()
    b, n = det.shape[0], det.shape[1]
    _lib.check(lib.sm_mask_rescore(_lib.ptr(feat), _lib.ptr(labels), _lib.ptr(det), _lib.ptr(ndet), b, n, int(hw),
                                   feat.shape[-1], _lib.ptr(out), _lib.stream_ptr()), "sm_mask_rescore")
    return out


# ------------------------------------------------------------------------------- training-path NCHW layers
def groupnorm_nchw_fwd(x, gamma, beta, groups, eps, relu):
    lib = _lib.load()
    b, c = x.shape[0], x.shape[1]
    hw = x.numel() // (b * c)
    y = torch.empty_like(x)
    stats = torch.empty(b, groups, 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.sm_groupnorm_nchw_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(stats), b, c,
                                         hw, groups, float(eps), int(relu), _lib.stream_ptr()), "sm_groupnorm_nchw_fwd")
    return y, stats


def groupnorm_nchw_bwd(x, y, dy, gamma, stats, groups, relu, need_dx=True, need_dw=True):
    lib = _lib.load()
    b, c = x.shape[0], x.shape[1]
    hw = x.numel() // (b * c)
    dx = torch.empty_like(x) if need_dx else None
    dg = torch.empty(c, dtype=torch.float32, device=x.device) if need_dw else None
    db = torch.empty(c, dtype=torch.float32, device=x.device) if need_dw else None
    scratch = torch.empty(b, groups, 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.sm_groupnorm_nchw_bwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(dy), _lib.ptr(gamma), _lib.ptr(stats),
                                         _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(scratch), b, c, hw, groups,
                                         int(relu), _lib.stream_ptr()), "sm_groupnorm_nchw_bwd")
    return dx, dg, db


def upsample_nchw(x, factor, backward=False):
    """x f32 [N,C,h,w] -> [N,C,h*f,w*f]; backward=True: x is dy [N,C,h*f,w*f] -> dx [N,C,h,w]."""
    lib = _lib.load()
    n, c = x.shape[0], x.shape[1]
    if not backward:
        h, w = x.shape[2], x.shape[3]
        y = torch.empty(n, c, h * factor, w * factor, dtype=torch.float32, device=x.device)
        _lib.check(lib.sm_upsample_bilinear_nchw_fwd(_lib.ptr(x), _lib.ptr(y), n * c, h, w, factor, _lib.stream_ptr()),
                   "sm_upsample_bilinear_nchw_fwd")
        return y
    h, w = x.shape[2] // factor, x.shape[3] // factor
    dx = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    _lib.check(lib.sm_upsample_bilinear_nchw_bwd(_lib.ptr(x), _lib.ptr(dx), n * c, h, w, factor, _lib.stream_ptr()),
               "sm_upsample_bilinear_nchw_bwd")
    return dx


def sgd_step(param, grad, buf, lr, momentum, weight_decay, first_step):
    lib = _lib.load()
    _lib.check(lib.sm_sgd_step(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(buf), param.numel(), float(lr), float(momentum),
                               float(weight_decay), int(first_step), _lib.stream_ptr()), "sm_sgd_step")
    # the kernel wrote through a raw pointer: tell autograd (and the launch-plan cache, which fingerprints
    # (data_ptr, _version) of every weight -- plan_cache.py) that the tensor changed
    torch.autograd.graph.increment_version(param)


# ------------------------------------------------------------------------------- training graph on row tensors
def _weight_prep_geometry(w, mode, cin_pad):
    co, ci, kh, kw = w.shape
    if mode == 0:
        cin_pad = cin_pad or ((ci + 7) // 8 * 8)
        rows, k = co, kh * kw * cin_pad
    elif mode == 1:
        cin_pad = co
        rows, k = ci, kh * kw * co
    else:
        cin_pad = co
        rows, k = kh * kw * ci, co
    tile = cout_tile(rows)
    return (rows + tile - 1) // tile * tile, (k + 63) // 64 * 64, cin_pad


def _weight_prep_launch(w, scale, mode, out, rows_pad, kp, cin_pad):
    lib = _lib.load()
    co, ci, kh, kw = w.shape
    wc = w.detach()
    if wc.dtype != torch.float32 or not wc.is_contiguous():
        wc = wc.float().contiguous()
    _lib.check(lib.sm_weight_prep(_lib.ptr(wc), _lib.ptr(scale), co, ci, kh, kw, mode, _lib.ptr(out), rows_pad, kp, cin_pad,
                                  _lib.stream_ptr()), "sm_weight_prep")


class _WPItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("out", C.c_void_p), ("co", C.c_int32), ("ci", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("mode", C.c_int32), ("kp", C.c_int32), ("cin_pad", C.c_int32),
                ("tc", C.c_int32), ("tiles_x", C.c_int32), ("pad", C.c_int32)]


class _WeightPrepCache:
    """bf16 operand layouts of the conv PARAMETERS of a training step, kept across steps: the buffers are allocated
    (zeroed: the padding never changes) once per (parameter, layout), and `refresh()` -- called at the start of a step --
    rewrites all of them whose parameter changed since (optimizer step, checkpoint load: `_version`) in ONE launch
    (sm_weight_prep_multi) instead of ~130 launches + memsets spread over forward and backward (2.3 ms of a 34 ms step).
    Temporaries (concatenated / folded weights) are not cached: they go through the single launch."""

    def __init__(self):
        self.entries = {}
        self.table = None          # (keys tuple, items tensor, blocks tensor, nblocks, lds)

    @staticmethod
    def _ver(e):
        w, sc = e["w"](), e["scale"]
        return None if w is None else (w._version, w.data_ptr(), None if sc is None else (sc._version, sc.data_ptr()))

    def get(self, w, scale, mode, cin_pad):
        import weakref
        rows_pad, kp, cin_pad = _weight_prep_geometry(w, mode, cin_pad)
        if not isinstance(w, torch.nn.Parameter) or w.dtype != torch.float32 or not w.is_contiguous():
            out = torch.empty(rows_pad, kp, dtype=BF16, device=w.device)
            _weight_prep_launch(w, scale, mode, out, rows_pad, kp, cin_pad)
            return out, rows_pad
        key = (id(w), mode, cin_pad, 0 if scale is None else id(scale))
        e = self.entries.get(key)
        if e is None or e["w"]() is not w:
            e = dict(w=weakref.ref(w), scale=scale, mode=mode, cin_pad=cin_pad, rows_pad=rows_pad, kp=kp,
                     out=torch.zeros(rows_pad, kp, dtype=BF16, device=w.device), ver=None)
            self.entries[key] = e
            self.table = None
        v = self._ver(e)
        if e["ver"] != v:
            _weight_prep_launch(w, scale, mode, e["out"], rows_pad, kp, cin_pad)
            e["ver"] = v
        return e["out"], rows_pad

    def refresh(self):
        dead = [k for k, e in self.entries.items() if e["w"]() is None]
        for k in dead:
            del self.entries[k]
            self.table = None
        stale = [(k, e) for k, e in self.entries.items() if e["ver"] != self._ver(e)]
        if not stale:
            return 0
        keys = tuple(k for k, _ in stale)
        ptrs = tuple((e["w"]().data_ptr(), 0 if e["scale"] is None else e["scale"].data_ptr()) for _, e in stale)
        if self.table is None or self.table[0] != (keys, ptrs):
            tab = (_WPItem * len(stale))()
            blocks, lds = [], 0
            for i, (_, e) in enumerate(stale):
                w = e["w"]()
                co, ci, kh, kw = w.shape
                tc = 32 if kh * kw <= 9 else 8
                t = tab[i]
                t.w, t.scale, t.out = w.data_ptr(), (0 if e["scale"] is None else e["scale"].data_ptr()), e["out"].data_ptr()
                t.co, t.ci, t.kh, t.kw, t.mode, t.kp, t.cin_pad, t.tc = co, ci, kh, kw, e["mode"], e["kp"], e["cin_pad"], tc
                t.tiles_x = (ci + tc - 1) // tc
                blocks.extend((i, j) for j in range(t.tiles_x * ((co + 31) // 32)))
                lds = max(lds, 4 * 32 * (tc * kh * kw + 1))
            dev = stale[0][1]["out"].device
            items = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            blk = torch.tensor(blocks, dtype=torch.int32, device=dev).contiguous()
            self.table = ((keys, ptrs), items, blk, len(blocks), lds)
        _, items, blk, nb, lds = self.table
        _lib.check(_lib.load().sm_weight_prep_multi(_lib.ptr(items), _lib.ptr(blk), nb, lds, _lib.stream_ptr()),
                   "sm_weight_prep_multi")
        for _, e in stale:
            e["ver"] = self._ver(e)
        return len(stale)


WEIGHT_PREP_CACHE = _WeightPrepCache()


def weight_prep(w, scale, mode, cin_pad=None):
    """f32 OIHW parameter (x scale[cout]) -> bf16 operand of sm_conv2d (mode 0) / the dX conv (mode 1) / the grad-column
    GEMM (mode 2) -- the layouts of prep_conv_weight applied to w, w.flip(2,3).permute(1,0,2,3) and
    w.permute(2,3,1,0).reshape(K,co,1,1).  Returns (tensor, rows_pad).  Parameters are served from WEIGHT_PREP_CACHE (the
    returned tensor is then a persistent buffer: do not write to it)."""
    return WEIGHT_PREP_CACHE.get(w, scale, mode, cin_pad)


class _GradSink:
    """Where parameter gradients go when they live in all-reduce buckets (dist_train.GradBucketer): the kernels that
    produce a parameter gradient (sm_wgrad_finish, sm_bias_grad_rows, sm_gn_bwd_rows) write it STRAIGHT into the
    parameter's slice of the flat bucket instead of into a fresh tensor that autograd would then add or copy there --
    what torch DDP's gradient_as_bucket_view buys MMDistributedDataParallel (M/mmdet/apis/train.py:135-139), minus
    its copy-in.  Keys are the parameters' data_ptr()s (saved tensors come back as other Python objects).

    Direct writes OVERWRITE, so they are only safe for a parameter that one backward op uses once per step.  The first
    step after attach() is a census: every op reports its use and returns its gradient the ordinary way (autograd adds
    it into the zeroed view); parameters counted once become direct from the second step on.  A second contribution in
    a later step still goes through autograd's in-place add.  Readiness of a gradient is NOT signalled from here: the
    bucketer counts a parameter when autograd's AccumulateGrad node for it has run (its post-accumulate hook), which
    is after every op that uses the parameter -- direct writers included."""

    def __init__(self):
        self.detach()

    def detach(self, owner=None):
        """owner None: forget everything; else only the views that `owner` (a GradBucketer) attached -- a second bucketer's
        attach() or remove() must not take the first one's views away"""
        if owner is None or not getattr(self, "owners", None):
            self.views, self.uses, self.direct, self.written, self.census, self.owners = {}, {}, set(), set(), True, {}
            return
        for k in self.owners.pop(id(owner), ()):
            self.views.pop(k, None)
            self.uses.pop(k, None)
            self.direct.discard(k)
            self.written.discard(k)

    def attach(self, views, owner=None):
        """register `views` {param.data_ptr(): gradient view}; views of other owners stay (a fresh census covers all)"""
        views = dict(views)
        if owner is None:
            self.detach()
        else:
            self.detach(owner)
            self.owners[id(owner)] = list(views.keys())
        self.views.update(views)
        self.uses, self.direct, self.census = {}, set(), True

    def begin_step(self):
        if self.census and self.uses:
            self.direct = {k for k, c in self.uses.items() if c == 1}
            self.census = False
        self.written.clear()

    def target(self, param):
        """the view to write `param`'s gradient into, or None (= return the gradient to autograd)"""
        if not self.views or param is None:
            return None
        k = param.data_ptr()
        v = self.views.get(k)
        if v is None:
            return None
        if self.census:
            self.uses[k] = self.uses.get(k, 0) + 1
            return None
        if k in self.direct and k not in self.written:
            return v
        return None

    def commit(self, param):
        self.written.add(param.data_ptr())


GRAD_SINK = _GradSink()


def wgrad_finish(gw_t, scale, co, ci, kh, kw, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty(co, ci, kh, kw, dtype=torch.float32, device=gw_t.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == co * ci * kh * kw
    _lib.check(lib.sm_wgrad_finish(_lib.ptr(gw_t), _lib.ptr(scale), co, ci, kh, kw, _lib.ptr(out), _lib.stream_ptr()),
               "sm_wgrad_finish")
    return out


def relu_bwd_bf16(g, y):
    lib = _lib.load()
    assert g.dtype == BF16 and y.dtype == BF16 and g.numel() == y.numel() and g.is_contiguous() and y.is_contiguous()
    out = torch.empty_like(g)
    _lib.check(lib.sm_relu_bwd_bf16(_lib.ptr(g), _lib.ptr(y), _lib.ptr(out), g.numel(), _lib.stream_ptr()), "sm_relu_bwd_bf16")
    return out


def bias_grad_rows(g, channels, out=None):
    lib = _lib.load()
    assert g.dtype == BF16 and g.dim() == 2 and g.is_contiguous()
    if out is None:
        out = torch.empty(channels, dtype=torch.float32, device=g.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == channels
    _lib.check(lib.sm_bias_grad_rows(_lib.ptr(g), g.shape[0], g.shape[1], channels, _lib.ptr(out), _lib.stream_ptr()),
               "sm_bias_grad_rows")
    return out


def gn_bwd_rows(x, dy, gamma, beta, stats, lv, channels, groups, eps, relu, dg=None, db=None):
    _check_gn_stats(stats)
    lib = _lib.load()
    nlev = len(lv)
    hw = (C.c_int32 * nlev)(*[h * w for h, w in lv.sizes])
    row0 = (C.c_int64 * nlev)(*lv.row0)
    dx = torch.empty_like(x)
    dg = torch.empty(channels, dtype=torch.float32, device=x.device) if dg is None else dg
    db = torch.empty(channels, dtype=torch.float32, device=x.device) if db is None else db
    bins = torch.empty(lv.batch * nlev * groups * 2, dtype=torch.float32, device=x.device)
    _lib.check(lib.sm_gn_bwd_rows(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats), lv.batch, nlev,
                                  hw, row0, channels, groups, eps, int(relu), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db),
                                  _lib.ptr(bins), _lib.stream_ptr()), "sm_gn_bwd_rows")
    return dx, dg, db


def upsample_bilinear_bwd_rows(gout, out_cstride, out_coff, batch, h, w, c, factor):
    lib = _lib.load()
    gin = torch.empty(batch * h * w, c, dtype=BF16, device=gout.device)
    _lib.check(lib.sm_upsample_bilinear_bwd_rows(_lib.ptr(gout), out_cstride, out_coff, batch, h, w, c, factor, _lib.ptr(gin),
                                                 _lib.stream_ptr()), "sm_upsample_bilinear_bwd_rows")
    return gin


def nearest_bwd_rows(g_fine, batch, fine_hw, coarse_hw, c):
    lib = _lib.load()
    out = torch.empty(batch * coarse_hw[0] * coarse_hw[1], c, dtype=BF16, device=g_fine.device)
    _lib.check(lib.sm_nearest_bwd_rows(_lib.ptr(g_fine), batch, fine_hw[0], fine_hw[1], coarse_hw[0], coarse_hw[1], c,
                                       _lib.ptr(out), _lib.stream_ptr()), "sm_nearest_bwd_rows")
    return out


def scatter_stride_rows(t, batch, hw, out_hw, stride, c):
    lib = _lib.load()
    out = torch.empty(batch * hw[0] * hw[1], c, dtype=BF16, device=t.device)
    _lib.check(lib.sm_scatter_stride_rows(_lib.ptr(t), batch, hw[0], hw[1], out_hw[0], out_hw[1], stride, c, _lib.ptr(out),
                                          _lib.stream_ptr()), "sm_scatter_stride_rows")
    return out
