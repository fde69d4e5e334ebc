"""Backward of DeformConv / ModulatedDeformConv (deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda,
modulated_deform_conv_cuda_backward; mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-488, 592-685).

The head's configuration (256 -> 256 channels, groups = deformable_groups = 1) runs as two MFMA implicit GEMMs without any
column buffer (csrc/orp_dcn_bwd_mfma.hip, `backward_mfma`); every other configuration uses the column formulation: HIP
sampling kernels (csrc/orp_dcn_bwd.hip) around two library GEMMs."""
import ctypes

import torch

from .. import _lib


class _BwdLevel(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("grad_output", ctypes.c_void_p),
                ("grad_input", ctypes.c_void_p), ("grad_offset", ctypes.c_void_p), ("height", ctypes.c_int),
                ("width", ctypes.c_int)]


USE_MFMA = True          # False: force the column formulation (comparisons in tests / tools)


def mfma_ok(weight, groups, deformable_groups):
    if not USE_MFMA:
        return False
    cout, cin_g, kh, kw = weight.shape
    return bool(_lib.lib().orp_dcn_backward_mfma_ok(cin_g * groups, cout, kh, kw, groups, deformable_groups))


_IO_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def backward_mfma(inputs, offsets, weight, grad_outputs, stride, padding, dilation, need_input=True, need_weight=True,
                  sparse_grad=False, masks=None):
    """All levels of ONE DeformConv layer in one call: lists of NCHW inputs / offsets / grad_outputs ->
    (grad_inputs, grad_offsets, grad_weight) -- with `masks` (DCNv2 modulation, one [B,kh*kw,Ho,Wo] per level)
    (grad_inputs, grad_offsets, grad_weight, grad_masks); grad_weight is summed over the levels in a fixed order.
    fp16 / bf16 inputs (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF branch): inputs, grad_outputs and grad_inputs
    travel in that type and are converted inside the layout passes of the call; the arithmetic is fp32 MFMA, the small
    tensors (offsets, masks, weight and their gradients) are converted here.
    sparse_grad: the caller expects grad_outputs to be zero almost everywhere (ORP_DCN_BWD_SPARSE, include/orp_hip.h):
    grad_input is then scattered with atomics instead of the (bitwise reproducible) region pass."""
    L = _lib.lib()
    dt = inputs[0].dtype
    if dt not in _IO_CODES:
        raise TypeError("backward_mfma: fp32 / fp16 / bf16 tensors only")
    w = weight.detach().float().contiguous()
    cout, cin, kh, kw = w.shape
    B = inputs[0].size(0)
    n = len(inputs)
    levels = (_BwdLevel * n)()
    mptr = (ctypes.c_void_p * n)() if masks is not None else None
    gmptr = (ctypes.c_void_p * n)() if masks is not None else None
    keep, gis, gos, gms = [], [], [], []
    for i, (x, off, go) in enumerate(zip(inputs, offsets, grad_outputs)):
        x, go = x.detach().to(dt).contiguous(), go.detach().to(dt).contiguous()
        off = off.detach().float().contiguous()
        gi = torch.empty_like(x) if need_input else None
        goff = torch.empty_like(off) if need_input else None
        keep.append((x, off, go))
        gis.append(gi)
        gos.append(goff)
        levels[i] = _BwdLevel(_lib.ptr(x), _lib.ptr(off), _lib.ptr(go), _lib.ptr(gi), _lib.ptr(goff), x.size(2), x.size(3))
        if masks is not None:
            m = masks[i].detach().float().contiguous()
            gm = torch.empty_like(m) if need_input else None
            keep.append(m); gms.append(gm)
            mptr[i] = m.data_ptr()
            gmptr[i] = gm.data_ptr() if gm is not None else None
    gw = torch.empty_like(w) if need_weight else None
    nbytes = L.orp_dcn_backward_workspace_bytes(levels, n, B, kh, kw, stride[0], stride[1], padding[0], padding[1],
                                                dilation[0], dilation[1])
    if nbytes == 0:
        raise ValueError("orp_dcn_backward_workspace_bytes: invalid geometry")
    with torch.cuda.device(w.device):
        ws = _lib.workspace(w.device, nbytes)
        rc = L.orp_dcn_backward_multi_ex(levels, mptr, gmptr, _IO_CODES[dt], n, B, cin, cout, _lib.ptr(w), _lib.ptr(gw),
                                         (1 | (2 if sparse_grad else 0)) if need_input else 0, kh, kw,
                                         stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                         _lib.ptr(ws), nbytes, _lib.stream_of(w))
    _lib.check(rc, "orp_dcn_backward_multi_ex")
    if dt != torch.float32:                                        # the small gradients follow their tensors' type
        gos = [g.to(offsets[i].dtype) if g is not None else None for i, g in enumerate(gos)]
        gms = [g.to(masks[i].dtype) if g is not None else None for i, g in enumerate(gms)]
        gw = gw.to(weight.dtype) if gw is not None else None
    return (gis, gos, gw, gms) if masks is not None else (gis, gos, gw)


def _geo(input, weight, stride, padding, dilation):
    B, C, H, W = input.shape
    kh, kw = weight.size(2), weight.size(3)
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    return B, C, H, W, kh, kw, Ho, Wo


def _im2col(input, offset, mask, weight, stride, padding, dilation, dg):
    B, C, H, W, kh, kw, Ho, Wo = _geo(input, weight, stride, padding, dilation)
    col = torch.empty((C * kh * kw, B * Ho * Wo), dtype=input.dtype, device=input.device)
    fn = _lib.lib().orp_dcn_im2col_f64 if input.dtype == torch.float64 else _lib.lib().orp_dcn_im2col
    with torch.cuda.device(input.device):
        rc = fn(_lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), B, C, H, W, kh, kw, stride[0],
                                       stride[1], padding[0], padding[1], dilation[0], dilation[1], dg, _lib.ptr(col),
                                       _lib.stream_of(input))
    _lib.check(rc, "orp_dcn_im2col")
    return col


def _grad_columns(weight, grad_output, groups):
    """W^T . grad_out per group -> [C*taps, B*Ho*Wo]."""
    B, Cout, Ho, Wo = grad_output.shape
    go = grad_output.permute(1, 0, 2, 3).reshape(groups, Cout // groups, B * Ho * Wo)
    w = weight.reshape(groups, Cout // groups, -1)
    return torch.bmm(w.transpose(1, 2), go).reshape(-1, B * Ho * Wo).contiguous()


def _col2im(gcol, input, offset, mask, weight, stride, padding, dilation, dg):
    B, C, H, W, kh, kw, Ho, Wo = _geo(input, weight, stride, padding, dilation)
    grad_input = torch.zeros_like(input)
    grad_offset = torch.empty_like(offset)
    grad_mask = torch.empty_like(mask) if mask is not None else None
    fn = _lib.lib().orp_dcn_col2im_f64 if input.dtype == torch.float64 else _lib.lib().orp_dcn_col2im
    with torch.cuda.device(input.device):
        rc = fn(_lib.ptr(gcol), _lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), B, C, H, W, kh,
                                       kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], dg,
                                       _lib.ptr(grad_input), _lib.ptr(grad_offset), _lib.ptr(grad_mask),
                                       _lib.stream_of(input))
    _lib.check(rc, "orp_dcn_col2im")
    return grad_input, grad_offset, grad_mask


def is_f64(*ts):
    """float64 tensors take the `double` branch of the reference's dispatch: the column formulation in double."""
    return any(t is not None and t.dtype == torch.float64 for t in ts)


def _prep(*ts):
    dt = torch.float64 if is_f64(*ts) else torch.float32
    return [t.detach().to(dt).contiguous() if t is not None else None for t in ts]


def forward_columns(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups):
    """The reference's own forward formulation (deform_conv_cuda.cpp:152-260, 490-567): columns = deformable_im2col (x mask),
    out = W . columns per group (+ bias) -- used for float64 tensors (sampling kernel here, double GEMM from the library)."""
    x, off, m, w, b = _prep(input, offset, mask, weight, bias)
    B, C, H, W, kh, kw, Ho, Wo = _geo(x, w, stride, padding, dilation)
    Cout = w.size(0)
    col = _im2col(x, off, m, w, stride, padding, dilation, deformable_groups)
    out = torch.bmm(w.reshape(groups, Cout // groups, -1), col.reshape(groups, -1, B * Ho * Wo))
    out = out.reshape(Cout, B, Ho, Wo).permute(1, 0, 2, 3).contiguous()
    if b is not None:
        out += b[None, :, None, None]
    return out


def _backward_input_nhwc(input, offset, weight, grad_output, stride, padding, dilation):
    """groups = deformable_groups = 1: position-major grad columns (one GEMM on NHWC grad_out) + the channel-parallel
    col2im kernel (coalesced rows, wave-reduced grad_offset)."""
    B, C, H, W, kh, kw, Ho, Wo = _geo(input, weight, stride, padding, dilation)
    Cout = weight.size(0)
    go = grad_output.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)                  # NHWC rows (view or one copy)
    w2 = weight.permute(0, 2, 3, 1).reshape(Cout, kh * kw * C)                       # [o][tap][c]
    gcol_t = torch.mm(go, w2)                                                        # [B*P, taps*C]
    x_nhwc = input.permute(0, 2, 3, 1).contiguous()
    gx_nhwc = torch.zeros_like(x_nhwc)
    grad_offset = torch.empty_like(offset)
    with torch.cuda.device(input.device):
        rc = _lib.lib().orp_dcn_col2im_nhwc(_lib.ptr(gcol_t), _lib.ptr(x_nhwc), _lib.ptr(offset), B, C, H, W, kh, kw,
                                            stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1],
                                            _lib.ptr(gx_nhwc), _lib.ptr(grad_offset), _lib.stream_of(input))
    _lib.check(rc, "orp_dcn_col2im_nhwc")
    return gx_nhwc.permute(0, 3, 1, 2).contiguous(), grad_offset


def backward_input(input, offset, weight, grad_output, stride, padding, dilation, groups, deformable_groups):
    if is_f64(input, offset, weight, grad_output):
        dt, dto = input.dtype, offset.dtype
        x, off, w, go = _prep(input, offset, weight, grad_output)
        gcol = _grad_columns(w, go, groups)
        gi, goff, _ = _col2im(gcol, x, off, None, w, stride, padding, dilation, deformable_groups)
        return gi.to(dt), goff.to(dto)
    if mfma_ok(weight, groups, deformable_groups):
        gi, go, _ = backward_mfma([input], [offset], weight, [grad_output], stride, padding, dilation, True, False)
        return gi[0], go[0]
    dt, dto = input.dtype, offset.dtype
    input, offset, weight, grad_output = _prep(input, offset, weight, grad_output)
    if groups == 1 and deformable_groups == 1:
        gi, go = _backward_input_nhwc(input, offset, weight, grad_output, stride, padding, dilation)
    else:
        gcol = _grad_columns(weight, grad_output, groups)
        gi, go, _ = _col2im(gcol, input, offset, None, weight, stride, padding, dilation, deformable_groups)
    return gi.to(dt), go.to(dto)


def backward_parameters(input, offset, weight, grad_output, stride, padding, dilation, groups, deformable_groups):
    if not is_f64(input, offset, weight, grad_output) and mfma_ok(weight, groups, deformable_groups):
        return backward_mfma([input], [offset], weight, [grad_output], stride, padding, dilation, False, True)[2]
    dt = weight.dtype
    input, offset, weight, grad_output = _prep(input, offset, weight, grad_output)
    col = _im2col(input, offset, None, weight, stride, padding, dilation, deformable_groups)
    return _grad_weight(col, weight, grad_output, groups).to(dt)


def _grad_weight(col, weight, grad_output, groups):
    B, Cout, Ho, Wo = grad_output.shape
    go = grad_output.permute(1, 0, 2, 3).reshape(groups, Cout // groups, B * Ho * Wo)
    colg = col.reshape(groups, -1, B * Ho * Wo)
    return torch.bmm(go, colg.transpose(1, 2)).reshape(weight.shape)


def modulated_backward(input, offset, mask, weight, grad_output, stride, padding, dilation, groups, deformable_groups,
                       with_bias):
    if not is_f64(input, offset, mask, weight, grad_output) and mfma_ok(weight, groups, deformable_groups):
        # DCNv2 on the MFMA implicit GEMMs (orp_dcn_backward_multi_ex): the modulation rides in the sample weights
        gis, gos, gw, gms = backward_mfma([input], [offset], weight, [grad_output], stride, padding, dilation, True, True,
                                          masks=[mask])
        gb = grad_output.float().sum(dim=(0, 2, 3)).to(grad_output.dtype) if with_bias else None
        return gis[0], gos[0], gms[0], gw, gb
    dt = input.dtype
    input, offset, mask, weight, grad_output = _prep(input, offset, mask, weight, grad_output)
    gcol = _grad_columns(weight, grad_output, groups)
    gi, go, gm = _col2im(gcol, input, offset, mask, weight, stride, padding, dilation, deformable_groups)
    col = _im2col(input, offset, mask, weight, stride, padding, dilation, deformable_groups)
    gw = _grad_weight(col, weight, grad_output, groups)
    gb = grad_output.sum(dim=(0, 2, 3)) if with_bias else None
    if dt != input.dtype:
        gi, go, gm, gw = gi.to(dt), go.to(dt), gm.to(dt), gw.to(dt)
        gb = gb.to(dt) if gb is not None else None
    return gi, go, gm, gw, gb
