"""Backward of DeformConv / ModulatedDeformConv (deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda,
modulated_deform_conv_cuda_backward; mmdet/ops/dcn/src/deform_conv_cuda.cpp:262-488, 592-685) in the column
formulation: HIP sampling kernels (csrc/orp_dcn_bwd.hip) around two library GEMMs."""
import torch

from .. import _lib


def _geo(input, weight, stride, padding, dilation):
    B, C, H, W = input.shape
    kh, kw = weight.size(2), weight.size(3)
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    return B, C, H, W, kh, kw, Ho, Wo


def _im2col(input, offset, mask, weight, stride, padding, dilation, dg):
    B, C, H, W, kh, kw, Ho, Wo = _geo(input, weight, stride, padding, dilation)
    col = torch.empty((C * kh * kw, B * Ho * Wo), dtype=torch.float32, device=input.device)
    with torch.cuda.device(input.device):
        rc = _lib.lib().orp_dcn_im2col(_lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), B, C, H, W, kh, kw, stride[0],
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
    with torch.cuda.device(input.device):
        rc = _lib.lib().orp_dcn_col2im(_lib.ptr(gcol), _lib.ptr(input), _lib.ptr(offset), _lib.ptr(mask), B, C, H, W, kh,
                                       kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], dg,
                                       _lib.ptr(grad_input), _lib.ptr(grad_offset), _lib.ptr(grad_mask),
                                       _lib.stream_of(input))
    _lib.check(rc, "orp_dcn_col2im")
    return grad_input, grad_offset, grad_mask


def _prep(*ts):
    return [t.detach().float().contiguous() if t is not None else None for t in ts]


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
    input, offset, weight, grad_output = _prep(input, offset, weight, grad_output)
    if groups == 1 and deformable_groups == 1:
        return _backward_input_nhwc(input, offset, weight, grad_output, stride, padding, dilation)
    gcol = _grad_columns(weight, grad_output, groups)
    gi, go, _ = _col2im(gcol, input, offset, None, weight, stride, padding, dilation, deformable_groups)
    return gi, go


def backward_parameters(input, offset, weight, grad_output, stride, padding, dilation, groups, deformable_groups):
    input, offset, weight, grad_output = _prep(input, offset, weight, grad_output)
    col = _im2col(input, offset, None, weight, stride, padding, dilation, deformable_groups)
    return _grad_weight(col, weight, grad_output, groups)


def _grad_weight(col, weight, grad_output, groups):
    B, Cout, Ho, Wo = grad_output.shape
    go = grad_output.permute(1, 0, 2, 3).reshape(groups, Cout // groups, B * Ho * Wo)
    colg = col.reshape(groups, -1, B * Ho * Wo)
    return torch.bmm(go, colg.transpose(1, 2)).reshape(weight.shape)


def modulated_backward(input, offset, mask, weight, grad_output, stride, padding, dilation, groups, deformable_groups,
                       with_bias):
    input, offset, mask, weight, grad_output = _prep(input, offset, mask, weight, grad_output)
    gcol = _grad_columns(weight, grad_output, groups)
    gi, go, gm = _col2im(gcol, input, offset, mask, weight, stride, padding, dilation, deformable_groups)
    col = _im2col(input, offset, mask, weight, stride, padding, dilation, deformable_groups)
    gw = _grad_weight(col, weight, grad_output, groups)
    gb = grad_output.sum(dim=(0, 2, 3)) if with_bias else None
    return gi, go, gm, gw, gb
