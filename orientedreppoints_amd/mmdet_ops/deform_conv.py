"""Mirror of mmdet/ops/dcn/deform_conv.py (DeformConvFunction :14-112, ModulatedDeformConvFunction :115-190,
DeformConv :197-255, DeformConvPack :258-330, ModulatedDeformConv :333-380, ModulatedDeformConvPack :383-440) on the
MI355X HIP library.

What changes under the same API: the forward is an MFMA implicit GEMM with no HBM `columns` buffer
(csrc/orp_dcn.hip); `deform_conv_multi` applies one layer to ALL FPN levels in a single launch (the head calls the
same DeformConv on 5 levels).  Channels-last (NHWC) inputs are consumed in place and produce channels-last outputs;
NCHW inputs are converted through a workspace and produce NCHW outputs.
"""
import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from .. import _lib, _packcache


class _DcnLevel(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("output", ctypes.c_void_p),
                ("height", ctypes.c_int), ("width", ctypes.c_int)]


_packed_cache = _packcache.new_cache("dcn_weight_fp32")


def _packed_weight(weight, cache=True):
    """[Cout,Cin,kh,kw] -> [kh*kw,Cin,Cout] ++ [kh*kw,Cin/4,Cout,4] (both kernel generations).

    `cache=True` (inference): the pack is cached on the live parameter object (_packcache.OwnerCache: weak reference +
    identity check + storage / version state).  `cache=False` (the autograd Functions pass it when the weight takes a
    gradient -- decided from `ctx.needs_input_grad`, because grad mode is always off inside Function.forward): the pack
    is rebuilt on every call, so an optimizer / EMA / `reset_parameters` writing through `.data` (no version bump) can
    never make forward and backward disagree (the pack costs ~1 % of the conv)."""
    w = weight.detach()
    state = _packcache.tensor_state(w)
    if cache:
        hit = _packed_cache.get(weight, state)
        if hit is not None:
            return _lib.keep_for_graph(hit)
    w = w.float().contiguous()
    cout, cin, kh, kw = w.shape
    packed = torch.empty((_lib.lib().orp_dcn_packed_weight_floats(cout, cin, kh, kw),), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().orp_dcn_pack_weight(_lib.ptr(w), cout, cin, kh, kw, _lib.ptr(packed), _lib.stream_of(w))
    _lib.check(rc, "orp_dcn_pack_weight")
    if not cache:
        return packed
    return _lib.keep_for_graph(_packed_cache.put(weight, state, packed))


def invalidate_packed_weights():
    """Drop EVERY cached weight pack and folded affine (fp32 / half DeformConv packs, head 1x1 packs, BatchNorm
    affines): call after writing parameters through `.data`, which bypasses the version counters."""
    _packcache.invalidate_all()


def _out_hw(h, w, weight, stride, padding, dilation):
    ho = (h + 2 * padding[0] - (dilation[0] * (weight.size(2) - 1) + 1)) // stride[0] + 1
    wo = (w + 2 * padding[1] - (dilation[1] * (weight.size(3) - 1) + 1)) // stride[1] + 1
    return ho, wo


def fast_path_ok(weight, groups, deformable_groups):
    cout, cin_g, kh, kw = weight.shape
    return bool(_lib.lib().orp_dcn_fast_path_ok(cin_g * groups, cout, kh, kw, groups, deformable_groups))


def deform_conv_forward_multi(inputs, offsets, weight, stride, padding, dilation, masks=None, bias=None, relu=False,
                              cache_pack=True):
    """One DeformConv layer over a list of feature maps (same batch / channels) in ONE launch.  fp32, no autograd.
    masks (list of [B,kh*kw,Ho,Wo], DCNv2 modulation) / bias ([Cout]) / relu (fused max(., 0)) are optional."""
    if inputs[0].dtype in _HALF_CODES and weight.dtype == inputs[0].dtype and half_path_ok(weight, 1, 1):
        return deform_conv_forward_multi_half(inputs, offsets, weight, stride, padding, dilation, masks, bias, relu,
                                              cache_pack)
    L = _lib.lib()
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    x0 = inputs[0]
    B, cin = x0.size(0), x0.size(1)
    cout, _, kh, kw = weight.shape
    nhwc = all(x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
               for x in inputs)
    packed = _packed_weight(weight, cache_pack)
    xs, offs, outs = [], [], []
    levels = (_DcnLevel * len(inputs))()
    for i, (x, off) in enumerate(zip(inputs, offsets)):
        assert x.size(0) == B and x.size(1) == cin
        x = x.detach().float()
        x = x if nhwc else x.contiguous()
        off = off.detach().float().contiguous()
        ho, wo = _out_hw(x.size(2), x.size(3), weight, stride, padding, dilation)
        if off.size(1) != 2 * kh * kw or off.size(2) != ho or off.size(3) != wo:
            raise ValueError("offset must be [B, 2*kh*kw, Ho, Wo]")
        out = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device,
                          memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        xs.append(x); offs.append(off); outs.append(out)
        levels[i] = _DcnLevel(x.data_ptr(), off.data_ptr(), out.data_ptr(), x.size(2), x.size(3))
    mask_ptrs, ms = None, []
    if masks is not None:
        mask_ptrs = (ctypes.c_void_p * len(inputs))()
        for i, m in enumerate(masks):
            m = m.detach().float().contiguous()
            if tuple(m.shape) != (B, kh * kw, outs[i].size(2), outs[i].size(3)):
                raise ValueError("mask must be [B, kh*kw, Ho, Wo]")
            ms.append(m)
            mask_ptrs[i] = m.data_ptr()
    b = bias.detach().float().contiguous() if bias is not None else None
    layout = 1 if nhwc else 0
    nbytes = L.orp_dcn_forward_workspace_bytes(levels, len(inputs), B, cin, layout)
    ws = _lib.workspace(x0.device, nbytes)
    with torch.cuda.device(x0.device):
        rc = L.orp_dcn_forward_multi_ex(levels, mask_ptrs, len(inputs), B, cin, cout, _lib.ptr(packed), _lib.ptr(b),
                                        1 if relu else 0, kh, kw, stride[0], stride[1], padding[0], padding[1],
                                        dilation[0], dilation[1], layout, layout, _lib.ptr(ws), ws.numel(),
                                        _lib.stream_of(x0))
    _lib.check(rc, "orp_dcn_forward_multi_ex")
    return outs


_HALF_CODES = {torch.float16: 1, torch.bfloat16: 2}
_packed_cache_h = _packcache.new_cache("dcn_weight_half")


def _packed_weight_h(weight, cache=True):
    """[Cout,Cin,kh,kw] fp16 / bf16 -> [tap][Cin/16][2][Cout][8] (orp_dcn_pack_weight_h), cached like the fp32 pack."""
    w = weight.detach()
    state = _packcache.tensor_state(w)
    if cache:
        hit = _packed_cache_h.get(weight, state)
        if hit is not None:
            return _lib.keep_for_graph(hit)
    w = w.contiguous()
    cout, cin, kh, kw = w.shape
    packed = torch.empty((cout * cin * kh * kw,), dtype=w.dtype, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.lib().orp_dcn_pack_weight_h(_lib.ptr(w), cout, cin, kh, kw, _lib.ptr(packed), _HALF_CODES[w.dtype],
                                              _lib.stream_of(w))
    _lib.check(rc, "orp_dcn_pack_weight_h")
    if not cache:
        return packed
    return _lib.keep_for_graph(_packed_cache_h.put(weight, state, packed))


def half_path_ok(weight, groups, deformable_groups):
    cout, cin_g, kh, kw = weight.shape
    return weight.dtype in _HALF_CODES and bool(_lib.lib().orp_dcn_half_path_ok(cin_g * groups, cout, kh, kw, groups,
                                                                               deformable_groups))


def deform_conv_forward_multi_half(inputs, offsets, weight, stride, padding, dilation, masks=None, bias=None, relu=False,
                                   cache_pack=True):
    """`deform_conv_forward_multi` for fp16 / bf16 tensors (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF
    branch; BASELINE configs[4]): v_mfma_f32_32x32x16_{f16,bf16}, fp32 bilinear combine and accumulation, outputs in the
    input dtype.  All tensors (inputs, offsets, masks, bias, weight) must share that dtype."""
    L = _lib.lib()
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    x0 = inputs[0]
    dt = x0.dtype
    code = _HALF_CODES[dt]
    B, cin = x0.size(0), x0.size(1)
    cout, _, kh, kw = weight.shape
    if weight.dtype != dt:
        raise TypeError("deform_conv (half): weight dtype %s != input dtype %s" % (weight.dtype, dt))
    nhwc = all(x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
               for x in inputs)
    packed = _packed_weight_h(weight, cache_pack)

    class _LevelH(ctypes.Structure):
        _fields_ = [("input", ctypes.c_void_p), ("offset", ctypes.c_void_p), ("output", ctypes.c_void_p),
                    ("height", ctypes.c_int), ("width", ctypes.c_int)]
    levels = (_LevelH * len(inputs))()
    keep, outs = [], []
    for i, (x, off) in enumerate(zip(inputs, offsets)):
        assert x.size(0) == B and x.size(1) == cin and x.dtype == dt
        x = x.detach()
        x = x if nhwc else x.contiguous()
        off = off.detach().to(dt).contiguous()
        ho, wo = _out_hw(x.size(2), x.size(3), weight, stride, padding, dilation)
        if off.size(1) != 2 * kh * kw or off.size(2) != ho or off.size(3) != wo:
            raise ValueError("offset must be [B, 2*kh*kw, Ho, Wo]")
        out = torch.empty((B, cout, ho, wo), dtype=dt, device=x.device,
                          memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        keep += [x, off]; outs.append(out)
        levels[i] = _LevelH(x.data_ptr(), off.data_ptr(), out.data_ptr(), x.size(2), x.size(3))
    mask_ptrs = None
    if masks is not None:
        mask_ptrs = (ctypes.c_void_p * len(inputs))()
        for i, m in enumerate(masks):
            m = m.detach().to(dt).contiguous()
            if tuple(m.shape) != (B, kh * kw, outs[i].size(2), outs[i].size(3)):
                raise ValueError("mask must be [B, kh*kw, Ho, Wo]")
            keep.append(m)
            mask_ptrs[i] = m.data_ptr()
    b = bias.detach().to(dt).contiguous() if bias is not None else None
    layout = 1 if nhwc else 0
    nbytes = L.orp_dcn_forward_h_workspace_bytes(levels, len(inputs), B, cin, layout)
    ws = _lib.workspace(x0.device, nbytes)
    with torch.cuda.device(x0.device):
        rc = L.orp_dcn_forward_multi_h(levels, mask_ptrs, len(inputs), B, cin, cout, _lib.ptr(packed), _lib.ptr(b),
                                       1 if relu else 0, kh, kw, stride[0], stride[1], padding[0], padding[1],
                                       dilation[0], dilation[1], layout, layout, code, _lib.ptr(ws), ws.numel(),
                                       _lib.stream_of(x0))
    _lib.check(rc, "orp_dcn_forward_multi_h")
    return outs


def deform_conv_forward_pair(inputs_a, inputs_b, offsets, weight_a, weight_b, stride, padding, dilation, masks=None,
                             bias_a=None, bias_b=None, relu=False, cache_pack=True, out_channels_last=None, amax=None):
    """Two DeformConv layers with the SAME offsets (the head's cls / refine pair) over a list of feature maps in ONE
    launch (`orp_dcn_forward_pair`): the bilinear coefficient table of every tile is built once for both layers.
    fp32, no autograd.  Returns (outs_a, outs_b).  Falls back to two `deform_conv_forward_multi` launches when the
    channel count is not a multiple of 256.  Outputs follow the inputs' memory format unless `out_channels_last` says
    otherwise (the head hands the towers' last layer over channels-last -- no transposition launch -- and wants NCHW
    outputs for its 1x1 convolutions).  amax: a `fused_norm.Amax` from the producer of channels-last inputs (slot 0: inputs_a,
    slot `stride`: inputs_b) -- the fp16-pieces mode then skips its range pre-pass."""
    L = _lib.lib()
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    x0 = inputs_a[0]
    B, cin = x0.size(0), x0.size(1)
    cout, _, kh, kw = weight_a.shape
    if cin % 256 != 0 or tuple(weight_b.shape) != tuple(weight_a.shape):
        return (deform_conv_forward_multi(inputs_a, offsets, weight_a, stride, padding, dilation, masks, bias_a, relu,
                                          cache_pack),
                deform_conv_forward_multi(inputs_b, offsets, weight_b, stride, padding, dilation, masks, bias_b, relu,
                                          cache_pack))
    nhwc = all(x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
               for x in list(inputs_a) + list(inputs_b))
    pa, pb = _packed_weight(weight_a, cache_pack), _packed_weight(weight_b, cache_pack)
    n = len(inputs_a)
    lev_a, lev_b = (_DcnLevel * n)(), (_DcnLevel * n)()
    keep, outs_a, outs_b = [], [], []
    for i in range(n):
        xa, xb, off = inputs_a[i].detach().float(), inputs_b[i].detach().float(), offsets[i].detach().float().contiguous()
        assert xa.shape == xb.shape and xa.size(0) == B and xa.size(1) == cin
        xa = xa if nhwc else xa.contiguous()
        xb = xb if nhwc else xb.contiguous()
        ho, wo = _out_hw(xa.size(2), xa.size(3), weight_a, stride, padding, dilation)
        if off.size(1) != 2 * kh * kw or off.size(2) != ho or off.size(3) != wo:
            raise ValueError("offset must be [B, 2*kh*kw, Ho, Wo]")
        out_cl = nhwc if out_channels_last is None else bool(out_channels_last)
        fmt = torch.channels_last if out_cl else torch.contiguous_format
        oa = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=xa.device, memory_format=fmt)
        ob = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=xa.device, memory_format=fmt)
        keep += [xa, xb, off]; outs_a.append(oa); outs_b.append(ob)
        lev_a[i] = _DcnLevel(xa.data_ptr(), off.data_ptr(), oa.data_ptr(), xa.size(2), xa.size(3))
        lev_b[i] = _DcnLevel(xb.data_ptr(), off.data_ptr(), ob.data_ptr(), xb.size(2), xb.size(3))
    mask_ptrs = None
    if masks is not None:
        mask_ptrs = (ctypes.c_void_p * n)()
        for i, m in enumerate(masks):
            m = m.detach().float().contiguous()
            if tuple(m.shape) != (B, kh * kw, outs_a[i].size(2), outs_a[i].size(3)):
                raise ValueError("mask must be [B, kh*kw, Ho, Wo]")
            keep.append(m)
            mask_ptrs[i] = m.data_ptr()
    ba = bias_a.detach().float().contiguous() if bias_a is not None else None
    bb = bias_b.detach().float().contiguous() if bias_b is not None else None
    layout = 1 if nhwc else 0
    nbytes = 2 * L.orp_dcn_forward_workspace_bytes(lev_a, n, B, cin, layout)
    ws = _lib.workspace(x0.device, nbytes)
    with torch.cuda.device(x0.device):
        if amax is not None and nhwc:
            rc = L.orp_dcn_forward_pair_amax(lev_a, lev_b, mask_ptrs, n, B, cin, cout, _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(ba),
                                             _lib.ptr(bb), 1 if relu else 0, kh, kw, stride[0], stride[1], padding[0],
                                             padding[1], dilation[0], dilation[1], layout, 1 if out_cl else 0, _lib.ptr(ws),
                                             ws.numel(), amax.bits.data_ptr(), int(amax.stride), _lib.stream_of(x0))
        else:
            rc = L.orp_dcn_forward_pair(lev_a, lev_b, mask_ptrs, n, B, cin, cout, _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(ba),
                                        _lib.ptr(bb), 1 if relu else 0, kh, kw, stride[0], stride[1], padding[0], padding[1],
                                        dilation[0], dilation[1], layout, 1 if out_cl else 0, _lib.ptr(ws), ws.numel(),
                                        _lib.stream_of(x0))
    _lib.check(rc, "orp_dcn_forward_pair")
    return outs_a, outs_b


class _DcnHeadLevel(ctypes.Structure):
    _fields_ = [("output_a", ctypes.c_void_p), ("output_b", ctypes.c_void_p), ("residual_b", ctypes.c_void_p)]


class _DcnHeads(ctypes.Structure):
    _fields_ = [("weight_a_packed", ctypes.c_void_p), ("bias_a", ctypes.c_void_p), ("k_a", ctypes.c_int),
                ("weight_b_packed", ctypes.c_void_p), ("bias_b", ctypes.c_void_p), ("k_b", ctypes.c_int),
                ("levels", ctypes.POINTER(_DcnHeadLevel))]


_packed_heads = _packcache.new_cache("dcn_head_1x1")


def _packed_head_weight(weight):
    """[k,256,1,1] -> the [256][20] pack of orp_dcn_forward_pair_heads, cached on the live parameter (inference only)."""
    w = weight.detach()
    state = _packcache.tensor_state(w)
    hit = _packed_heads.get(weight, state)
    if hit is not None:
        return _lib.keep_for_graph(hit)
    k = w.size(0)
    w2 = w.float().reshape(k, 256).contiguous()
    packed = torch.empty((_lib.lib().orp_dcn_head_packed_floats(),), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().orp_dcn_pack_head_weight(_lib.ptr(w2), k, _lib.ptr(packed), _lib.stream_of(w2)),
                   "orp_dcn_pack_head_weight")
    return _lib.keep_for_graph(_packed_heads.put(weight, state, packed))


def pair_heads_ok(conv_a, conv_b, head_a, head_b, x):
    """The head's refinement stage can run as ONE launch (`deform_conv_forward_pair_heads`)."""
    def one_by_one(m):
        w = m.weight
        return (w.dim() == 4 and w.size(1) == 256 and w.size(2) == 1 and w.size(3) == 1 and w.size(0) <= 20 and
                tuple(m.stride) == (1, 1) and tuple(m.padding) == (0, 0) and m.groups == 1)
    return bool(x.is_cuda and x.dtype == torch.float32 and tuple(conv_a.weight.shape) == (256, 256, 3, 3) and
                tuple(conv_b.weight.shape) == (256, 256, 3, 3) and conv_a.stride == conv_b.stride and
                conv_a.padding == conv_b.padding and conv_a.dilation == conv_b.dilation and
                conv_a.groups == conv_b.groups == 1 and conv_a.deformable_groups == conv_b.deformable_groups == 1 and
                one_by_one(head_a) and one_by_one(head_b))


def deform_conv_forward_pair_heads(inputs_a, inputs_b, offsets, conv_a, conv_b, head_a, head_b, residuals_b=None):
    """relu(DeformConv_a(x_a)) -> head_a (1x1) and relu(DeformConv_b(x_b)) -> head_b (1x1) (+ residuals_b) for all levels
    in ONE launch (`orp_dcn_forward_pair_heads`): the head's refinement stage, orientedreppoints_head.py:164-170.  The
    256-channel DeformConv outputs are never materialised.  fp32, no autograd.  Returns (outs_a, outs_b)."""
    L = _lib.lib()
    stride, padding, dilation = _pair(conv_a.stride), _pair(conv_a.padding), _pair(conv_a.dilation)
    x0 = inputs_a[0]
    B = x0.size(0)
    kh, kw = conv_a.weight.size(2), conv_a.weight.size(3)
    ka, kb = head_a.weight.size(0), head_b.weight.size(0)
    pa, pb = _packed_weight(conv_a.weight), _packed_weight(conv_b.weight)
    ha, hb = _packed_head_weight(head_a.weight), _packed_head_weight(head_b.weight)
    ba = head_a.bias.detach().float().contiguous() if head_a.bias is not None else None
    bb = head_b.bias.detach().float().contiguous() if head_b.bias is not None else None
    n = len(inputs_a)
    lev_a, lev_b, hl = (_DcnLevel * n)(), (_DcnLevel * n)(), (_DcnHeadLevel * n)()
    keep, outs_a, outs_b = [], [], []
    for i in range(n):
        xa, xb = inputs_a[i].detach().float().contiguous(), inputs_b[i].detach().float().contiguous()
        off = offsets[i].detach().float().contiguous()
        ho, wo = _out_hw(xa.size(2), xa.size(3), conv_a.weight, stride, padding, dilation)
        if xa.shape != xb.shape or xa.size(1) != 256 or tuple(off.shape) != (B, 2 * kh * kw, ho, wo):
            raise ValueError("deform_conv_forward_pair_heads: [B,256,H,W] inputs and [B,2*kh*kw,Ho,Wo] offsets expected")
        oa = torch.empty((B, ka, ho, wo), dtype=torch.float32, device=xa.device)
        ob = torch.empty((B, kb, ho, wo), dtype=torch.float32, device=xa.device)
        r = None
        if residuals_b is not None:
            r = residuals_b[i].detach().float().contiguous()
            if r.shape != ob.shape:
                raise ValueError("deform_conv_forward_pair_heads: residual must match the second head's output")
        keep += [xa, xb, off, r]
        outs_a.append(oa); outs_b.append(ob)
        lev_a[i] = _DcnLevel(xa.data_ptr(), off.data_ptr(), None, xa.size(2), xa.size(3))
        lev_b[i] = _DcnLevel(xb.data_ptr(), off.data_ptr(), None, xb.size(2), xb.size(3))
        hl[i] = _DcnHeadLevel(oa.data_ptr(), ob.data_ptr(), r.data_ptr() if r is not None else None)
    heads = _DcnHeads(ha.data_ptr(), ba.data_ptr() if ba is not None else None, ka, hb.data_ptr(),
                      bb.data_ptr() if bb is not None else None, kb, hl)
    nbytes = 2 * L.orp_dcn_forward_workspace_bytes(lev_a, n, B, 256, 0)
    ws = _lib.workspace(x0.device, nbytes)
    with torch.cuda.device(x0.device):
        rc = L.orp_dcn_forward_pair_heads(lev_a, lev_b, n, B, _lib.ptr(pa), _lib.ptr(pb), ctypes.byref(heads), kh, kw,
                                          stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], 0,
                                          _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
    _lib.check(rc, "orp_dcn_forward_pair_heads")
    return outs_a, outs_b


def _forward_direct(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups):
    x = input.detach().float().contiguous()
    off = offset.detach().float().contiguous()
    w = weight.detach().float().contiguous()
    m = mask.detach().float().contiguous() if mask is not None else None
    b = bias.detach().float().contiguous() if bias is not None else None
    B, cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    ho, wo = _out_hw(H, W, w, stride, padding, dilation)
    out = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.lib().orp_dcn_forward_direct(_lib.ptr(x), _lib.ptr(off), _lib.ptr(m), _lib.ptr(w), _lib.ptr(b),
                                               _lib.ptr(out), B, cin, H, W, cout, kh, kw, stride[0], stride[1],
                                               padding[0], padding[1], dilation[0], dilation[1], groups,
                                               deformable_groups, _lib.stream_of(x))
    _lib.check(rc, "orp_dcn_forward_direct")
    return out


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step
        ctx.save_for_backward(input, offset, weight)
        DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride)
        if not input.is_cuda:
            raise NotImplementedError
        cur_im2col_step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        from . import deform_conv_backward as bw
        if bw.is_f64(input, offset, weight):
            # the `double` branch of AT_DISPATCH_FLOATING_TYPES_AND_HALF (deform_conv_cuda_kernel.cu:259): computed in double
            out = bw.forward_columns(input, offset, None, weight, None, ctx.stride, ctx.padding, ctx.dilation, groups,
                                     deformable_groups)
        elif fast_path_ok(weight, groups, deformable_groups):
            out = deform_conv_forward_multi([input], [offset], weight, ctx.stride, ctx.padding, ctx.dilation,
                                            cache_pack=not ctx.needs_input_grad[2])[0]
        else:
            out = _forward_direct(input, offset, None, weight, None, ctx.stride, ctx.padding, ctx.dilation, groups,
                                  deformable_groups)
        return out.to(input.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        from . import deform_conv_backward as bw
        input, offset, weight = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise NotImplementedError
        grad_input = grad_offset = grad_weight = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_offset = bw.backward_input(input, offset, weight, grad_output, ctx.stride, ctx.padding,
                                                        ctx.dilation, ctx.groups, ctx.deformable_groups)
        if ctx.needs_input_grad[2]:
            grad_weight = bw.backward_parameters(input, offset, weight, grad_output, ctx.stride, ctx.padding,
                                                 ctx.dilation, ctx.groups, ctx.deformable_groups)
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        """(B, Cout, Ho, Wo) of the convolution; a non-positive extent is the reference's 'input is too small' error."""
        spatial = tuple((input.size(ax + 2) + 2 * padding[ax] - (dilation[ax] * (weight.size(ax + 2) - 1) + 1)) // stride[ax] + 1
                        for ax in range(input.dim() - 2))
        shape = (input.size(0), weight.size(0)) + spatial
        if min(shape) <= 0:
            raise ValueError('convolution input is too small (output would be {})'.format('x'.join(str(v) for v in shape)))
        return shape


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias if bias is not None else input.new_empty(1))
        from . import deform_conv_backward as bw
        if bw.is_f64(input, offset, mask, weight):
            out = bw.forward_columns(input, offset, mask, weight, bias, _pair(stride), _pair(padding), _pair(dilation), groups,
                                     deformable_groups)
        elif fast_path_ok(weight, groups, deformable_groups):
            # DCNv2 on the MFMA implicit GEMM: the modulation scalar is folded into the bilinear weights of the tile
            out = deform_conv_forward_multi([input], [offset], weight, stride, padding, dilation, masks=[mask],
                                            bias=bias, cache_pack=not ctx.needs_input_grad[3])[0]
        else:
            out = _forward_direct(input, offset, mask, weight, bias, _pair(stride), _pair(padding), _pair(dilation),
                                  groups, deformable_groups)
        return out.to(input.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        from . import deform_conv_backward as bw
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        gi, go, gm, gw, gb = bw.modulated_backward(input, offset, mask, weight, grad_output, _pair(ctx.stride),
                                                   _pair(ctx.padding), _pair(ctx.dilation), ctx.groups,
                                                   ctx.deformable_groups, ctx.with_bias)
        return (gi, go, gm, gw, gb, None, None, None, None, None)


class DeformConvPairFunction(Function):
    """The head's two DeformConvs (same offsets) over ALL levels as one autograd node: forward = ONE pair launch
    (`orp_dcn_forward_pair`), backward = one `orp_dcn_backward_multi` call per layer (MFMA implicit GEMMs, every level in
    the same launches), the two layers' offset gradients added.  Inputs after the constants: weight_a, weight_b, then
    n inputs of layer a, n inputs of layer b, n offsets.  Returns the 2n outputs."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, stride, padding, dilation, n, weight_a, weight_b, *tensors):
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        # n may carry the callers' expectation about the two layers' output gradients (deform_conv_pair's sparse_grad)
        n, ctx.sparse_a, ctx.sparse_b = (n if isinstance(n, tuple) else (n, False, False))
        ctx.n = n
        xa, xb, offs = list(tensors[:n]), list(tensors[n:2 * n]), list(tensors[2 * n:])
        ctx.save_for_backward(weight_a, weight_b, *tensors)
        oa, ob = deform_conv_forward_pair(xa, xb, offs, weight_a, weight_b, ctx.stride, ctx.padding, ctx.dilation,
                                          cache_pack=not (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]))
        return tuple(oa) + tuple(ob)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        from . import deform_conv_backward as bw
        n = ctx.n
        weight_a, weight_b = ctx.saved_tensors[:2]
        tensors = ctx.saved_tensors[2:]
        xa, xb, offs = list(tensors[:n]), list(tensors[n:2 * n]), list(tensors[2 * n:])
        need_in = any(ctx.needs_input_grad[6:])
        res = []
        for w, xs, gs, wi, sparse in ((weight_a, xa, grads[:n], 4, ctx.sparse_a), (weight_b, xb, grads[n:], 5, ctx.sparse_b)):
            gs = [g if g is not None else torch.zeros((x.size(0), w.size(0)) + tuple(o.shape[2:]), device=x.device)
                  for g, x, o in zip(gs, xs, offs)]
            res.append(bw.backward_mfma(xs, offs, w, gs, ctx.stride, ctx.padding, ctx.dilation, need_input=need_in,
                                        need_weight=ctx.needs_input_grad[wi], sparse_grad=sparse))
        (gia, goa, gwa), (gib, gob, gwb) = res
        goff = [a + b for a, b in zip(goa, gob)] if need_in else [None] * n
        if not need_in:
            gia, gib = [None] * n, [None] * n
        return (None, None, None, None, gwa, gwb) + tuple(gia) + tuple(gib) + tuple(goff)


def deform_conv_pair(inputs_a, inputs_b, offsets, weight_a, weight_b, stride=1, padding=0, dilation=1,
                     sparse_grad=(False, False)):
    """Autograd-capable pair launch over a list of levels (training path of the head): (outs_a, outs_b).
    sparse_grad[i]: layer i's output gradient is expected to be zero almost everywhere (the head's point-refinement
    branch only receives gradient at positive points) -> its backward scatters with atomics (ORP_DCN_BWD_SPARSE)."""
    n = len(inputs_a)
    outs = DeformConvPairFunction.apply(stride, padding, dilation, (n, bool(sparse_grad[0]), bool(sparse_grad[1])),
                                        weight_a, weight_b, *inputs_a, *inputs_b, *offsets)
    return list(outs[:n]), list(outs[n:])


def pair_autograd_ok(conv_a, conv_b, x):
    """True when the two DeformConv modules can run as one `deform_conv_pair` node on x's device / dtype."""
    from . import deform_conv_backward as bw
    same = (conv_a.stride == conv_b.stride and conv_a.padding == conv_b.padding and conv_a.dilation == conv_b.dilation and
            conv_a.weight.shape == conv_b.weight.shape and conv_a.groups == conv_b.groups == 1 and
            conv_a.deformable_groups == conv_b.deformable_groups == 1)
    return bool(same and x.is_cuda and x.dtype == torch.float32 and conv_a.weight.size(1) % 256 == 0 and
                fast_path_ok(conv_a.weight, 1, 1) and bw.mfma_ok(conv_a.weight, 1, 1) and
                min(x.size(2), x.size(3)) >= conv_a.kernel_size[0])


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


def _fan_in_uniform_(weight, in_channels, kernel_size):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with fan_in = Cin * kh * kw (what both reference modules initialise with)."""
    bound = 1.0 / math.sqrt(in_channels * kernel_size[0] * kernel_size[1])
    weight.data.uniform_(-bound, bound)


def _adopt_legacy_offset_keys(state_dict, prefix, local_metadata):
    """Checkpoints written before module version 2 store the offset convolution as `<name>_offset.{weight,bias}`; move them
    to `<name>.conv_offset.*` unless the new key is already there."""
    if local_metadata.get('version', None) not in (None, 1):
        return
    for leaf in ('weight', 'bias'):
        old, new = prefix[:-1] + '_offset.' + leaf, prefix + 'conv_offset.' + leaf
        if new not in state_dict and old in state_dict:
            state_dict[new] = state_dict.pop(old)


class DeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        assert not bias
        for name, ch in (('in_channels', in_channels), ('out_channels', out_channels)):
            assert ch % groups == 0, '{} {} cannot be divisible by groups {}'.format(name, ch, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.transposed, self.output_padding = False, _single(0)             # the attributes nn.Conv2d consumers look for
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        _fan_in_uniform_(self.weight, self.in_channels, self.kernel_size)

    def forward(self, x, offset):
        # inputs smaller than the kernel are padded, as the reference does (deform_conv.py:239-255)
        input_pad = (x.size(2) < self.kernel_size[0] or x.size(3) < self.kernel_size[1])
        if input_pad:
            pad_h = max(self.kernel_size[0] - x.size(2), 0)
            pad_w = max(self.kernel_size[1] - x.size(3), 0)
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                          self.deformable_groups)
        if input_pad:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out

    def forward_multi(self, xs, offsets, relu=False):
        """All FPN levels in one launch (inference / no-grad fast path); relu=True fuses the ReLU the head applies to
        both DeformConv outputs into the kernel's epilogue."""
        if fast_path_ok(self.weight, self.groups, self.deformable_groups) and not torch.is_grad_enabled():
            return deform_conv_forward_multi(xs, offsets, self.weight, self.stride, self.padding, self.dilation,
                                             relu=relu)
        outs = [self.forward(x, o) for x, o in zip(xs, offsets)]
        return [F.relu(o) for o in outs] if relu else outs


class DeformConvPack(DeformConv):
    """DeformConv that predicts its own offsets with a zero-initialised convolution of the same geometry."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        taps = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = nn.Conv2d(self.in_channels, 2 * taps * self.deformable_groups, kernel_size=self.kernel_size,
                                     stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):
        for t in (self.conv_offset.weight, self.conv_offset.bias):
            t.data.zero_()

    def forward(self, x):
        return deform_conv(x, self.conv_offset(x), self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        _adopt_legacy_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)


class ModulatedDeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation  # (kept as given, like the reference: ints or pairs)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.with_bias = bias
        self.transposed, self.output_padding = False, _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.register_parameter('bias', nn.Parameter(torch.Tensor(out_channels)) if bias else None)
        self.reset_parameters()

    def reset_parameters(self):
        _fan_in_uniform_(self.weight, self.in_channels, self.kernel_size)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """DCNv2 with its own predictor: one zero-initialised convolution emits 2 offset maps and 1 modulation logit per tap."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        taps = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = nn.Conv2d(self.in_channels, 3 * taps * self.deformable_groups, kernel_size=self.kernel_size,
                                     stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):
        for t in (self.conv_offset.weight, self.conv_offset.bias):
            t.data.zero_()

    def forward(self, x):
        dy, dx, logit = torch.chunk(self.conv_offset(x), 3, dim=1)
        return modulated_deform_conv(x, torch.cat((dy, dx), dim=1), torch.sigmoid(logit), self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups, self.deformable_groups)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        _adopt_legacy_offset_keys(state_dict, prefix, local_metadata)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
