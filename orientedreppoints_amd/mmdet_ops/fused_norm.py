"""Fused normalisation + activation passes (include/orp_hip.h `orp_groupnorm_act_multi`, `orp_affine_act`,
`orp_bias_act_multi`): the GroupNorm+ReLU of the dense-head ConvModules for all FPN levels in one launch pair, the
eval-mode BatchNorm (+ residual) + ReLU of the ResNet bottlenecks as one pass, and the bias / ReLU / residual / base-offset
passes around the head's output convolutions for all levels in one launch.  Those are inference-only (callers use them
under torch.no_grad()).  Training: `group_norm_act_train` is the autograd-capable GroupNorm(+ReLU) over a list of tensors
(one launch pair forward, one pair + a parameter reduction backward, `orp_groupnorm_act_multi_train / _backward`)."""
import ctypes

import torch

from .. import _lib, _packcache


class _NormLevel(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("output", ctypes.c_void_p), ("height", ctypes.c_int),
                ("width", ctypes.c_int)]


def group_norm_act_multi(xs, gn, relu=True, inplace=True, nhwc=None):
    """[GroupNorm(+ReLU)(x) for x in xs] -- xs: list of [B,C,H,W] fp32 CUDA tensors (FPN levels, possibly of several
    towers: up to 16); gn: one nn.GroupNorm for all of them, or a list with one module per tensor (same num_groups and
    eps).  ONE launch pair.
    nhwc='only': the results come back as channels-last tensors ONLY (same logical shape, memory [B,H,W,C]) -- written
    transposed by the normalisation's second pass; nhwc='both': (NCHW list, channels-last list).  What the head's DeformConv
    consumes without a transposition launch."""
    L = _lib.lib()
    x0 = xs[0]
    B, C = x0.size(0), x0.size(1)
    gns = list(gn) if isinstance(gn, (list, tuple)) else [gn] * len(xs)
    if len(gns) != len(xs) or any(g.num_groups != gns[0].num_groups or g.eps != gns[0].eps
                                  for g in {id(g): g for g in gns}.values()):
        raise ValueError("group_norm_act_multi: one GroupNorm per tensor, equal num_groups / eps")
    levels = (_NormLevel * len(xs))()
    gam = (ctypes.c_void_p * len(xs))()
    bet = (ctypes.c_void_p * len(xs))()
    ins, outs, keep, seen = [], [], [], {}
    for i, x in enumerate(xs):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == C):
            raise ValueError("group_norm_act_multi expects fp32 CUDA [B,C,H,W] tensors with equal B and C")
        x = x.detach().contiguous()
        y = x if inplace else torch.empty_like(x)
        ins.append(x); outs.append(y)
        levels[i] = _NormLevel(x.data_ptr(), y.data_ptr(), x.size(2), x.size(3))
        ptrs = seen.get(id(gns[i]))
        if ptrs is None:                                   # once per distinct module, not per tensor
            g_ = gns[i].weight.detach().float().contiguous()
            b_ = gns[i].bias.detach().float().contiguous()
            keep += [g_, b_]
            ptrs = seen[id(gns[i])] = (g_.data_ptr(), b_.data_ptr())
        gam[i], bet[i] = ptrs
    nbytes = L.orp_groupnorm_workspace_bytes(levels, len(xs), B, C, gns[0].num_groups)
    ws = _lib.workspace(x0.device, nbytes)
    if nhwc is not None:
        if nhwc not in ('only', 'both') or C % 32 != 0 or 32 % (C // gns[0].num_groups) != 0:
            raise ValueError("group_norm_act_multi(nhwc=...): 'only' | 'both', channels % 32 == 0, 32 % (channels / groups) == 0")
        cl = [torch.empty(x.shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last) for x in ins]
        cl_ptrs = (ctypes.c_void_p * len(xs))(*[t.data_ptr() for t in cl])
        if nhwc == 'only':
            for i in range(len(xs)):
                levels[i].output = None
        with torch.cuda.device(x0.device):
            rc = L.orp_groupnorm_act_multi_nhwc(levels, gam, bet, cl_ptrs, len(xs), B, C, gns[0].num_groups, float(gns[0].eps),
                                                1 if relu else 0, _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
        _lib.check(rc, "orp_groupnorm_act_multi_nhwc")
        return cl if nhwc == 'only' else (outs, cl)
    with torch.cuda.device(x0.device):
        rc = L.orp_groupnorm_act_multi_ex(levels, gam, bet, len(xs), B, C, gns[0].num_groups, float(gns[0].eps),
                                          1 if relu else 0, _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
    _lib.check(rc, "orp_groupnorm_act_multi_ex")
    return outs


_affine_cache = _packcache.new_cache("bn_affine")


def _bn_affine(bn):
    """eval-mode BatchNorm as (scale, shift): y = x*scale + shift; cached on the live module (_packcache.OwnerCache:
    weak reference + identity check), keyed by the storage / version state of its four tensors."""
    def st(t):
        return _packcache.tensor_state(t) if t is not None else None
    state = (st(bn.weight), st(bn.bias), st(bn.running_mean), st(bn.running_var), float(bn.eps))
    hit = _affine_cache.get(bn, state)
    if hit is None:
        with torch.no_grad():
            rstd = torch.rsqrt(bn.running_var.float() + bn.eps)
            w = bn.weight.float() if bn.weight is not None else torch.ones_like(rstd)
            b = bn.bias.float() if bn.bias is not None else torch.zeros_like(rstd)
            scale = (w * rstd).contiguous()
            shift = (b - bn.running_mean.float() * scale).contiguous()
        hit = _affine_cache.put(bn, state, (scale, shift))
    return _lib.keep_for_graph(hit[0]), _lib.keep_for_graph(hit[1])


def bn_act(x, bn, residual=None, relu=True):
    """relu?(BatchNorm_eval(x) (+ residual)) in ONE pass, in place on x ([B,C,H,W] fp32 CUDA, contiguous)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        raise ValueError("bn_act expects a contiguous fp32 CUDA [B,C,H,W] tensor")
    if residual is not None and not (residual.shape == x.shape and residual.is_contiguous()
                                     and residual.dtype == torch.float32):
        raise ValueError("bn_act: residual must match x")
    scale, shift = _bn_affine(bn)
    B, C, H, W = x.shape
    with torch.cuda.device(x.device):
        rc = _lib.lib().orp_affine_act(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(x),
                                       B, C, H * W, 1 if relu else 0, _lib.stream_of(x))
    _lib.check(rc, "orp_affine_act")
    return x


class _BiasLevel(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("output", ctypes.c_void_p),
                ("output2", ctypes.c_void_p), ("height", ctypes.c_int), ("width", ctypes.c_int)]


def bias_act_multi(xs, bias, relu=False, residuals=None, sub=None):
    """ys[i] = relu?(xs[i] + bias[c] (+ residuals[i])), in place on xs; with `sub` ([C]) additionally returns
    zs[i] = ys[i] - sub[c].  xs: fp32 CUDA [B,C,H,W] tensors (one per FPN level), ONE launch for all of them.
    Same per-element operation order as the separate framework passes (bias add, residual add, ReLU, subtraction)."""
    x0 = xs[0]
    B, C = x0.size(0), x0.size(1)
    levels = (_BiasLevel * len(xs))()
    ys, zs = [], []
    keep = []
    for i, x in enumerate(xs):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == C):
            raise ValueError("bias_act_multi expects fp32 CUDA [B,C,H,W] tensors with equal B and C")
        x = x.detach().contiguous()
        r = None
        if residuals is not None:
            r = residuals[i].detach().contiguous()
            if r.shape != x.shape or r.dtype != torch.float32:
                raise ValueError("bias_act_multi: residual must match x")
            keep.append(r)
        z = torch.empty_like(x) if sub is not None else None
        ys.append(x); zs.append(z)
        levels[i] = _BiasLevel(x.data_ptr(), r.data_ptr() if r is not None else None, x.data_ptr(),
                               z.data_ptr() if z is not None else None, x.size(2), x.size(3))
    b = bias.detach().float().contiguous() if bias is not None else None
    s = sub.detach().float().reshape(-1).contiguous() if sub is not None else None
    if (b is not None and b.numel() != C) or (s is not None and s.numel() != C):
        raise ValueError("bias_act_multi: bias / sub must have C elements")
    with torch.cuda.device(x0.device):
        rc = _lib.lib().orp_bias_act_multi(levels, len(xs), B, C, _lib.ptr(b), _lib.ptr(s), 1 if relu else 0,
                                           _lib.stream_of(x0))
    _lib.check(rc, "orp_bias_act_multi")
    return (ys, zs) if sub is not None else ys


_packed_1x1 = _packcache.new_cache("conv1x1_weight")


def _packed_1x1_weight(weight):
    """[Cout,Cin,1,1] -> the [Cin][32] pack of orp_conv1x1_multi, cached on the live parameter (inference only)."""
    w = weight.detach()
    state = _packcache.tensor_state(w)
    hit = _packed_1x1.get(weight, state)
    if hit is not None:
        return _lib.keep_for_graph(hit)
    cout, cin = w.size(0), w.size(1)
    w2 = w.float().reshape(cout, cin).contiguous()
    packed = torch.empty((_lib.lib().orp_conv1x1_packed_floats(cin),), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().orp_conv1x1_pack_weight(_lib.ptr(w2), cout, cin, _lib.ptr(packed), _lib.stream_of(w2)),
                   "orp_conv1x1_pack_weight")
    return _lib.keep_for_graph(_packed_1x1.put(weight, state, packed))


def conv1x1_ok(conv, x):
    w = conv.weight
    return bool(x.is_cuda and x.dtype == torch.float32 and w.dim() == 4 and w.size(2) == 1 and w.size(3) == 1 and
                tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0) and conv.groups == 1 and
                _lib.lib().orp_conv1x1_ok(w.size(1), w.size(0)))


def conv1x1_multi(xs, conv, relu=False, residuals=None, sub=None):
    """ys[i] = relu?(conv(xs[i]) (+ residuals[i])) for a 1x1 nn.Conv2d with at most 32 output channels (bias included),
    all FPN levels in ONE launch; with `sub` ([Cout]) also returns zs[i] = ys[i] - sub.  Inference only (no autograd).
    Same epilogue order as the framework passes it replaces (bias add, residual add, ReLU, subtraction)."""
    x0 = xs[0]
    B, cin = x0.size(0), x0.size(1)
    cout = conv.weight.size(0)
    packed = _packed_1x1_weight(conv.weight)
    levels = (_BiasLevel * len(xs))()
    ys, zs, keep = [], [], []
    for i, x in enumerate(xs):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == cin):
            raise ValueError("conv1x1_multi expects fp32 CUDA [B,Cin,H,W] tensors with equal B and Cin")
        x = x.detach().contiguous()
        y = torch.empty((B, cout, x.size(2), x.size(3)), dtype=torch.float32, device=x.device)
        r = None
        if residuals is not None:
            r = residuals[i].detach().contiguous()
            if r.shape != y.shape or r.dtype != torch.float32:
                raise ValueError("conv1x1_multi: residual must match the output")
        z = torch.empty_like(y) if sub is not None else None
        keep += [x, r]
        ys.append(y); zs.append(z)
        levels[i] = _BiasLevel(x.data_ptr(), r.data_ptr() if r is not None else None, y.data_ptr(),
                               z.data_ptr() if z is not None else None, x.size(2), x.size(3))
    b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    s = sub.detach().float().reshape(-1).contiguous() if sub is not None else None
    if s is not None and s.numel() != cout:
        raise ValueError("conv1x1_multi: sub must have Cout elements")
    with torch.cuda.device(x0.device):
        rc = _lib.lib().orp_conv1x1_multi(levels, len(xs), B, cin, cout, _lib.ptr(packed), _lib.ptr(b), _lib.ptr(s),
                                          1 if relu else 0, _lib.stream_of(x0))
    _lib.check(rc, "orp_conv1x1_multi")
    return (ys, zs) if sub is not None else ys


SMALL_LEVEL_POSITIONS = 1024        # H*W up to which a level goes through conv3x3_multi's HIP kernel (32 x 32 at 1024^2)


def conv3x3_multi(xs, conv, split_k=False):
    """[conv(x) without bias for x in xs] for 3x3 / stride 1 or 2 / pad 1 / groups 1 nn.Conv2d modules, inference only
    (split_k: launches with few positions also split K over the grid, fixed-order sum of the partial images);
    conv: one module for all tensors or a list with one module per tensor (equal channel counts).  The small levels
    (H*W <= SMALL_LEVEL_POSITIONS) share ONE launch of the exact-fp32 MFMA kernel `orp_conv3x3_small_multi_strided` (the
    framework would issue an im2col + GEMM pair per level), the big levels stay on the library (Winograd)."""
    import torch.nn.functional as F
    from .deform_conv import _packed_weight
    convs = list(conv) if isinstance(conv, (list, tuple)) else [conv] * len(xs)
    w0 = convs[0].weight
    cout, cin = w0.size(0), w0.size(1)
    L = _lib.lib()

    def eligible(c):
        w = c.weight
        return (tuple(w.shape) == (cout, cin, 3, 3) and tuple(c.stride) in ((1, 1), (2, 2)) and tuple(c.padding) == (1, 1)
                and tuple(c.dilation) == (1, 1) and c.groups == 1 and w.dtype == torch.float32)
    ok = bool(L.orp_conv3x3_small_ok(cin, cout)) and all(eligible(c) for c in {id(c): c for c in convs}.values())
    outs = [None] * len(xs)
    small = []
    for i, x in enumerate(xs):
        if ok and x.is_cuda and x.dtype == torch.float32 and x.size(2) * x.size(3) <= SMALL_LEVEL_POSITIONS:
            small.append(i)
        else:
            c = convs[i]
            outs[i] = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
    if small:
        B = xs[small[0]].size(0)
        levels = (_NormLevel * len(small))()
        wts = (ctypes.c_void_p * len(small))()
        strides = (ctypes.c_int * len(small))()
        keep, seen = [], {}
        for k, i in enumerate(small):
            x = xs[i].detach().contiguous()
            st = int(convs[i].stride[0])
            y = torch.empty((B, cout, (x.size(2) - 1) // st + 1, (x.size(3) - 1) // st + 1), dtype=torch.float32,
                            device=x.device)
            wp = seen.get(id(convs[i]))
            if wp is None:                                 # once per distinct module
                packed = _packed_weight(convs[i].weight)
                keep.append(packed)
                wp = seen[id(convs[i])] = packed.data_ptr()
            keep.append(x); outs[i] = y
            levels[k] = _NormLevel(x.data_ptr(), y.data_ptr(), x.size(2), x.size(3))
            wts[k] = wp
            strides[k] = st
        x0 = xs[small[0]]
        nbytes = L.orp_conv3x3_small_workspace_bytes(levels, strides, len(small), B, cout) if split_k else 0
        ws = _lib.workspace(x0.device, nbytes) if nbytes else None
        with torch.cuda.device(x0.device):
            rc = L.orp_conv3x3_small_multi_strided(levels, wts, strides, len(small), B, cin, cout, _lib.ptr(ws),
                                                   ws.numel() if ws is not None else 0, _lib.stream_of(x0))
        _lib.check(rc, "orp_conv3x3_small_multi_strided")
    return outs


class _GroupNormActTrain(torch.autograd.Function):
    """ys[i] = relu?(GroupNorm(xs[i])) for n tensors (the FPN levels of one tower layer) with autograd: forward = the
    inference launch pair + stored (mean, rstd); backward = two launches for every grad_input + one per distinct module
    for (dgamma, dbeta), all sums in a fixed order.  Inputs: n, groups, eps, relu, owner (tensor i uses parameter set
    owner[i]), then the n tensors, then the distinct gammas, then the distinct betas."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, n, groups, eps, relu, owner, *tensors):
        L = _lib.lib()
        xs = [t.detach().contiguous() for t in tensors[:n]]
        m = (len(tensors) - n) // 2
        gammas = [t.detach().float().contiguous() for t in tensors[n:n + m]]
        betas = [t.detach().float().contiguous() for t in tensors[n + m:]]
        x0 = xs[0]
        B, C = x0.size(0), x0.size(1)
        levels = (_NormLevel * n)()
        gam = (ctypes.c_void_p * n)()
        bet = (ctypes.c_void_p * n)()
        ys = []
        for i, x in enumerate(xs):
            if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == C):
                raise ValueError("group_norm_act_train expects fp32 CUDA [B,C,H,W] tensors with equal B and C")
            y = torch.empty_like(x)
            ys.append(y)
            levels[i] = _NormLevel(x.data_ptr(), y.data_ptr(), x.size(2), x.size(3))
            gam[i], bet[i] = gammas[owner[i]].data_ptr(), betas[owner[i]].data_ptr()
        stats = torch.empty((n * B * groups, 2), dtype=torch.float32, device=x0.device)
        nbytes = L.orp_groupnorm_workspace_bytes(levels, n, B, C, groups)
        ws = _lib.workspace(x0.device, nbytes)
        with torch.cuda.device(x0.device):
            rc = L.orp_groupnorm_act_multi_train(levels, gam, bet, n, B, C, groups, float(eps), 1 if relu else 0,
                                                 _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
        _lib.check(rc, "orp_groupnorm_act_multi_train")
        ctx.save_for_backward(stats, *xs, *gammas, *betas, *ys)      # ys: the ReLU mask (the next layer keeps them alive anyway)
        ctx.meta = (n, m, groups, bool(relu), tuple(owner))
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        L = _lib.lib()
        n, m, groups, relu, owner = ctx.meta
        saved = ctx.saved_tensors
        stats, xs, gammas, betas = saved[0], saved[1:1 + n], saved[1 + n:1 + n + m], saved[1 + n + m:1 + n + 2 * m]
        ys = saved[1 + n + 2 * m:]
        x0 = xs[0]
        B, C = x0.size(0), x0.size(1)
        levels = (_NormLevel * n)()
        dys = (ctypes.c_void_p * n)()
        dxs = (ctypes.c_void_p * n)()
        gam = (ctypes.c_void_p * n)()
        bet = (ctypes.c_void_p * n)()
        dg = (ctypes.c_void_p * n)()
        db = (ctypes.c_void_p * n)()
        keep, gxs = [], []
        dgam = [torch.empty_like(g) for g in gammas]
        dbet = [torch.empty_like(b) for b in betas]
        for i, x in enumerate(xs):
            g = grads[i]
            g = torch.zeros_like(x) if g is None else g.detach().float().contiguous()
            gx = torch.empty_like(x)
            keep.append(g); gxs.append(gx)
            levels[i] = _NormLevel(x.data_ptr(), ys[i].data_ptr(), x.size(2), x.size(3))
            dys[i], dxs[i] = g.data_ptr(), gx.data_ptr()
            gam[i], bet[i] = gammas[owner[i]].data_ptr(), betas[owner[i]].data_ptr()
            dg[i], db[i] = dgam[owner[i]].data_ptr(), dbet[owner[i]].data_ptr()
        nbytes = L.orp_groupnorm_backward_workspace_bytes(levels, n, B, C, groups)
        ws = _lib.workspace(x0.device, nbytes)
        with torch.cuda.device(x0.device):
            rc = L.orp_groupnorm_act_multi_backward(levels, dys, dxs, gam, bet, dg, db, n, B, C, groups, 1 if relu else 0,
                                                    _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
        _lib.check(rc, "orp_groupnorm_act_multi_backward")
        return (None, None, None, None, None) + tuple(gxs) + tuple(dgam) + tuple(dbet)


def group_norm_act_train(xs, gn, relu=True):
    """[relu?(GroupNorm(x)) for x in xs] with autograd, ONE launch pair forward for all tensors (up to 16).  gn: one
    nn.GroupNorm for all of them or a list with one module per tensor (equal num_groups / eps; modules may repeat)."""
    gns = list(gn) if isinstance(gn, (list, tuple)) else [gn] * len(xs)
    if len(gns) != len(xs) or any(g.num_groups != gns[0].num_groups or g.eps != gns[0].eps for g in gns):
        raise ValueError("group_norm_act_train: one GroupNorm per tensor, equal num_groups / eps")
    distinct, owner = [], []
    for g in gns:
        for k, d in enumerate(distinct):
            if d is g:
                owner.append(k)
                break
        else:
            owner.append(len(distinct)); distinct.append(g)
    outs = _GroupNormActTrain.apply(len(xs), gns[0].num_groups, gns[0].eps, bool(relu), tuple(owner), *xs,
                                    *[d.weight for d in distinct], *[d.bias for d in distinct])
    return list(outs)


# ---- channels-last tower path (round 4): conv_split_multi -> group_norm_act_multi_cl -> ... -----------------------------------

class _ConvLevel(ctypes.Structure):
    _fields_ = [("input_a", ctypes.c_void_p), ("input_b", ctypes.c_void_p), ("output_a", ctypes.c_void_p),
                ("output_b", ctypes.c_void_p), ("height", ctypes.c_int), ("width", ctypes.c_int)]


def _is_cl(x):
    """memory is [B, H, W, C] (channels-last strides of the logical [B, C, H, W] tensor)"""
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


def _ranges_wanted():
    """the library's arithmetic mode is the fp16-pieces one (orp_dcn_get_split_mode() == 3): producers leave their ranges"""
    return _lib.lib().orp_dcn_get_split_mode() == 3


class Amax(object):
    """Device-side range hand-over between a producer and the fp16-pieces convolution that reads its outputs: `bits` is an int32
    CUDA tensor of float bits (upper bounds of max |x| per slot, written by atomicMax), `stride` 0 = both layers of a pair
    launch read slot 0, 1 = layer b reads slot 1.  None anywhere in the chain simply means the convolution takes the maximum
    itself (a pre-pass over its inputs)."""
    __slots__ = ('bits', 'stride', 'count')

    def __new__(cls, bits, stride=0, count=1):
        if bits is None:
            return None
        return object.__new__(cls)

    def __init__(self, bits, stride=0, count=1):
        # count > 1 (conv_split_gn's hand-over): layer k's range is the maximum of `count` words at bits[k * stride]
        self.bits, self.stride, self.count = bits, stride, count


def to_channels_last_multi(xs, amax_slots=None, amax_into=None, force_ranges=False):
    """[x in channels-last memory for x in xs] -- NCHW-contiguous fp32 CUDA tensors [B,C,H,W] with equal B and C, ONE launch
    (`orp_nchw_to_nhwc_multi`); tensors that already are channels-last pass through.
    amax_slots (one slot index per tensor): also returns an int32 tensor with max |x| per slot as float bits, or None when a
    tensor passed through untouched (its range is not known here); amax_into = (bits tensor, slots): accumulate into an
    existing one instead (all tensors of xs must then be converted here or the result is None again)."""
    L = _lib.lib()
    todo = [i for i, x in enumerate(xs) if not _is_cl(x)]
    outs = list(xs)
    want_amax = amax_slots is not None or amax_into is not None
    ranges = want_amax and (force_ranges or _ranges_wanted())      # (only the fp16-pieces kernels read them)
    if not todo:
        return (outs, None) if want_amax else outs
    x0 = xs[todo[0]]
    B, C = x0.size(0), x0.size(1)
    levels = (_NormLevel * len(todo))()
    keep = []
    for k, i in enumerate(todo):
        x = xs[i].detach()
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == C):
            raise ValueError("to_channels_last_multi expects fp32 CUDA [B,C,H,W] tensors with equal B and C")
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        keep.append(x); outs[i] = y
        levels[k] = _NormLevel(x.data_ptr(), y.data_ptr(), x.size(2), x.size(3))
    if ranges and len(todo) == len(xs):
        if amax_into is not None:
            bits, slots_all, reset = amax_into[0], amax_into[1], 0
        else:
            slots_all, reset = list(amax_slots), 1
            bits = torch.empty(max(slots_all) + 1, dtype=torch.int32, device=x0.device)
        slots = (ctypes.c_int * len(todo))(*[int(slots_all[i]) for i in todo])
        with torch.cuda.device(x0.device):
            rc = L.orp_nchw_to_nhwc_multi_amax(levels, len(todo), B, C, slots, bits.data_ptr(), bits.numel(), reset,
                                               _lib.stream_of(x0))
        _lib.check(rc, "orp_nchw_to_nhwc_multi_amax")
        return outs, bits
    with torch.cuda.device(x0.device):
        rc = L.orp_nchw_to_nhwc_multi(levels, len(todo), B, C, _lib.stream_of(x0))
    _lib.check(rc, "orp_nchw_to_nhwc_multi")
    return (outs, None) if want_amax else outs


def conv_split_ok(conv, x=None):
    """the module is a bias-free-or-not kh x kw nn.Conv2d (groups 1, fp32) of a shape `orp_conv_split_multi` takes"""
    w = conv.weight
    ok = (isinstance(conv, torch.nn.Conv2d) and conv.groups == 1 and w.dtype == torch.float32 and w.is_cuda and
          isinstance(conv.padding, tuple) and
          bool(_lib.lib().orp_conv_split_ok(w.size(1), w.size(0), w.size(2), w.size(3))))
    if ok and x is not None:
        ok = x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(1) == w.size(1)
    return ok


def conv_split_weights(xs_a, weights_a, xs_b=None, weight_b=None, biases_a=None, bias_b=None, stride=(1, 1), padding=(1, 1),
                       dilation=(1, 1), relu=False, out_channels_last=True, nprod=None, cache_pack=True, amax=None):
    """The launch behind conv_split_multi, on tensors: weights_a = one [Cout,Cin,kh,kw] weight for all of xs_a, or a list with
    one weight per tensor (then no second layer); cache_pack=False re-packs the weights on every call (training: the
    optimizer writes them between calls); amax: an `Amax` from the producer of the inputs (fp16-pieces mode: no range
    pre-pass), or None."""
    from .deform_conv import _packed_weight
    L = _lib.lib()
    pair = xs_b is not None
    per_level = list(weights_a) if isinstance(weights_a, (list, tuple)) else None
    w = per_level[0] if per_level is not None else weights_a
    cout, cin, kh, kw = w.shape
    if per_level is not None and (pair or len(per_level) != len(xs_a) or any(tuple(t.shape) != tuple(w.shape) for t in per_level)):
        raise ValueError("conv_split: one weight per tensor, equal shapes, no second layer")
    if pair and (tuple(weight_b.shape) != tuple(w.shape) or len(xs_b) != len(xs_a)):
        raise ValueError("conv_split: the two layers must have equal shapes")
    if nprod is None:
        nprod = L.orp_dcn_get_split_mode() or 6
    x0 = xs_a[0]
    B = x0.size(0)
    n = len(xs_a)
    levels = (_ConvLevel * n)()
    keep, outs_a, outs_b = [], [], []
    st, pd, dl = tuple(stride), tuple(padding), tuple(dilation)
    fmt = torch.channels_last if out_channels_last else torch.contiguous_format
    for i in range(n):
        xa = xs_a[i].detach()
        xb = xs_b[i].detach() if pair else None
        for x in (xa, xb):
            if x is None:
                continue
            if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == cin and _is_cl(x)):
                raise ValueError("conv_split expects channels-last fp32 CUDA [B,%d,H,W] tensors" % cin)
            if xb is not None and x.shape != xa.shape:
                raise ValueError("conv_split: the two layers' inputs must have equal shapes")
        H, W = xa.size(2), xa.size(3)
        ho = (H + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
        wo = (W + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
        oa = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=xa.device, memory_format=fmt)
        ob = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=xa.device, memory_format=fmt) if pair else None
        keep += [xa, xb]; outs_a.append(oa); outs_b.append(ob)
        levels[i] = _ConvLevel(xa.data_ptr(), xb.data_ptr() if pair else None, oa.data_ptr(), ob.data_ptr() if pair else None, H, W)

    def f32(t):
        return t.detach().float().contiguous() if t is not None else None
    ws = _lib.workspace(x0.device, 256)                    # nprod = 3: max |x| of the inputs (a pre-pass writes it there)
    am_ptr = amax.bits.data_ptr() if amax is not None else None
    am_stride = int(amax.stride) if amax is not None else 0
    keep.append(amax.bits if amax is not None else None)
    if per_level is not None:
        wts = (ctypes.c_void_p * n)()
        bs = (ctypes.c_void_p * n)()
        for i, wt in enumerate(per_level):
            pk = _packed_weight(wt, cache_pack)
            bi = f32(biases_a[i]) if biases_a is not None else None
            keep += [pk, bi]
            wts[i] = pk.data_ptr()
            bs[i] = bi.data_ptr() if bi is not None else None
        with torch.cuda.device(x0.device):
            rc = L.orp_conv_split_multi_ex(levels, wts, bs, n, B, cin, cout, 1 if relu else 0, kh, kw, st[0], st[1], pd[0], pd[1],
                                           dl[0], dl[1], 1 if out_channels_last else 0, int(nprod), _lib.ptr(ws), ws.numel(),
                                           am_ptr, _lib.stream_of(x0))
        _lib.check(rc, "orp_conv_split_multi_ex")
        return outs_a
    pa = _packed_weight(w, cache_pack)
    pb = _packed_weight(weight_b, cache_pack) if pair else None
    ba, bb = f32(biases_a), f32(bias_b) if pair else None
    with torch.cuda.device(x0.device):
        rc = L.orp_conv_split_multi(levels, n, B, cin, cout, _lib.ptr(pa), _lib.ptr(pb), _lib.ptr(ba), _lib.ptr(bb),
                                    1 if relu else 0, kh, kw, st[0], st[1], pd[0], pd[1], dl[0], dl[1],
                                    1 if out_channels_last else 0, int(nprod), _lib.ptr(ws), ws.numel(), am_ptr, am_stride,
                                    _lib.stream_of(x0))
    _lib.check(rc, "orp_conv_split_multi")
    return (outs_a, outs_b) if pair else outs_a


def conv_split_multi(xs_a, conv_a, xs_b=None, conv_b=None, bias=False, relu=False, out_channels_last=True, nprod=None, amax=None):
    """[conv_a(x) for x in xs_a] (and [conv_b(x) for x in xs_b]: two layers of equal shape in ONE launch -- the two towers'
    layer k) over all FPN levels, `orp_conv_split_multi`: fp32 on the bf16 matrix pipe with every operand split exactly
    into three bf16 pieces, fp32 accumulation.  xs_*: channels-last fp32 CUDA tensors (logical [B,C,H,W]); conv_a: one
    nn.Conv2d for all tensors, or a list with one per tensor (the FPN's output convolutions; no second layer then); bias=True
    adds the modules' biases in the epilogue, relu fuses the activation; outputs channels-last or NCHW.  No autograd
    (training: conv_split_train).  nprod: 6 | 9 partial products (None: the library's DeformConv split mode, 6 when off)."""
    per_level = list(conv_a) if isinstance(conv_a, (list, tuple)) else None
    c0 = per_level[0] if per_level is not None else conv_a
    mods = (per_level or [c0]) + ([conv_b] if conv_b is not None else [])
    if any(c.stride != c0.stride or c.padding != c0.padding or c.dilation != c0.dilation for c in mods):
        raise ValueError("conv_split_multi: equal strides / paddings / dilations")
    if per_level is not None:
        return conv_split_weights(xs_a, [c.weight for c in per_level], None, None,
                                  [c.bias for c in per_level] if bias else None, None, c0.stride, c0.padding, c0.dilation,
                                  relu, out_channels_last, nprod, amax=amax)
    return conv_split_weights(xs_a, c0.weight, xs_b, conv_b.weight if conv_b is not None else None,
                              c0.bias if bias else None, conv_b.bias if (bias and conv_b is not None) else None,
                              c0.stride, c0.padding, c0.dilation, relu, out_channels_last, nprod, amax=amax)


def conv_split_gn_ok(conv_a, conv_b, gn_a, gn_b, x):
    """the two towers' layer k can run as `conv_split_gn`: stride-1 'same' bias-free convolutions of one shape that
    orp_conv_split_multi takes, at most 512 input channels, affine GroupNorms with equal groups / eps whose group size divides 32"""
    import torch.nn as nn
    for c in (conv_a, conv_b):
        if not (conv_split_ok(c, x) and c.bias is None and tuple(c.stride) == (1, 1) and c.weight.size(1) <= 512 and
                tuple(c.weight.shape) == tuple(conv_a.weight.shape) and c.padding == conv_a.padding and c.dilation == conv_a.dilation and
                2 * c.padding[0] == c.dilation[0] * (c.weight.size(2) - 1) and 2 * c.padding[1] == c.dilation[1] * (c.weight.size(3) - 1)):
            return False
    cout = conv_a.weight.size(0)
    for g in (gn_a, gn_b):
        if not (isinstance(g, nn.GroupNorm) and g.affine and g.num_groups == gn_a.num_groups and g.eps == gn_a.eps and
                g.num_channels == cout and cout % g.num_groups == 0 and 32 % (cout // g.num_groups) == 0 and 1024 % cout == 0):
            return False
    return True


def conv_split_gn(xs_a, conv_a, xs_b, conv_b, gn_a, gn_b, coef_in=None, amax=None, nprod=None, materialize=False):
    """One layer of BOTH towers -- conv -> GroupNorm -> ReLU over all FPN levels (reference head :91-113, ConvModule) -- with the
    normalisation fused AROUND the convolution launch (`orp_conv_split_multi_gn`): the statistics leave the convolution's epilogue
    per tile, `orp_conv_split_gn_finish` merges them into per-(tensor, image, channel) coefficients (a, b), and the NEXT layer
    applies relu(x a + b) while it reads.  xs_*: channels-last fp32 tensors; coef_in: the previous layer's coefficients (None: the
    inputs are taken as they are).  Returns (outs_a, outs_b, coef, amax): RAW convolution outputs + their coefficients -- or, with
    materialize=True (the towers' last layer), the normalised + ReLU'd outputs (in place) and coef None.  amax: `Amax` of the
    normalised outputs for the fp16-pieces arithmetic (None in the other modes)."""
    from .deform_conv import _packed_weight
    L = _lib.lib()
    n = len(xs_a)
    w = conv_a.weight
    cout, cin, kh, kw = w.shape
    if nprod is None:
        nprod = L.orp_dcn_get_split_mode() or 6
    x0 = xs_a[0]
    B = x0.size(0)
    levels = (_ConvLevel * n)()
    outs_a, outs_b, keep = [], [], []
    for i in range(n):
        xa, xb = xs_a[i].detach(), xs_b[i].detach()
        for x in (xa, xb):
            if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == cin and _is_cl(x) and
                    x.shape == xa.shape):
                raise ValueError("conv_split_gn expects channels-last fp32 CUDA [B,%d,H,W] tensors" % cin)
        H, W = xa.size(2), xa.size(3)
        oa = torch.empty((B, cout, H, W), dtype=torch.float32, device=xa.device, memory_format=torch.channels_last)
        ob = torch.empty((B, cout, H, W), dtype=torch.float32, device=xa.device, memory_format=torch.channels_last)
        keep += [xa, xb]; outs_a.append(oa); outs_b.append(ob)
        levels[i] = _ConvLevel(xa.data_ptr(), xb.data_ptr(), oa.data_ptr(), ob.data_ptr(), H, W)
    G = gn_a.num_groups
    pf = int(L.orp_conv_split_gn_partial_floats(levels, n, B, G, 2))
    partials = torch.empty((pf,), dtype=torch.float32, device=x0.device)
    ws = _lib.workspace(x0.device, 256)
    pa, pb = _packed_weight(conv_a.weight), _packed_weight(conv_b.weight)
    am_ptr = amax.bits.data_ptr() if amax is not None else None
    pd, dl = conv_a.padding, conv_a.dilation
    with torch.cuda.device(x0.device):
        rc = L.orp_conv_split_multi_gn(levels, n, B, cin, cout, _lib.ptr(pa), _lib.ptr(pb), kh, kw, pd[0], pd[1], dl[0], dl[1], int(nprod),
                                       _lib.ptr(coef_in), 1, _lib.ptr(partials), pf, G, _lib.ptr(ws), ws.numel(), am_ptr,
                                       int(amax.stride) if amax is not None else 0, int(amax.count) if amax is not None else 0,
                                       _lib.stream_of(x0))
    _lib.check(rc, "orp_conv_split_multi_gn")
    coef = torch.empty((2 * n, B, cout, 2), dtype=torch.float32, device=x0.device)
    ranges = _ranges_wanted()
    per_set = n * B * G
    bound = torch.empty((2, per_set), dtype=torch.int32, device=x0.device) if ranges else None
    gam = (ctypes.c_void_p * (2 * n))()
    bet = (ctypes.c_void_p * (2 * n))()
    for k, g in enumerate((gn_a, gn_b)):
        g_, b_ = g.weight.detach().float().contiguous(), g.bias.detach().float().contiguous()
        keep += [g_, b_]
        for i in range(n):
            gam[k * n + i], bet[k * n + i] = g_.data_ptr(), b_.data_ptr()
    with torch.cuda.device(x0.device):
        rc = L.orp_conv_split_gn_finish(levels, n, B, cout, G, 2, float(gn_a.eps), gam, bet, _lib.ptr(partials), _lib.ptr(coef),
                                        bound.data_ptr() if bound is not None else None, _lib.stream_of(x0))
    _lib.check(rc, "orp_conv_split_gn_finish")
    if not materialize:
        return outs_a, outs_b, coef, (Amax(bound, per_set, per_set) if bound is not None else None)
    nl = (_NormLevel * (2 * n))()
    for i, t in enumerate(outs_a + outs_b):
        nl[i] = _NormLevel(t.data_ptr(), t.data_ptr(), t.size(2), t.size(3))
    slots = torch.empty((2,), dtype=torch.int32, device=x0.device) if bound is not None else None
    with torch.cuda.device(x0.device):
        rc = L.orp_affine_act_multi_cl(nl, 2 * n, B, cout, _lib.ptr(coef), 1, bound.data_ptr() if bound is not None else None, 2,
                                       per_set, slots.data_ptr() if slots is not None else None, _lib.stream_of(x0))
    _lib.check(rc, "orp_affine_act_multi_cl")
    return outs_a, outs_b, None, (Amax(slots, 1) if slots is not None else None)


class _WgradLevel(ctypes.Structure):
    _fields_ = [("input", ctypes.c_void_p), ("grad_output", ctypes.c_void_p), ("height", ctypes.c_int), ("width", ctypes.c_int)]


def conv_wgrad_split_ok(weight, padding, dilation):
    cout, cin, kh, kw = weight.shape
    return (bool(_lib.lib().orp_conv_wgrad_split_ok(cin, cout, kh, kw)) and weight.is_cuda and weight.dtype == torch.float32 and
            2 * padding[0] == dilation[0] * (kh - 1) and 2 * padding[1] == dilation[1] * (kw - 1))


def conv_wgrad_split(xs, grad_outs, weight_shape, padding=(1, 1), dilation=(1, 1), amax_x=None, amax_g=None):
    """grad_weight [Cout,Cin,kh,kw] of a stride-1 'same' 256 -> 256 convolution summed over a list of (input, grad_output) pairs
    (the FPN levels of one layer), `orp_conv_wgrad_split`: NCHW fp32 CUDA tensors, ONE launch, fixed summation order.
    amax_x / amax_g: one-element int32 tensors (float bits of an upper bound of max |input| / max |grad_output|), or None."""
    L = _lib.lib()
    cout, cin, kh, kw = weight_shape
    x0 = xs[0]
    B = x0.size(0)
    n = len(xs)
    levels = (_WgradLevel * n)()
    keep = []
    for i in range(n):
        x, g = xs[i].detach().float().contiguous(), grad_outs[i].detach().float().contiguous()
        if not (x.is_cuda and x.dim() == 4 and x.size(0) == B and x.size(1) == cin and tuple(g.shape) == (B, cout, x.size(2), x.size(3))):
            raise ValueError("conv_wgrad_split expects NCHW fp32 CUDA inputs [B,%d,H,W] and grad_outputs [B,%d,H,W]" % (cin, cout))
        keep += [x, g]
        levels[i] = _WgradLevel(x.data_ptr(), g.data_ptr(), x.size(2), x.size(3))
    gw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=x0.device)
    nbytes = L.orp_conv_wgrad_split_workspace_bytes(levels, n, B, kh, kw)
    ws = _lib.workspace(x0.device, nbytes)
    have = amax_x is not None and amax_g is not None
    with torch.cuda.device(x0.device):
        rc = L.orp_conv_wgrad_split(levels, n, B, cin, cout, kh, kw, padding[0], padding[1], dilation[0], dilation[1],
                                    amax_x.data_ptr() if have else None, amax_g.data_ptr() if have else None,
                                    gw.data_ptr(), _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
    _lib.check(rc, "orp_conv_wgrad_split")
    return gw


class _ConvSplitTrain(torch.autograd.Function):
    """ys = [conv(x, W_k)] for the levels of one tower layer (or of the two towers' layer k, or of the FPN's output
    convolutions with a weight per level) as ONE autograd node on the bf16-split kernel.  forward: inputs transposed to
    channels-last in one launch, one convolution launch writing NCHW.  backward: grad_input = the same kernel on the
    transposed grad_outputs with the flipped, transposed weights (stride 1: correlation with W[o][c][kh-1-i][kw-1-j] as
    [c][o], padding dil*(k-1) - pad); grad_weight = the library's weight-gradient kernel per level on the channels-last
    tensors (torch.ops.aten.convolution_backward), summed over the levels that share a weight."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        n, nw, groups, padding, dilation = meta             # n inputs, nw weights, groups[i] = weight index of input i
        ws, xs = tensors[:nw], tensors[nw:]
        def key(x):                                           # the towers' first layer reads the same FPN output twice
            return (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
        seen = {}
        for x in xs:
            seen.setdefault(key(x), len(seen))
        uniq = [None] * len(seen)
        for x in xs:
            uniq[seen[key(x)]] = x
        # the transposition also leaves max |x| (fp16-pieces mode: no range pre-pass in the convolution): one slot per layer
        # of a pair launch -- or one for everything when the two layers read the same tensors
        pair = nw == 2 and n % 2 == 0 and list(groups) == [0] * (n // 2) + [1] * (n // 2)
        split_slots = pair and len(uniq) == n
        slots = [(1 if (split_slots and i >= n // 2) else 0) for i in range(len(uniq))]
        uniq_cl, bits = to_channels_last_multi([u.detach().float() for u in uniq], amax_slots=slots, force_ranges=True)
        cl = [uniq_cl[seen[key(x)]] for x in xs]
        amax = Amax(bits, 1 if split_slots else 0) if bits is not None else None
        outs = _ConvSplitTrain._run(cl, ws, groups, padding, dilation, nw, amax)
        ctx.meta = meta
        ctx.x_bits = (bits, split_slots)                      # ranges of the inputs, for the weight-gradient kernel
        # the weight gradient reads ONE copy of the inputs: NCHW for the layers that take orp_conv_wgrad_split, channels-last for
        # the library's kernel -- only what the chosen routes need is kept (both copies doubled the saved activations)
        routes = [_ConvSplitTrain._wgrad_split_route(ws[k], [i for i in range(n) if groups[i] == k], padding, dilation) for k in range(nw)]
        ctx.routes = tuple(routes)
        keep_cl = cl if not all(routes) else []
        keep_x = [x.detach() for x in xs] if any(routes) else []
        ctx.kept = (len(keep_cl), len(keep_x))
        ctx.save_for_backward(*ws, *keep_cl, *keep_x)
        return tuple(outs)

    @staticmethod
    def _wgrad_split_route(w, idx, padding, dilation):
        return len(idx) <= 8 and conv_wgrad_split_ok(w, padding, dilation)

    @staticmethod
    def _run(cl, ws, groups, padding, dilation, nw, amax=None):
        """one launch: a single weight, two weights (pair: first / second half of the tensors), or one weight per tensor"""
        n = len(cl)
        if nw == 1:
            return conv_split_weights(cl, ws[0], padding=padding, dilation=dilation, out_channels_last=False, cache_pack=False,
                                      amax=amax)
        if nw == 2 and n % 2 == 0 and list(groups) == [0] * (n // 2) + [1] * (n // 2):
            a, b = conv_split_weights(cl[:n // 2], ws[0], cl[n // 2:], ws[1], padding=padding, dilation=dilation,
                                      out_channels_last=False, cache_pack=False, amax=amax)
            return a + b
        if amax is not None and amax.stride != 0:
            amax = None                                      # (a layer per tensor reads one slot for all of them)
        return conv_split_weights(cl, [ws[g] for g in groups], padding=padding, dilation=dilation, out_channels_last=False,
                                  cache_pack=False, amax=amax)

    @staticmethod
    def backward(ctx, *grads):
        n, nw, groups, padding, dilation = ctx.meta
        saved = ctx.saved_tensors
        ncl, nx = ctx.kept
        ws, cl, xs_nchw = saved[:nw], saved[nw:nw + ncl], saved[nw + ncl:nw + ncl + nx]
        kh, kw = ws[0].size(2), ws[0].size(3)
        pair = nw == 2 and n % 2 == 0 and list(groups) == [0] * (n // 2) + [1] * (n // 2)
        g_cl, bits = to_channels_last_multi([g.detach().float() for g in grads],
                                            amax_slots=[(1 if (pair and i >= n // 2) else 0) for i in range(n)], force_ranges=True)
        gxs = [None] * n
        if any(ctx.needs_input_grad[1 + nw:]):
            wt = [w.detach().flip(2, 3).transpose(0, 1).contiguous() for w in ws]
            pad_t = (dilation[0] * (kh - 1) - padding[0], dilation[1] * (kw - 1) - padding[1])
            gxs = _ConvSplitTrain._run(g_cl, wt, groups, pad_t, dilation, nw,
                                       Amax(bits, 1 if pair else 0) if bits is not None else None)
        gws = [None] * nw
        for k in range(nw):
            if not ctx.needs_input_grad[1 + k]:
                continue
            idx = [i for i in range(n) if groups[i] == k]
            if ctx.routes[k]:
                # one launch for the layer's levels, straight from the NCHW tensors (positions = the contraction axis)
                xb, x_split = ctx.x_bits
                sx = 1 if (x_split and idx[0] >= n // 2) else 0
                sg = 1 if (pair and idx[0] >= n // 2) else 0
                have = xb is not None and bits is not None
                gws[k] = conv_wgrad_split([xs_nchw[i] for i in idx], [grads[i] for i in idx], tuple(ws[k].shape), padding, dilation,
                                          xb[sx:sx + 1] if have else None, bits[sg:sg + 1] if have else None)
                continue
            acc = None
            for i in idx:
                gw = torch.ops.aten.convolution_backward(g_cl[i], cl[i], ws[k], None, [1, 1], list(padding), list(dilation), False,
                                                         [0, 0], 1, [False, True, False])[1]
                acc = gw if acc is None else acc + gw
            gws[k] = acc
        return (None, *gws, *gxs)


def conv_split_train_ok(convs, x, allow_bias=False):
    """the modules' convolutions can run as a conv_split_train node: split mode on (ORP_TRAIN_SPLIT=0 switches the training
    route off for A/B timing), no autocast, stride 1, fp32 nn.Conv2d of one shape that `orp_conv_split_multi` takes (bias-free
    unless the caller adds the bias itself), fp32 CUDA input"""
    from .. import switches
    if not switches.TRAIN_SPLIT or _lib.lib().orp_dcn_get_split_mode() == 0 or torch.is_autocast_enabled():
        return False
    c0 = convs[0]
    for c in convs:
        if not (conv_split_ok(c, x) and (allow_bias or c.bias is None) and tuple(c.stride) == (1, 1) and tuple(c.weight.shape) == tuple(c0.weight.shape)
                and c.padding == c0.padding and c.dilation == c0.dilation and c.weight.size(0) % 64 == 0 and c.weight.size(1) % 64 == 0
                and c.dilation[0] * (c.weight.size(2) - 1) >= c.padding[0] and c.dilation[1] * (c.weight.size(3) - 1) >= c.padding[1]):
            return False
    return True


def conv_split_train(xs, convs):
    """[convs[i](x_i)] with autograd as one node; xs: NCHW fp32 CUDA tensors (up to 8 per distinct layer in a pair / single
    launch, see _ConvSplitTrain._run), convs: one module, or a list with one module per tensor (tensors sharing a module are
    one layer).  Layouts a launch takes: all tensors one layer; first half / second half two layers (the two towers' layer
    k); a layer per tensor (the FPN)."""
    convs = list(convs) if isinstance(convs, (list, tuple)) else [convs] * len(xs)
    order, groups = {}, []
    for c in convs:
        if id(c) not in order:
            order[id(c)] = len(order)
        groups.append(order[id(c)])
    mods = [None] * len(order)
    for c in convs:
        mods[order[id(c)]] = c
    c0 = mods[0]
    meta = (len(xs), len(mods), tuple(groups), tuple(c0.padding), tuple(c0.dilation))
    return list(_ConvSplitTrain.apply(meta, *[m.weight for m in mods], *xs))


def group_norm_act_multi_cl(xs, gn, relu=True, inplace=True, amax_slots=None):
    """[GroupNorm(+ReLU)(x) for x in xs] for CHANNELS-LAST fp32 CUDA tensors (logical [B,C,H,W], memory [B,H,W,C]; up to 16:
    both towers' levels), results channels-last -- `orp_groupnorm_act_multi_cl`, three launches for all tensors.  gn: one
    nn.GroupNorm or a list with one per tensor (equal num_groups / eps).  amax_slots (a slot index per tensor): returns
    (outs, int32 tensor of float bits: an upper bound of max |y| per slot, from the statistics pass) for `Amax`."""
    L = _lib.lib()
    x0 = xs[0]
    B, C = x0.size(0), x0.size(1)
    gns = list(gn) if isinstance(gn, (list, tuple)) else [gn] * len(xs)
    if len(gns) != len(xs) or any(g.num_groups != gns[0].num_groups or g.eps != gns[0].eps
                                  for g in {id(g): g for g in gns}.values()):
        raise ValueError("group_norm_act_multi_cl: one GroupNorm per tensor, equal num_groups / eps")
    G = gns[0].num_groups
    if 1024 % C != 0 or C % G != 0 or (C // G) % 4 != 0:
        raise ValueError("group_norm_act_multi_cl: 1024 % channels == 0 and (channels / groups) % 4 == 0")
    levels = (_NormLevel * len(xs))()
    gam = (ctypes.c_void_p * len(xs))()
    bet = (ctypes.c_void_p * len(xs))()
    outs, keep, seen = [], [], {}
    for i, x in enumerate(xs):
        x = x.detach()
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(0) == B and x.size(1) == C and _is_cl(x)):
            raise ValueError("group_norm_act_multi_cl expects channels-last fp32 CUDA [B,C,H,W] tensors with equal B and C")
        y = x if inplace else torch.empty_like(x, memory_format=torch.channels_last)
        keep.append(x); outs.append(y)
        levels[i] = _NormLevel(x.data_ptr(), y.data_ptr(), x.size(2), x.size(3))
        ptrs = seen.get(id(gns[i]))
        if ptrs is None:
            g_ = gns[i].weight.detach().float().contiguous()
            b_ = gns[i].bias.detach().float().contiguous()
            keep += [g_, b_]
            ptrs = seen[id(gns[i])] = (g_.data_ptr(), b_.data_ptr())
        gam[i], bet[i] = ptrs
    nbytes = L.orp_groupnorm_cl_workspace_bytes(levels, len(xs), B, C, G)
    ws = _lib.workspace(x0.device, nbytes)
    if amax_slots is not None and not _ranges_wanted():
        amax_slots, no_ranges = None, True
    else:
        no_ranges = False
    if amax_slots is not None:
        slots = (ctypes.c_int * len(xs))(*[int(v) for v in amax_slots])
        bits = torch.empty(max(int(v) for v in amax_slots) + 1, dtype=torch.int32, device=x0.device)
        with torch.cuda.device(x0.device):
            rc = L.orp_groupnorm_act_multi_cl_amax(levels, gam, bet, len(xs), B, C, G, float(gns[0].eps), 1 if relu else 0, slots,
                                                   bits.data_ptr(), bits.numel(), _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
        _lib.check(rc, "orp_groupnorm_act_multi_cl_amax")
        return outs, bits
    with torch.cuda.device(x0.device):
        rc = L.orp_groupnorm_act_multi_cl(levels, gam, bet, len(xs), B, C, G, float(gns[0].eps), 1 if relu else 0,
                                          _lib.ptr(ws), ws.numel(), _lib.stream_of(x0))
    _lib.check(rc, "orp_groupnorm_act_multi_cl")
    return (outs, None) if no_ranges else outs
