"""OrientedRepPointsHead -- host-side mirror of mmdet/models/anchor_heads/orientedreppoints_head.py:20-781 with the
same constructor arguments, parameter names (cls_convs.{i}.conv/gn, reg_convs, reppoints_cls_conv, reppoints_cls_out,
reppoints_pts_init_conv/out, reppoints_pts_refine_conv/out) and method signatures, on the MI355X HIP operators.

What is re-designed rather than translated:
  * forward(): the towers / 1x1 heads stay stock PyTorch convs (run level by level), but the two DeformConvs are
    applied to ALL five FPN levels in one launch each (mmdet_ops.deform_conv.forward_multi) when no gradient is
    needed -- the reference launches im2col + GEMM per level;
  * get_bboxes_single(): the min-area-rect decode (rect * stride + centre) is one fused stream-ordered kernel call
    for all levels of an image instead of five blocking minaerarect calls with device->host->device copies;
  * the training-side methods (loss, APAA) live in orientedreppoints_head_train.py.
"""
from __future__ import division

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mmdet_ops.deform_conv import DeformConv
from ..mmdet_ops.minarea_rect import minaerarect_decode
from .core import PointGenerator, fused_postprocess, multi_apply, multiclass_rnms, multiclass_rnms_static
from .layers import ConvModule, bias_init_with_prob, normal_init
from .registry import HEADS, build_loss


@HEADS.register_module
class OrientedRepPointsHead(nn.Module):

    def __init__(self, num_classes, in_channels, feat_channels=256, point_feat_channels=256, stacked_convs=3,
                 num_points=9, gradient_mul=0.1, point_strides=[8, 16, 32, 64, 128], point_base_scale=4,
                 conv_cfg=None, norm_cfg=None,
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_rbox_init=dict(type='GIoULoss', loss_weight=0.375),
                 loss_rbox_refine=dict(type='GIoULoss', loss_weight=1.0),
                 loss_spatial_init=dict(type='SpatialBorderLoss', loss_weight=0.05),
                 loss_spatial_refine=dict(type='SpatialBorderLoss', loss_weight=0.1),
                 center_init=True, top_ratio=0.4):
        super(OrientedRepPointsHead, self).__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.feat_channels = feat_channels
        self.point_feat_channels = point_feat_channels
        self.stacked_convs = stacked_convs
        self.num_points = num_points
        self.gradient_mul = gradient_mul
        self.point_base_scale = point_base_scale
        self.point_strides = point_strides
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ['FocalLoss']
        self.loss_cls = build_loss(loss_cls)
        self.loss_rbox_init = build_loss(loss_rbox_init)
        self.loss_rbox_refine = build_loss(loss_rbox_refine)
        self.loss_spatial_init = build_loss(loss_spatial_init)
        self.loss_spatial_refine = build_loss(loss_spatial_refine)
        self.center_init = center_init
        self.top_ratio = top_ratio
        self.cls_out_channels = self.num_classes - 1 if self.use_sigmoid_cls else self.num_classes
        self.point_generators = [PointGenerator() for _ in self.point_strides]
        # the DCN kernel is the sqrt(num_points) x sqrt(num_points) grid the points are initialised on
        self.dcn_kernel = int(np.sqrt(num_points))
        self.dcn_pad = int((self.dcn_kernel - 1) / 2)
        assert self.dcn_kernel * self.dcn_kernel == num_points, 'The points number should be a square number.'
        assert self.dcn_kernel % 2 == 1, 'The points number should be an odd square number.'
        dcn_base = np.arange(-self.dcn_pad, self.dcn_pad + 1).astype(np.float64)
        dcn_base_y = np.repeat(dcn_base, self.dcn_kernel)
        dcn_base_x = np.tile(dcn_base, self.dcn_kernel)
        dcn_base_offset = np.stack([dcn_base_y, dcn_base_x], axis=1).reshape((-1))
        self.dcn_base_offset = torch.tensor(dcn_base_offset).view(1, -1, 1, 1)
        self._init_layers()

    def _init_layers(self):
        self.relu = nn.ReLU(inplace=True)
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else self.feat_channels
            self.cls_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                                             conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
            self.reg_convs.append(ConvModule(chn, self.feat_channels, 3, stride=1, padding=1,
                                             conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        pts_out_dim = 2 * self.num_points
        self.reppoints_cls_conv = DeformConv(self.feat_channels, self.point_feat_channels, self.dcn_kernel, 1,
                                             self.dcn_pad)
        self.reppoints_cls_out = nn.Conv2d(self.point_feat_channels, self.cls_out_channels, 1, 1, 0)
        self.reppoints_pts_init_conv = nn.Conv2d(self.feat_channels, self.point_feat_channels, 3, 1, 1)
        self.reppoints_pts_init_out = nn.Conv2d(self.point_feat_channels, pts_out_dim, 1, 1, 0)
        self.reppoints_pts_refine_conv = DeformConv(self.feat_channels, self.point_feat_channels, self.dcn_kernel, 1,
                                                    self.dcn_pad)
        self.reppoints_pts_refine_out = nn.Conv2d(self.point_feat_channels, pts_out_dim, 1, 1, 0)

    def init_weights(self):
        for m in self.cls_convs:
            normal_init(m.conv, std=0.01)
        for m in self.reg_convs:
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.reppoints_cls_conv, std=0.01)
        normal_init(self.reppoints_cls_out, std=0.01, bias=bias_cls)
        normal_init(self.reppoints_pts_init_conv, std=0.01)
        normal_init(self.reppoints_pts_init_out, std=0.01)
        normal_init(self.reppoints_pts_refine_conv, std=0.01)
        normal_init(self.reppoints_pts_refine_out, std=0.01)

    # ---- forward -------------------------------------------------------------------------------------------------
    def _towers(self, x):
        cls_feat, pts_feat = x, x
        for cls_conv in self.cls_convs:
            cls_feat = cls_conv(cls_feat)
        for reg_conv in self.reg_convs:
            pts_feat = reg_conv(pts_feat)
        pts_out_init = self.reppoints_pts_init_out(self.relu(self.reppoints_pts_init_conv(pts_feat)))
        return cls_feat, pts_feat, pts_out_init

    def _base_offset_on(self, x):
        """dcn_base_offset on x's device / dtype, cached (the reference re-uploads the 18 floats every forward with
        `.type_as(x)`: a blocking H2D copy per call, and illegal inside a stream capture)."""
        key = (x.device, x.dtype)
        cache = self.__dict__.setdefault('_base_offset_cache', {})
        t = cache.get(key)
        if t is None:
            t = self.dcn_base_offset.to(device=x.device, dtype=x.dtype)
            cache[key] = t
        return t

    def _fused_towers_ok(self, feats):
        x = feats[0]
        if not (x.is_cuda and x.dtype == torch.float32):
            return False
        for m in list(self.cls_convs) + list(self.reg_convs):
            if not (m.with_norm and isinstance(m.norm, nn.GroupNorm) and m.norm.affine and m.with_activation
                    and isinstance(m.conv, nn.Conv2d) and m.conv.bias is None):
                return False
        if len(feats) > 16:
            return False
        for m in (self.reppoints_pts_init_conv, self.reppoints_pts_init_out, self.reppoints_cls_out,
                  self.reppoints_pts_refine_out):
            if not (isinstance(m, nn.Conv2d) and m.bias is not None and m.weight.dtype == torch.float32):
                return False
        return True

    def _nhwc_norm_ok(self, channels):
        """the towers' last GroupNorm has a shape `orp_groupnorm_act_multi_nhwc` takes (32-channel tiles of whole groups);
        otherwise the hand-over is off and the NCHW normalisation + transposition route runs"""
        for m in (self.cls_convs[-1], self.reg_convs[-1]) if len(self.cls_convs) and len(self.reg_convs) else ():
            g = m.norm.num_groups
            if channels % 32 != 0 or channels % g != 0 or 32 % (channels // g) != 0:
                return False
        return bool(len(self.cls_convs) and len(self.reg_convs))

    @staticmethod
    def _conv_nobias(m, x):
        return F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)

    @staticmethod
    def _tower_multi(convs, feats, last_nhwc=None):
        """One tower over all levels, layer by layer: big-level convolutions on the library, the small levels in one HIP
        launch, then ONE GroupNorm+ReLU launch pair right behind them while the activations are cache-resident.
        (Running both towers in lockstep -- one small-level launch / one normalisation pair for ten tensors -- was
        measured 0.5-1 % slower: the pair then reads 178 MB written two convolutions earlier.)"""
        from ..mmdet_ops.fused_norm import conv3x3_multi, group_norm_act_multi
        cur = list(feats)
        for k, m in enumerate(convs):
            # last_nhwc ('only' | 'both'): the LAST layer's normalisation writes its result channels-last (as well), the
            # layout the DeformConv gathers from -- instead of a separate transposition launch over both towers' outputs
            last = last_nhwc if k == len(convs) - 1 else None
            cur = group_norm_act_multi(conv3x3_multi(cur, m.conv), m.norm, relu=True, inplace=True, nhwc=last)
        return cur

    def _split_towers_ok(self, feats):
        """The channels-last tower path applies: library split mode on (ORP_DCN_SPLIT != 0), `split_towers` not switched off
        (attribute, or ORP_TOWER_SPLIT=0 for A/B timing), equal-depth towers of stride-1 convolutions of a shape
        `orp_conv_split_multi` takes, GroupNorm shapes the channels-last kernels take."""
        from .. import _lib, switches
        from ..mmdet_ops.fused_norm import conv_split_ok
        on = getattr(self, 'split_towers', None)
        if on is None:
            # automatic: on at every input size, as the reference runs its towers at every size (head :91-113, :148-171);
            # ORP_TOWER_SPLIT=0 switches the path off (A/B timing).  (Round 4 kept pyramids below 8 x 8 off this path because
            # graph replays there disagreed with eager: that was packed-fp32 VALU code of ANY kernel miscomputing next to dense
            # MFMAs -- DESIGN.md 4.5 --, fixed in the library's build, not a property of this path.)
            on = switches.TOWER_SPLIT
        if not on or _lib.lib().orp_dcn_get_split_mode() == 0 or len(self.cls_convs) != len(self.reg_convs) or len(feats) > 8:
            return False
        x = feats[0]
        mods = list(self.cls_convs) + list(self.reg_convs)
        if not mods:
            return False
        c0, n0 = mods[0].conv, mods[0].norm
        for m in mods:
            c = m.conv
            if not (conv_split_ok(c, x) and tuple(c.stride) == (1, 1) and tuple(c.weight.shape) == tuple(c0.weight.shape) and
                    c.padding == c0.padding and c.dilation == c0.dilation and c.weight.size(0) == c.weight.size(1) and
                    2 * c.padding[0] == c.dilation[0] * (c.weight.size(2) - 1) and
                    2 * c.padding[1] == c.dilation[1] * (c.weight.size(3) - 1) and
                    m.norm.num_groups == n0.num_groups and m.norm.eps == n0.eps):
                return False
        C, G = x.size(1), n0.num_groups
        if 1024 % C != 0 or C % G != 0 or (C // G) % 4 != 0:
            return False
        p = self.reppoints_pts_init_conv
        return conv_split_ok(p, x) and tuple(p.stride) == (1, 1)

    def _towers_split(self, feats):
        """Both towers and the init branch's 3x3 convolution, channels-last (see forward).  Returns (classification tower
        output, regression tower output) as channels-last tensors -- what the DeformConv pair launch gathers from --,
        relu(reppoints_pts_init_conv(.)) NCHW for the 1x1 output convolution, and the towers' outputs' range (`Amax`)."""
        from ..mmdet_ops.fused_norm import Amax, conv_split_multi, group_norm_act_multi_cl, to_channels_last_multi
        n = len(feats)
        # ranges for the fp16-pieces mode ride along (Amax: upper bounds of max |x| per tensor set, left on the device by the
        # producers -- the FPN's normalisation, the transposition of the extra levels, each tower layer's normalisation), so
        # that no convolution needs a pass of its own over its inputs; any gap in the chain just means that pre-pass runs
        fpn_bits = getattr(feats, 'orp_amax', None)         # mmdet_models/fpn.py FpnOutputs: the bound of its channels-last outputs
        cur = list(feats)
        todo = [i for i, f in enumerate(cur) if not f.is_contiguous(memory_format=torch.channels_last)]
        am = None
        if not todo:
            am = Amax(fpn_bits, 0) if fpn_bits is not None else None
        elif len(todo) == n:
            cur, bits = to_channels_last_multi(cur, amax_slots=[0] * n)
            am = Amax(bits, 0) if bits is not None else None
        elif fpn_bits is not None:                           # the extra levels: transposed here, their maxima merged into the slot
            conv, bits = to_channels_last_multi([cur[i] for i in todo], amax_into=(fpn_bits, [0] * len(todo)))
            for k, i in enumerate(todo):
                cur[i] = conv[k]
            am = Amax(fpn_bits, 0) if bits is not None else None
        else:
            cur = to_channels_last_multi(cur)
        cls_cur = reg_cur = cur
        from .. import switches
        from ..mmdet_ops.fused_norm import conv_split_gn, conv_split_gn_ok
        fuse = getattr(self, 'fuse_tower_norm', None)          # None: automatic (ORP_TOWER_GN_FUSE=0 switches it off, A/B timing)
        if fuse is None:
            fuse = switches.TOWER_GN_FUSE
        fuse = bool(fuse) and all(conv_split_gn_ok(a.conv, b.conv, a.norm, b.norm, cur[0]) for a, b in zip(self.cls_convs, self.reg_convs))
        if fuse:
            # conv -> GroupNorm -> ReLU with the normalisation fused AROUND the convolution launches: statistics from the
            # convolution's epilogue, the affine + ReLU applied by the next layer as it reads (two launches per layer instead of
            # four, the normalised tensors of the inner layers never reach HBM); the last layer's is materialised in place
            coef, depth = None, len(self.cls_convs)
            for k, (a, b) in enumerate(zip(self.cls_convs, self.reg_convs)):
                cls_cur, reg_cur, coef, am = conv_split_gn(cls_cur, a.conv, reg_cur, b.conv, a.norm, b.norm, coef_in=coef, amax=am,
                                                           materialize=(k == depth - 1))
        else:
            for a, b in zip(self.cls_convs, self.reg_convs):
                oa, ob = conv_split_multi(cls_cur, a.conv, reg_cur, b.conv, amax=am)
                both, bits = group_norm_act_multi_cl(oa + ob, [a.norm] * n + [b.norm] * n, relu=True, amax_slots=[0] * n + [1] * n)
                am = Amax(bits, 1) if bits is not None else None
                cls_cur, reg_cur = both[:n], both[n:]
        hid = conv_split_multi(reg_cur, self.reppoints_pts_init_conv, bias=True, relu=True, out_channels_last=False,
                               amax=Amax(am.bits[1:], 0) if am is not None else None)
        return cls_cur, reg_cur, hid, am                     # am: the range of (cls_cur, reg_cur) for the DeformConv pair launch

    def _dcn_pair(self, cls_feats, pts_feats, offsets, out_channels_last=None, amax=None):
        a, b = self.reppoints_cls_conv, self.reppoints_pts_refine_conv
        from ..mmdet_ops.deform_conv import deform_conv_forward_pair, fast_path_ok
        same = (a.stride == b.stride and a.padding == b.padding and a.dilation == b.dilation and
                a.weight.shape == b.weight.shape)
        # (fp32 tensors only: half / bfloat16 / double tensors go layer by layer through forward_multi, which picks the half kernel
        #  -- the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF half branch -- or the double column formulation)
        if same and cls_feats[0].is_cuda and cls_feats[0].dtype == torch.float32 and a.weight.dtype == torch.float32 and \
                fast_path_ok(a.weight, a.groups, a.deformable_groups) and fast_path_ok(b.weight, b.groups, b.deformable_groups):
            return deform_conv_forward_pair(cls_feats, pts_feats, offsets, a.weight, b.weight, a.stride, a.padding,
                                            a.dilation, relu=True, out_channels_last=out_channels_last, amax=amax)
        assert out_channels_last is None
        return a.forward_multi(cls_feats, offsets, relu=True), b.forward_multi(pts_feats, offsets, relu=True)

    def forward_single(self, x):
        """One level, autograd-capable (reference forward_single, head :148-171)."""
        dcn_base_offset = self._base_offset_on(x)
        cls_feat, pts_feat, pts_out_init = self._towers(x)
        pts_out_init_grad_mul = (1 - self.gradient_mul) * pts_out_init.detach() + self.gradient_mul * pts_out_init
        dcn_offset = pts_out_init_grad_mul - dcn_base_offset
        dcn_cls_feat = self.reppoints_cls_conv(cls_feat, dcn_offset)
        cls_out = self.reppoints_cls_out(self.relu(dcn_cls_feat))
        pts_out_refine = self.reppoints_pts_refine_out(self.relu(self.reppoints_pts_refine_conv(pts_feat, dcn_offset)))
        pts_out_refine = pts_out_refine + pts_out_init.detach()
        return cls_out, pts_out_init, pts_out_refine, x

    def _tower_train(self, convs, feats):
        """One tower over all levels, layer by layer, with autograd: the library convolution per level, then GroupNorm +
        ReLU of all levels as ONE autograd node (one launch pair forward, three launches backward instead of ~8 per
        level; mmdet_ops/fused_norm.py group_norm_act_train)."""
        from ..mmdet_ops.fused_norm import group_norm_act_train
        cur = list(feats)
        for m in convs:
            cur = group_norm_act_train([m.conv(x) for x in cur], m.norm, relu=True)
        return cur

    def _towers_train_pair(self, feats):
        """Both towers with autograd, layer by layer: conv_split_train (one node for the ten tensors) -> GroupNorm + ReLU (one
        node for the ten tensors)."""
        from ..mmdet_ops.fused_norm import conv_split_train, group_norm_act_train
        n = len(feats)
        a_cur = b_cur = list(feats)
        for a, b in zip(self.cls_convs, self.reg_convs):
            outs = conv_split_train(a_cur + b_cur, [a.conv] * n + [b.conv] * n)
            ys = group_norm_act_train(outs, [a.norm] * n + [b.norm] * n, relu=True)
            a_cur, b_cur = ys[:n], ys[n:]
        return a_cur, b_cur

    def forward_train_multi(self, feats):
        """Training forward with the two DeformConvs of ALL levels as one autograd node (one pair launch forward, the MFMA
        backward over all levels at once); per level the same operations as forward_single, in the same order."""
        from ..mmdet_ops.deform_conv import deform_conv_pair
        dcn_base_offset = self._base_offset_on(feats[0])
        cls_feats, pts_feats, inits, offsets = [], [], [], []
        fused_gn = self._fused_towers_ok(feats)
        hid = None
        if fused_gn:
            from ..mmdet_ops.fused_norm import conv_split_train, conv_split_train_ok
            tower_convs = [m.conv for m in list(self.cls_convs) + list(self.reg_convs)]
            if len(self.cls_convs) == len(self.reg_convs) and len(feats) <= 8 and conv_split_train_ok(tower_convs, feats[0]):
                # the two towers' layer k over all levels: ONE convolution node on the bf16-split kernel (forward and
                # grad_input; grad_weight stays on the library), then the GroupNorm + ReLU node of both towers' tensors
                cls_all, pts_all = self._towers_train_pair(feats)
            else:
                cls_all, pts_all = self._tower_train(self.cls_convs, feats), self._tower_train(self.reg_convs, feats)
            pc = self.reppoints_pts_init_conv
            if len(feats) <= 8 and conv_split_train_ok([pc], feats[0], allow_bias=True):     # (orp_split::kMaxLevels tensors per launch)
                hid = conv_split_train(pts_all, pc)
                if pc.bias is not None:
                    hid = [h + pc.bias.view(1, -1, 1, 1) for h in hid]
        for i, x in enumerate(feats):
            if fused_gn:
                cls_feat, pts_feat = cls_all[i], pts_all[i]
                init_mid = hid[i] if hid is not None else self.reppoints_pts_init_conv(pts_feat)
                pts_out_init = self.reppoints_pts_init_out(self.relu(init_mid))
            else:
                cls_feat, pts_feat, pts_out_init = self._towers(x)
            grad_mul = (1 - self.gradient_mul) * pts_out_init.detach() + self.gradient_mul * pts_out_init
            cls_feats.append(cls_feat); pts_feats.append(pts_feat); inits.append(pts_out_init)
            offsets.append(grad_mul - dcn_base_offset)
        a, b = self.reppoints_cls_conv, self.reppoints_pts_refine_conv
        # the classification branch's gradient is dense (focal loss at every point), the refinement branch's only lives at
        # the positive points: its DeformConv backward takes the sparse route -- a few hundred active positions scattered
        # with fp32 atomics (0.59 ms instead of 1.17 ms), which makes grad_input of THAT branch order-dependent in its last
        # bits.  `deterministic_backward = True` on the head (or ORP_DETERMINISTIC=1) keeps the fixed-order region pass for
        # both branches: bitwise reproducible gradients at the price of that half millisecond.
        from .. import switches
        det = getattr(self, 'deterministic_backward', None)
        if det is None:
            det = switches.DETERMINISTIC
        dcn_cls, dcn_pts = deform_conv_pair(cls_feats, pts_feats, offsets, a.weight, b.weight, a.stride, a.padding,
                                            a.dilation, sparse_grad=(False, not det))
        cls_outs = [self.reppoints_cls_out(torch.relu(c)) for c in dcn_cls]
        refines = [self.reppoints_pts_refine_out(torch.relu(p)) + init.detach() for p, init in zip(dcn_pts, inits)]
        return cls_outs, inits, refines, list(feats)

    def __getstate__(self):
        # copy.deepcopy / pickle of a head that has run with its towers on two streams: the stream object is per process (and cannot be
        # pickled); the copy creates its own on first use
        state = self.__dict__.copy()
        state['_side_stream'] = None
        return state

    def forward(self, feats):
        if torch.is_grad_enabled():
            from ..mmdet_ops.deform_conv import pair_autograd_ok
            if all(pair_autograd_ok(self.reppoints_cls_conv, self.reppoints_pts_refine_conv, x) for x in feats):
                return self.forward_train_multi(feats)
            return multi_apply(self.forward_single, feats)
        # inference: same arithmetic, but every layer covers all levels at once: the tower ConvModules run their
        # GroupNorm+ReLU as one fused HIP launch pair over the five levels, each DeformConv is ONE launch
        dcn_base_offset = self._base_offset_on(feats[0])
        fused = self._fused_towers_ok(feats)
        if fused:
            # channels-last hand-over to the DeformConv pair launch (round 4): the last GroupNorm of each tower writes its
            # result transposed -- the classification tower ONLY so (nothing else reads it), the regression tower both ways
            # (the init branch's 3x3 convolution reads NCHW) -- which removes the pair launch's transposition kernel
            # (one more read + write of both towers' outputs per image).
            a_, b_ = self.reppoints_cls_conv, self.reppoints_pts_refine_conv
            hand = getattr(self, 'nhwc_handover', True)         # (attribute: False switches the hand-over off, A/B timing)
            hand = bool(hand) and not getattr(self, 'fuse_output_convs', False) and a_.weight.size(1) % 256 == 0 and \
                tuple(a_.weight.shape) == tuple(b_.weight.shape) and a_.stride == b_.stride and a_.padding == b_.padding and \
                a_.dilation == b_.dilation and a_.groups == 1 and b_.groups == 1 and a_.deformable_groups == 1 and \
                b_.deformable_groups == 1 and min(min(f.size(2), f.size(3)) for f in feats) > 1 and self._nhwc_norm_ok(feats[0].size(1))
            from ..mmdet_ops.fused_norm import bias_act_multi, conv3x3_multi
            split = hand and self._split_towers_ok(feats)
            side = None
            two = getattr(self, 'tower_streams', None)                # None: default on (attribute False: off; PipelinedInference sets it)
            if not split and (two if two is not None else True):
                # the two towers are independent chains: the classification tower runs on a second stream (a fork / join in
                # a captured graph), so that its small-level kernels fill the CUs the other tower's leave idle
                cur = torch.cuda.current_stream(feats[0].device)
                if getattr(self, '_side_stream', None) is None:
                    self._side_stream = torch.cuda.Stream(device=feats[0].device)
                side = self._side_stream
                side.wait_stream(cur)
            if split:
                # the towers on the bf16 matrix pipe (round 4, csrc/orp_conv_split.hip): channels-last all the way, the two
                # towers' layer k = ONE launch over all levels, GroupNorm+ReLU of both towers' ten tensors in one launch
                # triple, the init branch's 3x3 convolution with bias + ReLU in its epilogue -- 10 launches for what were
                # 2 x 3 x (5 convolutions + 2) + 6; no second stream
                cls_feats, pts_dcn_in, hid, dcn_amax = self._towers_split(feats)
            else:
                if side is not None:
                    with torch.cuda.stream(side):
                        cls_feats = self._tower_multi(self.cls_convs, feats, 'only' if hand else None)
                else:
                    cls_feats = self._tower_multi(self.cls_convs, feats, 'only' if hand else None)
                pts_feats = self._tower_multi(self.reg_convs, feats, 'both' if hand else None)
                pts_dcn_in = pts_feats
                if hand:
                    pts_feats, pts_dcn_in = pts_feats
                # bias-carrying convolutions: the convolution runs without its bias, ONE launch per layer then adds it for
                # all levels together with what follows (ReLU / `- dcn_base_offset` / `+ pts_out_init`), same op order
                hid = bias_act_multi(conv3x3_multi(pts_feats, self.reppoints_pts_init_conv),
                                     self.reppoints_pts_init_conv.bias, relu=True)
            from ..mmdet_ops.fused_norm import conv1x1_multi, conv1x1_ok
            one_by_one = all(conv1x1_ok(m, hid[0]) for m in (self.reppoints_pts_init_out, self.reppoints_cls_out,
                                                              self.reppoints_pts_refine_out))
            if one_by_one:                      # 1x1 output convolution + bias + `- dcn_base_offset`: one launch, all levels
                inits, offsets = conv1x1_multi(hid, self.reppoints_pts_init_out, sub=dcn_base_offset)
            else:
                inits, offsets = bias_act_multi([self._conv_nobias(self.reppoints_pts_init_out, h) for h in hid],
                                                self.reppoints_pts_init_out.bias, sub=dcn_base_offset)
        else:
            cls_feats, pts_feats, inits = [], [], []
            for x in feats:
                cls_feat, pts_feat, pts_out_init = self._towers(x)
                cls_feats.append(cls_feat); pts_feats.append(pts_feat); inits.append(pts_out_init)
            # (1-g)*p.detach() + g*p is p up to one rounding; without autograd the two extra passes are skipped
            offsets = [init - dcn_base_offset for init in inits]
        if fused and side is not None:
            torch.cuda.current_stream(feats[0].device).wait_stream(side)
            for t in cls_feats:
                t.record_stream(torch.cuda.current_stream(feats[0].device))
        if fused and one_by_one and getattr(self, 'fuse_output_convs', False):
            # (off by default -- measured round 2: the fused epilogue adds ~45 us to the DeformConv launch, as much as the two
            # bandwidth-bound 1x1 launches it replaces cost, and those overlap with other work in throughput mode)
            from ..mmdet_ops.deform_conv import deform_conv_forward_pair_heads, pair_heads_ok
            if pair_heads_ok(self.reppoints_cls_conv, self.reppoints_pts_refine_conv, self.reppoints_cls_out,
                             self.reppoints_pts_refine_out, cls_feats[0]):
                # the whole refinement stage in ONE launch: both DeformConvs, ReLU, both 1x1 output convolutions and
                # `+ pts_out_init`; the 256-channel DeformConv outputs never reach HBM
                cls_outs, refines = deform_conv_forward_pair_heads(
                    cls_feats, pts_feats, offsets, self.reppoints_cls_conv, self.reppoints_pts_refine_conv,
                    self.reppoints_cls_out, self.reppoints_pts_refine_out, residuals_b=inits)
                return cls_outs, inits, refines, list(feats)
        # both DeformConvs take the same offsets: ONE launch for the two layers and all levels, ReLU fused in the epilogue
        if fused and hand:
            dcn_cls, dcn_pts = self._dcn_pair(cls_feats, pts_dcn_in, offsets, out_channels_last=False,
                                              amax=dcn_amax if split else None)
        else:
            dcn_cls, dcn_pts = self._dcn_pair(cls_feats, pts_feats, offsets)
        if fused and one_by_one:
            cls_outs = conv1x1_multi(dcn_cls, self.reppoints_cls_out)
            refines = conv1x1_multi(dcn_pts, self.reppoints_pts_refine_out, residuals=inits)
        elif fused:
            cls_outs = bias_act_multi([self._conv_nobias(self.reppoints_cls_out, c) for c in dcn_cls],
                                      self.reppoints_cls_out.bias)
            refines = bias_act_multi([self._conv_nobias(self.reppoints_pts_refine_out, p) for p in dcn_pts],
                                     self.reppoints_pts_refine_out.bias, residuals=inits)
        else:
            cls_outs, refines = [], []
            for c, p, init in zip(dcn_cls, dcn_pts, inits):
                cls_outs.append(self.reppoints_cls_out(c))
                refines.append(self.reppoints_pts_refine_out(p) + init)
        return cls_outs, inits, refines, list(feats)

    # ---- test-time decode + NMS ------------------------------------------------------------------------------------
    def get_bboxes(self, cls_scores, pts_preds_init, pts_preds_refine, base_feats, img_metas, cfg, rescale=False,
                   nms=True, static=False):
        assert len(cls_scores) == len(pts_preds_refine)
        num_levels = len(cls_scores)
        device = cls_scores[0].device
        mlvl_points = [self.point_generators[i].grid_points(cls_scores[i].size()[-2:], self.point_strides[i], device)
                       for i in range(num_levels)]
        result_list = []
        for img_id in range(len(img_metas)):
            cls_score_list = [cls_scores[i][img_id].detach() for i in range(num_levels)]
            points_pred_list = [pts_preds_refine[i][img_id].detach() for i in range(num_levels)]
            img_shape = img_metas[img_id]['img_shape']
            scale_factor = img_metas[img_id]['scale_factor']
            result_list.append(self.get_bboxes_single(cls_score_list, points_pred_list, mlvl_points, img_shape,
                                                      scale_factor, cfg, rescale, nms, static=static))
        return result_list

    def get_bboxes_single(self, cls_scores, points_preds, mlvl_points, img_shape, scale_factor, cfg, rescale=False,
                          nms=True, static=False):
        assert len(cls_scores) == len(points_preds) == len(mlvl_points)
        if nms and static and self.use_sigmoid_cls and not rescale and cfg.nms.get('type', 'rnms') == 'rnms' \
                and cfg.get('fused_postprocess', True) and cls_scores[0].is_cuda:
            # decode -> selection -> NMS -> packing on the fused HIP kernels (same detections, same order)
            return fused_postprocess(cls_scores, points_preds, self.point_strides, cfg, self.num_points)
        lvl_pts, lvl_scores, lvl_centers, lvl_strides = [], [], [], []
        for i_lvl, (cls_score, points_pred, points) in enumerate(zip(cls_scores, points_preds, mlvl_points)):
            assert cls_score.size()[-2:] == points_pred.size()[-2:]
            cls_score = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels)
            scores = cls_score.sigmoid() if self.use_sigmoid_cls else cls_score.softmax(-1)
            points_pred = points_pred.permute(1, 2, 0).reshape(-1, 2 * self.num_points)
            nms_pre = cfg.get('nms_pre', -1)
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                if self.use_sigmoid_cls:
                    max_scores, _ = scores.max(dim=1)
                else:
                    max_scores, _ = scores[:, 1:].max(dim=1)
                _, topk_inds = max_scores.topk(nms_pre)
                points = points[topk_inds, :]
                points_pred = points_pred[topk_inds, :]
                scores = scores[topk_inds, :]
            # (y, x) -> (x, y) pairs, grid units
            pts_xy = points_pred.reshape(-1, self.num_points, 2).flip(-1).reshape(-1, 2 * self.num_points)
            lvl_pts.append(pts_xy)
            lvl_scores.append(scores)
            lvl_centers.append(points[:, :2])
            lvl_strides.append(points[:, 2])
        pts_all = torch.cat(lvl_pts)
        centers = torch.cat(lvl_centers).contiguous()
        strides = torch.cat(lvl_strides).contiguous()
        # one fused kernel for all levels: hull -> min-area rect -> corners * stride + centre  (head :746-749)
        mlvl_bboxes = minaerarect_decode(pts_all, centers, strides)
        mlvl_reppoints = pts_all * strides[:, None] + centers.repeat(1, self.num_points)
        if rescale:
            mlvl_bboxes /= mlvl_bboxes.new_tensor(scale_factor)
            mlvl_reppoints /= mlvl_reppoints.new_tensor(scale_factor)
        mlvl_scores = torch.cat(lvl_scores)
        if self.use_sigmoid_cls:
            padding = mlvl_scores.new_zeros(mlvl_scores.shape[0], 1)
            mlvl_scores = torch.cat([padding, mlvl_scores], dim=1)
        if nms and static:
            # sync-free, fixed-shape variant: one packed device tensor (see core.multiclass_rnms_static)
            return multiclass_rnms_static(mlvl_bboxes, mlvl_scores, cfg.score_thr, cfg.nms, cfg.max_per_img,
                                          mlvl_reppoints, capacity=cfg.get('static_capacity', 8192))
        if nms:
            return multiclass_rnms(mlvl_bboxes, mlvl_scores, cfg.score_thr, cfg.nms, cfg.max_per_img,
                                   multi_reppoints=mlvl_reppoints)
        return mlvl_bboxes, mlvl_scores

    def loss(self, *args, **kwargs):
        from .orientedreppoints_head_train import head_loss
        return head_loss(self, *args, **kwargs)
