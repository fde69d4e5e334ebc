"""FPN neck on stock PyTorch-ROCm (mmdet/models/necks/fpn.py:11-178: start_level, add_extra_convs, num_outs, GN)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import ConvModule, xavier_init
from .registry import NECKS


class FpnOutputs(tuple):
    """The neck's outputs: a tuple of per-level tensors, plus `orp_amax` -- None, or an int32 CUDA tensor holding (float bits)
    an upper bound of max |x| over the channels-last levels, left by their normalisation for the head's first tower layer."""
    orp_amax = None


@NECKS.register_module
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None):
        super(FPN, self).__init__()
        assert isinstance(in_channels, list)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        self.add_extra_convs, self.extra_convs_on_inputs = add_extra_convs, extra_convs_on_inputs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg if not no_norm_on_lateral else None,
                                                 act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))
        extra_levels = num_outs - self.backbone_end_level + self.start_level
        if add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.extra_convs_on_inputs:
                    ic = self.in_channels[self.backbone_end_level - 1]
                else:
                    ic = out_channels
                self.fpn_convs.append(ConvModule(ic, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def _fused_ok(self, inputs):
        """Inference on the GPU with GroupNorm'ed, activation-free ConvModules (the DOTA configs): the lateral / output
        normalisations of all levels run as ONE launch pair each, the small output levels on the HIP convolution."""
        x = inputs[0]
        if not (x.is_cuda and x.dtype == torch.float32):
            return False
        used = len(self.lateral_convs)
        mods = list(self.lateral_convs) + list(self.fpn_convs[:used])
        return all(m.with_norm and isinstance(m.norm, nn.GroupNorm) and m.norm.affine and not m.with_activation
                   and isinstance(m.conv, nn.Conv2d) and m.conv.bias is None
                   and m.norm.num_groups == mods[0].norm.num_groups and m.norm.eps == mods[0].norm.eps for m in mods)

    def _split_ok(self, laterals):
        """The output convolutions take the channels-last bf16-split path: library split mode on (ORP_DCN_SPLIT != 0),
        `split_convs` not switched off (attribute, or ORP_FPN_SPLIT=0 for A/B timing), stride-1 'same' convolutions of one
        shape that `orp_conv_split_multi_ex` takes, GroupNorm shapes the channels-last kernels take."""
        from .. import _lib, switches
        from ..mmdet_ops.fused_norm import conv_split_ok
        on = getattr(self, 'split_convs', None)
        if on is None:
            on = switches.FPN_SPLIT                           # automatic: on at every input size (ORP_FPN_SPLIT=0: off, A/B timing)
        used = len(self.lateral_convs)
        if not on or _lib.lib().orp_dcn_get_split_mode() == 0 or used > 8:
            return False
        c0 = self.fpn_convs[0].conv
        for fc in self.fpn_convs[:used]:
            c = fc.conv
            if not (conv_split_ok(c, laterals[0]) and tuple(c.stride) == (1, 1) and tuple(c.weight.shape) == tuple(c0.weight.shape)
                    and c.padding == c0.padding and c.dilation == c0.dilation and
                    2 * c.padding[0] == c.dilation[0] * (c.weight.size(2) - 1) and
                    2 * c.padding[1] == c.dilation[1] * (c.weight.size(3) - 1)):
                return False
        C, G = c0.weight.size(0), self.fpn_convs[0].norm.num_groups
        return 1024 % C == 0 and C % G == 0 and (C // G) % 4 == 0 and min(min(t.size(2), t.size(3)) for t in laterals) > 1

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        used = len(self.lateral_convs)
        fused = self._fused_ok(inputs)
        split_amax = None
        train = fused and torch.is_grad_enabled()     # autograd: the same launch pairs as one node per normalisation layer
        if train:
            from ..mmdet_ops.fused_norm import group_norm_act_train
            laterals = group_norm_act_train([lc.conv(inputs[i + self.start_level])
                                             for i, lc in enumerate(self.lateral_convs)],
                                            [lc.norm for lc in self.lateral_convs], relu=False)
        elif fused:
            from ..mmdet_ops.fused_norm import conv3x3_multi, group_norm_act_multi
            laterals = group_norm_act_multi([lc.conv(inputs[i + self.start_level])
                                             for i, lc in enumerate(self.lateral_convs)],
                                            [lc.norm for lc in self.lateral_convs], relu=False, inplace=True)
        else:
            laterals = [lc(inputs[i + self.start_level]) for i, lc in enumerate(self.lateral_convs)]
        for i in range(used - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                              mode='nearest')
        if train:
            from ..mmdet_ops.fused_norm import conv_split_train, conv_split_train_ok
            out_convs = [fc.conv for fc in self.fpn_convs[:used]]
            if used <= 8 and conv_split_train_ok(out_convs, laterals[0]):     # one node, a layer per level (bf16-split kernel)
                conv_outs = conv_split_train(laterals, out_convs)
            else:
                conv_outs = [c(l) for c, l in zip(out_convs, laterals)]
            outs = group_norm_act_train(conv_outs, [fc.norm for fc in self.fpn_convs[:used]], relu=False)
        elif fused and self._split_ok(laterals):
            # the output convolutions of all levels in ONE launch on the bf16 matrix pipe (csrc/orp_conv_split.hip: fp32 in /
            # out, operands split exactly into three bf16 pieces), a layer of its own per level; channels-last from here on --
            # the layout the head's towers read (their own transposition launch goes away)
            from ..mmdet_ops.fused_norm import Amax, conv_split_multi, group_norm_act_multi_cl, to_channels_last_multi
            # (ranges for the fp16-pieces arithmetic ride along on the device: the transposition leaves max |x| of the
            #  laterals, the normalisation a bound of its outputs' for the head's first tower layer -- FpnOutputs.orp_amax)
            cl, bits = to_channels_last_multi(laterals, amax_slots=[0] * used)
            conv = conv_split_multi(cl, [fc.conv for fc in self.fpn_convs[:used]], amax=Amax(bits, 0) if bits is not None else None)
            outs, split_amax = group_norm_act_multi_cl(conv, [fc.norm for fc in self.fpn_convs[:used]], relu=False,
                                                       amax_slots=[0] * used)
        elif fused:
            outs = group_norm_act_multi(conv3x3_multi(laterals, [fc.conv for fc in self.fpn_convs[:used]]),
                                        [fc.norm for fc in self.fpn_convs[:used]], relu=False, inplace=True)
        else:
            outs = [self.fpn_convs[i](laterals[i]) for i in range(used)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for i in range(self.num_outs - used):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                extra = self._extra_fused if (fused and not train) else (lambda m, x: m(x))
                if self.extra_convs_on_inputs:
                    outs.append(extra(self.fpn_convs[used], inputs[self.backbone_end_level - 1]))
                else:
                    outs.append(extra(self.fpn_convs[used], outs[-1]))
                for i in range(used + 1, self.num_outs):
                    if self.relu_before_extra_convs:
                        outs.append(extra(self.fpn_convs[i], F.relu(outs[-1])))
                    else:
                        outs.append(extra(self.fpn_convs[i], outs[-1]))
        res = FpnOutputs(outs)
        res.orp_amax = split_amax        # None, or the device-side range of the channels-last outputs (the first `used` levels)
        return res

    @staticmethod
    def _extra_fused(m, x):
        """One extra level (stride-2 ConvModule, GroupNorm, no activation) at inference: the fixed-order HIP convolution
        when the map is small (the library's split-K kernel for these shapes sums with atomics: P6 / P7 would differ
        by ~2e-6 from run to run and detections near score_thr would come and go), then the fused GroupNorm."""
        from ..mmdet_ops.fused_norm import conv3x3_multi, group_norm_act_multi
        c = m.conv
        if not (m.with_norm and isinstance(m.norm, nn.GroupNorm) and not m.with_activation and c.bias is None and
                tuple(c.kernel_size) == (3, 3)):
            return m(x)
        return group_norm_act_multi(conv3x3_multi([x], c, split_k=True), [m.norm], relu=False, inplace=True)[0]
