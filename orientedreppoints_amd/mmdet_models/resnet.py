"""ResNet backbone on stock PyTorch-ROCm (mmdet/models/backbones/resnet.py:306-520 interface: depth, num_stages,
out_indices, frozen_stages, norm_cfg, style, norm_eval).  Parameter names follow torchvision / the released
checkpoints (conv1, bn1, layer{1..4}.{i}.conv{1,2,3}, bn{1,2,3}, downsample.{0,1}).  Out of the HIP hot path: this
is the part BASELINE.json says stays stock."""
import torch
import torch.nn as nn

from .layers import build_conv_layer, build_norm_layer, constant_init, kaiming_init
from .registry import BACKBONES


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 norm_cfg=dict(type='BN'), dcn=None):
        super(Bottleneck, self).__init__()
        assert style in ['pytorch', 'caffe']
        assert dcn is None or isinstance(dcn, dict)
        if style == 'pytorch':
            conv1_stride, conv2_stride = 1, stride
        else:
            conv1_stride, conv2_stride = stride, 1
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=conv1_stride, bias=False)
        self.add_module('bn1', build_norm_layer(norm_cfg, planes)[1])
        # dcn = dict(type='DCN' | 'DCNv2', deformable_groups=1, fallback_on_stride=False) turns conv2 into the
        # offset-predicting deformable convolution (mmdet/models/backbones/resnet.py:140-170); its parameters are
        # conv2.weight and conv2.conv_offset.{weight,bias}, as in the released DCN checkpoints
        self.with_dcn = dcn is not None
        fallback_on_stride = False
        if self.with_dcn:
            # as the reference does it (resnet.py:145-148): the flag is POPPED from the dict the backbone hands to every block
            # of every stage, so only the first block built from it sees the flag -- which decides the set of
            # conv2.conv_offset keys a reference-trained checkpoint has -- and a set flag means a plain conv2 whatever its stride
            fallback_on_stride = dcn.pop('fallback_on_stride', False)
            dcn = dict(dcn)
            if 'modulated' in dcn:                                   # the older spelling of the same choice
                dcn.setdefault('type', 'DCNv2' if dcn.pop('modulated') else 'DCN')
            dcn.setdefault('type', 'DCN')
        use_dcn = self.with_dcn and not fallback_on_stride
        self.conv2 = build_conv_layer(dcn if use_dcn else None, planes, planes, kernel_size=3, stride=conv2_stride,
                                      padding=dilation, dilation=dilation, bias=False)
        self.conv2_is_dcn = use_dcn
        self.add_module('bn2', build_norm_layer(norm_cfg, planes)[1])
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module('bn3', build_norm_layer(norm_cfg, planes * self.expansion)[1])
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def _fused_ok(self, x):
        # inference only: eval-mode BatchNorm is a per-channel affine -> one fused HIP pass with the ReLU / residual add
        return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and \
            isinstance(self.bn1, nn.BatchNorm2d) and not self.bn1.training and not self.bn3.training

    def _forward_fused(self, x):
        from ..mmdet_ops.fused_norm import bn_act
        out = bn_act(self.conv1(x).contiguous(), self.bn1, relu=True)
        out = bn_act(self.conv2(out).contiguous(), self.bn2, relu=True)
        out = self.conv3(out).contiguous()
        if self.downsample is not None:
            identity = bn_act(self.downsample[0](x).contiguous(), self.downsample[1], relu=False)
        else:
            identity = x.contiguous()
        return bn_act(out, self.bn3, residual=identity, relu=True)

    def forward(self, x):
        if self._fused_ok(x):
            return self._forward_fused(x)
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), gcb=None, stage_with_gcb=None, gen_attention=None,
                 stage_with_gen_attention=None, with_cp=False, zero_init_residual=True):
        super(ResNet, self).__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet'.format(depth))
        assert gcb is None and gen_attention is None, 'context / attention blocks are not part of the DOTA configs'
        assert conv_cfg is None, 'the backbone convolutions are stock nn.Conv2d (dcn= selects the deformable conv2)'
        # one private copy per model build: the blocks pop 'fallback_on_stride' from the dict they share (the reference's pop-once
        # semantics, which decide the checkpoint's conv_offset key set) -- the CALLER's config object is left as it was, so a second
        # backbone built from the same config sees the flag again (round-5 advisor)
        self.dcn = dict(dcn) if dcn is not None else None
        self.stage_with_dcn = tuple(stage_with_dcn) if stage_with_dcn is not None else (False,) * num_stages
        if dcn is not None:
            assert len(self.stage_with_dcn) == num_stages
        self.depth, self.num_stages, self.out_indices = depth, num_stages, out_indices
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.zero_init_residual = zero_init_residual
        block, stage_blocks = self.arch_settings[depth]
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.add_module('bn1', build_norm_layer(norm_cfg, 64)[1])
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, num_blocks in enumerate(stage_blocks[:num_stages]):
            planes = 64 * 2 ** i
            stride, dilation = strides[i], dilations[i]
            downsample = None
            if stride != 1 or inplanes != planes * block.expansion:
                downsample = nn.Sequential(
                    nn.Conv2d(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                    build_norm_layer(norm_cfg, planes * block.expansion)[1])
            stage_dcn = self.dcn if (self.dcn is not None and self.stage_with_dcn[i]) else None
            layers = [block(inplanes, planes, stride, dilation, downsample, style, norm_cfg, stage_dcn)]
            inplanes = planes * block.expansion
            for _ in range(1, num_blocks):
                layers.append(block(inplanes, planes, 1, dilation, None, style, norm_cfg, stage_dcn))
            name = 'layer{}'.format(i + 1)
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = inplanes

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in [self.conv1, self.bn1]:
                for param in m.parameters():
                    param.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, 'layer{}'.format(i))
            m.eval()
            for param in m.parameters():
                param.requires_grad = False

    def init_weights(self, pretrained=None):
        """`pretrained`: a local checkpoint file, or a model-zoo URI of the reference configs ('torchvision://resnet50',
        'open-mmlab://...').  There is no network here: a URI is resolved through the environment variable
        ORP_PRETRAINED_DIR (<dir>/<name>.pth) and, when no such file exists, a warning says that the backbone -- whose
        stem and layer1 the DOTA configs FREEZE -- stays at its random initialisation."""
        if isinstance(pretrained, str):
            import os
            import warnings
            path = pretrained
            if '://' in pretrained:
                name = pretrained.split('://', 1)[1].replace('/', '_')
                root = os.environ.get('ORP_PRETRAINED_DIR', '')
                path = os.path.join(root, name + '.pth') if root else ''
            if path and os.path.isfile(path):
                sd = torch.load(path, map_location='cpu')
                sd = sd.get('state_dict', sd)
                missing, unexpected = self.load_state_dict(sd, strict=False)
                unexpected = [k for k in unexpected if not k.startswith('fc.')]      # the classifier head is not part of it
                if missing or unexpected:
                    warnings.warn('ResNet.init_weights(%r): missing keys %s, unexpected keys %s'
                                  % (pretrained, list(missing)[:8], unexpected[:8]))
                return
            warnings.warn('ResNet.init_weights: pretrained=%r was NOT loaded (no such file%s); the backbone keeps its '
                          'random initialisation, frozen_stages=%d of it frozen'
                          % (pretrained, '' if '://' not in pretrained else '; set ORP_PRETRAINED_DIR', self.frozen_stages))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                constant_init(m, 1)
        if self.dcn is not None:                                      # resnet.py:478-482: offsets (and masks) start at zero
            for m in self.modules():
                if isinstance(m, Bottleneck) and hasattr(m.conv2, 'conv_offset'):
                    constant_init(m.conv2.conv_offset, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    constant_init(m.bn3, 0)

    def forward(self, x):
        if (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and \
                isinstance(self.bn1, nn.BatchNorm2d) and not self.bn1.training:
            from ..mmdet_ops.fused_norm import bn_act
            x = self.maxpool(bn_act(self.conv1(x).contiguous(), self.bn1, relu=True))
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super(ResNet, self).train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
