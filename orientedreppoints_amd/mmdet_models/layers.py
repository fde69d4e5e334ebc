"""ConvModule / norm / init helpers the dense head and FPN are built from (mmdet/ops/conv_module.py:11-140,
mmdet/ops/norm.py:3-55, mmcv.cnn init functions).  Stock PyTorch modules; parameter names match the released
checkpoints (`<name>.conv.weight`, `<name>.gn.weight`, ...)."""
import numpy as np
import torch
import torch.nn as nn

norm_cfg_table = {'BN': ('bn', nn.BatchNorm2d), 'SyncBN': ('bn', nn.SyncBatchNorm), 'GN': ('gn', nn.GroupNorm)}


def build_norm_layer(cfg, num_features, postfix=''):
    assert isinstance(cfg, dict) and 'type' in cfg
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    if layer_type not in norm_cfg_table:
        raise KeyError('Unrecognized norm type {}'.format(layer_type))
    abbr, norm_layer = norm_cfg_table[layer_type]
    name = abbr + str(postfix)
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type != 'GN':
        layer = norm_layer(num_features, **cfg_)
    else:
        assert 'num_groups' in cfg_
        layer = norm_layer(num_channels=num_features, **cfg_)
    for param in layer.parameters():
        param.requires_grad = requires_grad
    return name, layer


def _conv_types():
    """The reference's conv registry (mmdet/ops/conv.py:6-12): 'Conv', 'DCN' (DeformConvPack), 'DCNv2'
    (ModulatedDeformConvPack).  'ConvWS' is not used by any DOTA config and is not provided."""
    from ..mmdet_ops.deform_conv import DeformConvPack, ModulatedDeformConvPack
    return {'Conv': nn.Conv2d, 'DCN': DeformConvPack, 'DCNv2': ModulatedDeformConvPack}


def build_conv_layer(cfg, *args, **kwargs):
    """mmdet/ops/conv.py:15-42: cfg None -> nn.Conv2d; dict(type='DCN' | 'DCNv2', deformable_groups=..) -> the
    offset-predicting DeformConv packs, whose sampling + contraction run on the HIP DeformConv kernels."""
    if cfg is None:
        cfg_ = dict(type='Conv')
    else:
        assert isinstance(cfg, dict) and 'type' in cfg
        cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    table = _conv_types()
    if layer_type not in table:
        raise KeyError('Unrecognized conv type {}'.format(layer_type))
    return table[layer_type](*args, **kwargs, **cfg_)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


class ConvModule(nn.Module):
    """conv -> norm -> ReLU block (order fixed to the one the configs use)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 order=('conv', 'norm', 'act')):
        super(ConvModule, self).__init__()
        assert tuple(order) == ('conv', 'norm', 'act')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = False if self.with_norm else True
        self.with_bias = bias
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                     dilation=dilation, groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg.get('type', 'ReLU') == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        kaiming_init(self.conv, nonlinearity='relu')
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm and isinstance(self.norm, nn.GroupNorm) and not torch.is_grad_enabled() \
                and x.is_cuda and x.dtype == torch.float32:
            # inference: GroupNorm (+ ReLU) as one fused HIP launch pair (mmdet_ops/fused_norm.py)
            from ..mmdet_ops.fused_norm import group_norm_act_multi
            return group_norm_act_multi([x], self.norm, relu=bool(activate and self.with_activation), inplace=True)[0]
        if norm and self.with_norm:
            x = self.norm(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x
