#!/usr/bin/env python3
"""Golden outputs of the reference's own SwinTransformer module (mmdet/models/backbones/swin_transformer.py), executed on
the CPU where it lies under stub parents (timm / mmcv_custom / mmdet.utils are absent here: DropPath, to_2tuple,
trunc_normal_ and the registry are stubbed; none of them carries arithmetic at drop rate 0 / eval).  A reduced configuration
(embed_dim 24, depths 2-2-2-2, window 7) keeps the fixture small; the WEIGHTS are not stored: both sides fill every parameter
from `fill_parameters` below (seeded, sorted key order), which also proves the state-dict key compatibility.  Inputs whose
feature maps are NOT multiples of the window / the 2 x 2 merge exercise every padding branch.

    python tests/golden/make_golden_swin.py      -> tests/golden/swin_py.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
CFG = dict(embed_dim=24, depths=[2, 2, 2, 2], num_heads=[2, 4, 4, 8], window_size=7, mlp_ratio=4., qkv_bias=True,
           qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, ape=False, patch_norm=True,
           out_indices=(1, 2, 3), use_checkpoint=False)
CASES = [(1, 150, 203, 5), (2, 224, 224, 6), (1, 61, 97, 7)]            # (batch, H, W, seed)


def fill_parameters(model, seed=1234):
    """Every parameter (sorted by name) drawn from a seeded normal; LayerNorm weights around 1.  Buffers stay as built."""
    rng = np.random.RandomState(seed)
    sd = model.state_dict()
    params = dict(model.named_parameters())
    for k in sorted(params):
        shape = tuple(params[k].shape)
        if k.endswith('norm.weight') or '.norm1.weight' in k or '.norm2.weight' in k or (k.startswith('norm') and k.endswith('weight')):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif k.endswith('bias'):
            v = 0.05 * rng.standard_normal(shape)
        elif 'relative_position_bias_table' in k:
            v = 0.5 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = rng.standard_normal(shape) / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    model.load_state_dict(sd)
    return sorted(sd.keys())


def inputs(case):
    b, h, w, seed = case
    return torch.from_numpy(np.random.RandomState(seed).standard_normal((b, 3, h, w)).astype(np.float32))


def load_reference_swin():
    def stub(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        m.__dict__.update(attrs)
        return m

    class _Registry:
        def register_module(self, cls=None):
            return cls if cls is not None else (lambda c: c)

    class DropPath(nn.Module):                      # stochastic depth: identity in eval mode (what the golden run uses)
        def __init__(self, p=0.):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert not self.training
            return x
    stub('timm'); stub('timm.models')
    stub('timm.models.layers', DropPath=DropPath, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
         trunc_normal_=nn.init.trunc_normal_)
    stub('mmcv_custom', load_checkpoint=None)
    stub('mmdet'); stub('mmdet.utils', get_root_logger=lambda *a, **k: None)
    stub('mmdet.models'); stub('mmdet.models.registry', BACKBONES=_Registry())
    stub('mmdet.models.backbones')
    spec = importlib.util.spec_from_file_location('mmdet.models.backbones.swin_transformer',
                                                  os.path.join(REF, 'mmdet/models/backbones/swin_transformer.py'))
    m = importlib.util.module_from_spec(spec)
    m.__package__ = 'mmdet.models.backbones'
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m.SwinTransformer


if __name__ == "__main__":
    Swin = load_reference_swin()
    torch.manual_seed(0)
    model = Swin(**CFG)
    model.eval()                                   # (the reference's train() override returns None)
    keys = fill_parameters(model)
    out = {'keys': np.array(keys)}
    with torch.no_grad():
        for ci, case in enumerate(CASES):
            for li, y in enumerate(model(inputs(case))):
                out['case%d_out%d' % (ci, li)] = y.numpy()
    # the full-size module's key set (Swin-T of the DOTA config): names + shapes only
    full = Swin(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], out_indices=(1, 2, 3))
    fsd = full.state_dict()
    out['full_keys'] = np.array(sorted(fsd.keys()))
    out['full_shapes'] = np.array([','.join(map(str, fsd[k].shape)) for k in sorted(fsd.keys())])
    np.savez_compressed(os.path.join(OUT, 'swin_py.npz'), **out)
    print("wrote swin_py.npz:", {k: v.shape for k, v in out.items() if k.startswith('case')})
