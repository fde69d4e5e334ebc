"""CPU: the plain-PyTorch Swin backbone (orientedreppoints_amd/mmdet_models/swin.py) against golden outputs of the reference's
own SwinTransformer module executed on the CPU (tests/golden/make_golden_swin.py -> swin_py.npz): same state-dict keys (the
released checkpoints load), same outputs on inputs that exercise every padding branch, stochastic depth / checkpointing /
frozen stages behave, and the Swin-T DOTA config builds and takes the head's inputs."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_swin", os.path.join(HERE, "golden", "make_golden_swin.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_swin_outputs_and_state_dict_keys_match_the_reference_module():
    from orientedreppoints_amd.mmdet_models.swin import SwinTransformer
    G = np.load(os.path.join(HERE, "golden", "swin_py.npz"))
    gen = _gen()
    model = SwinTransformer(**gen.CFG).eval()
    keys = gen.fill_parameters(model)                       # load_state_dict inside: names and shapes must match
    assert keys == list(G["keys"])
    with torch.no_grad():
        for ci, case in enumerate(gen.CASES):
            outs = model(gen.inputs(case))
            assert len(outs) == 3
            for li, y in enumerate(outs):
                want = G["case%d_out%d" % (ci, li)]
                assert tuple(y.shape) == want.shape
                assert np.max(np.abs(y.numpy() - want)) <= 2e-5 * max(1.0, float(np.max(np.abs(want)))), (ci, li)
    # the full Swin-T of configs/dota/orientedrepoints_swin_tiny_demo.py: the released checkpoint's key set and shapes
    full = SwinTransformer(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], out_indices=(1, 2, 3))
    sd = full.state_dict()
    assert sorted(sd.keys()) == list(G["full_keys"])
    assert [",".join(map(str, sd[k].shape)) for k in sorted(sd.keys())] == list(G["full_shapes"])


def test_swin_training_features():
    from orientedreppoints_amd.mmdet_models.swin import SwinTransformer, StochasticDepth
    gen = _gen()
    cfg = dict(gen.CFG)
    torch.manual_seed(0)
    a = SwinTransformer(**cfg)
    gen.fill_parameters(a)
    cfg_cp = dict(cfg); cfg_cp['use_checkpoint'] = True
    b = SwinTransformer(**cfg_cp)
    b.load_state_dict(a.state_dict())
    x = gen.inputs((2, 96, 80, 3))
    # activation checkpointing recomputes the same forward: same outputs and gradients (stochastic depth off for the comparison)
    for m in list(a.modules()) + list(b.modules()):
        if isinstance(m, StochasticDepth):
            m.p = 0.0
    a.train(); b.train()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    sum(o.sum() for o in a(xa)).backward()
    sum(o.sum() for o in b(xb)).backward()
    assert torch.allclose(xa.grad, xb.grad, atol=1e-5)
    ga = dict(a.named_parameters())['layers.2.blocks.1.attn.qkv.weight'].grad
    gb = dict(b.named_parameters())['layers.2.blocks.1.attn.qkv.weight'].grad
    assert torch.allclose(ga, gb, atol=1e-5) and float(ga.abs().max()) > 0
    # stochastic depth: rates grow linearly to drop_path_rate, identity in eval, per-sample drop with rescaling in train
    rates = [m.p for m in SwinTransformer(**cfg).modules() if isinstance(m, StochasticDepth)]
    assert len(rates) == 7 and abs(rates[-1] - 0.2) < 1e-6 and rates == sorted(rates)
    sd = StochasticDepth(0.5).train()
    y = sd(torch.ones(4000, 3))
    assert set(np.unique(y.numpy()).tolist()) == {0.0, 2.0} and 0.4 < float((y[:, 0] == 0).float().mean()) < 0.6
    assert torch.equal(sd.eval()(torch.ones(5, 3)), torch.ones(5, 3))
    # frozen stages: patch embedding + the first stage without gradient and in eval mode after .train()
    f = SwinTransformer(frozen_stages=2, **cfg).train()
    assert not f.patch_embed.proj.weight.requires_grad and not f.layers[0].blocks[0].attn.qkv.weight.requires_grad
    assert f.layers[1].blocks[0].attn.qkv.weight.requires_grad and not f.layers[0].training and f.layers[1].training


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs/dota"), reason="reference tree not present")
def test_swin_tiny_dota_config_builds_and_feeds_the_neck():
    from orientedreppoints_amd.mmdet_models import Config, build_detector
    cfg = Config.fromfile("/root/reference/configs/dota/orientedrepoints_swin_tiny_demo.py")
    cfg.model.pretrained = None
    m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg).eval()
    with torch.no_grad():
        feats = m.neck(m.backbone(torch.randn(1, 3, 128, 160)))
    assert [tuple(f.shape[1:]) for f in feats] == [(256, 16, 20), (256, 8, 10), (256, 4, 5), (256, 2, 3), (256, 1, 2)]
