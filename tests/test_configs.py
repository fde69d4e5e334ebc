"""CPU: the reference's configs/dota/*.py load UNCHANGED through the mmcv-free Config and every model `type=` string
resolves in the registries; orientedreppoints_amd/dota_configs.py carries the same values (skipped without the
reference tree)."""
import os

import pytest

REF_CFG = "/root/reference/configs/dota"


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present")
def test_reference_configs_load_unchanged_and_build():
    from orientedreppoints_amd.mmdet_models import Config, build_detector
    from orientedreppoints_amd import dota_configs
    names = sorted(os.listdir(REF_CFG))
    assert len(names) == 3
    for f in names:
        cfg = Config.fromfile(os.path.join(REF_CFG, f))
        assert cfg.model.type == 'OrientedRepPointsDetector'
        assert cfg.test_cfg.nms.type == 'rnms' and cfg.dist_params.backend == 'nccl'
        if cfg.model.backbone.type == 'ResNet':
            m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
            keys = set(m.state_dict().keys())
            for k in ('bbox_head.cls_convs.0.conv.weight', 'bbox_head.cls_convs.0.gn.weight',
                      'bbox_head.reppoints_cls_conv.weight', 'bbox_head.reppoints_cls_out.bias',
                      'bbox_head.reppoints_pts_init_conv.weight', 'bbox_head.reppoints_pts_refine_out.weight',
                      'backbone.layer1.0.conv1.weight', 'neck.lateral_convs.0.conv.weight', 'neck.fpn_convs.4.gn.weight'):
                assert k in keys, k
    cfg = Config.fromfile(os.path.join(REF_CFG, 'orientedrepoints_r50_demo.py'))
    ours = dict(dota_configs.r50_model)
    theirs = dict(cfg.model)
    assert theirs['bbox_head'] == ours['bbox_head']
    assert theirs['neck'] == ours['neck']
    assert {k: v for k, v in theirs['backbone'].items()} == ours['backbone']
    assert dict(cfg.test_cfg) == dota_configs.test_cfg
    assert dict(cfg.train_cfg) == dota_configs.train_cfg
