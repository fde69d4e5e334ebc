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
        # every config builds -- ResNet-50 / 101 and Swin-T (its `pretrained` path is a file of the authors' machine: the
        # builder warns and keeps the random initialisation)
        m = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        keys = set(m.state_dict().keys())
        first = 'backbone.layer1.0.conv1.weight' if cfg.model.backbone.type == 'ResNet' else \
            'backbone.layers.0.blocks.0.attn.relative_position_bias_table'
        for k in ('bbox_head.cls_convs.0.conv.weight', 'bbox_head.cls_convs.0.gn.weight',
                  'bbox_head.reppoints_cls_conv.weight', 'bbox_head.reppoints_cls_out.bias',
                  'bbox_head.reppoints_pts_init_conv.weight', 'bbox_head.reppoints_pts_refine_out.weight',
                  first, 'neck.lateral_convs.0.conv.weight',
                  # (the Swin config has no extra FPN convolutions: levels 4 and 5 are max-pooled)
                  'neck.fpn_convs.%d.gn.weight' % (4 if cfg.model.backbone.type == 'ResNet' else 2)):
            assert k in keys, k
    cfg = Config.fromfile(os.path.join(REF_CFG, 'orientedrepoints_r50_demo.py'))
    ours = dict(dota_configs.r50_model)
    theirs = dict(cfg.model)
    assert theirs['bbox_head'] == ours['bbox_head']
    assert theirs['neck'] == ours['neck']
    assert {k: v for k, v in theirs['backbone'].items()} == ours['backbone']
    assert dict(cfg.test_cfg) == dota_configs.test_cfg
    assert dict(cfg.train_cfg) == dota_configs.train_cfg
    cfg = Config.fromfile(os.path.join(REF_CFG, 'orientedrepoints_swin_tiny_demo.py'))
    theirs = dict(cfg.model)
    for part in ('backbone', 'neck', 'bbox_head'):
        assert dict(theirs[part]) == dota_configs.swin_t_model[part], part
    assert cfg.optimizer.type == dota_configs.swin_t_optimizer['type'] and cfg.optimizer.lr == dota_configs.swin_t_optimizer['lr']
    assert cfg.optimizer.weight_decay == dota_configs.swin_t_optimizer['weight_decay']
    assert tuple(cfg.optimizer.paramwise_cfg.custom_keys.keys()) == dota_configs.swin_t_optimizer['no_decay_keys']


def test_dcn_conv_types_and_resnet_stage_with_dcn():
    """mmdet/ops/conv.py:6-12 ('DCN' / 'DCNv2' conv types) and mmdet/models/backbones/resnet.py:118-161,365-410
    (`dcn=`, `stage_with_dcn`): the model route to DeformConvPack / ModulatedDeformConvPack, with the parameter names of the
    released DCN checkpoints (conv2.weight, conv2.conv_offset.{weight,bias}) and zero-initialised offset predictors."""
    import torch
    from orientedreppoints_amd.mmdet_models import build_backbone
    from orientedreppoints_amd.mmdet_models.layers import ConvModule, build_conv_layer
    from orientedreppoints_amd.mmdet_ops.deform_conv import DeformConvPack, ModulatedDeformConvPack
    assert isinstance(build_conv_layer(dict(type='DCN'), 8, 8, 3, padding=1), DeformConvPack)
    m = build_conv_layer(dict(type='DCNv2', deformable_groups=2), 8, 8, 3, padding=1, bias=False)
    assert isinstance(m, ModulatedDeformConvPack) and m.conv_offset.out_channels == 2 * 27 and m.bias is None
    assert isinstance(build_conv_layer(None, 8, 8, 3), torch.nn.Conv2d)
    with pytest.raises(KeyError):
        build_conv_layer(dict(type='ConvWS'), 8, 8, 3)
    cm = ConvModule(8, 8, 3, padding=1, conv_cfg=dict(type='DCN'), norm_cfg=dict(type='GN', num_groups=2))
    assert isinstance(cm.conv, DeformConvPack)
    for dcn, cls, mult in ((dict(type='DCNv2', deformable_groups=1, fallback_on_stride=False), ModulatedDeformConvPack, 27),
                           (dict(modulated=False, deformable_groups=1, fallback_on_stride=False), DeformConvPack, 18)):
        net = build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                                  style='pytorch', dcn=dcn, stage_with_dcn=(False, True, True, True)))
        net.init_weights(None)
        assert isinstance(net.layer1[0].conv2, torch.nn.Conv2d)
        for name in ('layer2', 'layer3', 'layer4'):
            for blk in getattr(net, name):
                assert isinstance(blk.conv2, cls) and blk.conv2.conv_offset.out_channels == mult
                assert float(blk.conv2.conv_offset.weight.detach().abs().max()) == 0.0
        assert net.layer2[0].conv2.stride == (2, 2) or net.layer2[0].conv2.stride == 2
        keys = set(net.state_dict().keys())
        assert {'layer2.0.conv2.weight', 'layer2.0.conv2.conv_offset.weight', 'layer2.0.conv2.conv_offset.bias'} <= keys
    net = build_backbone(dict(type='ResNet', depth=50, dcn=dict(type='DCN', fallback_on_stride=True),
                              stage_with_dcn=(False, True, True, True)))
    assert isinstance(net.layer2[0].conv2, torch.nn.Conv2d) and isinstance(net.layer2[1].conv2, DeformConvPack)
    # the reference POPS the flag from the dict every block of every stage receives (resnet.py:145-148): only the first block
    # built from it falls back; the stride-2 first blocks of the later stages are deformable -- the key set of its checkpoints
    assert isinstance(net.layer3[0].conv2, DeformConvPack) and isinstance(net.layer4[0].conv2, DeformConvPack)
    assert 'layer2.0.conv2.conv_offset.weight' not in net.state_dict() and 'layer3.0.conv2.conv_offset.weight' in net.state_dict()


def test_pipelined_inference_refuses_the_combination_that_stalls_the_device():
    """A model.half() detector with FOUR captured graphs in flight stalled the device when the library was restricted to its reproducible
    solvers (torch.backends.cudnn.deterministic = True; round 6, docs/notebook/round6.md 8) -- in the default mode it runs four deep at the
    fp32 rate.  The constructor raises for that combination before anything is captured (no GPU needed to see it)."""
    import torch
    from orientedreppoints_amd import dota_configs
    from orientedreppoints_amd.mmdet_models import ConfigDict, PipelinedInference, build_detector
    model = build_detector(ConfigDict(dota_configs.r50_model), train_cfg=None, test_cfg=ConfigDict(dota_configs.test_cfg)).eval().half()
    img = torch.zeros(1, 3, 64, 64, dtype=torch.float16)
    metas = [dict(img_shape=(64, 64, 3), pad_shape=(64, 64, 3), scale_factor=1.0, flip=False)]
    flag = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        with pytest.raises(ValueError, match="at most two graphs in flight"):
            PipelinedInference(model, img, metas, depth=4)
    finally:
        torch.backends.cudnn.deterministic = flag


def test_detector_with_a_side_stream_can_be_deep_copied_and_pickled():
    """A head that has run its two towers on two streams (ORP_DCN_SPLIT=0, or training) holds a torch.cuda.Stream; copy.deepcopy(model)
    failed with "cannot pickle 'Stream' object" (round 6: found by running the GPU suite under ORP_DCN_SPLIT=0).  The stream is per
    process: copies and pickles drop it and create their own on first use."""
    import copy, io
    import torch
    from orientedreppoints_amd import dota_configs
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    model = build_detector(ConfigDict(dota_configs.r50_model), train_cfg=None, test_cfg=ConfigDict(dota_configs.test_cfg)).eval()

    class NotPicklable(object):                              # what a Stream looks like to copy / pickle
        def __reduce_ex__(self, protocol):
            raise TypeError("cannot pickle 'Stream' object")
    model.bbox_head._side_stream = NotPicklable()
    twin = copy.deepcopy(model)
    assert twin.bbox_head._side_stream is None and isinstance(model.bbox_head._side_stream, NotPicklable)
    assert set(twin.state_dict().keys()) == set(model.state_dict().keys())
    torch.save(model, io.BytesIO())
