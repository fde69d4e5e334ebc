"""Model / test settings of the DOTA R-50 and R-101 configs as plain data, for bench.py and the GPU tests (the
reference tree is not present on the GPU box).  Values are those of configs/dota/orientedrepoints_r50_demo.py:1-67
and orientedrepoints_r101_demo.py; tests/test_configs.py checks them against the reference files when the tree is
available and checks that the reference configs themselves load unchanged through mmdet_models.Config."""

_norm_cfg = dict(type='GN', num_groups=32, requires_grad=True)


def _model(depth):
    return dict(
        type='OrientedRepPointsDetector',
        pretrained=None,
        backbone=dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=True), style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs=True, num_outs=5, norm_cfg=_norm_cfg),
        bbox_head=dict(
            type='OrientedRepPointsHead', num_classes=16, in_channels=256, feat_channels=256,
            point_feat_channels=256, stacked_convs=3, num_points=9, gradient_mul=0.3,
            point_strides=[8, 16, 32, 64, 128], point_base_scale=2, norm_cfg=_norm_cfg,
            loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
            loss_rbox_init=dict(type='GIoULoss', loss_weight=0.375),
            loss_rbox_refine=dict(type='GIoULoss', loss_weight=1.0),
            loss_spatial_init=dict(type='SpatialBorderLoss', loss_weight=0.05),
            loss_spatial_refine=dict(type='SpatialBorderLoss', loss_weight=0.1),
            top_ratio=0.4))


r50_model = _model(50)
r101_model = _model(101)


def _with_backbone(base, backbone, neck=None):
    m = dict(base)
    m['backbone'] = backbone
    if neck is not None:
        m['neck'] = neck
    return m


# configs/dota/orientedrepoints_swin_tiny_demo.py:4-53 (Swin-T; three backbone outputs, FPN levels 4 / 5 max-pooled)
swin_t_model = _with_backbone(
    r50_model,
    dict(type='SwinTransformer', embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.,
         qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, ape=False, patch_norm=True,
         out_indices=(1, 2, 3), use_checkpoint=False),
    dict(type='FPN', in_channels=[192, 384, 768], out_channels=256, num_outs=5, norm_cfg=_norm_cfg))
# the optimizer of that config (:128-131): AdamW, no weight decay on norms / position-bias tables
swin_t_optimizer = dict(type='AdamW', lr=0.0001, betas=(0.9, 0.999), weight_decay=0.05,
                        no_decay_keys=('absolute_pos_embed', 'relative_position_bias_table', 'norm'))

# not a reference config file: the R-50 model with the backbone's DCN option of mmdet/models/backbones/resnet.py:365-410
# switched on (conv2 of stages 2-4 = ModulatedDeformConvPack) -- the model route to the DCNv2 kernels BASELINE configs[4] names
r50_dcnv2_model = _with_backbone(
    r50_model,
    dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
         norm_cfg=dict(type='BN', requires_grad=True), style='pytorch',
         dcn=dict(type='DCNv2', deformable_groups=1, fallback_on_stride=False), stage_with_dcn=(False, True, True, True)))

train_cfg = dict(
    init=dict(assigner=dict(type='PointAssigner', scale=4, pos_num=1), allowed_border=-1, pos_weight=-1, debug=False),
    refine=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.1, neg_iou_thr=0.1, min_pos_iou=0,
                              ignore_iof_thr=-1),
                allowed_border=-1, pos_weight=-1, debug=False))

test_cfg = dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(type='rnms', iou_thr=0.4), max_per_img=2000)
