"""Host-side mirror of the reference's operator API for the dense-head hot path (mmdet/ops/* in
LiWentomng/OrientedRepPoints), bound to the MI355X HIP library through include/orp_hip.h.

Same names, argument meaning and error behaviour as the reference wrappers, so `from mmdet.ops import X` call sites
can be re-pointed here unchanged.
"""
from .nms_wrapper import rnms, rnms_cuda, poly_nms_gpu, soft_rnms  # noqa: F401
from .minarea_rect import minaerarect  # noqa: F401
from .iou_wrapper import convex_iou, convex_overlaps, convex_giou  # noqa: F401
from .chamfer_distance import ChamferDistance2D, Chamfer2D  # noqa: F401
from .point_justify import pointsJf, points_in_quad_aligned  # noqa: F401
from .sigmoid_focal_loss import SigmoidFocalLoss, sigmoid_focal_loss  # noqa: F401
from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack,  # noqa: F401
                          deform_conv, modulated_deform_conv, deform_conv_forward_multi, deform_conv_forward_multi_half, deform_conv_forward_pair,
                          deform_conv_pair)
from .box_iou_rotated import box_iou_rotated  # noqa: F401
