"""Host-side mirror of the reference's model / registry surface for the dense-head hot path.  Importing this package
registers every `type=` string the DOTA configs use for the model (detector, backbone, neck, head, losses,
assigners)."""
from .config import Config, ConfigDict  # noqa: F401
from .registry import (BACKBONES, NECKS, HEADS, LOSSES, DETECTORS, BBOX_ASSIGNERS, build_detector, build_loss,  # noqa: F401
                       build_head, build_backbone, build_neck, build_assigner)
from . import resnet, swin, fpn, losses, assigners, orientedreppoints_head, detector  # noqa: F401
from .orientedreppoints_head import OrientedRepPointsHead  # noqa: F401
from .detector import OrientedRepPointsDetector  # noqa: F401
from .graph_inference import GraphedInference, PipelinedInference  # noqa: F401
