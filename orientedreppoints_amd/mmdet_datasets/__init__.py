"""Data side of the hot path: the rotated-box pipeline transforms of
mmdet/datasets/pipelines/{transforms.py:43-270, poly_transforms.py:15-546} and `DotaDataset` (mmdet/datasets/dota.py:6-83 on
coco.py:34-76 / custom.py) WITHOUT cv2 / mmcv / pycocotools: numpy for the box arithmetic (exactly the reference's
expressions), torch CPU ops for image resampling.  Pure host code; nothing here touches the GPU."""
from .geometry import box_points, get_best_begin_point, min_area_rect, poly2rbox, rbox2poly  # noqa: F401
from .pipelines import (PIPELINES, HSVAugment, rbbox_flip, rbbox_mapping_back, Collect, Compose, CorrectBox, CorrectRBBox, DefaultFormatBundle, ImageToTensor,  # noqa: F401
                        LoadAnnotations, LoadImageFromFile, MultiScaleFlipAug, Normalize, Pad, PolyRandomFlip,
                        PolyRandomRotate, PolyResize, RotateRandomFlip, RotateResize, build_pipeline)
from .dota import DATASETS, DotaDataset, build_dataset  # noqa: F401
