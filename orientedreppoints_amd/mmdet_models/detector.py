"""OrientedRepPointsDetector (mmdet/models/detectors/orientedreppoints_detector.py:9-46 + single_stage.py:13-50 +
base.py:97-149): backbone + neck on stock PyTorch-ROCm, dense head on the HIP operators."""
import torch
import torch.nn as nn

from .core import rbbox2result, rbbox2result_packed
from .registry import DETECTORS, build_backbone, build_head, build_neck


@DETECTORS.register_module
class OrientedRepPointsDetector(nn.Module):

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super(OrientedRepPointsDetector, self).__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        self.bbox_head = build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return self.neck is not None

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        self.bbox_head.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_dummy(self, img):
        return self.bbox_head(self.extract_feat(img))

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_rbboxes_ignore=None):
        x = self.extract_feat(img)
        outs = self.bbox_head(x)
        loss_inputs = tuple(outs) + (gt_bboxes, gt_labels, img_metas, self.train_cfg)
        return self.bbox_head.loss(*loss_inputs, gt_rbboxes_ignore=gt_rbboxes_ignore)

    def simple_test(self, img, img_metas, rescale=False):
        """Reference behaviour: the results of the first image (base.py asserts imgs_per_gpu == 1 at test)."""
        return self.simple_test_batch(img, img_metas, rescale)[0]

    def simple_test_batch(self, img, img_metas, rescale=False):
        """Extension for BASELINE config 4 (2 img/GPU): the per-image loop already present in get_bboxes is surfaced."""
        x = self.extract_feat(img)
        outs = self.bbox_head(x)
        bbox_inputs = tuple(outs) + (img_metas, self.test_cfg, rescale)
        if img.is_cuda and self.test_cfg.get('static_postprocess', True) and \
                self.test_cfg.nms.get('type', 'rnms') == 'rnms' and not torch.is_grad_enabled():
            # decode -> multiclass NMS -> packing without a single host synchronisation; ONE D2H per image at the end
            packed = self.bbox_head.get_bboxes(*bbox_inputs, static=True)
            results = [rbbox2result_packed(p, self.bbox_head.num_classes) for p in packed]
            if all(r is not None for r in results):
                return results
        bbox_list = self.bbox_head.get_bboxes(*bbox_inputs)
        return [rbbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)
                for det_bboxes, det_labels in bbox_list]

    def forward_test(self, imgs, img_metas, **kwargs):
        if isinstance(imgs, (list, tuple)):
            assert len(imgs) == 1, 'aug_test is not part of the hot path build'
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        with torch.no_grad():
            return self.forward_test(img, img_meta, **kwargs)
