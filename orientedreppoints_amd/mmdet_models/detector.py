"""OrientedRepPointsDetector (mmdet/models/detectors/orientedreppoints_detector.py:9-46 + single_stage.py:13-50 +
base.py:97-149): backbone + neck on stock PyTorch-ROCm, dense head on the HIP operators."""
import torch
import torch.nn as nn

from .core import multiclass_rnms, rbbox2result, rbbox2result_packed
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

    # ---- test-time augmentation (orientedreppoints_detector.py:49-144) -----------------------------------------------------
    def extract_feats(self, imgs):
        assert isinstance(imgs, (list, tuple))
        return [self.extract_feat(img) for img in imgs]

    @staticmethod
    def rbbox_flip(rbboxes, img_shape, direction='horizontal'):
        """Mirror (..., 8k) corner rows inside an image of img_shape = (height, width): x -> w - x - 1 (or y -> h - y - 1)."""
        assert rbboxes.shape[-1] % 8 == 0
        if direction not in ('horizontal', 'vertical'):
            raise ValueError('Invalid flipping direction "{}"'.format(direction))
        axis = 0 if direction == 'horizontal' else 1
        extent = img_shape[1] if direction == 'horizontal' else img_shape[0]
        flipped = rbboxes.clone()
        flipped[..., axis::2] = extent - rbboxes[..., axis::2] - 1
        return flipped

    def rbox_mapping_back(self, rboxes, img_shape, scale_factor, flip):
        """A view's boxes -> the original image: undo the flip, then the resize."""
        return (self.rbbox_flip(rboxes, img_shape) if flip else rboxes) / scale_factor

    def merge_aug_results(self, aug_bboxes, aug_scores, img_metas):
        """Concatenate the views' candidates in original-image coordinates.  aug_bboxes: per view [n, 8k]; aug_scores: per
        view [n, classes + 1] or None; img_metas: per view a one-element list (one image per GPU at test)."""
        recovered = [self.rbox_mapping_back(b, m[0]['img_shape'], m[0]['scale_factor'], m[0]['flip'])
                     for b, m in zip(aug_bboxes, img_metas)]
        bboxes = torch.cat(recovered, dim=0)
        if aug_scores is None:
            return bboxes
        return bboxes, torch.cat(aug_scores, dim=0)

    def aug_test(self, imgs, img_metas, rescale=False):
        """Every view through backbone / neck / head and the decode WITHOUT NMS (fused HIP decode: hull -> min-area rect ->
        image space), candidates of all views mapped back and ONE multiclass rotated NMS over their union.  As in the
        reference, boxes are in the original image's scale when `rescale`, else multiplied by the first view's factor."""
        aug_bboxes, aug_scores = [], []
        for img, img_meta in zip(imgs, img_metas):
            outs = self.bbox_head(self.extract_feat(img))
            det_bboxes, det_scores = self.bbox_head.get_bboxes(*(tuple(outs) + (img_meta, self.test_cfg, False, False)))[0]
            aug_bboxes.append(det_bboxes)
            aug_scores.append(det_scores)
        merged_bboxes, merged_scores = self.merge_aug_results(aug_bboxes, aug_scores, img_metas)
        det_bboxes, det_labels = multiclass_rnms(merged_bboxes, merged_scores, self.test_cfg.score_thr, self.test_cfg.nms,
                                                 self.test_cfg.max_per_img)
        if not rescale:
            det_bboxes = det_bboxes.clone()
            det_bboxes[:, :8] *= img_metas[0][0]['scale_factor']
        return rbbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:97-126: a list of views -> simple_test for one, aug_test for several; a bare tensor is one view."""
        if isinstance(imgs, (list, tuple)):
            if len(imgs) != len(img_metas):
                raise ValueError('num of augmentations ({}) != num of image meta ({})'.format(len(imgs), len(img_metas)))
            if len(imgs) == 1:
                return self.simple_test(imgs[0], img_metas[0], **kwargs)
            assert imgs[0].size(0) == 1, 'aug_test: one image per GPU'
            return self.aug_test(list(imgs), list(img_metas), **kwargs)
        return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        with torch.no_grad():
            return self.forward_test(img, img_meta, **kwargs)
