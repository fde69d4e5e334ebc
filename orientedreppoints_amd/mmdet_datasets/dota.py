"""DotaDataset (mmdet/datasets/dota.py:6-83 on coco.py:34-76, custom.py:60-150): COCO-style json with 8-coordinate `bbox`
entries, parsed with the standard library instead of pycocotools."""
import json
import os.path as osp

import numpy as np
from torch.utils.data import Dataset

from .pipelines import Compose, _Registry

DATASETS = _Registry('dataset')


def build_dataset(cfg):
    args = dict(cfg)
    t = args.pop('type')
    cls = DATASETS.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError('%s is not in the dataset registry' % t)
    return cls(**args)


@DATASETS.register_module
class DotaDataset(Dataset):

    CLASSES = ('plane', 'baseball-diamond', 'bridge', 'ground-track-field', 'small-vehicle', 'large-vehicle', 'ship',
               'tennis-court', 'basketball-court', 'storage-tank', 'soccer-ball-field', 'roundabout', 'harbor',
               'swimming-pool', 'helicopter')

    def __init__(self, ann_file, pipeline, data_root=None, img_prefix='', seg_prefix=None, proposal_file=None,
                 test_mode=False, filter_empty_gt=True):
        self.ann_file, self.data_root, self.img_prefix = ann_file, data_root, img_prefix
        self.seg_prefix, self.proposal_file = seg_prefix, proposal_file
        self.test_mode, self.filter_empty_gt = test_mode, filter_empty_gt
        if self.data_root is not None:
            if not osp.isabs(self.ann_file):
                self.ann_file = osp.join(self.data_root, self.ann_file)
            if not (self.img_prefix is None or osp.isabs(self.img_prefix)):
                self.img_prefix = osp.join(self.data_root, self.img_prefix)
        self.img_infos = self.load_annotations(self.ann_file)
        self.proposals = None
        if not test_mode:
            valid_inds = self._filter_imgs()
            self.img_infos = [self.img_infos[i] for i in valid_inds]
        if not self.test_mode:
            self._set_group_flag()
        self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self.img_infos)

    # ---- coco.py:34-76 without pycocotools -------------------------------------------------------------------------
    def load_annotations(self, ann_file):
        with open(ann_file) as f:
            data = json.load(f)
        self.cat_ids = sorted(c['id'] for c in data.get('categories', []))        # COCO.getCatIds(): sorted ids
        self.cat2label = {cat_id: i + 1 for i, cat_id in enumerate(self.cat_ids)}
        self._anns_of = {}
        for ann in data.get('annotations', []):
            self._anns_of.setdefault(ann['image_id'], []).append(ann)
        self.img_ids = [im['id'] for im in data.get('images', [])]               # dict order == file order
        img_infos = []
        for im in data.get('images', []):
            info = dict(im)
            info['filename'] = info['file_name']
            img_infos.append(info)
        return img_infos

    def get_ann_info(self, idx):
        img_id = self.img_infos[idx]['id']
        return self._parse_ann_info(self.img_infos[idx], self._anns_of.get(img_id, []))

    def _filter_imgs(self, min_size=32):
        valid_inds = []
        ids_with_ann = set(self._anns_of.keys())
        for i, img_info in enumerate(self.img_infos):
            if self.filter_empty_gt and self.img_ids[i] not in ids_with_ann:
                continue
            if min(img_info['width'], img_info['height']) >= min_size:
                valid_inds.append(i)
        return valid_inds

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i in range(len(self)):
            if self.img_infos[i]['width'] / self.img_infos[i]['height'] > 1:
                self.flag[i] = 1

    # ---- dota.py:32-82 -----------------------------------------------------------------------------------------------
    def _parse_ann_info(self, img_info, ann_info):
        gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks_ann = [], [], [], []
        for ann in ann_info:
            if ann.get('ignore', False):
                continue
            bbox = ann['bbox']
            if ann.get('iscrowd', False):
                gt_bboxes_ignore.append(bbox)
            else:
                gt_bboxes.append(bbox)
                gt_labels.append(self.cat2label[ann['category_id']])
                gt_masks_ann.append(ann.get('segmentation'))
        if gt_bboxes:
            gt_bboxes = np.array(gt_bboxes, dtype=np.float32)
            gt_labels = np.array(gt_labels, dtype=np.int64)
        else:
            gt_bboxes = np.zeros((0, 8), dtype=np.float32)
            gt_labels = np.array([], dtype=np.int64)
        gt_bboxes_ignore = np.array(gt_bboxes_ignore, dtype=np.float32) if gt_bboxes_ignore else \
            np.zeros((0, 8), dtype=np.float32)
        seg_map = img_info['filename'].replace('jpg', 'png')
        return dict(bboxes=gt_bboxes, labels=gt_labels, bboxes_ignore=gt_bboxes_ignore, masks=gt_masks_ann, seg_map=seg_map)

    # ---- custom.py:110-150 ---------------------------------------------------------------------------------------------
    def pre_pipeline(self, results):
        results['img_prefix'] = self.img_prefix
        results['seg_prefix'] = self.seg_prefix
        results['proposal_file'] = self.proposal_file
        results['bbox_fields'] = []
        results['mask_fields'] = []
        results['seg_fields'] = []

    def _rand_another(self, idx):
        pool = np.where(self.flag == self.flag[idx])[0]
        return np.random.choice(pool)

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    def prepare_train_img(self, idx):
        results = dict(img_info=self.img_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.img_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)
