"""Development aid (GPU box): orp_box_iou_rotated against the reference compiled for the host and against the oracle restatement, BIT BY
BIT, on 2.16 M pairs (6 draws of 900 x 400 rotated boxes, 300 of them placed on top of each other): 0 differing (round 6) -- the test
held 1e-4 until then."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.mmdet_ops import box_iou_rotated
from oracle import orp_oracle as O
dev = torch.device('cuda:0')
tot = bad_ref = bad_orc = 0
for seed in range(6):
    a = S.gen_rboxes(900, 31 + seed).astype(np.float32); b = S.gen_rboxes(400, 62 + seed).astype(np.float32)
    b[:300, :2] = a[:300, :2] + np.random.RandomState(seed).uniform(-8, 8, (300, 2)).astype(np.float32)
    got = box_iou_rotated(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    ref = O.box_iou_rotated(a, b, use_ref=True); orc = O.box_iou_rotated(a, b)
    tot += got.size
    bad_ref += int(np.count_nonzero(got.view(np.uint32) != ref.view(np.uint32)))
    bad_orc += int(np.count_nonzero(got.view(np.uint32) != orc.view(np.uint32)))
    print(seed, 'nonzero ious', int((ref > 0).sum()), 'max|got-ref|', float(np.abs(got - ref).max()), 'oracle vs ref bits differ', int(np.count_nonzero(orc.view(np.uint32) != ref.view(np.uint32))))
print('pairs', tot, 'GPU vs reference-compiled-for-host: differing', bad_ref, '| GPU vs oracle restatement: differing', bad_orc)
