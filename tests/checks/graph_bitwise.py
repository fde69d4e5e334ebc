"""Development aid (GPU box): captured-graph replays of the whole inference step against the eager call, STAGE BY STAGE AND BIT BY BIT.

Round 4 left two graph-replay anomalies that were only ever seen at the level of detections (where one ulp can flip a min-area-rect
tie).  This harness copies every stage's tensors -- backbone outputs, FPN outputs, tower outputs, DeformConv outputs, the head's
outputs, and (fp16-pieces arithmetic) every range word at its consumer -- into buffers allocated BEFORE the capture (no allocation
inside the capture, so the graph pool's block reuse is the product's), for the eager run and for every captured graph, and compares
them with torch.equal.  DEPTH graphs are replayed concurrently on their own streams, as PipelinedInference does.

  SIZE=256 BATCH=2 DEPTH=3 ITERS=200 NIMG=6 SPLIT=on|off|auto MODE=6|3|0 EAGER_BETWEEN=0|1 STASH=1|0 python tests/checks/graph_bitwise.py
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from orientedreppoints_amd import _lib
from orientedreppoints_amd.dota_configs import r50_model, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, build_detector
import importlib
FN = importlib.import_module('orientedreppoints_amd.mmdet_ops.fused_norm')
DC = importlib.import_module('orientedreppoints_amd.mmdet_ops.deform_conv')

SIZE = int(os.environ.get('SIZE', '256'))
BATCH = int(os.environ.get('BATCH', '2'))
DEPTH = int(os.environ.get('DEPTH', '3'))
ITERS = int(os.environ.get('ITERS', '200'))
NIMG = int(os.environ.get('NIMG', '6'))
SPLIT = os.environ.get('SPLIT', 'on')
MODE = int(os.environ.get('MODE', '6'))
EAGER_BETWEEN = os.environ.get('EAGER_BETWEEN', '0') == '1'
STASH = os.environ.get('STASH', '1') == '1'
DET = os.environ.get('DETERMINISTIC', '1') == '1'

dev = torch.device('cuda:0')
L = _lib.lib()
torch.manual_seed(0)
torch.backends.cudnn.deterministic = DET
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
head = model.bbox_head
with torch.no_grad():
    head.reppoints_cls_out.weight.normal_(0, 0.05)
    head.reppoints_cls_out.bias.fill_(-3.3)
    head.reppoints_pts_init_out.bias.copy_(torch.tensor(
        [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]], dtype=torch.float32, device=dev).reshape(-1) * 2.0)
if SPLIT != 'auto':
    head.split_towers = SPLIT == 'on'
    model.neck.split_convs = SPLIT == 'on'
head.tower_streams = False
assert L.orp_dcn_set_split_mode(MODE) == 0
metas = [dict(img_shape=(SIZE, SIZE, 3), pad_shape=(SIZE, SIZE, 3), scale_factor=1.0, flip=False)] * BATCH

# ---- stage stash -----------------------------------------------------------------------------------------------------------------
CUR = None            # dict name -> list of buffers of the set being filled (eager set / one per graph), or None: product path only
COUNTER = [0]


def stash(name, tensors):
    if CUR is None or not STASH:
        return
    bufs = CUR.setdefault(name, None)
    if bufs is None:
        assert not torch.cuda.is_current_stream_capturing(), "stash buffers must exist before the capture (%s)" % name
        bufs = CUR[name] = [torch.empty_like(t) for t in tensors]
    for b, t in zip(bufs, tensors):
        b.copy_(t)


CURLOG = [None]       # the range-word log of the set being filled: the library's launch counter restarts with every forward


def _pre(m, inp):
    COUNTER[0] = 0
    PP[0] = PP[1] = PP[2] = 0
    if CURLOG[0] is not None:
        L.orp_debug_amax_log(ctypes.c_void_p(CURLOG[0].data_ptr()), LOGCAP)


LOGCAP = 32
model.backbone.register_forward_pre_hook(_pre)
model.backbone.register_forward_hook(lambda m, inp, out: stash('backbone', list(out)))
model.neck.register_forward_hook(lambda m, inp, out: stash('fpn', list(out)))
head.register_forward_hook(lambda m, inp, out: (stash('head_cls', list(out[0])), stash('head_init', list(out[1])),
                                                stash('head_refine', list(out[2]))) and None)
_towers_split = head._towers_split


def towers_split(feats):
    r = _towers_split(feats)
    stash('tower_cls', list(r[0])); stash('tower_reg', list(r[1])); stash('tower_hid', list(r[2]))
    return r


head._towers_split = towers_split
_dcn_pair = head._dcn_pair


def dcn_pair(*a, **k):
    r = _dcn_pair(*a, **k)
    stash('dcn_cls', list(r[0])); stash('dcn_pts', list(r[1]))
    return r


head._dcn_pair = dcn_pair
_csw = FN.conv_split_weights


def conv_split_weights(*a, **k):
    am = k.get('amax')
    n = COUNTER[0]; COUNTER[0] += 1
    if am is not None:
        stash('range_before_conv%02d' % n, [am.bits])
    r = _csw(*a, **k)
    if am is not None:
        stash('range_after_conv%02d' % n, [am.bits])
    return r


FN.conv_split_weights = conv_split_weights
_dfp = DC.deform_conv_forward_pair


def deform_conv_forward_pair(*a, **k):
    am = k.get('amax')
    if am is not None:
        stash('range_before_dcn', [am.bits])
    r = _dfp(*a, **k)
    if am is not None:
        stash('range_after_dcn', [am.bits])
    return r


DC.deform_conv_forward_pair = deform_conv_forward_pair

# ---- the post-processing of every image (core.fused_postprocess): candidates, decoded boxes, compacted pairs, NMS keep list ----
CORE = importlib.import_module('orientedreppoints_amd.mmdet_models.core')
MR = importlib.import_module('orientedreppoints_amd.mmdet_ops.minarea_rect')
NW = importlib.import_module('orientedreppoints_amd.mmdet_ops.nms_wrapper')
PP = [0, 0, 0]
_sel, _mrd, _rbd = CORE.select_candidates, MR.minaerarect_decode, NW.rnms_batched_device


def select_candidates(*a, **k):
    r = _sel(*a, **k)
    stash('pp%d_cand' % PP[0], [r]); PP[0] += 1
    return r


def minaerarect_decode(pts_xy, centers, strd):
    r = _mrd(pts_xy, centers, strd)
    stash('pp%d_pts' % PP[1], [pts_xy]); stash('pp%d_boxes' % PP[1], [r]); PP[1] += 1
    return r


def rnms_batched_device(dets, seg, cap, thr):
    stash('pp%d_seg' % PP[2], [seg])
    stash('pp%d_dets' % PP[2], [dets])
    keep, num = _rbd(dets, seg, cap, thr)
    stash('pp%d_num' % PP[2], [num]); stash('pp%d_keep' % PP[2], [keep]); PP[2] += 1
    return keep, num


CORE.select_candidates = select_candidates
MR.minaerarect_decode = minaerarect_decode
NW.rnms_batched_device = rnms_batched_device

def new_log():
    return torch.full((LOGCAP * 4,), -1, dtype=torch.int32, device=dev)


def eager(img, store):
    global CUR
    CUR = store
    CURLOG[0] = store.setdefault('_log', [new_log()])[0]
    with torch.no_grad():
        res = model.simple_test_batch(img, metas)
    L.orp_debug_amax_log(None, 0)
    CUR = None
    CURLOG[0] = None
    return res


imgs = [torch.randn(BATCH, 3, SIZE, SIZE, device=dev, generator=torch.Generator(device=dev).manual_seed(10 + i)) for i in range(NIMG)]

# eager references (a first call allocates the eager set's buffers, then one call per image)
EAG = {}
eager(imgs[0], EAG)
want, want_det = [], []
for im in imgs:
    want_det.append(eager(im, EAG))
    torch.cuda.synchronize()
    want.append({k: [t.clone() for t in v] for k, v in EAG.items()})
# is the eager path itself bitwise reproducible?
unstable = set()
for j, im in enumerate(imgs):
    eager(im, EAG)
    torch.cuda.synchronize()
    for k, v in EAG.items():
        a_, b_ = v, want[j][k]
        if k.startswith('pp') and k.endswith('_keep'):                # entries [0, num) are written, the tail is whatever the allocator handed out
            na, nb = int(EAG[k[:-4] + 'num'][0].reshape(-1)[0]), int(want[j][k[:-4] + 'num'][0].reshape(-1)[0])
            a_, b_ = [v[0].reshape(-1)[:na]], [want[j][k][0].reshape(-1)[:nb]]
        if k.startswith('pp') and k.endswith('_dets'):                # rows [0, seg[1]) are written
            na, nb = int(EAG[k[:-4] + 'seg'][0][1]), int(want[j][k[:-4] + 'seg'][0][1])
            a_, b_ = [v[0][:na]], [want[j][k][0][:nb]]
        if any(a.shape != b.shape or not torch.equal(a, b) for a, b in zip(a_, b_)):
            unstable.add(k)
print("config: size %d batch %d depth %d mode %d split %s eager_between %s stash %s deterministic-library %s; stages: %s"
      % (SIZE, BATCH, DEPTH, MODE, SPLIT, EAGER_BETWEEN, STASH, DET, sorted(k for k in EAG if not k.startswith('range') and k != '_log')))
print("eager run vs eager run: stages that are NOT bitwise reproducible: %s" % (sorted(unstable) or 'none'))
print("detections per image (eager): %s" % [[sum(len(c) for c in r) for r in w] for w in want_det])

# captured graphs, each with its own stash set (buffers allocated by an eager call first)
slots, sets = [], []
for d in range(DEPTH):
    S = {}
    eager(imgs[0], S)
    CUR = S
    CURLOG[0] = S['_log'][0]
    gi = GraphedInference(model, imgs[0], metas)
    L.orp_debug_amax_log(None, 0)
    CUR = None
    CURLOG[0] = None
    slots.append(gi); sets.append(S)
streams = [torch.cuda.Stream(device=dev) for _ in range(DEPTH)]


def describe(name, got, ref):
    out = []
    for i, (a, b) in enumerate(zip(got, ref)):
        if a.shape != b.shape:
            out.append("%s[%d]: %d valid entries vs eager %d" % (name, i, a.shape[0], b.shape[0]))
            continue
        if torch.equal(a, b):
            continue
        if a.dim() <= 2 and a.dtype != torch.int32:
            ne = (a != b)
            rows = torch.nonzero(ne.reshape(ne.shape[0], -1).any(dim=1)).reshape(-1).tolist()
            out.append("%s[%d] %s: rows that differ %s; first: %s vs eager %s" % (name, i, tuple(a.shape), rows[:12],
                       a[rows[0]].reshape(-1)[:10].tolist(), b[rows[0]].reshape(-1)[:10].tolist()))
            continue
        if a.dtype == torch.int32 and a.numel() > 64:
            ne = torch.nonzero(a.reshape(-1) != b.reshape(-1)).reshape(-1).tolist()
            out.append("%s[%d]: %d int32 entries differ, first at %s: %s vs eager %s" % (name, i, len(ne), ne[:8], a.reshape(-1)[ne[:8]].tolist(),
                                                                                        b.reshape(-1)[ne[:8]].tolist()))
            continue
        if a.dtype == torch.int32:
            out.append("%s[%d]: %s vs eager %s" % (name, i, [hex(v & 0xffffffff) for v in a.reshape(-1).tolist()], [hex(v & 0xffffffff) for v in b.reshape(-1).tolist()]))
            continue
        d = (a.float() - b.float()).abs()
        ne = a != b
        msg = "%s[%d] %s: %d of %d elements differ, max |diff| %.3e (scale %.3e), nan %d" % (
            name, i, tuple(a.shape), int(ne.sum()), a.numel(), float(torch.nan_to_num(d).max()), float(b.abs().max()), int(torch.isnan(a).sum()))
        if a.dim() == 4:
            pos = ne.any(dim=1)                                   # [B, H, W]
            npos = int(pos.sum())
            ch = ne.sum(dim=1)[pos]
            msg += "; positions touched %d of %d, channels per touched position min %d max %d" % (
                npos, pos.numel(), int(ch.min()), int(ch.max()))
            idx = torch.nonzero(pos)[:6].tolist()
            msg += "; first positions (b,h,w) %s" % idx
        out.append(msg)
    return out


rng = np.random.RandomState(1)
bad_by_stage, bad_det, shown, t0 = {}, 0, 0, time.time()
order_names = ['backbone', 'fpn', 'tower_cls', 'tower_reg', 'tower_hid', 'dcn_cls', 'dcn_pts', 'head_cls', 'head_init', 'head_refine']
for it in range(ITERS):
    pick = [int(rng.randint(0, NIMG)) for _ in range(DEPTH)]
    cur = torch.cuda.current_stream(dev)
    for d in range(DEPTH):
        s = streams[d]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            slots[d].static_img.copy_(imgs[pick[d]], non_blocking=True)
            slots[d].graph.replay()
    torch.cuda.synchronize()
    for d in range(DEPTH):
        j = pick[d]
        first = None
        names = [n for n in order_names if n in sets[d]] + sorted(k for k in sets[d] if k.startswith('range')) + ['_log']
        lines = []
        names += sorted(k for k in sets[d] if k.startswith('pp'))
        for n in names:
            if n not in want[j]:
                continue
            got_n, want_n = sets[d][n], want[j][n]
            if n.startswith('pp') and n.endswith('_dets'):           # rows [0, seg[1]) are written
                ng, nw = int(sets[d][n[:-4] + 'seg'][0][1]), int(want[j][n[:-4] + 'seg'][0][1])
                got_n, want_n = [got_n[0][:ng]], [want_n[0][:nw]]
            if n.startswith('pp') and n.endswith('_keep'):           # entries [0, num) are written
                ng, nw = int(sets[d][n[:-4] + 'num'][0].reshape(-1)[0]), int(want[j][n[:-4] + 'num'][0].reshape(-1)[0])
                got_n, want_n = [got_n[0].reshape(-1)[:ng]], [want_n[0].reshape(-1)[:nw]]
            if any(a.shape != b.shape or not torch.equal(a, b) for a, b in zip(got_n, want_n)):
                bad_by_stage[n] = bad_by_stage.get(n, 0) + 1
                if first is None and not n.startswith('range') and n != '_log':
                    first = n
                lines += describe(n, got_n, want_n)
        from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
        res = [rbbox2result_packed(p, head.num_classes) for p in slots[d].packed]
        det_same = all(r is not None for r in res) and all(
            a.shape == b.shape and np.array_equal(a, b) for gr, wr in zip(res, want_det[j]) for a, b in zip(gr, wr))
        if not det_same:
            bad_det += 1
        if (lines or not det_same) and shown < 12:
            shown += 1
            print("iteration %d graph %d image %d: first differing stage %s; detections identical: %s (graph %s eager %s)" % (
                it, d, j, first, det_same, [sum(len(c) for c in r) if r is not None else None for r in res],
                [sum(len(c) for c in r) for r in want_det[j]]))
            for ln in lines[:14]:
                print("    " + ln)
    if EAGER_BETWEEN:
        eager(imgs[int(rng.randint(0, NIMG))], EAG)
torch.cuda.synchronize()
print("RESULT size %d batch %d depth %d mode %d split %s eager_between %s stash %s: %d iterations x %d graphs in %.1f s; replays whose "
      "detections differ from eager: %d; stages differing (count of replays): %s" % (
          SIZE, BATCH, DEPTH, MODE, SPLIT, EAGER_BETWEEN, STASH, ITERS, DEPTH, time.time() - t0, bad_det,
          {k: v for k, v in sorted(bad_by_stage.items())} or 'none'))
