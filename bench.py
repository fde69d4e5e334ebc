#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the Oriented RepPoints dense-head hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): OrientedRepPoints R-50 FPN inference, 1024x1024 DOTA patch, 15 classes,
bs = 1 per GPU.  A "step" is one full `simple_test`: stock PyTorch-ROCm ResNet-50 + FPN convolutions (eval BatchNorm +
residual + ReLU as one fused HIP pass), the dense head (tower convs: library on the two big levels, one HIP MFMA launch
for the three small ones; GroupNorm+ReLU and the bias passes as fused HIP launches; both DeformConvs on the HIP MFMA
kernel), decode (fused min-area-rect kernel), multiclass rotated NMS
(HIP mask + on-device sweep) and rbbox2result (the D2H the reference's test loop also pays).  Inputs are resident in
HBM before the timed region.  Random-init weights and a synthetic image (no network): because a random-init head
scores ~0.01 everywhere and predicts zero-size point sets, two biases are calibrated ONCE before timing so that the
post-processing sees a realistic dense scene (see `calibrate_head`) -- the compute of every layer is unchanged.

The K steps are timed three times: launched eagerly from Python (`eager_ms_per_step`; this loop also provides the live
HIP-event kernel timings), as ONE hipGraph replay per step with the host waiting for each result
(`mmdet_models.GraphedInference`, `graph_replay_ms`: same kernels, same detections, ~220 launches leave the host), and in
throughput mode (`mmdet_models.PipelinedInference`, `pipelined_ms_per_step`): `--pipeline` captured graphs in flight on
their own streams, results fetched asynchronously and ALL collected inside the timed bracket -- `value` / `mode` report the
last one that works.

One JSON line on rank 0: metric/value = whole-job images/sec; plus `roofline` for the dominant hot-path kernel (the
DeformConv implicit GEMM, timed live with HIP events inside liborp_hip.so over the timed region, MFMA-bound), `nms`
(the rotated-IoU + NMS stage: us/img, mask/sweep kernel times), `nms_batched_16_images` (the (image x class)-batched
form of BASELINE.md section 3) and `cpu_baseline` (polyiou + py_cpu_nms_poly on the host cores on one image's detections:
the reference's own polyiou.cpp when oracle/_ref is present, the oracle's C port otherwise, plus the fast / Pool / fp32
variants of SURVEY 8d).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from orientedreppoints_amd import _lib  # noqa: E402
from orientedreppoints_amd.dota_configs import (r50_model, r101_model, swin_t_model, r50_dcnv2_model,  # noqa: E402
                                                swin_t_optimizer, test_cfg as TEST_CFG)
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD at 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), 16x the fp32 matrix rate
IMG = 1024                       # --size
MODELS = {'r50': r50_model, 'r101': r101_model, 'swin_t': swin_t_model, 'r50_dcnv2': r50_dcnv2_model}
MODEL_LABEL = {'r50': 'R-50', 'r101': 'R-101', 'swin_t': 'Swin-T', 'r50_dcnv2': 'R-50-DCNv2'}
TRAIN_SIZES = None               # --mode train with --size a,b,...: the multi-scale patch sizes (configs[4]: 1024,1536)
TARGET_DETS = 2000               # (point, class) pairs above score_thr per image: the "dense scene" of BASELINE configs


def calibrate_head(model, img, target=TARGET_DETS):
    """Make the random-init head emit a realistic post-processing load without touching any layer's compute:
      * reppoints_pts_init_out.bias <- a 3x3 grid (in feature-grid units, (y,x) order) whose extent varies per channel
        through the existing random weights (std raised to 0.05) -> decoded boxes have non-zero size / orientation;
      * reppoints_cls_out.bias      <- shifted PER CLASS so that each of the 15 classes contributes ~target / 15 of the
        (point, class) scores above score_thr: the timed NMS then sees the class-offset coordinates of multiclass_rnms
        (bbox_nms.py:156-158) the fp32 IoU arithmetic has to reproduce, and the per-class CPU baseline is 15-way.
        (Round 2 shifted one common bias: all ~2 000 detections fell into one class.)
    """
    head = model.bbox_head
    with torch.no_grad():
        base = torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
                            dtype=torch.float32, device=img.device).reshape(-1) * 2.0
        head.reppoints_pts_init_out.bias.copy_(base)
        head.reppoints_pts_init_out.weight.normal_(0, 0.05)
        head.reppoints_pts_refine_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        feats = model.extract_feat(img)
        cls_outs, _, _, _ = head(feats)
        C = cls_outs[0].size(1)
        logits = torch.cat([c.permute(1, 0, 2, 3).reshape(C, -1) for c in cls_outs], 1)       # [C, B*N]
        thr_logit = float(np.log(TEST_CFG['score_thr'] / (1 - TEST_CFG['score_thr'])))
        per_class = [target // C + (1 if c < target % C else 0) for c in range(C)]
        for c in range(C):
            k = max(1, min(per_class[c], logits.size(1) - 1))
            kth = torch.topk(logits[c], k).values[-1]
            head.reppoints_cls_out.bias[c] += thr_logit - kth + 1e-4


PMC_FILE = 'profiles/r06_pmc.json'


def load_pmc():
    """(counters dict, provenance note) of the committed rocprofv3 --pmc passes -- or ({}, why not) when the passes were
    collected on a different build of liborp_hip.so than the one that runs now (`orp_version` carries a hash of the
    kernel sources): stale counters are not reported."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return {}, '%s not present' % PMC_FILE
    try:
        pmc = json.load(open(path))
    except Exception as e:   # noqa: BLE001
        return {}, '%s unreadable: %s' % (PMC_FILE, e)
    have = _lib.lib().orp_version().decode()
    want = pmc.get('_build', {}).get('orp_version')
    if want != have:
        return {}, '%s was collected on build "%s", this run uses "%s": traffic not reported' % (PMC_FILE, want, have)
    return pmc, '%s (rocprofv3 --pmc passes on this build, collected separately)' % PMC_FILE


def read_prof(slot):
    tot = ctypes.c_double(0)
    cnt = ctypes.c_int(0)
    _lib.lib().orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p),
                                ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
    return tot.value, cnt.value


def nms_inputs_of_one_image(model, img, metas):
    """The [M,9] class-offset dets multiclass_rnms hands to rnms for this image (for the CPU baseline + byte counts)."""
    captured = {}
    from orientedreppoints_amd.mmdet_ops import nms_wrapper
    orig = nms_wrapper.rnms

    def spy(dets, iou_thr, device_id=None):
        captured['dets'] = dets.detach().clone()
        captured['thr'] = iou_thr
        return orig(dets, iou_thr, device_id)
    nms_wrapper.rnms = spy
    from orientedreppoints_amd.mmdet_models import orientedreppoints_head as head_mod
    orig_mc = head_mod.multiclass_rnms

    def spy_mc(multi_bboxes, multi_scores, score_thr, *a, **k):
        captured['labels'] = (multi_scores[:, 1:] > score_thr).nonzero()[:, 1].detach().clone()
        return orig_mc(multi_bboxes, multi_scores, score_thr, *a, **k)
    head_mod.multiclass_rnms = spy_mc
    static = model.test_cfg.get('static_postprocess', True)
    model.test_cfg['static_postprocess'] = False      # the reference-shaped path hands the [M,9] dets to `rnms`
    try:
        with torch.no_grad():
            model.simple_test(img, metas)
    finally:
        nms_wrapper.rnms = orig
        head_mod.multiclass_rnms = orig_mc
        model.test_cfg['static_postprocess'] = static
    return captured


def _timed(fn, budget_s, max_reps=64):
    total, reps, out = 0.0, 0, None
    while total < budget_s and reps < max_reps:
        t0 = time.perf_counter()
        out = fn()
        total += time.perf_counter() - t0
        reps += 1
    return total / reps, out, reps


def _pool_nms(args):
    """Pool worker (module level: picklable): one class's detections through the CPU NMS."""
    from oracle import orp_oracle as O
    d64, thr, use_ref = args
    return len(O.ref_py_cpu_nms_poly(d64, thr) if use_ref else O.py_cpu_nms_poly(d64, thr))


def cpu_baselines(dets_np, labels_np, thr, budget_s=6.0):
    """The CPU side of SURVEY 8d on ONE image's detections (the [M,9] class-offset set multiclass_rnms hands to rnms),
    each variant repeated until ~budget_s of CPU work is spent:
      B0  polyiou.cpp iou_poly + the py_cpu_nms_poly greedy loop (ResultMerge.py:18-41), fp64, 1 core -- with the
          REFERENCE's own polyiou.cpp compiled in oracle/_ref when that library travelled here (kind "reference"), and
          with the oracle's C restatement of it (kind "port");
      B0' py_cpu_nms_poly_fast (ResultMerge_multi_process.py:60-121: polyiou only where the HBBs overlap), 1 core;
      B1  the per-class multiprocessing.Pool split of ResultMerge_multi_process.py:225-231 over the host cores (<= 16);
      B2  rnms_cpu.cpp's fp32 rotate_iou, hard NMS (soft_rnms method 0 semantics), 1 core;
      B3  microseconds per polyiou call.
    The loops are C (no SWIG / numpy overhead per pair): a conservative -- i.e. fast -- statement of the reference."""
    import multiprocessing as mp
    from oracle import orp_oracle as O
    O.build()
    have_ref = O.ref() is not None and hasattr(O.ref(), 'ref_py_cpu_nms_poly')
    d64 = np.ascontiguousarray(dets_np, np.float64)
    M = d64.shape[0]
    v = {}
    dt_port, keep, reps = _timed(lambda: O.py_cpu_nms_poly(d64, thr), budget_s)
    v['B0 polyiou + py_cpu_nms_poly, oracle C port, 1 core'] = dict(us_per_img=dt_port * 1e6, kept=len(keep), repeats=reps)
    dt_ref = None
    if have_ref:
        dt_ref, keep_r, reps = _timed(lambda: O.ref_py_cpu_nms_poly(d64, thr), budget_s)
        v['B0 polyiou + py_cpu_nms_poly, reference polyiou.cpp, 1 core'] = dict(us_per_img=dt_ref * 1e6, kept=len(keep_r), repeats=reps)
    dt_fast, keep_f, reps = _timed(lambda: O.py_cpu_nms_poly_fast(d64, thr, use_ref=have_ref), min(budget_s, 2.0))
    v["B0' py_cpu_nms_poly_fast (HBB prefilter), 1 core"] = dict(us_per_img=dt_fast * 1e6, kept=len(keep_f), repeats=reps)
    if have_ref:
        d32 = np.ascontiguousarray(dets_np, np.float32)
        ds = np.ascontiguousarray(d32[O.sort_order(d32[:, 8])])
        dt2, keep2, reps = _timed(lambda: O.ref_rnms_cpu_hard(ds, thr), budget_s)
        v['B2 rnms_cpu.cpp rotate_iou fp32 hard NMS, 1 core'] = dict(us_per_img=dt2 * 1e6, kept=len(keep2), repeats=reps)
        rng = np.random.RandomState(0)
        ia, ib = rng.randint(0, M, 200000), rng.randint(0, M, 200000)
        pa, pb = np.ascontiguousarray(d64[ia, :8]), np.ascontiguousarray(d64[ib, :8])
        t0 = time.perf_counter(); O.ref_polyiou_many(pa, pb); t1 = time.perf_counter() - t0
        v['B3 polyiou.cpp iou_poly'] = dict(us_per_iou=t1 / 200000 * 1e6, pairs=200000)
    cores = max(1, min(16, os.cpu_count() or 1))          # the reference uses Pool(16)
    if labels_np is not None and cores > 1:
        parts = [(np.ascontiguousarray(d64[labels_np == c]), thr, have_ref) for c in np.unique(labels_np)]
        try:
            ctx = mp.get_context('fork')
            with ctx.Pool(cores) as pool:
                pool.map(_pool_nms, parts)                # warm the workers
                dt1, kept1, reps = _timed(lambda: sum(pool.map(_pool_nms, parts)), min(budget_s, 3.0))
            v['B1 per-class Pool(%d)' % cores] = dict(us_per_img=dt1 * 1e6, kept=kept1, repeats=reps, cores=cores,
                                                      classes=len(parts))
        except Exception as e:   # noqa: BLE001
            v['B1 per-class Pool(%d)' % cores] = 'failed: %s' % (str(e)[:120],)
    head = dt_ref if dt_ref is not None else dt_port
    b1 = next((x for k, x in v.items() if k.startswith('B1') and isinstance(x, dict)), None)
    note = ("`value` is the reference's plain greedy loop on one core -- the SLOWEST CPU variant.  The realistic CPU figures for this "
            "stage are B0' (the reference's own HBB-prefiltered loop, 1 core): %.0f us/img%s; the GPU stage should be read against "
            "those, not against `value`" % (dt_fast * 1e6, (" and B1 (its per-class Pool(%d)): %.0f us/img" % (b1['cores'], b1['us_per_img']))
                                              if b1 else ""))
    return dict(value=head * 1e6, unit='us/img (rotated-IoU + poly NMS stage)', cores=1, note=note,
                kind='reference' if dt_ref is not None else 'port',
                sample='1 image, %d class-offset detections (the set multiclass_rnms hands to rnms), thr %.2f; every '
                       'variant repeated for <= %.0f s of CPU work' % (M, thr, budget_s),
                host_cores=os.cpu_count(), variants=v)


def batched_nms_line(dev, images=16, boxes=2000, thr=0.4):
    """BASELINE.md section 3 / configs[3]: the rotated NMS of a 16-image batch as (image x class) segments in ONE
    orp_rnms_batched launch sequence (dense clustered scenes of `boxes` detections per image, 15 classes)."""
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_batched_device
    parts, sizes = [], []
    for im in range(images):
        d, lab = S.gen_dense_scene(boxes, 100 + im)
        for c in range(15):
            sel = d[lab == c]
            parts.append(sel); sizes.append(len(sel))
    d = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(dev)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)
    max_seg = int(max(sizes))
    for _ in range(3):
        keep, num = rnms_batched_device(d, off, max_seg, thr)
    _lib.lib().orp_profile_enable(1)
    for sl in (0, 1, 2):
        read_prof(sl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 20
    for _ in range(iters):
        keep, num = rnms_batched_device(d, off, max_seg, thr)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    k = {name: (lambda t: t[0] / t[1] * 1e3 if t[1] else None)(read_prof(sl)) for name, sl in (('mask', 0), ('sweep', 1), ('rank_prepare', 2))}
    _lib.lib().orp_profile_enable(0)
    M = int(d.shape[0])
    pairs = float(sum(n * (n - 1) / 2 for n in sizes))
    alg = 36.0 * M + 8.0 * int(num.sum().item())          # read 36 B per box, write 8 B per kept index
    return dict(images=images, segments=len(sizes), boxes=M, max_segment=max_seg, kept=int(num.sum().item()),
                us_per_batch=us, us_per_img=us / images, kernel_us=k, pairs=pairs, gpairs_per_s=pairs / us / 1e3,
                algorithmic_bytes=alg, hbm_gbs=alg / us / 1e3, hbm_frac=alg / us / 1e3 / HBM_PEAK_GBS,
                note='one launch sequence (rank+prepare, mask, sweep) for all %d (image x class) segments; per-segment '
                     'keep sets are checked against the oracle in tests/test_gpu_parity.py' % len(sizes))


def per_op_table(dev, budget_s=2.0):
    """Per-op microseconds at the configs[1] shapes next to the CPU reference port (oracle/, 1 host core) of the same op
    -- BASELINE.json: "per-op us reported next to the CPU reference".  GPU: HIP-event pairs around >= 10 launches after
    warm-up; CPU: the oracle timed once AT THE SAME SIZE (nothing is extrapolated; an op whose CPU port needs more than
    ~10 s has no CPU figure)."""
    from oracle import orp_oracle as O
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.mmdet_ops import convex_iou, deform_conv_forward_multi, minaerarect
    from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
    O.build()

    def gpu_us(fn, iters=10, warm=2):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def cpu_us(fn):
        t0 = time.perf_counter()
        fn()
        return (time.perf_counter() - t0) * 1e6

    out = {}
    # min-area-rect decode of the <= 5344 candidates of one image
    pts = S.gen_pointsets(5344, 2).astype(np.float32)
    tp = torch.from_numpy(pts).to(dev)
    out['minaerarect_5344_sets'] = dict(gpu_us=gpu_us(lambda: minaerarect(tp)), cpu_us=cpu_us(lambda: O.minarearect(pts)),
                                        cpu_sample='full size')
    # refine-stage assigner IoU: all 21824 point sets of an image x 32 gts (grid-ordered, as the head produces them)
    ar = []
    for st in (8, 16, 32, 64, 128):
        nn = IMG // st
        yy, xx = np.meshgrid(np.arange(nn), np.arange(nn), indexing='ij')
        ar.append(np.stack([xx.reshape(-1) * st + st / 2.0, yy.reshape(-1) * st + st / 2.0], 1))
    around = np.concatenate(ar)
    pall = np.ascontiguousarray(S.gen_pointsets(len(around), 6, around=around), np.float32)
    gts = S.gen_gts(32, 3).astype(np.float32)
    tpa, tg = torch.from_numpy(pall).to(dev), torch.from_numpy(gts).to(dev)
    out['convex_iou_21824x32'] = dict(gpu_us=gpu_us(lambda: convex_iou(tpa, tg), iters=5),
                                      cpu_us=cpu_us(lambda: O.convex_iou(pall, gts)), cpu_sample='full size')
    # rotated NMS of a 2000-box class-offset dense scene (fp32 reference arithmetic on both sides)
    d = S.gen_dense_scene(2000, 1)[0].astype(np.float32)
    td = torch.from_numpy(d).to(dev)
    out['rnms_2000_boxes'] = dict(gpu_us=gpu_us(lambda: rnms_device(td, 0.4)), cpu_us=cpu_us(lambda: O.rnms(d, 0.4)),
                                  cpu_sample='full size (fp32 devrIoU port)')
    # DeformConv forward: GPU = all five levels of one image in one launch; CPU = the 16x16 level, scaled by positions
    torch.manual_seed(0)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    xs = [torch.randn(1, 256, IMG // st, IMG // st, device=dev).contiguous(memory_format=torch.channels_last)
          for st in (8, 16, 32, 64, 128)]
    offs = [torch.randn(1, 18, IMG // st, IMG // st, device=dev) * 2 for st in (8, 16, 32, 64, 128)]
    # (the CPU port needs ~40 s for the whole launch: no CPU figure here, none extrapolated; the 32 x 32 level alone follows)
    out['deform_conv_21824_positions'] = dict(
        gpu_us=gpu_us(lambda: deform_conv_forward_multi(xs, offs, w, 1, 1, 1)), cpu_us=None,
        cpu_sample='not timed (tens of seconds); see deform_conv_32x32_level')
    xc, oc, wc = xs[2].contiguous().cpu().numpy(), offs[2].cpu().numpy(), w.cpu().numpy()
    out['deform_conv_32x32_level'] = dict(
        gpu_us=gpu_us(lambda: deform_conv_forward_multi(xs[2:3], offs[2:3], w, 1, 1, 1)),
        cpu_us=cpu_us(lambda: O.dcn_forward(xc, oc, wc, 1, 1, 1)),
        cpu_sample='full size: the 1 024 positions of the 32 x 32 level, 256 -> 256 channels (4.7 % of the launch above)')
    # the same layer in fp16 / bf16 (the reference's half dispatch; BASELINE configs[4]): v_mfma_f32_32x32x16, fp32 accumulate
    for dt, nm in ((torch.float16, 'fp16'), (torch.bfloat16, 'bf16')):
        hx, ho, hw = [x.to(dt) for x in xs], [o.to(dt) for o in offs], w.to(dt)
        out['deform_conv_%s_21824_positions' % nm] = dict(gpu_us=gpu_us(lambda: deform_conv_forward_multi(hx, ho, hw, 1, 1, 1)),
                                                         cpu_us=None, cpu_sample='(no CPU half path in the reference)')
    # ---- the training-path ops (configs[2] shapes: 21824 points, 64 gts, 5000 positives) --------------------------
    from orientedreppoints_amd.mmdet_ops import (ChamferDistance2D, box_iou_rotated, convex_giou, points_in_quad_aligned,
                                                 sigmoid_focal_loss)
    from orientedreppoints_amd.mmdet_ops.apaa import max_iou_assign, point_assign
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)   # noqa: E731
    P = 5000
    pp, gg = S.gen_pointsets(P, 2).astype(np.float32), S.gen_gts(P, 3).astype(np.float32)
    tpp, tgg = t32(pp), t32(gg)
    out['convex_giou_5000_pairs'] = dict(gpu_us=gpu_us(lambda: convex_giou(tpp, tgg)),
                                         cpu_us=cpu_us(lambda: O.convex_giou(pp, gg)), cpu_sample='full size')
    out['points_in_quad_5000x9'] = dict(gpu_us=gpu_us(lambda: points_in_quad_aligned(tpp, tgg)),
                                        cpu_us=cpu_us(lambda: O.points_in_quad_aligned(pp, gg)), cpu_sample='full size')
    rng = np.random.RandomState(0)
    ca, cb_ = (rng.rand(P, 40, 2) * 100).astype(np.float32), (rng.rand(P, 40, 2) * 100).astype(np.float32)
    tca, tcb = t32(ca), t32(cb_)
    out['chamfer_5000x40x40'] = dict(gpu_us=gpu_us(lambda: ChamferDistance2D(tca, tcb)),
                                     cpu_us=cpu_us(lambda: O.chamfer_forward(ca, cb_)), cpu_sample='full size')
    N = 2 * 21824
    lg = rng.randn(N, 15).astype(np.float32); lb = rng.randint(0, 16, N).astype(np.int64)
    tlg, tlb = t32(lg), torch.from_numpy(lb).to(dev)
    out['sigmoid_focal_43648x15'] = dict(gpu_us=gpu_us(lambda: sigmoid_focal_loss(tlg, tlb, 2.0, 0.25)),
                                         cpu_us=cpu_us(lambda: O.focal_forward(lg, lb, 2.0, 0.25)), cpu_sample='full size')
    pts3 = np.concatenate([np.concatenate([a_ - st / 2.0, np.full((len(a_), 1), st)], 1)
                           for a_, st in zip(ar, (8, 16, 32, 64, 128))]).astype(np.float32)
    g64 = S.gen_gts(64, 5).astype(np.float32)
    tp3, tg64 = t32(pts3), t32(g64)
    out['point_assign_21824x64'] = dict(gpu_us=gpu_us(lambda: point_assign(tp3, tg64)),
                                        cpu_us=cpu_us(lambda: O.point_assign(pts3, g64)), cpu_sample='full size')
    ovl = (rng.rand(21824, 64) * 0.3).astype(np.float32) * (rng.rand(21824, 64) < 0.02)      # [N, K], mostly zero
    tov = t32(ovl)
    out['max_iou_assign_21824x64'] = dict(gpu_us=gpu_us(lambda: max_iou_assign(tov, 0.1, 0.1)),
                                          cpu_us=cpu_us(lambda: O.max_iou_assign(ovl, 0.1, 0.1)), cpu_sample='full size')
    rb = S.gen_rboxes(1000, 1).astype(np.float32)
    trb = t32(rb)
    out['box_iou_rotated_1000x1000'] = dict(gpu_us=gpu_us(lambda: box_iou_rotated(trb, trb)),
                                            cpu_us=cpu_us(lambda: O.box_iou_rotated(rb, rb)), cpu_sample='full size')
    # ---- roofline entry per op: algorithmic bytes -> HBM GB/s and its fraction of 8 TB/s (the formal bound of
    #      these scan-type ops; at these sizes they are ALU / latency limited), pair rate where the op is pairwise, and -- when
    #      profiles/<round>_pmc.json was collected on THIS build (tools/pmc_all.sh on the same shapes) -- the counters that say
    #      what the kernel is really bound by: VALU busy, VALU issue fraction, resident waves per SIMD, HBM traffic
    P_ = 5000
    alg = {
        'minaerarect_5344_sets': (104.0 * 5344, None, 'minarearect'),
        'convex_iou_21824x32': (72.0 * 21824 + 32.0 * 32 + 4.0 * 21824 * 32, 21824.0 * 32, 'convex_iou'),
        'rnms_2000_boxes': (36.0 * 2000 + 8.0 * 2000 * 32, 2000.0 * 1999 / 2, 'nms_mask'),
        'convex_giou_5000_pairs': (180.0 * P_, float(P_), 'convex_giou'),
        'points_in_quad_5000x9': (140.0 * P_, None, 'points_in_quad_aligned'),
        'chamfer_5000x40x40': (1280.0 * P_, P_ * 1600.0, 'chamfer_nn'),
        'sigmoid_focal_43648x15': (4.0 * N * 15 * 2 + 8.0 * N, None, 'focal_fwd'),
        'point_assign_21824x64': (12.0 * 21824 + 32.0 * 64 + 8.0 * 21824, 21824.0 * 64, 'point_assign_gt'),
        'max_iou_assign_21824x64': (4.0 * 21824 * 64 + 12.0 * 21824, None, 'max_iou_assign'),
        'box_iou_rotated_1000x1000': (20.0 * 2000 + 4.0e6, 1.0e6, 'box_iou_rotated'),
    }
    pmc, _note = load_pmc()
    for name, (nbytes, pairs, key) in alg.items():
        e = out.get(name)
        if not e or not e.get('gpu_us'):
            continue
        t = e['gpu_us'] * 1e-6
        r = dict(algorithmic_bytes=nbytes, hbm_gbs=round(nbytes / t / 1e9, 2), hbm_frac=nbytes / t / 1e9 / HBM_PEAK_GBS)
        if pairs:
            r['gpairs_per_s'] = round(pairs / t / 1e9, 3)
        c = (pmc or {}).get(key)
        if c:
            for k_ in ('valu_busy_frac', 'valu_issue_frac', 'waves_per_simd', 'hbm_bytes_per_launch', 'kernel_us_at_2p4ghz'):
                if k_ in c:
                    r[k_] = c[k_]
        e['roofline'] = r
    return out


def dcn_pair_modes_us(dev, batch, img, modes):
    """The head's DeformConv pair launch (both layers, five levels of `batch` img x img images, NCHW in / out as the towers
    hand the features over) in each arithmetic mode of the library: average kernel time from the library's own HIP events."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
    L = _lib.lib()
    g = torch.Generator(device='cpu').manual_seed(11)
    sizes = [img // s_ for s_ in (8, 16, 32, 64, 128)]
    fa = [torch.randn(batch, 256, n, n, generator=g).to(dev) for n in sizes]
    fb = [torch.randn(batch, 256, n, n, generator=g).to(dev) for n in sizes]
    of = [(torch.randn(batch, 18, n, n, generator=g) * 2).to(dev) for n in sizes]
    w1, w2 = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev), (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(dev)
    out = {}
    try:
        for m in modes:
            L.orp_dcn_set_split_mode(int(m))
            for _ in range(3):
                deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
            torch.cuda.synchronize()
            L.orp_profile_enable(1)
            read_prof(3)
            for _ in range(20):
                deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
            torch.cuda.synchronize()
            ms, n = read_prof(3)
            out[int(m)] = (ms / n * 1e3) if n else None
    finally:
        L.orp_dcn_set_split_mode(-1)
        L.orp_profile_enable(0)
    return out


def train_probe(dev, model_name='r50', steps=10, warmup=4, gts=64):
    """BASELINE configs[2] at its per-GPU load inside the default run (rank 0, N = 1): `steps` SGD iterations of the R-50
    detector on 2 synthetic 1024^2 images x `gts` polygons (forward, APAA losses, backward, clip, step), wall time per
    iteration and the library's own HIP events around the hot-path kernels of the training step."""
    from orientedreppoints_amd import dist_utils as D
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.dota_configs import train_cfg as TRAIN_CFG
    L = _lib.lib()
    torch.manual_seed(0)
    model = build_detector(ConfigDict(MODELS[model_name]), train_cfg=ConfigDict(TRAIN_CFG),
                           test_cfg=ConfigDict(TEST_CFG)).to(dev).train()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=1e-4)
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=False)
    g = torch.Generator(device='cpu').manual_seed(4321)
    batch = 2
    data = dict(
        img=torch.randn(batch, 3, 1024, 1024, generator=g).to(dev),
        img_meta=[dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)] * batch,
        gt_bboxes=[torch.from_numpy(S.gen_polys(gts, 40 + i, wh=(16, 120))[:, :8].astype(np.float32)).to(dev) for i in range(batch)],
        gt_labels=[torch.randint(1, 16, (gts,), generator=g).to(dev) for _ in range(batch)])
    for _ in range(warmup):
        log_vars = D.train_step(model, opt, data, hook)
    torch.cuda.synchronize()
    slots = dict(dcn_fwd=3, dcn_bwd_all=8, dcn_bwd_input_gemm=9, dcn_bwd_scatter=10, dcn_bwd_weight=11, convex_iou=5,
                 convex_giou=6, minarearect=4, tower_fpn_conv_fwd_and_grad_input=12, tower_fpn_conv_grad_weight=13)
    L.orp_profile_enable(1)
    for sl in slots.values():
        read_prof(sl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        log_vars = D.train_step(model, opt, data, hook)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.orp_profile_enable(0)
    ev = {}
    for name, sl in slots.items():
        ms, n = read_prof(sl)
        ev[name] = dict(us_per_step=round(ms / steps * 1e3, 1), launches_per_step=round(n / steps, 2)) if n else None
    del model, opt
    torch.cuda.empty_cache()
    return {'workload': 'BASELINE configs[2] per-GPU load: train step, 2 img/GPU x %d gts, 1024x1024, R-50 FPN, SGD, f32' % gts,
            'steps': steps, 'warmup': warmup, 'ms_per_step': round(elapsed / steps * 1e3, 3),
            'images_per_s': round(batch * steps / elapsed, 2), 'loss': round(float(log_vars['loss']), 4),
            'dcn_forward_mode': int(L.orp_dcn_get_split_mode()), 'hip_events': ev,
            'note': 'the HIP-event slots above (this library\'s DeformConv, tower / FPN convolution and convex kernels) are about a '
                    'quarter of the step; roughly half of it is the library\'s forward / backward convolutions of the backbone '
                    '(stock PyTorch-ROCm by design), the rest losses, assigners, optimizer and normalisation passes'}


def quick_config(dev, model_name, batch, size, steps=10, warmup=3, depth=4, half=None):
    """A compact line for another BASELINE configuration at its per-GPU load (rank 0, N = 1, after the headline loops): the same
    measurement as the headline -- captured graphs, `depth` in flight for `value`, one at a time for `value_serial` -- with fewer
    steps; every replay's detections are compared with the eager step's.  half = torch.float16 / torch.bfloat16: the whole detector
    (weights and activations) in that type -- the DeformConvs on `orp_dcn_forward_half`, the post-processing on fp32 copies of the head's
    outputs (the reference dispatches its operators over float AND half: deform_conv_cuda_kernel.cu:259).  The library's half-precision
    solvers are not reproducible and its deterministic ones stall the device with four graphs in flight (docs/notebook/round6.md 8), so a
    half configuration is measured in the default library mode and says so in `replays_identical_to_eager`."""
    import copy
    from orientedreppoints_amd.mmdet_models import GraphedInference, PipelinedInference
    torch.manual_seed(0)
    model = build_detector(ConfigDict(MODELS[model_name]), train_cfg=None, test_cfg=ConfigDict(copy.deepcopy(TEST_CFG))).to(dev).eval()
    g = torch.Generator(device='cpu').manual_seed(4321)
    img = torch.randn(batch, 3, size, size, generator=g).to(dev)
    metas = [dict(img_shape=(size, size, 3), pad_shape=(size, size, 3), scale_factor=1.0, flip=False) for _ in range(batch)]
    calibrate_head(model, img[:1])
    if half is not None:
        model, img = model.to(half), img.to(half)
    det_flag = torch.backends.cudnn.deterministic
    if half is not None and det_flag:
        raise RuntimeError('half configuration skipped: torch.backends.cudnn.deterministic is set')
    try:
        with torch.no_grad():
            ref = model.simple_test_batch(img, metas)
            same_ = lambda ra, rb: all(a.shape == b.shape and np.array_equal(a, b) for r, q in zip(ra, rb) for a, b in zip(r, q))   # noqa: E731
            if half is None and not all(same_(ref, model.simple_test_batch(img, metas)) for _ in range(2)):
                torch.backends.cudnn.deterministic = True       # (1536^2: the library's default solvers accumulate with atomics)
                ref = model.simple_test_batch(img, metas)
        gi = GraphedInference(model, img, metas)
        for _ in range(warmup):
            r = gi(img)
        identical = same_(r, ref)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gi(img)
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t0) / steps * 1e3
        del gi
        pi = PipelinedInference(model, img, metas, depth=depth)
        got = [r for r in (pi.submit(img) for _ in range(warmup + depth)) if r is not None] + pi.flush()
        identical = identical and all(same_(g_, ref) for g_ in got)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(pi.submit(img) is not None for _ in range(steps)) + len(pi.flush())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        assert n == steps
        return dict(workload='%s FPN inference, %dx%d, bs=%d/GPU%s' % (MODEL_LABEL[model_name], size, size, batch,
                                                                        '' if half is None else ', model.to(%s)' % str(half).split('.')[-1]),
                    value=batch / ms * 1e3,
                    value_serial=batch / serial_ms * 1e3, unit='images/s', ms_per_step=ms, graph_replay_ms=serial_ms, steps=steps,
                    images_in_flight=depth, detections=int(sum(sum(len(c) for c in r) for r in ref)),
                    replays_identical_to_eager=bool(identical), library_deterministic_mode=bool(torch.backends.cudnn.deterministic))
    finally:
        torch.backends.cudnn.deterministic = det_flag


def half_config_in_a_subprocess(depth, timeout=180):
    """The fp16-model line of `other_configs`, measured by a CHILD process under a timeout: a configuration one library flag away from a
    device stall (see quick_config) must not be able to take the headline line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--half-config', '1', '--pipeline', str(depth)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=timeout,
                           env={k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')})
    except subprocess.TimeoutExpired:
        return 'failed: no result within %d s (child killed)' % timeout
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not lines:
        return 'failed: %s' % ((r.stderr or r.stdout)[-200:],)
    return json.loads(lines[-1])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    """`python bench.py --gpus N` started by hand (no RANK in the environment) launches its own N ranks, one per GPU,
    exactly as the reference's tools/dist_train.sh:9-10 does (`python -m torch.distributed.launch --nproc_per_node=N`):
    re-exec through torch.distributed.run on 127.0.0.1.  Under a launcher (RANK set) this is a no-op."""
    if args.gpus <= 1 or 'RANK' in os.environ:
        return
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    sys.exit(subprocess.call(cmd, env=env))


def init_ranks(args):
    """(rank, local_rank, world, device, dist-or-None).  The process group exists iff world > 1; `--gpus` must be the
    world size (a launcher started with a different --nproc-per-node is a usage error, not a silent 1-rank run)."""
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    cpu = args.device == 'cpu'
    if cpu:
        dev = torch.device('cpu')
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (or --device cpu --dry for the launcher test)'
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo' if cpu else 'nccl', rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    return rank, local_rank, world, dev, dist


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize()


def timed_steps(step, args, dev, dist, world):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize; MAX over ranks (seconds)."""
    for _ in range(args.warmup):
        step()
    _sync(dev)
    if dist is not None:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _sync(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main_dry(args):
    """`--dry` (with --device cpu: the gloo launcher test that runs without a GPU): the rank / barrier / max-over-ranks
    / JSON plumbing of the benchmark around a stand-in step that does no hot-path work.  Never a measurement."""
    rank, local_rank, world, dev, dist = init_ranks(args)
    x = torch.randn(64, 64, device=dev)

    def step():
        return (x @ x).sum()
    if args.mode == 'train':
        # the exchange of `--mode train` (dist_utils.train_step + the overlapped bucketed all-reduce) around a stand-in
        # model whose second head only rank 0 uses: the ranks' autograd graphs differ, as with images without positives
        from orientedreppoints_amd import dist_utils as D

        class Standin(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.body = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1))
                self.a, self.b = torch.nn.Conv2d(8, 2, 1), torch.nn.Conv2d(8, 2, 1)

            def forward(self, img, use_b):
                f = self.body(img)
                return {'loss_cls': self.a(f).pow(2).mean() + (self.b(f).abs().mean() if use_b else 0.0)}
        torch.manual_seed(0)
        net = Standin().to(dev)
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)
        hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, bucket_size_mb=0.001)
        data = dict(img=torch.randn(2, 3, 16, 16, device=dev), use_b=(rank == 0))

        def step():                                           # noqa: F811
            return D.train_step(net, opt, data, hook)
    elapsed = timed_steps(step, args, dev, dist, world)
    if args.mode == 'train' and dist is not None:             # every rank holds the same parameters after the steps
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(flat, ref), 'ranks diverged in the dry training exchange'
    n = world
    if dist is not None:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)                                # every rank took part
        n = int(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'DRY RUN (launcher plumbing only, no hot-path work)', 'value': args.batch * args.steps * world / elapsed,
                          'unit': 'steps/s', 'n_gpus': n, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': 'dry', 'mode': args.mode, 'device': args.device,
                                     'parallelism': 'replicas x%d' % world}}))
    if dist is not None:
        dist.destroy_process_group()


def main_train(args):
    """`--mode train`: BASELINE configs[2] -- one SGD iteration (forward, APAA losses, backward with the bucketed
    gradient all-reduce over RCCL overlapped for N > 1, optimizer step) on 2 synthetic 1024x1024 images with `--gts` polygons each per GPU.
    Not the headline metric (that is inference images/sec); same JSON contract, weak scaling."""
    from orientedreppoints_amd import dist_utils as D
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.dota_configs import train_cfg as TRAIN_CFG

    rank, local_rank, world, dev, dist = init_ranks(args)
    _lib.lib()
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)
    batch = args.batch if args.batch > 1 else 2          # configs[2]: imgs_per_gpu = 2

    torch.manual_seed(0)                                  # identical initial weights on every rank
    model = build_detector(ConfigDict(MODELS[args.model]), train_cfg=ConfigDict(TRAIN_CFG),
                           test_cfg=ConfigDict(TEST_CFG)).to(dev).train()
    if args.model == 'swin_t':
        # the Swin config's optimizer: AdamW, weight decay 0.05 except norms / position-bias tables (paramwise_cfg)
        oc = swin_t_optimizer
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        plain = [p for n, p in named if not any(k in n for k in oc['no_decay_keys'])]
        nodecay = [p for n, p in named if any(k in n for k in oc['no_decay_keys'])]
        opt = torch.optim.AdamW([dict(params=plain), dict(params=nodecay, weight_decay=0.0)], lr=oc['lr'],
                                betas=oc['betas'], weight_decay=oc['weight_decay'])
    else:
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9,
                              weight_decay=1e-4)
    # N > 1: bucketed all-reduce overlapped with backward (32 MB buckets over RCCL / xGMI)
    amp_dtype = dict(f32=None, fp16=torch.float16, bf16=torch.bfloat16)[args.dtype]
    scaler = torch.amp.GradScaler('cuda') if args.dtype == 'fp16' else None
    hook = D.DistOptimizerHook(grad_clip=dict(max_norm=35, norm_type=2), overlap=True, scaler=scaler)
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    sizes = TRAIN_SIZES or [IMG]                            # multi-scale: the steps cycle through the patch sizes
    datas = [dict(
        img=torch.randn(batch, 3, sz, sz, generator=g).to(dev),
        img_meta=[dict(img_shape=(sz, sz, 3), pad_shape=(sz, sz, 3), scale_factor=1.0, flip=False)] * batch,
        gt_bboxes=[torch.from_numpy(S.gen_polys(args.gts, 40 + i + 7 * rank, wh=(16, 120))[:, :8]
                                    .astype(np.float32) * (sz / 1024.0)).to(dev) for i in range(batch)],
        gt_labels=[torch.randint(1, 16, (args.gts,), generator=g).to(dev) for _ in range(batch)]) for sz in sizes]
    backbone_graphed = False
    if args.graph_backbone and amp_dtype is None and len(sizes) == 1 and args.model in ('r50', 'r101'):
        backbone_graphed = D.graph_backbone(model, datas[0]['img'])
    count = [0]

    def step():
        # parse_losses all-reduces + .item()s the logged scalars every iteration as the reference's batch_processor does
        data = datas[count[0] % len(datas)]
        count[0] += 1
        return D.train_step(model, opt, data, hook, autocast_dtype=amp_dtype)

    for _ in range(args.warmup):
        log_vars = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    _lib.lib().orp_profile_enable(1)
    for s in range(16):
        read_prof(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        log_vars = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _lib.lib().orp_profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = {name: read_prof(slot) for name, slot in (('dcn_fwd', 3), ('minarearect', 4))}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    out = {
        'metric': 'training images/sec (OrientedRepPoints %s FPN, %s DOTA patches, APAA on, %s step)'
                  % (args.model, '/'.join('%dx%d' % (z, z) for z in sizes), 'AdamW' if args.model == 'swin_t' else 'SGD'),
        'value': round(batch * args.steps * world / elapsed, 3),
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': {'f32': 'f32', 'fp16': 'fp16 autocast (hot-path operators f32, GradScaler)',
                  'bf16': 'bf16 autocast (hot-path operators f32)'}[args.dtype],
        'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[%s]: train step, %d img/GPU x %d gts, %s, 15 classes'
                               % ('4' if args.model == 'swin_t' else '2', batch, args.gts,
                                  ' / '.join('%dx%d' % (z, z) for z in sizes)),
                   'model': args.model, 'patch_sizes': sizes,
                   'imgs_per_gpu': batch, 'gts_per_image': args.gts,
                   'parallelism': 'dp%d (image-parallel, bucketed gradient all-reduce overlapped with backward)' % world},
        'loss': round(float(log_vars['loss']), 4),
        'backbone_graphed': backbone_graphed,
        'hip_events_ms_per_step': {k: (round(v[0] / args.steps, 3) if v[1] else None) for k, v in prof.items()},
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', choices=sorted(MODELS), default='r50',
                    help='r50 (default, BASELINE configs[1]); r101 with --batch 2 is the per-GPU load of configs[3]')
    ap.add_argument('--size', type=str, default='1024',
                    help='square patch size: 1024 (configs[1]) or 1536 (configs[4] shapes); --mode train accepts a list '
                         '"1024,1536" = multi-scale training, the steps cycle through the sizes')
    ap.add_argument('--device', choices=('cuda', 'cpu'), default='cuda', help='cpu only together with --dry (gloo)')
    ap.add_argument('--dry', action='store_true',
                    help='launcher / rank / timing plumbing only, stand-in step (the CPU test of the N > 1 path)')
    ap.add_argument('--batch', type=int, default=1, help='images per GPU per step (config 1: 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--half-config', type=int, default=0, help=argparse.SUPPRESS)     # internal: the fp16-model line, run by a child process
    ap.add_argument('--train-probe', type=int, default=1,
                    help='1 (default): the line also carries `train` = 10 SGD iterations of BASELINE configs[2] at its per-GPU load '
                         '(rank 0, N = 1, after the timed inference loops)')
    ap.add_argument('--pipeline', type=int, default=4,
                    help='captured graphs in flight in the throughput measurement (PipelinedInference); 1 = off')
    ap.add_argument('--graph', type=int, default=1,
                    help='1 (default): after the eager loop, time the same K steps as ONE hipGraph replay each (device '
                         'part captured once) and report that as `value`; 0: eager only')
    ap.add_argument('--cudnn-benchmark', type=int, default=0,
                    help='torch.backends.cudnn.benchmark (MIOpen find mode), the reference\'s cfg.cudnn_benchmark '
                         '(tools/test.py:108-110)')
    ap.add_argument('--mode', choices=('test', 'train'), default='test',
                    help="test (default): the headline inference step; train: one SGD iteration of BASELINE configs[2] "
                         "(2 img/GPU, APAA on), gradients all-reduced over RCCL for N > 1")
    ap.add_argument('--gts', type=int, default=64, help='--mode train: ground-truth polygons per image')
    ap.add_argument('--graph-backbone', type=int, default=0,
                    help='--mode train, f32, one patch size: 1 = the backbone forward + backward replayed as two hipGraphs '
                         '(dist_utils.graph_backbone).  Measured round 6: 31.2 ms per step against 29.7 eager -- the step is not '
                         'bound by the launch rate of the backbone -- so the default stays 0')
    ap.add_argument('--dtype', choices=('f32', 'fp16', 'bf16'), default='f32',
                    help='--mode train only: f32 (default, the reference\'s arithmetic) or torch.autocast in fp16 (with a '
                         'GradScaler) / bf16: library convolutions in half, hot-path operators on fp32-cast inputs')
    args = ap.parse_args()
    global IMG, TRAIN_SIZES
    size_list = [int(z) for z in str(args.size).split(',')]
    IMG = args.size = size_list[0]
    if len(size_list) > 1:
        if args.mode != 'train':
            raise SystemExit('bench.py: a list of sizes is only valid with --mode train')
        TRAIN_SIZES = size_list
    if args.device == 'cpu' and not args.dry:
        raise SystemExit('bench.py: the hot path has no CPU fallback; --device cpu is only valid with --dry')
    if args.half_config:                                  # child of half_config_in_a_subprocess: one compact line, nothing else
        assert torch.cuda.is_available()
        _lib.lib()
        print(json.dumps(quick_config(torch.device('cuda', 0), 'r50', 1, 1024, depth=max(args.pipeline, 1), half=torch.float16)))
        return
    maybe_spawn(args)                                     # --gpus N by hand -> N ranks (no-op under a launcher)
    if args.dry:
        return main_dry(args)
    if args.mode == 'train':
        return main_train(args)

    rank, local_rank, world, dev, dist = init_ranks(args)
    distributed = world > 1
    _lib.lib()     # fail loudly if the HIP library is missing
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)

    torch.manual_seed(0)
    model = build_detector(ConfigDict(MODELS[args.model]), train_cfg=None, test_cfg=ConfigDict(TEST_CFG)).to(dev).eval()
    g = torch.Generator(device='cpu').manual_seed(1234 + rank)
    img = torch.randn(args.batch, 3, IMG, IMG, generator=g).to(dev)
    metas = [dict(img_shape=(IMG, IMG, 3), pad_shape=(IMG, IMG, 3), scale_factor=1.0, flip=False)
             for _ in range(args.batch)]
    calibrate_head(model, img[:1])

    def step():
        with torch.no_grad():
            return model.simple_test_batch(img, metas)

    # the eager loop (kernel timings, roofline) runs the launch sequence of the throughput mode's graphs: one stream, no
    # tower fork -- the fork belongs to the one-image-at-a-time replay below
    model.bbox_head.tower_streams = False
    for _ in range(max(1, args.warmup)):                    # (--warmup 0: one untimed step still, the reference result of the checks below)
        res = step()
    ndet = int(sum(sum(len(c) for c in r) for r in res))
    # the step is meant to be bitwise reproducible (fixed-order sums in every HIP kernel; the library convolutions the
    # detector keeps are deterministic for its shapes): checked, reported, and the replay checks below then compare
    # detection COUNTS with a 0.5 % allowance so that a library kernel that is not cannot cost the throughput figure
    def same(ra, rb):
        return all(a.shape == b.shape and np.array_equal(a, b) for r, q in zip(ra, rb) for a, b in zip(r, q))

    step_reproducible = all(same(res, step()) for _ in range(3))
    library_deterministic_mode = False
    if not step_reproducible:
        # some library convolution of this shape accumulates with atomics (1536^2: the stride-2 / 48^2 3x3 convolutions of
        # layer2 / layer4 get a split-K solver, tests/checks/determinism_modules.py): ask MIOpen for reproducible solvers
        # only (torch.backends.cudnn.deterministic -> MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC) and measure THAT
        torch.backends.cudnn.deterministic = True
        library_deterministic_mode = True
        for _ in range(max(3, args.warmup)):
            res = step()
        ndet = int(sum(sum(len(c) for c in r) for r in res))
        step_reproducible = all(same(res, step()) for _ in range(3))
    count_slack = 0 if step_reproducible else max(2, ndet // 200)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    _lib.lib().orp_profile_enable(1)
    for s in range(16):
        read_prof(s)       # reset
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _lib.lib().orp_profile_enable(0)
    model.bbox_head.tower_streams = None                    # default (fork on) for the single-graph replay
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same K steps as ONE hipGraph replay each (mmdet_models/graph_inference.py): same kernels, same results, the
    # ~240 launches per step leave the host.  This is the deployment path and the reported `value` when capture works;
    # the eager loop above stays in the JSON (`eager_ms_per_step`) and provides the live HIP-event kernel timings.
    graph_ms = None
    pipe_ms = None
    replays_identical = None
    eager_elapsed = elapsed
    if args.graph:
        ok = 1
        try:
            from orientedreppoints_amd.mmdet_models import GraphedInference
            gi = GraphedInference(model, img, metas)
            for _ in range(max(1, args.warmup)):
                gres = gi(img)
            ngraph = int(sum(sum(len(c) for c in r) for r in gres))
            if abs(ngraph - ndet) > count_slack:
                raise RuntimeError('graph replay returned %d detections, eager %d' % (ngraph, ndet))
        except Exception as e:   # noqa: BLE001  (report, keep the eager measurement)
            ok = 0
            graph_ms = 'failed: %s' % (str(e)[:200],)
        if distributed:                                     # every rank replays, or none does
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            tg = time.perf_counter()
            for _ in range(args.steps):
                gi(img)
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            g_elapsed = time.perf_counter() - tg
            if distributed:
                t = torch.tensor([g_elapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                g_elapsed = float(t.item())
            graph_ms = g_elapsed / args.steps * 1e3
            elapsed = g_elapsed
        # ---- throughput mode: two captured graphs in flight (the tail of image i overlaps the backbone of image i + 1);
        # every image runs the complete step, results are collected one submit later and all of them inside the bracket
        if ok and args.pipeline > 1:
            pok = 1
            try:
                from orientedreppoints_amd.mmdet_models import PipelinedInference
                del gi
                pi = PipelinedInference(model, img, metas, depth=args.pipeline)
                got = [r for r in (pi.submit(img) for _ in range(args.warmup + args.pipeline)) if r is not None] + pi.flush()
                if any(abs(int(sum(sum(len(c) for c in r) for r in g)) - ndet) > count_slack for g in got):
                    raise RuntimeError('pipelined replay returned a different detection count')
                replays_identical = bool(step_reproducible and all(same(g, res) for g in got))
            except Exception as e:   # noqa: BLE001
                pok = 0
                pipe_ms = 'failed: %s' % (str(e)[:200],)
            if distributed:
                flag = torch.tensor([pok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                pok = int(flag.item())
            if pok:
                torch.cuda.synchronize()
                if distributed:
                    dist.barrier()
                tp = time.perf_counter()
                n_res = 0
                for _ in range(args.steps):
                    n_res += pi.submit(img) is not None
                n_res += len(pi.flush())
                torch.cuda.synchronize()
                if distributed:
                    dist.barrier()
                p_elapsed = time.perf_counter() - tp
                assert n_res == args.steps
                if distributed:
                    t = torch.tensor([p_elapsed], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    p_elapsed = float(t.item())
                pipe_ms = p_elapsed / args.steps * 1e3
                elapsed = p_elapsed

    prof = {name: read_prof(slot) for name, slot in
            (('nms_mask', 0), ('nms_sweep', 1), ('nms_rank_prepare', 2), ('dcn_fwd', 3), ('minarearect', 4), ('conv_split', 12))}
    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    total_imgs = args.batch * args.steps * world
    value = total_imgs / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    # one image at a time per GPU (a replay is submitted only after the previous result has been read): comparable with a
    # latency-style baseline and with round 1's `value`; `value` itself is the throughput mode when that ran
    value_serial = (total_imgs / (graph_ms * 1e-3 * args.steps)) if isinstance(graph_ms, float) else \
        total_imgs / eager_elapsed

    # ---- stage timings on rank 0 (outside the timed region): rotated-IoU+NMS us/img -------------------------
    cap = nms_inputs_of_one_image(model, img[:1], metas[:1])
    dets = cap.get('dets')
    M = int(dets.shape[0]) if dets is not None else 0
    nms_us = None
    nms_graph_us = None
    if M > 0:
        from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
        for _ in range(3):
            rnms_device(dets, cap['thr'])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rnms_device(dets, cap['thr'])
        e1.record()
        torch.cuda.synchronize()
        nms_us = e0.elapsed_time(e1) / 20 * 1e3
        # the same call as the deployment runs it: inside a captured graph (no launch gaps between its three kernels)
        nms_graph_us = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rnms_device(dets, cap['thr'])
            torch.cuda.current_stream().wait_stream(side)
            g_nms = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_nms):
                for _ in range(10):
                    rnms_device(dets, cap['thr'])
            g_nms.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(4):
                g_nms.replay()
            e1.record()
            torch.cuda.synchronize()
            nms_graph_us = e0.elapsed_time(e1) / 40 * 1e3
            del g_nms
        except Exception as ex:   # noqa: BLE001
            nms_graph_us = 'failed: %s' % (str(ex)[:120],)

    # ---- roofline of the dominant hot-path kernel: the DeformConv implicit GEMM (exact-fp32 MFMA) -------------------
    # ONE launch per image runs both DeformConvs of the head (cls + refine share their offsets: orp_dcn_forward_pair) over
    # all FPN levels.  algorithmic flops per launch = layers * 2 * positions * Cout * Cin * taps;
    # algorithmic bytes = 4 * (layers * (x + packed weights + out) + offsets).  `traffic` = HBM bytes per launch from the
    # rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), null
    # when no PMC pass of this round's kernel is committed.
    dcn_ms, dcn_n = prof['dcn_fwd']
    roof = None
    npos = args.batch * sum((IMG // s_) ** 2 for s_ in (8, 16, 32, 64, 128))
    cin = cout = 256
    layers = 2
    if dcn_n > 0:
        avg_s = dcn_ms / dcn_n * 1e-3
        flops = layers * 2.0 * npos * cout * cin * 9
        alg_bytes = 4.0 * (layers * (npos * cin + 9 * cin * cout + npos * cout) + npos * 18)
        mode = int(_lib.lib().orp_dcn_get_split_mode())          # 0 = exact-fp32 MFMA, 6 / 9 = products of three bf16 pieces, 3 = two fp16 pieces
        pmc, pmc_note = load_pmc()
        d = pmc.get('dcn_fwd_split' if mode else 'dcn_fwd_pair', {}) if pmc else {}
        traffic = d.get('hbm_bytes_per_launch') if d and args.batch == d.get('batch') and IMG == d.get('img', 1024) else None
        tiles = sum((args.batch * (IMG // s_) ** 2 + 95) // 96 for s_ in (8, 16, 32, 64, 128))
        ksplit = os.environ.get('ORP_DCN_KSPLIT') != '0' and tiles > 256 and tiles * 9 * 105 <= ((tiles + 255) // 256) * 18 * 128 * 100   # the library's rule (csrc/orp_dcn.hip)
        # the SAME pair launch in the other arithmetic mode, on the same shapes (random feature maps), HIP events inside the
        # library around the kernel: so that both the figure of the instruction actually issued and the exact-fp32 figure
        # are in the line whichever mode the timed loops ran in
        modes_us = dcn_pair_modes_us(dev, args.batch, IMG, (0, 3, 6, 9))
        exact_us = modes_us.get(0)
        exact = dict(kernel='dcn_fwd_mfma2_kernel<3, nchw, 2 layers, tap-granular split>' if ksplit
                     else 'dcn_fwd_mfma2_kernel<3, nchw, 2 layers>', instruction='v_mfma_f32_32x32x2_f32',
                     avg_launch_us=exact_us, achieved=(flops / (exact_us * 1e-6) / 1e12) if exact_us else None,
                     peak=FP32_MFMA_PEAK_TFLOPS, frac=(flops / (exact_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS) if exact_us else None)
        common = dict(unit='TFLOP/s', traffic=traffic, traffic_source=pmc_note, avg_launch_us=avg_s * 1e6, launches=dcn_n,
                      layers_per_launch=layers, positions_per_launch=npos, algorithmic_flops_per_launch=flops,
                      algorithmic_bytes_per_launch=alg_bytes, algorithmic_tflops=flops / avg_s / 1e12,
                      pair_launch_us_by_mode={str(k): v for k, v in modes_us.items()}, arithmetic_mode=mode)
        if mode == 0:
            roof = dict(kernel=exact['kernel'], bound='mfma', achieved=flops / avg_s / 1e12, peak=FP32_MFMA_PEAK_TFLOPS,
                        frac=flops / avg_s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        note='exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense = the fp32 vector peak; a register-'
                             'operand microbenchmark of the instruction sustains 146-156 TFLOP/s on this part: '
                             'tests/checks/mfma_rate.hip); one launch per image = cls + refine DeformConv over all levels', **common)
        else:
            issued = mode * flops / avg_s / 1e12
            f16 = mode == 3
            by_mode = {str(k): (k * flops / (v * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS) for k, v in modes_us.items() if k and v}
            roof = dict(kernel='dcn_fwd_split_kernel<MT 3, %d products, nchw, 2 layers as grid halves>' % mode, bound='mfma',
                        achieved=issued, peak=BF16_MFMA_PEAK_TFLOPS, frac=issued / BF16_MFMA_PEAK_TFLOPS,
                        instruction='v_mfma_f32_32x32x16_f16' if f16 else 'v_mfma_f32_32x32x16_bf16',
                        products_per_fp32_multiply=mode, frac_by_mode_back_to_back=by_mode,
                        frac_of_fp32_mfma_peak_equivalent=flops / avg_s / 1e12 / FP32_MFMA_PEAK_TFLOPS, exact_fp32=exact,
                        note=('fp32 tensors, fp32 accumulation; every fp32 operand carried as TWO fp16 pieces (11 + 11 significant bits, '
                              '|v - (hi + lo)| <= 2^-22 |v|, after an exact power-of-two range scaling from max |x| of the launch\'s '
                              'inputs), the product formed from hi*hi + hi*lo + lo*hi on the 16-bit matrix pipe, each exact in the fp32 '
                              'accumulator (csrc/orp_dcn_split.hip; error vs the fp64 oracle below the exact-fp32 kernel\'s own, '
                              'tests/test_gpu_dcn_split.py). `achieved` counts the MFMA flops actually issued (3 x the algorithmic fp32 '
                              'flops) against the 2.5 PFLOP/s dense peak: with half the matrix work of the six-product mode (three '
                              'bf16 pieces, operands exact; `frac_by_mode_back_to_back`) the launch is no longer bound by the matrix '
                              'pipe alone -- 55 us of it are outside the K loop (range pre-pass, coefficient tables, 45 MB of output '
                              'stores) -- so `frac` is LOWER than in that mode while the launch is faster; `algorithmic_tflops` / '
                              '`frac_of_fp32_mfma_peak_equivalent` are the fp32-equivalent rate (the exact-fp32 MFMA kernel, kept as '
                              '`exact_fp32`, cannot exceed 157.3 TFLOP/s)') if f16 else
                             ('fp32 tensors, fp32 accumulation; every fp32 operand split EXACTLY into three bf16 pieces and the '
                              'product formed from %d partial products on the bf16 matrix pipe (csrc/orp_dcn_split.hip). '
                              '`achieved` counts the bf16 MFMA flops actually issued (%d x the algorithmic fp32 flops) against '
                              'the 2.5 PFLOP/s dense bf16 peak; `algorithmic_tflops` is the fp32-equivalent rate (the exact-'
                              'fp32 MFMA kernel, kept as `exact_fp32`, cannot exceed 157.3). The kernel runs at the 1.4 kW '
                              'socket power cap (2.14 GHz instead of 2.4; a register-operand microbenchmark of the instruction '
                              'sustains 1.7-2.0 PFLOP/s there: tests/checks/mfma_rate_bf16.hip, clock_under_split.sh)'
                              % (mode, mode)), **common)
    # the head's tower / FPN output convolutions on the same kernel (PLAIN instantiation, csrc/orp_conv_split.hip): per image
    # 3 pair launches (both towers' layer k) + the init branch's convolution over all five levels + the FPN's three output
    # convolutions in one launch; HIP events inside the library, summed over the step's launches
    cs_ms, cs_n = prof['conv_split']
    if roof is not None and cs_n > 0:
        mode = int(_lib.lib().orp_dcn_get_split_mode()) or 6
        npos3 = args.batch * sum((IMG // s_) ** 2 for s_ in (8, 16, 32))
        layers_all, layers_fpn = 7, 1                      # 256 -> 256 3x3 layers over all five levels / over the first three
        cflops = 2.0 * cout * cin * 9 * (layers_all * npos + layers_fpn * npos3)
        per_step = cs_n / float(dcn_n)                     # one DeformConv pair launch per step
        us_step = cs_ms * 1e3 / dcn_n
        roof['tower_and_fpn_convolutions'] = dict(
            kernel='dcn_fwd_split_kernel<MT 3, %d products, PLAIN (no offsets)>' % mode, launches_per_step=per_step,
            instruction='v_mfma_f32_32x32x16_f16' if mode == 3 else 'v_mfma_f32_32x32x16_bf16',
            us_per_step=us_step, algorithmic_flops_per_step=cflops, algorithmic_tflops=cflops / (us_step * 1e-6) / 1e12,
            achieved=mode * cflops / (us_step * 1e-6) / 1e12, peak=BF16_MFMA_PEAK_TFLOPS,
            frac=mode * cflops / (us_step * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            note='the same bf16-split kernel without offsets: the head\'s seven 256->256 3x3 tower convolutions (the two towers\' '
                 'layer k as grid halves of one launch) and the FPN\'s three output convolutions (a layer per level), channels-'
                 'last; the library (Winograd + small-level kernel) took 246 us per layer, ORP_TOWER_SPLIT=0 / ORP_FPN_SPLIT=0 '
                 'switch back')
    if roof is not None:
        tw = roof.get('tower_and_fpn_convolutions')
        roof['dominant_by_time'] = ('tower_and_fpn_convolutions: %.0f us per step in %.0f launches (frac %.3f); the DeformConv pair launch '
                                    'of the top-level fields: %.0f us' % (tw['us_per_step'], tw['launches_per_step'], tw['frac'],
                                                                         roof['avg_launch_us'])) if tw else 'the DeformConv pair launch'
    # ---- the rotated-IoU + NMS stage (HBM is the formal bound, the work is fp32 VALU) --------------------------------
    mask_ms, mask_n = prof['nms_mask']
    nms = None
    if mask_n > 0 and M > 0:
        avg_s = mask_ms / mask_n * 1e-3
        cb = (M + 63) // 64
        alg_bytes = 36.0 * M + 8.0 * M * cb       # SURVEY 8d, mask formulation
        pairs = M * (M - 1) / 2.0
        nms = dict(stage_us_per_img=nms_us, stage_us_per_img_in_graph=nms_graph_us, boxes=M, mask_kernel_us=avg_s * 1e6,
                   sweep_kernel_us=(prof['nms_sweep'][0] / prof['nms_sweep'][1] * 1e3) if prof['nms_sweep'][1] else None,
                   algorithmic_bytes_per_launch=alg_bytes, hbm_gbs=alg_bytes / avg_s / 1e9,
                   hbm_frac=alg_bytes / avg_s / 1e9 / HBM_PEAK_GBS, pairs_per_launch=pairs,
                   gpairs_per_s=pairs / avg_s / 1e9,
                   note='bit-exact fp32 triangle-fan IoU: ALU-bound, 0.6 MB of traffic per launch')
        # counters of HEAD's mask kernel from the committed rocprofv3 --pmc passes (2 000-box dense scene of the same shape)
        try:
            pmc, pmc_note = load_pmc()
            pm = pmc.get('nms_mask', {})
            c = pm.get('counters', {})
            simd_cycles = c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0          # GRBM is summed over the 8 XCDs; 1024 SIMDs
            # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md, "SQ PMC units")
            nms.update(traffic=pm.get('hbm_bytes_per_launch'), valu_busy_frac=4.0 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles,
                       waves_per_simd=4.0 * c['SQ_WAVE_CYCLES'] / simd_cycles,
                       lds_bank_conflict_frac=(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'])
                       if c.get('SQ_LDS_IDX_ACTIVE') else None, traffic_source=pmc_note,
                       # this kernel's real roofline: a VALU instruction of a 64-wide wave occupies its SIMD for 4 cycles, so
                       # 4 * SQ_INSTS_VALU / SIMD-cycles of the launch is the fraction of the chip's VALU issue slots it used
                       valu_issue_frac=(4.0 * c['SQ_INSTS_VALU'] / simd_cycles) if c.get('SQ_INSTS_VALU') else None,
                       valu_insts_per_launch=c.get('SQ_INSTS_VALU'))
        except Exception:   # noqa: BLE001
            pass
    per_op = None
    if not args.no_cpu_baseline:
        try:
            per_op = per_op_table(dev)
        except Exception as e:   # noqa: BLE001  (a secondary table must not take the headline line down)
            per_op = 'failed: %s' % (str(e)[:200],)
    cpu = None
    if not args.no_cpu_baseline and M > 0:
        labels = cap.get('labels')
        cpu = cpu_baselines(dets.cpu().numpy(), labels.cpu().numpy() if labels is not None else None, cap['thr'])
        cpu['gpu_stage_us_per_img'] = nms_us
    batched = None
    if not args.no_cpu_baseline:
        try:
            batched = batched_nms_line(dev)
        except Exception as e:   # noqa: BLE001
            batched = 'failed: %s' % (str(e)[:200],)

    other = None
    if not args.no_cpu_baseline and world == 1 and (args.model, IMG, args.batch) == ('r50', 1024, 1):
        # BASELINE configs[3] at its per-GPU load (R-101, two images per GPU) and the 1536^2 shapes of configs[4], compact
        other = {}
        for key, (mname, b_, sz) in (('configs[3] per-GPU load', ('r101', 2, 1024)), ('1536^2 patches', ('r50', 1, 1536))):
            try:
                other[key] = quick_config(dev, mname, b_, sz, depth=max(args.pipeline, 1))
            except Exception as e:   # noqa: BLE001
                other[key] = 'failed: %s' % (str(e)[:200],)
    train = None
    if not args.no_cpu_baseline and world == 1 and args.train_probe:
        try:
            train = train_probe(dev)
        except Exception as e:   # noqa: BLE001
            train = 'failed: %s' % (str(e)[:200],)
    if other is not None:
        # last, and in a child process under a timeout (every measurement of this line is done by now): the whole detector in fp16
        torch.cuda.synchronize()
        other['fp16 model'] = half_config_in_a_subprocess(max(args.pipeline, 1))
    out = {
        'metric': 'images/sec (%dx%d DOTA, %s FPN)' % (IMG, IMG, MODEL_LABEL[args.model]), 'value': value,
        'value_serial': value_serial, 'unit': 'images/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: OrientedRepPoints %s FPN inference, %dx%d patch, 15 classes, '
                               'bs=%d/GPU, nms_pre=2000, score_thr=0.05, rnms iou_thr=0.4, max_per_img=2000; '
                               'random-init weights, head biases calibrated to ~%d dets/img'
                               % ('configs[1]' if (args.model, IMG, args.batch) == ('r50', 1024, 1) else
                                  ('configs[3] per-GPU load' if (args.model, args.batch) == ('r101', 2) else 'variant'),
                                  MODEL_LABEL[args.model], IMG, IMG, args.batch, TARGET_DETS),
                   'global_batch': args.batch * world, 'parallelism': 'replicas x%d (image-parallel, no collective)' % world,
                   'images_in_flight_per_gpu': args.pipeline if isinstance(pipe_ms, float) else 1},
        'detections_per_step': ndet, 'step_bitwise_reproducible': bool(step_reproducible), 'library_deterministic_mode': library_deterministic_mode, 'nms_classes_present': int(len(set(cap['labels'].tolist()))) if cap.get('labels') is not None else None,
        'library_build': _lib.lib().orp_version().decode(),
        # (a library swapped in through the environment -- dev aid -- must show in the line)
        'library_path': _lib.LIB_PATH if os.environ.get('ORP_HIP_LIB') else 'in-tree (orientedreppoints_amd/csrc/liborp_hip.so)',
        'rotated_iou_nms_us_per_img': nms_us, 'nms_boxes': M,
        'kernel_us': {k: (v[0] / v[1] * 1e3 if v[1] else None) for k, v in prof.items()},
        'mode': ('hipgraph replay, %d images in flight' % args.pipeline) if isinstance(pipe_ms, float) else
                'hipgraph replay' if isinstance(graph_ms, float) else 'eager',
        'eager_ms_per_step': round(eager_elapsed / args.steps * 1e3, 4),
        'graph_replay_ms': graph_ms,
        'pipelined_ms_per_step': pipe_ms,
        'replays_identical_to_eager': replays_identical,
        'arithmetic': {0: 'exact fp32 MFMA', 3: 'fp32 as two fp16 pieces, 3 products (default)', 6: 'fp32 as three bf16 pieces, 6 products',
                       9: 'fp32 as three bf16 pieces, 9 products'}.get(int(_lib.lib().orp_dcn_get_split_mode())),
        'roofline': roof, 'nms': nms, 'nms_batched_16_images': batched, 'per_op_us': per_op, 'train': train,
        'other_configs': other,
        'cpu_baseline': cpu,
    }
    print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
