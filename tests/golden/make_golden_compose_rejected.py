#!/usr/bin/env python3
"""Goldens for the post-processing scenes tests/golden/make_golden_compose.py REJECTED.

make_golden_compose.py draws a scene, re-runs it with its float inputs perturbed by a few ulps and only accepts it when every
discrete outcome is unchanged -- so the asserting end-to-end tests demand "same detections, same labels, same order" on scenes
that survive rounding noise.  The round-5 verdict (weak 2) asks what happens on the scenes that filter threw away.  This script
runs the reference's own get_bboxes / multiclass_rnms / rbbox2result (same loader as make_golden_compose.py, nothing copied) on
every rejected draw -- unperturbed -- and stores detections and labels; tests/test_gpu_compose.py runs the HIP path on the same
draws and REPORTS (does not assert) how many scenes / detections differ.

    python tests/golden/make_golden_compose_rejected.py        ->  tests/golden/compose_rejected_py.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_compose as M  # noqa: E402
import compose_inputs as CI  # noqa: E402


def main():
    acc = np.load(os.path.join(HERE, 'compose_py.npz'))
    R = M.load_reference()
    h = M.make_head(R)
    g = {}
    for name, (size, seed, kw, max_per_img) in M.PP_SCENES.items():
        accepted = int(acc['pp_%s_seed' % name])
        seeds = list(range(seed, accepted, 100))
        g['pp_%s_seeds' % name] = np.array(seeds, dtype=np.int64)
        for s in seeds:
            cls, pts = CI.postprocess_scene(size, s, **kw)
            M._PERTURB['on'] = False
            dets, labels, _ = M.run_postprocess(R, h, cls, pts, max_per_img, size)
            g['pp_%s_%d_dets' % (name, s)] = dets.astype(np.float32)
            g['pp_%s_%d_labels' % (name, s)] = labels.astype(np.int16)
            print('rejected scene %-9s seed %4d: %d detections' % (name, s, dets.shape[0]))
    out = os.path.join(HERE, 'compose_rejected_py.npz')
    np.savez_compressed(out, **g)
    print('written', out, '%.1f KB' % (os.path.getsize(out) / 1024))


if __name__ == '__main__':
    main()
