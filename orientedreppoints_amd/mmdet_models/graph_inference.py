"""Inference of a detector as ONE hipGraph replay per call (MI355X deployment path; no counterpart in the reference,
whose test loop launches ~2000 framework kernels per image from Python).

The whole device part of `simple_test_batch` -- backbone, FPN, dense head, decode, multiclass rotated NMS, packing --
is fixed-shape and free of host synchronisation (`get_bboxes(..., static=True)`, mmdet_models/core.py), so it is
captured once for a given input shape and replayed: ~240 kernel launches become one graph launch, and the step no
longer depends on how fast the host can issue them.  Per call: one device copy of the image into the captured input
buffer, the replay, one D2H copy of the packed detections per image (`rbbox2result_packed`, what the reference's
`rbbox2result` also pays).  Results are those of `simple_test_batch`.
"""
import torch

from .core import rbbox2result_packed


class GraphedInference(object):

    def __init__(self, model, img, img_metas, warmup=3):
        """model: an eval-mode OrientedRepPointsDetector on a GPU; img [B,3,H,W] (its shape / dtype are captured);
        img_metas: the B meta dicts used for every later call.  Raises if the path is not capturable."""
        if model.training or not img.is_cuda:
            raise ValueError("GraphedInference needs an eval-mode model and a CUDA image")
        if model.test_cfg.nms.get('type', 'rnms') != 'rnms' or not model.test_cfg.get('static_postprocess', True):
            raise ValueError("GraphedInference needs the static rnms post-processing")
        self.model, self.metas = model, list(img_metas)
        self.num_classes = model.bbox_head.num_classes
        self.static_img = img.clone()
        head = model.bbox_head

        def device_part():
            outs = head(model.extract_feat(self.static_img))
            return head.get_bboxes(*(tuple(outs) + (self.metas, model.test_cfg, False)), static=True)

        side = torch.cuda.Stream(device=img.device)
        side.wait_stream(torch.cuda.current_stream(img.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):               # library algorithm selection, workspaces, weight packing
                device_part()
        torch.cuda.current_stream(img.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only this thread's calls are checked during capture (a process-group watchdog thread polling its
        # events must not abort it)
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.packed = device_part()

    def __call__(self, img):
        """The per-image result lists of `simple_test_batch(img, img_metas)`."""
        self.static_img.copy_(img, non_blocking=True)
        self.graph.replay()
        results = [rbbox2result_packed(p, self.num_classes) for p in self.packed]
        if any(r is None for r in results):               # more pairs above score_thr than the static capacity holds
            with torch.no_grad():
                return self.model.simple_test_batch(img, self.metas)
        return results
