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

from .. import _lib
from .core import rbbox2result_packed


def _param_fingerprint(tensors):
    """(storage address, version) of every parameter and buffer: changes when load_state_dict / an optimizer step /
    .to() replaced or rewrote them (writes through `.data` bypass the version counter: call `recapture()` after those)."""
    fp = 0
    for t in tensors:
        fp = (fp * 1000003 + t.data_ptr() + 7919 * t._version) & 0xFFFFFFFFFFFF
    return fp


class GraphedInference(object):
    """Owns everything its graph touches: the input buffer, the packed output, the scratch (allocated from the graph's
    private memory pool while capturing: `_lib.workspace`), and references to the cached packed DeformConv / 3x3 weights
    and folded BatchNorm affines the captured kernels read (`_lib.keep_for_graph`) -- so no later eager call (a large
    merge NMS growing the shared scratch, the capacity-overflow fallback, a training step, a cache eviction) can free
    or move memory a replay uses.  If the model's parameters change, the next call re-captures."""

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
        self.warmup = max(1, warmup)
        self.captures = 0
        self._capture()

    def _device_part(self):
        model, head = self.model, self.model.bbox_head
        outs = head(model.extract_feat(self.static_img))
        return head.get_bboxes(*(tuple(outs) + (self.metas, model.test_cfg, False)), static=True)

    def _capture(self):
        dev = self.static_img.device
        self.graph = None                                     # release the previous graph (and its pool) first
        self.packed = None
        self._refs = []
        prev, _lib._keepalive = _lib._keepalive, self._refs
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):                  # library algorithm selection, weight packing, BN folding
                    self._device_part()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are checked during capture (a process-group watchdog thread polling
            # its events must not abort it)
            with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self.packed = self._device_part()
            self.graph = graph
        finally:
            _lib._keepalive = prev
        self._tensors = list(self.model.parameters()) + list(self.model.buffers())
        self._fingerprint = _param_fingerprint(self._tensors)
        self.captures += 1

    def recapture(self):
        """Force a new capture (after parameter writes the version counters do not see, e.g. through `.data`)."""
        self._capture()

    def __call__(self, img):
        """The per-image result lists of `simple_test_batch(img, img_metas)`."""
        self.static_img.copy_(img, non_blocking=True)
        self.graph.replay()
        # checked while the replay runs (keeps ~50 us of host work off the critical path): if the weights changed, the
        # packs / folded affines baked into the graph are stale -> capture again and redo this call
        if _param_fingerprint(self._tensors) != self._fingerprint:
            self._capture()
            self.graph.replay()
        results = [rbbox2result_packed(p, self.num_classes) for p in self.packed]
        if any(r is None for r in results):               # more pairs above score_thr than the static capacity holds
            with torch.no_grad():
                return self.model.simple_test_batch(img, self.metas)
        return results


class PipelinedInference(object):
    """Throughput mode of `GraphedInference`: `depth` (default 2) captured graphs, each with its own input / output /
    scratch, replayed on their own streams with the results fetched asynchronously into pinned host buffers.  `submit(img)`
    queues an image and returns the results of the image submitted `depth` calls earlier (None while the pipe fills);
    `flush()` returns what is still in flight, oldest first.  The tail of one image (decode, NMS: many small kernels
    that leave most CUs idle) then overlaps the backbone of the next one, and the host never waits on the image it has
    just queued.  Every image still runs the complete step and produces the same detections as `GraphedInference`."""

    def __init__(self, model, img, img_metas, depth=2, warmup=3, _allow_half=False):
        dev = img.device
        if (depth > 2 and not _allow_half and torch.backends.cudnn.deterministic
                and any(p.dtype != torch.float32 for p in model.parameters())):
            # measured (round 6, tests/checks/half_pipeline_probe.py): with the library restricted to its reproducible solvers
            # (torch.backends.cudnn.deterministic = True) a model.half() detector replays correctly from one or two captured graphs, but
            # with FOUR in flight the device stops making progress (a replay's completion event never signals; either half-precision
            # DeformConv kernel).  In the default library mode the same model runs four deep at the fp32 model's rate (338 images/s).
            # Refused here rather than hung there.
            raise ValueError("PipelinedInference: with torch.backends.cudnn.deterministic = True a half / bfloat16 model supports at most "
                             "two graphs in flight (depth <= 2); switch the flag off or lower the depth")
        self.model, self.metas, self.depth = model, list(img_metas), depth
        self.num_classes = model.bbox_head.num_classes
        # with several images in flight the two towers of ONE image need no second stream (measured: 225 vs 218 img/s)
        head = model.bbox_head
        prev = getattr(head, 'tower_streams', None)
        head.tower_streams = False
        try:
            self.slots = [GraphedInference(model, img, img_metas, warmup) for _ in range(depth)]
        finally:
            head.tower_streams = prev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.host = [[torch.empty(p.shape, dtype=p.dtype, pin_memory=True) for p in s.packed] for s in self.slots]
        self.pending = [None] * depth
        self.count = 0

    def _collect(self, k):
        img = self.pending[k]
        self.pending[k] = None
        self.events[k].synchronize()
        results = [rbbox2result_packed(h, self.num_classes) for h in self.host[k]]
        if any(r is None for r in results):               # static capacity overflow: the reference-shaped path, synchronously
            with torch.no_grad():
                return self.model.simple_test_batch(img, self.metas)
        return results

    def submit(self, img):
        k = self.count % self.depth
        out = self._collect(k) if self.pending[k] is not None else None
        slot, s = self.slots[k], self.streams[k]
        if _param_fingerprint(slot._tensors) != slot._fingerprint:
            head = self.model.bbox_head
            prev, head.tower_streams = getattr(head, 'tower_streams', None), False
            try:
                slot._capture()
            finally:
                head.tower_streams = prev
            self.host[k] = [torch.empty(p.shape, dtype=p.dtype, pin_memory=True) for p in slot.packed]
        s.wait_stream(torch.cuda.current_stream(img.device))     # the image may have been produced on the caller's stream
        with torch.cuda.stream(s):
            slot.static_img.copy_(img, non_blocking=True)
            slot.graph.replay()
            for h, p in zip(self.host[k], slot.packed):
                h.copy_(p, non_blocking=True)
            self.events[k].record(s)
        self.pending[k] = img
        self.count += 1
        return out

    def flush(self):
        outs = []
        for j in range(self.depth):
            k = (self.count + j) % self.depth
            if self.pending[k] is not None:
                outs.append(self._collect(k))
        return outs
