"""Mirror of mmdet/ops/point_justify (`pointsJf(points, polygons, output) -> int`, points_justify.cpp:17-39) plus the
aligned form the SpatialBorderLoss needs (row i against quad i), which replaces 9 x (M x M) matrices + torch.diag
(mmdet/models/losses/spatial_border_loss.py:24-67)."""
import torch

from .. import _lib


def pointsJf(points, polygons, output):
    """In-place fill of output[M,K] with 1.0 (inside) / 0.0.  Requires CUDA + contiguous, like CHECK_INPUT."""
    for t, n in ((points, "points"), (polygons, "polygons"), (output, "output")):
        _lib.require_cuda(t, n)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % n)
    if points.size(1) != 2:
        print("wrong points size")
        return 0
    dts = {points.dtype, polygons.dtype, output.dtype}
    if dts == {torch.float64}:
        # scalar_t = double of AT_DISPATCH_FLOATING_TYPES (points_justify_kernel.cu:107): the kernel copies every coordinate into a
        # `struct point { float x, y; }` (:20-23, 36-50) and tests in float -- so the double instantiation IS the float arithmetic on
        # the rounded coordinates, its 0 / 1 flags stored as double.  Same here, bit for bit.
        out32 = torch.empty(output.shape, dtype=torch.float32, device=output.device)
        r = pointsJf(points.float(), polygons.float(), out32)
        output.copy_(out32)
        return r
    if dts != {torch.float32}:
        raise TypeError("pointsJf: float32 (or all-float64) tensors only")
    rows, cols = points.size(0), polygons.size(0)
    with torch.cuda.device(points.device):
        rc = _lib.lib().orp_points_justify(_lib.ptr(points), rows, _lib.ptr(polygons), cols, _lib.ptr(output),
                                           _lib.stream_of(points))
    _lib.check(rc, "orp_points_justify")
    return 1


def points_in_quad_aligned(pts18, quads):
    """pts18 [M,18], quads [M,8] -> [M,9] inside flags (the diagonal of nine pointsJf calls)."""
    _lib.require_cuda(pts18, "pts18")
    p = pts18.detach().float().reshape(-1, 18).contiguous()
    q = quads.detach().float().reshape(-1, 8).contiguous()
    assert p.size(0) == q.size(0)
    out = torch.empty((p.size(0), 9), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.lib().orp_points_in_quad_aligned(_lib.ptr(p), _lib.ptr(q), p.size(0), _lib.ptr(out),
                                                   _lib.stream_of(p))
    _lib.check(rc, "orp_points_in_quad_aligned")
    return out
