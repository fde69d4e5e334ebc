"""Image operations of the pipeline without cv2 / mmcv (mmcv.imrescale / imresize / imflip / impad / imnormalize,
cv2.warpAffine): HWC numpy arrays in and out, resampling on torch CPU ops with OpenCV's half-pixel-centre convention."""
import numpy as np
import torch
import torch.nn.functional as F

_MODES = {'nearest': 'nearest', 'bilinear': 'bilinear', 'bicubic': 'bicubic', 'area': 'area'}


def _to_chw(img):
    t = torch.from_numpy(np.ascontiguousarray(img)).float()
    if t.dim() == 2:
        t = t[:, :, None]
    return t.permute(2, 0, 1)[None]


def _from_chw(t, like):
    out = t[0].permute(1, 2, 0).numpy()
    if like.ndim == 2:
        out = out[:, :, 0]
    if like.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    else:
        out = out.astype(like.dtype)
    return out


def imresize(img, size, return_scale=False, interpolation='bilinear'):
    """size = (w, h) (mmcv convention) -> resized image (and w_scale, h_scale)."""
    h, w = img.shape[:2]
    mode = _MODES.get(interpolation, 'bilinear')
    kw = dict(align_corners=False) if mode in ('bilinear', 'bicubic') else {}
    out = _from_chw(F.interpolate(_to_chw(img), size=(int(size[1]), int(size[0])), mode=mode, **kw), img)
    if not return_scale:
        return out
    return out, size[0] / w, size[1] / h


def rescale_size(old_size, scale):
    """mmcv.rescale_size: scale = float factor or (long edge, short edge) bound, aspect ratio kept."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        scale_factor = scale
    else:
        max_long_edge, max_short_edge = max(scale), min(scale)
        scale_factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    return int(w * float(scale_factor) + 0.5), int(h * float(scale_factor) + 0.5), scale_factor


def imrescale(img, scale, return_scale=False, interpolation='bilinear'):
    h, w = img.shape[:2]
    new_w, new_h, scale_factor = rescale_size((w, h), scale)
    out = imresize(img, (new_w, new_h), interpolation=interpolation)
    return (out, scale_factor) if return_scale else out


def imflip(img, direction='horizontal'):
    direction = direction if isinstance(direction, str) else str(np.asarray(direction).reshape(-1)[0])
    assert direction in ('horizontal', 'vertical')
    return np.flip(img, axis=1 if direction == 'horizontal' else 0).copy()


def impad(img, shape, pad_val=0):
    out = np.full(tuple(shape) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1], ...] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):
    pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
    pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, (pad_h, pad_w), pad_val)


def imnormalize(img, mean, std, to_rgb=True):
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    return (img - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)


def warp_affine(img, M, dsize, interpolation='bilinear'):
    """cv2.warpAffine(img, M, (w, h)): dst(x, y) = src(M^-1 [x, y, 1]); constant-0 border; pixel centres at integers."""
    w, h = int(dsize[0]), int(dsize[1])
    Minv = np.linalg.inv(np.vstack([np.asarray(M, np.float64), [0, 0, 1]]))[:2]
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    sx = Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]
    sy = Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]
    H, W = img.shape[:2]
    grid = np.stack([(sx + 0.5) / W * 2 - 1, (sy + 0.5) / H * 2 - 1], -1)            # align_corners=False coordinates
    t = F.grid_sample(_to_chw(img), torch.from_numpy(grid).float()[None], mode='nearest' if interpolation == 'nearest' else 'bilinear',
                      padding_mode='zeros', align_corners=False)
    return _from_chw(t, img)


def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """cv2.getRotationMatrix2D: positive angle = counter-clockwise (origin top-left)."""
    a = np.deg2rad(angle_deg)
    alpha, beta = scale * np.cos(a), scale * np.sin(a)
    cx, cy = float(center[0]), float(center[1])
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)
