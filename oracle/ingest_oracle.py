"""CPU restatement of the reference's OSCD ingest arithmetic (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

PARITY UNPINNED: the reference's city_loader (utils/dataloaders.py:86-111) calls rasterio and cv2.resize, neither of
which is installed in the build image, so this file could not be checked against the reference itself.  It restates
cv2's published float INTER_LINEAR algorithm (OpenCV modules/imgproc/src/resize.cpp: resizeGeneric_, HResizeLinear,
VResizeLinear; coordinate rule fx = (dx + 0.5) * scale - 0.5, border weights (1, 0)) and anchors on the reference's call
site: normalise in float32 first, resize second, dsize = (label width, label height).
"""
import numpy as np


def _taps(n_src, n_dst):
    scale = n_src / n_dst                                   # double, like cv2's inv_scale
    f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo], f[lo] = 0, 0.0
    hi = s >= n_src - 1
    s[hi], f[hi] = n_src - 1, 0.0
    return s, np.minimum(s + 1, n_src - 1), f


def resize_linear_f32(img, width, height):
    """cv2.resize(img, (width, height)) for a 2-D float32 image, default interpolation."""
    img = np.asarray(img, dtype=np.float32)
    if img.shape == (height, width):
        return img.copy()                                   # cv2 copies when the size does not change
    sy, sy1, fy = _taps(img.shape[0], height)
    sx, sx1, fx = _taps(img.shape[1], width)
    one = np.float32(1.0)
    rows0 = img[sy][:, sx] * (one - fx)[None, :] + img[sy][:, sx1] * fx[None, :]      # horizontal pass
    rows1 = img[sy1][:, sx] * (one - fx)[None, :] + img[sy1][:, sx1] * fx[None, :]
    return (rows0 * (one - fy)[:, None] + rows1 * fy[:, None]).astype(np.float32)      # vertical pass


def ingest_band(band, mean, std, width, height):
    """utils/dataloaders.py:94-98 for one band: astype(float32), (band - mean) / std, cv2.resize(band, (width, height))."""
    b = (band.astype(np.float32) - np.float32(mean)) / np.float32(std)
    return resize_linear_f32(b, width, height)


def gray_from_rgb(rgb):
    """cv2.imread(path, 0) on a colour PNG: fixed-point BT.601 luma (OpenCV color_rgb: R2Y=4899, G2Y=9617, B2Y=1868, shift 14)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)
