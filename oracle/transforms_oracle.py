"""CPU oracle of the non-augmenting input transform (TEST INFRASTRUCTURE ONLY: imported by tests/ only).

Restates `MyTransform(width, height, gray, augment=False)` of the reference
(rightLaneNetwork/dataManagement/myTransforms.py:6-31): Resize -> [ToGray] -> Normalize -> ToTensorV2.
The arithmetic lives in third-party libraries that are NOT under /root/reference and not installed here
(albumentations 0.5.2 -> cv2.resize INTER_LINEAR for 8-bit images, cv2 INTER_NEAREST for masks,
cv2.cvtColor RGB2GRAY, albumentations.Normalize with max_pixel_value 255), so this file restates their published
algorithms and **parity is unpinned**: no fixture of the reference covers it.
"""
import numpy as np

MEAN = (0.485, 0.456, 0.406)  # albumentations.Normalize defaults, applied to the stored (BGR) channel order
STD = (0.229, 0.224, 0.225)


def _lin_coef(dst, src):
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * (src / dst) - 0.5).astype(np.float32)
    i = np.floor(f).astype(np.int64)
    f = f - i.astype(np.float32)
    lo = i < 0
    i[lo], f[lo] = 0, 0.0
    hi = i >= src - 1
    i[hi], f[hi] = src - 1, 0.0
    i1 = np.minimum(i + 1, src - 1)
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return i, i1, c0, c1


def resize_linear_u8(img, h, w):
    """img uint8 [hs, ws, C] -> uint8 [h, w, C]; cv2.resize(..., interpolation=INTER_LINEAR) fixed-point path."""
    hs, ws = img.shape[:2]
    x0, x1, a0, a1 = _lin_coef(w, ws)
    y0, y1, b0, b1 = _lin_coef(h, hs)
    s = img.astype(np.int64)
    rows0 = s[y0][:, x0] * a0[None, :, None] + s[y0][:, x1] * a1[None, :, None]
    rows1 = s[y1][:, x0] * a0[None, :, None] + s[y1][:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (rows0 >> 4)) >> 16) + ((b1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_nearest(mask, h, w):
    hs, ws = mask.shape[:2]
    sy = np.minimum(np.floor(np.arange(h) * (hs / h)).astype(np.int64), hs - 1)
    sx = np.minimum(np.floor(np.arange(w) * (ws / w)).astype(np.int64), ws - 1)
    return mask[sy][:, sx]


def to_gray(img):
    v = img.astype(np.int64)
    g = (v[..., 0] * 4899 + v[..., 1] * 9617 + v[..., 2] * 1868 + (1 << 13)) >> 14
    return np.repeat(g[..., None], 3, axis=-1).astype(np.uint8)


def transform(img, label=None, width=160, height=120, gray=False):
    """-> (float32 [3, height, width], int64 [height, width] or None)"""
    r = resize_linear_u8(img, height, width)
    if gray:
        r = to_gray(r)
    mean = np.array(MEAN, np.float32) * np.float32(255.0)
    inv = np.float32(1.0) / (np.array(STD, np.float32) * np.float32(255.0))
    x = (r.astype(np.float32) - mean) * inv
    y = None if label is None else resize_nearest(label, height, width).astype(np.int64)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), y
