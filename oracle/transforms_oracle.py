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


# ---------------------------------------------------------------------------------------------------------------
# augmenting branch (myTransforms.py:8-13): the per-pixel arithmetic for GIVEN random parameters (the parameter table
# is drawn by the caller; see sim2real_lane_segment_amd/dataManagement/myTransforms.py:sample_aug_params).
# Same published definitions as the device kernels; parity with albumentations/cv2 unpinned.
# ---------------------------------------------------------------------------------------------------------------
AUG_NP = 80


def hsv_shift_u8(img, dh, ds, dv):
    f32 = np.float32
    c = img.astype(np.int64)
    c0, c1, c2 = c[..., 0], c[..., 1], c[..., 2]
    vmax = np.maximum(c0, np.maximum(c1, c2))
    vmin = np.minimum(c0, np.minimum(c1, c2))
    d = vmax - vmin
    dsafe = np.where(d == 0, 1, d).astype(f32)
    h_r = f32(30.0) * (c1 - c2).astype(f32) / dsafe
    h_g = f32(60.0) + f32(30.0) * (c2 - c0).astype(f32) / dsafe
    h_b = f32(120.0) + f32(30.0) * (c0 - c1).astype(f32) / dsafe
    hf = np.where(vmax == c0, h_r, np.where(vmax == c1, h_g, h_b)).astype(f32)
    hf = np.where(hf < 0, hf + f32(180.0), hf).astype(f32)
    hf = np.where(d == 0, f32(0.0), hf).astype(f32)
    H = np.rint(hf).astype(np.int64)
    H = np.where(H >= 180, H - 180, H)
    S = np.where(vmax == 0, 0, np.rint(f32(255.0) * d.astype(f32) / np.where(vmax == 0, 1, vmax).astype(f32))).astype(np.int64)
    V = vmax
    h2 = np.fmod(H.astype(f32) + f32(dh), f32(180.0)).astype(f32)
    h2 = np.where(h2 < 0, h2 + f32(180.0), h2).astype(f32)
    H2 = h2.astype(np.int64)
    S2 = np.minimum(np.maximum(S.astype(f32) + f32(ds), f32(0)), f32(255)).astype(np.int64)
    V2 = np.minimum(np.maximum(V.astype(f32) + f32(dv), f32(0)), f32(255)).astype(np.int64)
    s = S2.astype(f32) * f32(1.0 / 255.0)
    v = V2.astype(f32) * f32(1.0 / 255.0)
    h6 = H2.astype(f32) * f32(1.0 / 30.0)
    sec = h6.astype(np.int64)
    f = (h6 - sec.astype(f32)).astype(f32)
    one = f32(1.0)
    pp = (v * (one - s)).astype(f32)
    qq = (v * (one - (s * f).astype(f32))).astype(f32)
    tt = (v * (one - (s * (one - f).astype(f32)).astype(f32))).astype(f32)
    r = np.choose(np.minimum(sec, 5), [v, qq, pp, pp, tt, v])
    g = np.choose(np.minimum(sec, 5), [tt, v, v, qq, pp, pp])
    b = np.choose(np.minimum(sec, 5), [pp, pp, tt, v, v, qq])
    out = np.stack([r, g, b], -1).astype(f32)
    return np.clip(np.rint(out * f32(255.0)), 0, 255).astype(np.uint8)


def _hash32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def gauss_noise(n_index, h, w, sigma, seed):
    """N(0, sigma) per (pixel, channel) of image n_index from the same counter hash as the device kernel."""
    f32 = np.float32
    idx = (np.uint64(n_index * h * w) + np.arange(h * w, dtype=np.uint64))[:, None] * np.uint64(3) + np.arange(3, dtype=np.uint64)
    idx &= np.uint64(0xFFFFFFFF)
    s1 = (np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    s2 = (np.uint64(seed) * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    r1 = _hash32((idx * np.uint64(2) + np.uint64(1) + s1) & np.uint64(0xFFFFFFFF))
    r2 = _hash32((idx * np.uint64(2) + np.uint64(2) + s2) & np.uint64(0xFFFFFFFF))
    u1 = ((r1 >> np.uint64(8)).astype(f32) + f32(1.0)) * f32(1.0 / 16777216.0)
    u2 = (r2 >> np.uint64(8)).astype(f32) * f32(1.0 / 16777216.0)
    g = np.sqrt(f32(-2.0) * np.log(u1)).astype(f32) * np.cos(f32(6.2831853) * u2).astype(f32) * f32(sigma)
    return g.reshape(h, w, 3).astype(f32)


def augment(img, label, params, n_index, width=160, height=120):
    """One image. params: float32[AUG_NP] (layout: include/rln.h). -> (float32 [3,h,w], int64 [h,w] or None)"""
    f32 = np.float32
    p = np.asarray(params, f32)
    cy, cx, ch, cw = int(p[3]), int(p[4]), int(p[5]), int(p[6])
    crop = hsv_shift_u8(img[cy:cy + ch, cx:cx + cw], p[0], p[1], p[2])     # pointwise: commutes with the crop
    r = resize_linear_u8(crop, height, width)
    if p[7] < 0.5:
        k = p[16:16 + 49].reshape(7, 7)
        pad = np.pad(r.astype(f32), ((3, 3), (3, 3), (0, 0)), mode="reflect")   # BORDER_REFLECT_101
        acc = np.zeros(r.shape, f32)
        for ky in range(7):
            for kx in range(7):
                if k[ky, kx] != 0:
                    # fmaf(w, s, acc): exact product (24 x 24 bits fit a double), one rounding to float32
                    acc = (np.float64(k[ky, kx]) * pad[ky:ky + height, kx:kx + width].astype(np.float64)
                           + acc.astype(np.float64)).astype(f32)
        v = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    else:
        g = gauss_noise(n_index, height, width, p[9], int(p[10]))
        v = np.clip(r.astype(f32) + g, 0, 255).astype(np.uint8)                 # truncation
    mean = np.array(MEAN, f32) * f32(255.0)
    inv = f32(1.0) / (np.array(STD, f32) * f32(255.0))
    x = (v.astype(f32) - mean) * inv
    y = None if label is None else resize_nearest(label[cy:cy + ch, cx:cx + cw], height, width).astype(np.int64)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), y


def overlay(frame, probs, colors=((0, 0, 0), (0, 255, 0), (255, 0, 0), (0, 0, 255)), paint=(1, 2, 3)):
    """makeDemoVideo.py:36-46 for one frame: pred = argmax (first maximum), frame_out = cv2.resize(frame, (w, h))
    [default INTER_LINEAR: the script's third positional argument is `dst`], frame_out[pred == k] = colors[k].
    frame uint8 [hs, ws, 3], probs float32 [K, h, w] -> (uint8 [h, w, 3], uint8 [h, w])."""
    k, h, w = probs.shape
    pred = np.argmax(probs, axis=0).astype(np.uint8)
    out = resize_linear_u8(frame, h, w).copy()
    for c in paint:
        out[pred == c] = np.array(colors[c], np.uint8)
    return out, pred
