"""TEST INFRASTRUCTURE -- numpy restatement of the batched augmentation kernels (csrc/augment.hip).

Only tests/ may import this; the product path (att-aspp-unet_amd/augment.py -> csrc/augment.hip) never does.
PARITY UNPINNED against albumentations / cv2 themselves (not importable here, no version pinned by the reference,
attention_aspp_unet_pipeline_stage.py:149-153): this file restates the same published algorithms as the kernels,
operation by operation (separately rounded fp32 multiplies and adds), so the GPU results must match it bit for bit.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
M64 = (1 << 64) - 1


def hash_uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    """common.h hash_uniform: splitmix-style counter hash -> fp32 in [0, 1)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & M64) + idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(F32) * F32(1.0 / 16777216.0)


def _reflect101(i, n):
    i = np.asarray(i).copy()
    if n == 1:
        return np.zeros_like(i)
    while True:
        bad = (i < 0) | (i >= n)
        if not bad.any():
            return i
        i = np.where(i < 0, -i, i)
        i = np.where(i >= n, 2 * (n - 1) - i, i)


def _round_half_even_sat(v):
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def warp_affine(img: np.ndarray, inv6: np.ndarray, nearest=False, border=0) -> np.ndarray:
    """cv2.warpAffine semantics through the dst -> src map ``inv6`` (exact bilinear weights)."""
    H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    m = np.asarray(inv6, np.float64)
    fx = (m[0] * xs + m[1] * ys) + m[2]
    fy = (m[3] * xs + m[4] * ys) + m[5]

    def at(xx, yy):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], border).astype(F32)
    if nearest:
        return at(np.floor(fx + 0.5).astype(np.int64), np.floor(fy + 0.5).astype(np.int64)).astype(np.uint8)
    x0d, y0d = np.floor(fx), np.floor(fy)
    far = (x0d < -2) | (y0d < -2) | (x0d > W + 1) | (y0d > H + 1)
    x0 = np.clip(x0d, -4, W + 4).astype(np.int64)
    y0 = np.clip(y0d, -4, H + 4).astype(np.int64)
    a, b = (fx - x0d).astype(F32), (fy - y0d).astype(F32)
    r0 = (at(x0, y0) * (F32(1) - a)).astype(F32) + (at(x0 + 1, y0) * a).astype(F32)
    r1 = (at(x0, y0 + 1) * (F32(1) - a)).astype(F32) + (at(x0 + 1, y0 + 1) * a).astype(F32)
    v = (r0.astype(F32) * (F32(1) - b)).astype(F32) + (r1.astype(F32) * b).astype(F32)
    return np.where(far, border, _round_half_even_sat(v.astype(F32))).astype(np.uint8)


def elastic_fields(seed: int, H: int, W: int, taps: np.ndarray) -> np.ndarray:
    """-> fp32 [2, H, W]: uniform(-1, 1) counter noise blurred by the separable filter ``taps`` (BORDER_REFLECT_101)."""
    idx = np.arange(2 * H * W, dtype=np.uint64)
    noise = ((hash_uniform(seed, idx) * F32(2)).astype(F32) + F32(-1)).astype(F32).reshape(2, H, W)
    k = len(taps)
    r = k // 2
    out = np.empty_like(noise)
    for p in range(2):
        tmp = np.zeros((H, W), F32)
        for t in range(k):                                     # horizontal pass, taps in order
            tmp = (tmp + (noise[p][:, _reflect101(np.arange(W) + t - r, W)] * taps[t]).astype(F32)).astype(F32)
        acc = np.zeros((H, W), F32)
        for t in range(k):
            acc = (acc + (tmp[_reflect101(np.arange(H) + t - r, H), :] * taps[t]).astype(F32)).astype(F32)
        out[p] = acc
    return out


def remap(img: np.ndarray, disp: np.ndarray, alpha: float, nearest=False) -> np.ndarray:
    """cv2.remap(img, x + alpha dx, y + alpha dy, BORDER_REFLECT_101)."""
    if alpha == 0:
        return img.copy()
    H, W = img.shape
    ys, xs = np.mgrid[0:H, 0:W]
    al = F32(alpha)
    fx = (xs.astype(F32) + (disp[0] * al).astype(F32)).astype(F32)
    fy = (ys.astype(F32) + (disp[1] * al).astype(F32)).astype(F32)
    if nearest:
        xi = _reflect101(np.floor(fx + F32(0.5)).astype(np.int64), W)
        yi = _reflect101(np.floor(fy + F32(0.5)).astype(np.int64), H)
        return img[yi, xi]
    x0f, y0f = np.floor(fx), np.floor(fy)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    a, b = (fx - x0f).astype(F32), (fy - y0f).astype(F32)
    xa, xb, ya, yb = _reflect101(x0, W), _reflect101(x0 + 1, W), _reflect101(y0, H), _reflect101(y0 + 1, H)
    g = lambda yy, xx: img[yy, xx].astype(F32)
    r0 = (g(ya, xa) * (F32(1) - a)).astype(F32) + (g(ya, xb) * a).astype(F32)
    r1 = (g(yb, xa) * (F32(1) - a)).astype(F32) + (g(yb, xb) * a).astype(F32)
    v = (r0.astype(F32) * (F32(1) - b)).astype(F32) + (r1.astype(F32) * b).astype(F32)
    return _round_half_even_sat(v.astype(F32))


def apply(imgs: np.ndarray, msks, p, taps: np.ndarray, clahe_fn, median_fn, train=True):
    """The batch transform of att-aspp-unet_amd/augment.py::apply on the host: ``imgs`` uint8 [N,H,W] -> (x fp32, y fp32)."""
    N, H, W = imgs.shape
    xs, ys = [], []
    for n in range(N):
        img = imgs[n]
        msk = None if msks is None else msks[n]
        if train:
            if p.flip[n]:
                img = img[:, ::-1]
                msk = None if msk is None else msk[:, ::-1]
            img = warp_affine(img, p.inv_mats[n], nearest=False)
            if msk is not None:
                msk = warp_affine(msk, p.inv_mats[n], nearest=True)
            img = p.luts[n][img]
            if p.elastic_alpha[n] != 0:
                d = elastic_fields(int(p.elastic_seed[n]), H, W, taps)
                img = remap(img, d, float(p.elastic_alpha[n]))
                if msk is not None:
                    msk = remap(msk, d, float(p.elastic_alpha[n]), nearest=True)
        if p.clahe[n]:
            img = clahe_fn(img, 1.0, 8)
        if p.median[n]:
            img = median_fn(img)
        xs.append(img.astype(F32) / F32(255))
        if msk is not None:
            ys.append(msk.astype(F32) / F32(255))
    return np.stack(xs), (np.stack(ys) if ys else None)
