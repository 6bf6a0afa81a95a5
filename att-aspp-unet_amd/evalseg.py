"""Integer-count Dice / IoU of eval_segmentation_batch.py:41-49 (the parity metric of SURVEY.md section 8, row a15).

``dice(a, b)`` / ``iou(a, b)`` take what the reference takes -- two mask arrays of any numeric dtype, binarised with
``> 0`` -- and return the same float64 quotients of INTEGER counts, ``(2*|a&b| + eps) / (|a| + |b| + eps)`` and
``(|a&b| + eps) / (|a| + |b| - |a&b| + eps)``.  Host numpy arrays are counted on the host (that is the reference's own
arithmetic); device tensors are counted by ``aau_seg_counts`` (csrc/imgproc.hip) -- exact 64-bit counts, one 24-byte
read back -- so a batch of predicted masks never leaves HBM to be scored.  ``hd95`` (evalseg:51-58) is host geometry on
one mask pair: the cross-shaped erosion is restated in numpy (cv2 is absent), the Euclidean distance transform is
scipy's, as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _bin(a):
    """evalseg:41."""
    return (np.asarray(a) > 0).astype(np.uint8)


def counts(a, b):
    """-> (|a|, |b|, |a & b|) as Python ints after ``> 0`` binarisation."""
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        dev = a.device if isinstance(a, torch.Tensor) and a.is_cuda else (b.device if isinstance(b, torch.Tensor) else None)
        if dev is not None and dev.type == "cuda":
            ta = torch.as_tensor(a, device=dev)
            tb = torch.as_tensor(b, device=dev)
            if ta.shape != tb.shape:
                raise ValueError(f"mask shapes differ: {tuple(ta.shape)} vs {tuple(tb.shape)}")
            return ops.seg_counts(ta, tb)
        a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
        b = b.cpu().numpy() if isinstance(b, torch.Tensor) else b
    a, b = _bin(a), _bin(b)
    return int(a.sum(dtype=np.int64)), int(b.sum(dtype=np.int64)), int((a & b).sum(dtype=np.int64))


def dice(a, b, eps=1e-7):
    """evalseg:43-45."""
    na, nb, inter = counts(a, b)
    return (2 * inter + eps) / (na + nb + eps)


def iou(a, b, eps=1e-7):
    """evalseg:47-49."""
    na, nb, inter = counts(a, b)
    return (inter + eps) / (na + nb - inter + eps)


def _erode_cross(a: np.ndarray) -> np.ndarray:
    """cv2.erode(a, [[0,1,0],[1,1,1],[0,1,0]]) of a 0/1 image: the minimum over the 4-neighbourhood; cv2's default border
    for erosion is +infinity, i.e. pixels outside the image do not erode the rim."""
    p = np.pad(a, 1, constant_values=1)
    return p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]


def hd95(a, b) -> float:
    """evalseg:51-58: 95th-percentile symmetric Hausdorff distance (pixels) between the inner boundaries of two masks;
    NaN when either is empty."""
    from scipy.ndimage import distance_transform_edt
    a = a.cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.cpu().numpy() if isinstance(b, torch.Tensor) else b
    a, b = _bin(a), _bin(b)
    if a.sum() == 0 or b.sum() == 0:
        return float("nan")
    ab, bb = a - _erode_cross(a), b - _erode_cross(b)
    dta, dtb = distance_transform_edt(1 - ab), distance_transform_edt(1 - bb)
    d1, d2 = dtb[ab.astype(bool)], dta[bb.astype(bool)]
    return float(max(np.percentile(d1, 95), np.percentile(d2, 95)))
