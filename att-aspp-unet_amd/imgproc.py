"""GPU-resident image work around the network.

The reference does all of this on the host, per frame, with cv2 / scikit-image / SciPy
(attention_aspp_unet_pipeline_stage.py:340-348 ``refine_mask``, :449-457 the per-slice body of ``predict`` /
``calibrate``, model_attention_aspp.py:14-31,75-80), with a device->host copy of every probability map in between.
Here the same steps are kernels over ``[N, H, W]`` batches in HBM (csrc/imgproc.hip); a sweep of hundreds of frames
goes from raw bytes to final masks without leaving the device.

All functions take and return CUDA tensors (uint8 masks hold 0 / 1).  ``refine_mask`` also accepts a numpy array and
then returns one, which makes it a drop-in for the reference function.  The algorithms are restated from the published
ones of the libraries the reference calls; cv2 and skimage are not importable in the build container, so against THEM
parity is unpinned -- the tests compare with the numpy / SciPy restatement in oracle/imgproc_ref.py.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _abi
from ._abi import check, fn


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _as3(x: torch.Tensor, dtype) -> tuple[torch.Tensor, tuple]:
    """-> contiguous [N, H, W] view of ``x`` ([H,W], [N,H,W] or [N,1,H,W]) and the shape to restore."""
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
        raise _abi.AauError("imgproc functions need CUDA tensors (no CPU fallback)")
    if x.dtype != dtype:
        raise _abi.AauError(f"expected dtype {dtype}, got {x.dtype}")
    shape = tuple(x.shape)
    if x.dim() == 2:
        x = x[None]
    elif x.dim() == 4 and x.shape[1] == 1:
        x = x[:, 0]
    elif x.dim() != 3:
        raise _abi.AauError(f"expected [H,W], [N,H,W] or [N,1,H,W], got {shape}")
    return x.contiguous(), shape


def _restore(y: torch.Tensor, shape: tuple, hw=None) -> torch.Tensor:
    hw = hw or tuple(y.shape[-2:])
    if len(shape) == 2:
        return y[0]
    if len(shape) == 4:
        return y.view(shape[0], 1, *hw)
    return y


# ---------------------------------------------------------------- f1: inference tail
def resize_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """cv2.resize(x, (W, H), interpolation=INTER_LINEAR) for fp32 or uint8 frames; ``size`` = (H, W)."""
    dt = x.dtype
    if dt not in (torch.float32, torch.uint8):
        raise _abi.AauError("resize_bilinear: fp32 or uint8")
    x3, shape = _as3(x, dt)
    N, Hs, Ws = x3.shape
    Hd, Wd = int(size[0]), int(size[1])
    if (Hd, Wd) == (Hs, Ws):
        return x.clone()
    out = torch.empty(N, Hd, Wd, dtype=dt, device=x.device)
    name = "aau_resize_bilinear_f32" if dt == torch.float32 else "aau_resize_bilinear_u8"
    check(fn(name)(x3.data_ptr(), Hs, Ws, out.data_ptr(), Hd, Wd, N, _stream()), name)
    return _restore(out, shape, (Hd, Wd))


def gaussian_blur5(x: torch.Tensor) -> torch.Tensor:
    """cv2.GaussianBlur(x, (5, 5), 0) on fp32 frames."""
    x3, shape = _as3(x, torch.float32)
    N, H, W = x3.shape
    out = torch.empty_like(x3)
    check(fn("aau_gauss5_f32")(x3.data_ptr(), out.data_ptr(), N, H, W, _stream()), "aau_gauss5_f32")
    return _restore(out, shape)


def threshold(x: torch.Tensor, thr: float) -> torch.Tensor:
    """(x > thr).astype(uint8)."""
    x3, shape = _as3(x, torch.float32)
    out = torch.empty(x3.shape, dtype=torch.uint8, device=x.device)
    check(fn("aau_threshold_u8")(x3.data_ptr(), float(thr), out.data_ptr(), x3.numel(), _stream()), "aau_threshold_u8")
    return _restore(out, shape)


def keep_largest_component(m: torch.Tensor, min_area: int = 1, conn8: bool = True) -> torch.Tensor:
    """The largest connected component of every frame (zeros if it is smaller than ``min_area``)."""
    m3, shape = _as3(m, torch.uint8)
    N, H, W = m3.shape
    out = torch.empty_like(m3)
    labels = torch.empty(N, H, W, dtype=torch.int32, device=m.device)
    sizes = torch.empty(N, H, W, dtype=torch.int32, device=m.device)
    best = torch.empty(N, dtype=torch.int64, device=m.device)
    check(fn("aau_cc_keep_largest")(m3.data_ptr(), out.data_ptr(), labels.data_ptr(), sizes.data_ptr(), best.data_ptr(), N, H, W,
                                    1 if conn8 else 0, int(min_area), _stream()), "aau_cc_keep_largest")
    return _restore(out, shape)


def label(m: torch.Tensor, conn8: bool = True) -> torch.Tensor:
    """Component labels: int32, the smallest linear pixel index of the component, -1 for background."""
    m3, shape = _as3(m, torch.uint8)
    N, H, W = m3.shape
    labels = torch.empty(N, H, W, dtype=torch.int32, device=m.device)
    check(fn("aau_cc_label")(m3.data_ptr(), labels.data_ptr(), N, H, W, 1 if conn8 else 0, _stream()), "aau_cc_label")
    return _restore(labels, shape)


def morph(m: torch.Tensor, shape_id: int, erode: bool) -> torch.Tensor:
    m3, shape = _as3(m, torch.uint8)
    N, H, W = m3.shape
    out = torch.empty_like(m3)
    check(fn("aau_morph")(m3.data_ptr(), out.data_ptr(), N, H, W, int(shape_id), 1 if erode else 0, _stream()), "aau_morph")
    return _restore(out, shape)


def close_ellipse7(m: torch.Tensor) -> torch.Tensor:
    """cv2.morphologyEx(m, MORPH_CLOSE, getStructuringElement(MORPH_ELLIPSE, (7, 7)))."""
    return morph(morph(m, 7, False), 7, True)


def dilate3(m: torch.Tensor) -> torch.Tensor:
    """scipy.ndimage.binary_dilation(m, structure=ones((3, 3)), iterations=1)."""
    return morph(m, 3, False)


def fill_holes(m: torch.Tensor) -> torch.Tensor:
    """scipy.ndimage.binary_fill_holes(m)."""
    m3, shape = _as3(m, torch.uint8)
    N, H, W = m3.shape
    out = torch.empty_like(m3)
    labels = torch.empty(N, H, W, dtype=torch.int32, device=m.device)
    flags = torch.empty(N, H, W, dtype=torch.int32, device=m.device)
    check(fn("aau_fill_holes")(m3.data_ptr(), out.data_ptr(), labels.data_ptr(), flags.data_ptr(), N, H, W, _stream()), "aau_fill_holes")
    return _restore(out, shape)


def refine_mask(m):
    """pipeline:340-348: drop components below max(20, 0.15 % of the frame), keep the largest one, close it with the 7x7
    ellipse, fill its holes.  Device tensor in -> device tensor out (a batch [N,H,W] is refined frame by frame in one
    set of launches); numpy in -> numpy out."""
    if isinstance(m, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray((m != 0).astype(np.uint8))).cuda()
        return refine_mask(t).cpu().numpy().astype(m.dtype if m.dtype != bool else np.uint8)
    m3, shape = _as3(m, torch.uint8)
    H, W = m3.shape[-2:]
    min_area = max(20, int(0.0015 * H * W))
    big = keep_largest_component(m3, min_area=min_area, conn8=True)     # zeros when nothing reaches min_area (:344)
    return _restore(fill_holes(close_ellipse7(big)), shape)


def postprocess_probability(prob512: torch.Tensor, out_hw, thr: float) -> torch.Tensor:
    """pipeline:455-457: resize the [N,512,512] probability maps back to the slice size, 5x5 Gaussian, threshold,
    refine_mask.  Returns uint8 [N, H, W] on the device."""
    p = resize_bilinear(prob512, out_hw)
    return refine_mask(threshold(gaussian_blur5(p), thr))


# ---------------------------------------------------------------- f2: input pipeline
def normalize_minmax(u8: torch.Tensor) -> torch.Tensor:
    """cv2.normalize(x, None, 0, 255, NORM_MINMAX) per frame (uint8 in, uint8 out)."""
    x3, shape = _as3(u8, torch.uint8)
    N, H, W = x3.shape
    out = torch.empty_like(x3)
    ws = torch.empty(2 * N, dtype=torch.int32, device=u8.device)
    check(fn("aau_normalize_minmax_u8")(x3.data_ptr(), out.data_ptr(), ws.data_ptr(), N, H, W, _stream()), "aau_normalize_minmax_u8")
    return _restore(out, shape)


def clahe(u8: torch.Tensor, clip_limit: float = 1.0, tiles: int = 8) -> torch.Tensor:
    """cv2.createCLAHE(clip_limit, (tiles, tiles)).apply(x)."""
    x3, shape = _as3(u8, torch.uint8)
    N, H, W = x3.shape
    out = torch.empty_like(x3)
    lut = torch.empty(N * tiles * tiles * 256, dtype=torch.uint8, device=u8.device)
    check(fn("aau_clahe_u8")(x3.data_ptr(), out.data_ptr(), lut.data_ptr(), N, H, W, float(clip_limit), int(tiles), _stream()), "aau_clahe_u8")
    return _restore(out, shape)


def median3(u8: torch.Tensor) -> torch.Tensor:
    """cv2.medianBlur(x, 3)."""
    x3, shape = _as3(u8, torch.uint8)
    N, H, W = x3.shape
    out = torch.empty_like(x3)
    check(fn("aau_median3_u8")(x3.data_ptr(), out.data_ptr(), N, H, W, _stream()), "aau_median3_u8")
    return _restore(out, shape)


def to_float(u8: torch.Tensor, max_value: float = 255.0) -> torch.Tensor:
    """albumentations ToFloat(max_value)."""
    x3, shape = _as3(u8, torch.uint8)
    out = torch.empty(x3.shape, dtype=torch.float32, device=u8.device)
    check(fn("aau_u8_to_f32")(x3.data_ptr(), out.data_ptr(), float(max_value), x3.numel(), _stream()), "aau_u8_to_f32")
    return _restore(out, shape)


def preprocess_frames(u8: torch.Tensor, size: int = 512, resize_first: bool = False) -> torch.Tensor:
    """The inference-side input pipeline of pipeline:449-451 / :492-494 for a stack of uint8 slices [N,H,W]:
    normalize(MINMAX) -> CLAHE(1.0, 8x8) -> medianBlur(3) -> Resize(size) -> ToFloat(255)  ->  fp32 [N,1,size,size].
    ``resize_first`` gives the validation transform of FetalACDataset (:156: Resize, CLAHE, MedianBlur, ToFloat)."""
    x3, _ = _as3(u8, torch.uint8)
    if resize_first:
        e = median3(clahe(resize_bilinear(x3, (size, size))))
    else:
        e = resize_bilinear(median3(clahe(normalize_minmax(x3))), (size, size))
    return to_float(e)[:, None]


# ---------------------------------------------------------------- f4: ROI crop / paste of the GC wrapper
def roi_origin(frames: torch.Tensor, R: int = 224) -> torch.Tensor:
    """model_attention_aspp.py:20-27: (x0, y0) int32 [N, 2] of the R x R window of every fp32 frame."""
    x3, _ = _as3(frames, torch.float32)
    N, H, W = x3.shape
    sums = torch.empty(4 * N, dtype=torch.float64, device=frames.device)
    org = torch.empty(N, 2, dtype=torch.int32, device=frames.device)
    check(fn("aau_roi_origin")(x3.data_ptr(), sums.data_ptr(), org.data_ptr(), N, H, W, int(R), _stream()), "aau_roi_origin")
    return org


def roi_crop(frames: torch.Tensor, origin: torch.Tensor, R: int = 224) -> torch.Tensor:
    x3, _ = _as3(frames, torch.float32)
    N, H, W = x3.shape
    out = torch.empty(N, R, R, dtype=torch.float32, device=frames.device)
    check(fn("aau_roi_crop")(x3.data_ptr(), origin.data_ptr(), out.data_ptr(), N, H, W, int(R), _stream()), "aau_roi_crop")
    return out


def roi_paste_sigmoid(logits: torch.Tensor, origin: torch.Tensor, hw) -> torch.Tensor:
    l3, _ = _as3(logits, torch.float32)
    N, R, _ = l3.shape
    H, W = int(hw[0]), int(hw[1])
    out = torch.empty(N, H, W, dtype=torch.float32, device=logits.device)
    check(fn("aau_roi_paste_sigmoid")(l3.data_ptr(), origin.data_ptr(), out.data_ptr(), N, H, W, R, _stream()), "aau_roi_paste_sigmoid")
    return out


def frame_areas(prob: torch.Tensor, thr: float) -> torch.Tensor:
    p3, _ = _as3(prob, torch.float32)
    N, H, W = p3.shape
    areas = torch.empty(N, dtype=torch.int32, device=prob.device)
    check(fn("aau_frame_areas")(p3.data_ptr(), float(thr), areas.data_ptr(), N, H, W, _stream()), "aau_frame_areas")
    return areas
