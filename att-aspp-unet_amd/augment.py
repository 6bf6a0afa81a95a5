"""Random training augmentations of ``FetalACDataset`` (attention_aspp_unet_pipeline_stage.py:149-153) on the GPU.

The reference's train transform (albumentations ``Compose``) after ``Resize(512, 512)``:

    HorizontalFlip(0.5)
    Affine(scale=(0.92, 1.08), rotate=(-7, 7), translate_percent=(0, 0.02), shear=0, p=0.7)
    RandomGamma(gamma_limit=(80, 120), p=0.3)
    RandomBrightnessContrast(brightness_limit=0.1, contrast_limit=0.1, p=0.3)
    ElasticTransform(8, 3, p=0.25)                  # alpha 8, sigma 3
    CLAHE(1.0, (8, 8))  MedianBlur(3)               # albumentations' default p = 0.5 EACH (also in the val transform)
    ToFloat(max_value=255)

Here the PARAMETERS of every frame are drawn on the host by a counter-based sampler (numpy Philox keyed by
``(seed, epoch)``, counter = frame index: the draw of a frame does not depend on batch composition, worker count or
order of execution) and the PIXEL WORK runs as batched HIP kernels over ``[N, H, W]`` uint8 frames (csrc/augment.hip):
a frame whose transform was not drawn gets the identity parameters, so every step is one launch per batch.
Masks follow the geometric steps (flip, affine, elastic) with nearest interpolation, as albumentations does.

PARITY UNPINNED: albumentations / cv2 are not importable in the build container and the reference pins no version.
What is restated (oracle/augment_ref.py is the bit-exact checker of the kernels):
  * Affine: skimage-style matrix about the centre (W/2 - 0.5, H/2 - 0.5), independent x / y scale and translation,
    cv2.warpAffine semantics (dst -> src through the inverse, INTER_LINEAR / INTER_NEAREST, BORDER_CONSTANT 0) with exact
    bilinear weights instead of cv2's 1/32-pixel fixed point;
  * RandomGamma / RandomBrightnessContrast: the uint8 look-up tables exactly as albumentations builds them (float64 power,
    float32 ramp, ``astype(uint8)`` truncation, ``brightness_by_max``), composed into one table per frame;
  * ElasticTransform: uniform(-1, 1) noise -> GaussianBlur(sigma, ksize = round(8 sigma + 1) | 1, cv2.getGaussianKernel,
    BORDER_REFLECT_101) x alpha -> cv2.remap(BORDER_REFLECT_101); no affine part (``alpha_affine`` is gone in the
    albumentations versions that also deprecate ``always_apply``, which the reference's SafeCLAHE works around);
  * albumentations' own RNG stream is NOT reproduced (it is a function of its version and of Python's ``random``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _abi, imgproc
from ._abi import check, fn

# pipeline:149-153
P_FLIP, P_AFFINE, P_GAMMA, P_BC, P_ELASTIC, P_CLAHE, P_MEDIAN = 0.5, 0.7, 0.3, 0.3, 0.25, 0.5, 0.5
SCALE_LIM, ROT_LIM, TRANS_LIM = (0.92, 1.08), (-7.0, 7.0), (0.0, 0.02)
GAMMA_LIM, BRIGHT_LIM, CONTRAST_LIM = (80.0, 120.0), 0.1, 0.1
ELASTIC_ALPHA, ELASTIC_SIGMA = 8.0, 3.0
NDRAW = 16          # uniforms reserved per frame (a fixed budget keeps the stream of a frame independent of what was drawn)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class FrameParams:
    """Parameters of a batch, host side (numpy).  ``identity`` entries mean "transform not drawn"."""
    flip: np.ndarray            # uint8 [N]
    inv_mats: np.ndarray        # float64 [N, 6]: dst -> src maps of cv2.warpAffine
    luts: np.ndarray            # uint8 [N, 256]
    elastic_alpha: np.ndarray   # float32 [N]: 0 = not drawn
    elastic_seed: np.ndarray    # uint64 [N]
    clahe: np.ndarray           # uint8 [N]
    median: np.ndarray          # uint8 [N]


def affine_matrix(sx, sy, rot_deg, tx, ty, H, W) -> np.ndarray:
    """Forward 3x3 matrix: to-centre^-1 . translate . rotate-scale . to-centre (albumentations Affine with shear 0)."""
    cx, cy = W / 2.0 - 0.5, H / 2.0 - 0.5
    r = math.radians(rot_deg)
    A = np.array([[sx * math.cos(r), -sy * math.sin(r), tx], [sx * math.sin(r), sy * math.cos(r), ty], [0, 0, 1.0]])
    T0 = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    T1 = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
    return T1 @ A @ T0


def gamma_lut(gamma: float) -> np.ndarray:
    """albumentations gamma_transform for uint8."""
    table = (np.arange(0, 256.0 / 255, 1.0 / 255) ** gamma) * 255
    return table.astype(np.uint8)


def brightness_contrast_lut(alpha: float, beta: float) -> np.ndarray:
    """albumentations _brightness_contrast_adjust_uint with beta_by_max=True."""
    lut = np.arange(0, 256, dtype=np.float32)
    if alpha != 1:
        lut *= np.float32(alpha)
    if beta != 0:
        lut += np.float32(beta * 255.0)
    return np.clip(lut, 0, 255).astype(np.uint8)


def gaussian_taps(sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma) as float32 with cv2's automatic ksize for float images."""
    ksize = int(round(sigma * 4 * 2 + 1)) | 1
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def sample(indices, H: int, W: int, seed: int, epoch: int = 0, train: bool = True) -> FrameParams:
    """Draw the parameters of the frames ``indices`` (dataset indices).  ``train=False``: the validation transform, in which
    only CLAHE and MedianBlur are random (pipeline:155)."""
    n = len(indices)
    flip = np.zeros(n, np.uint8)
    inv = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float64), (n, 1))
    luts = np.tile(np.arange(256, dtype=np.uint8), (n, 1))
    ealpha = np.zeros(n, np.float32)
    eseed = np.zeros(n, np.uint64)
    clahe = np.zeros(n, np.uint8)
    median = np.zeros(n, np.uint8)
    for k, idx in enumerate(indices):
        # one Philox stream per (seed, epoch, frame): counter-based, so any worker can draw any frame
        bg = np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, int(epoch) & 0xFFFFFFFFFFFFFFFF], counter=[int(idx), 0, 0, 0])
        u = np.random.Generator(bg).random(NDRAW)
        uni = lambda j, lo, hi: lo + (hi - lo) * float(u[j])
        clahe[k], median[k] = u[14] < P_CLAHE, u[15] < P_MEDIAN
        if not train:
            continue
        flip[k] = u[0] < P_FLIP
        if u[1] < P_AFFINE:
            M = affine_matrix(uni(2, *SCALE_LIM), uni(3, *SCALE_LIM), uni(4, *ROT_LIM), uni(5, *TRANS_LIM) * W,
                              uni(6, *TRANS_LIM) * H, H, W)
            inv[k] = np.linalg.inv(M)[:2].reshape(6)
        lut = np.arange(256, dtype=np.uint8)
        if u[7] < P_GAMMA:
            lut = gamma_lut(uni(8, *GAMMA_LIM) / 100.0)[lut]
        if u[9] < P_BC:
            lut = brightness_contrast_lut(1.0 + uni(10, -CONTRAST_LIM, CONTRAST_LIM), uni(11, -BRIGHT_LIM, BRIGHT_LIM))[lut]
        luts[k] = lut
        if u[12] < P_ELASTIC:
            ealpha[k] = ELASTIC_ALPHA
            eseed[k] = np.uint64(int(u[13] * 2.0 ** 53)) ^ np.uint64((int(seed) * 0x9E3779B97F4A7C15 + int(idx)) & 0xFFFFFFFFFFFFFFFF)
    return FrameParams(flip, inv, luts, ealpha, eseed, clahe, median)


# ---------------------------------------------------------------- kernels (batched, [N, H, W] uint8 on the device)
def _u8(x):
    if not isinstance(x, torch.Tensor) or x.device.type != "cuda" or x.dtype != torch.uint8 or x.dim() != 3:
        raise _abi.AauError("augment: expected a CUDA uint8 tensor [N, H, W]")
    return x.contiguous()


def _dev(a: np.ndarray, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def hflip_frames(x, flags):
    x = _u8(x); N, H, W = x.shape
    out = torch.empty_like(x)
    check(fn("aau_hflip_frames_u8")(x.data_ptr(), out.data_ptr(), flags.data_ptr(), N, H, W, _stream()), "aau_hflip_frames_u8")
    return out


def warp_affine(x, inv_mats, nearest=False, border=0):
    x = _u8(x); N, H, W = x.shape
    out = torch.empty_like(x)
    check(fn("aau_warp_affine_u8")(x.data_ptr(), out.data_ptr(), inv_mats.data_ptr(), N, H, W, int(nearest), int(border), _stream()),
          "aau_warp_affine_u8")
    return out


def apply_lut(x, luts):
    x = _u8(x); N, H, W = x.shape
    out = torch.empty_like(x)
    check(fn("aau_lut_u8")(x.data_ptr(), out.data_ptr(), luts.data_ptr(), N, H * W, _stream()), "aau_lut_u8")
    return out


def elastic_fields(seeds, N, H, W, sigma=ELASTIC_SIGMA):
    """-> blurred displacement fields fp32 [N, 2, H, W] (unit amplitude; remap multiplies by the frame's alpha)."""
    noise = torch.empty(N, 2, H, W, dtype=torch.float32, device=seeds.device)
    check(fn("aau_elastic_noise")(seeds.data_ptr(), noise.data_ptr(), N, H, W, _stream()), "aau_elastic_noise")
    taps = _dev(gaussian_taps(sigma), seeds.device)
    out, tmp = torch.empty_like(noise), torch.empty_like(noise)
    check(fn("aau_gauss_sep_f32")(noise.data_ptr(), out.data_ptr(), tmp.data_ptr(), taps.data_ptr(), taps.numel(), 2 * N, H, W,
                                  _stream()), "aau_gauss_sep_f32")
    return out


def remap(x, disp, alpha, nearest=False):
    x = _u8(x); N, H, W = x.shape
    out = torch.empty_like(x)
    check(fn("aau_remap_u8")(x.data_ptr(), out.data_ptr(), disp.data_ptr(), alpha.data_ptr(), N, H, W, int(nearest), _stream()),
          "aau_remap_u8")
    return out


def select_frames(a, b, flags):
    a, b = _u8(a), _u8(b); N, H, W = a.shape
    out = torch.empty_like(a)
    check(fn("aau_select_frames_u8")(a.data_ptr(), b.data_ptr(), flags.data_ptr(), out.data_ptr(), N, H * W, _stream()),
          "aau_select_frames_u8")
    return out


def apply(img_u8: torch.Tensor, msk_u8, p: FrameParams, train: bool = True):
    """The transform of pipeline:149-155 after Resize on a batch: ``img_u8`` / ``msk_u8`` uint8 [N, S, S] on the device
    (``msk_u8`` may be None) -> (x fp32 [N,1,S,S] in [0,1], y fp32 [N,1,S,S] in {0,1} or None)."""
    img = _u8(img_u8)
    N, H, W = img.shape
    dev = img.device
    msk = _u8(msk_u8) if msk_u8 is not None else None
    if train:
        flags = _dev(p.flip, dev)
        img = hflip_frames(img, flags)
        inv = _dev(p.inv_mats, dev)
        img = warp_affine(img, inv, nearest=False)
        if msk is not None:
            msk = warp_affine(hflip_frames(msk, flags), inv, nearest=True)
        img = apply_lut(img, _dev(p.luts, dev))
        if p.elastic_alpha.any():
            disp = elastic_fields(_dev(p.elastic_seed.view(np.int64), dev), N, H, W)
            alpha = _dev(p.elastic_alpha, dev)
            img = remap(img, disp, alpha, nearest=False)
            if msk is not None:
                msk = remap(msk, disp, alpha, nearest=True)
    img = select_frames(imgproc.clahe(img, 1.0, 8), img, _dev(p.clahe, dev))
    img = select_frames(imgproc.median3(img), img, _dev(p.median, dev))
    x = imgproc.to_float(img).view(N, 1, H, W)
    y = None if msk is None else (msk.float() / 255.0).view(N, 1, H, W)
    return x, y
