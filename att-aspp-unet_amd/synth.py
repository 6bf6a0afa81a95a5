"""Synthetic fetal-abdomen ultrasound frames (host side, deterministic).

There is no dataset on the build or GPU machines, so tests and ``bench.py``
feed the path with phantoms of the shape the reference loader produces
(``attention_aspp_unet_pipeline_stage.py:143-170``): ``x`` is ``[B,1,H,W]`` fp32
in [0,1] quantised to k/255 (``ToFloat(max_value=255)``, ``:154``) and ``y`` is a
``[B,1,H,W]`` {0,1} mask (``:170``).  A frame is a dark sector-scan background,
a bright-rimmed ellipse with a darker interior and multiplicative Rayleigh-like
speckle; a fraction of the frames (default 20 %, README.md:18) is negative, i.e.
has no ellipse and an empty mask.
"""
from __future__ import annotations

import math

import torch


def make_frames(batch: int, size: int = 512, *, seed: int = 2025, neg_frac: float = 0.2,
                force_pattern: str | None = None):
    """Return ``(x, y)`` CPU tensors ``[batch,1,size,size]`` fp32.

    ``force_pattern``: optional string of 'p'/'n' per sample overriding the random
    positive/negative draw (e.g. ``"pn"``, ``"nn"`` for the all-negative case).
    """
    g = torch.Generator().manual_seed(seed)
    H = W = size
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                            torch.arange(W, dtype=torch.float32), indexing="ij")
    xs = torch.empty(batch, 1, H, W)
    ys = torch.zeros(batch, 1, H, W)
    # sector ("fan") geometry: apex above the image, 70 degree opening
    ax, ay = W / 2.0, -0.15 * H
    rad = torch.sqrt((xx - ax) ** 2 + (yy - ay) ** 2)
    ang = torch.atan2(xx - ax, yy - ay)
    fan = ((ang.abs() < math.radians(38)) & (rad > 0.2 * H) & (rad < 1.12 * H)).float()
    for b in range(batch):
        u = torch.rand(8, generator=g)
        if force_pattern is not None:
            positive = force_pattern[b % len(force_pattern)] == "p"
        else:
            positive = bool(u[0] >= neg_frac)
        tissue = 0.18 + 0.10 * torch.sin(rad / (0.06 * H) + 6.28 * u[1]) * torch.cos(ang * 9 + u[2])
        img = fan * tissue
        if positive:
            cx = (0.35 + 0.30 * u[3]) * W
            cy = (0.40 + 0.25 * u[4]) * H
            a = (0.14 + 0.12 * u[5]) * W
            bax = a * (0.65 + 0.30 * u[6])
            th = math.pi * float(u[7])
            ct, st = math.cos(th), math.sin(th)
            xr = (xx - cx) * ct + (yy - cy) * st
            yr = -(xx - cx) * st + (yy - cy) * ct
            r = torch.sqrt((xr / a) ** 2 + (yr / bax) ** 2)
            inside = (r <= 1.0).float()
            rim = torch.exp(-((r - 1.0) / 0.06) ** 2)
            img = img * (1 - 0.45 * inside) + 0.55 * rim * fan
            ys[b, 0] = inside * fan
        e1 = torch.randn(H, W, generator=g)
        e2 = torch.randn(H, W, generator=g)
        speckle = torch.sqrt(e1 ** 2 + e2 ** 2) / 1.2533  # Rayleigh, unit mean
        img = (img * (0.55 + 0.45 * speckle)).clamp_(0, 1)
        xs[b, 0] = torch.round(img * 255.0) / 255.0
    return xs, ys
