"""Criterion and metrics of the reference (attention_aspp_unet_pipeline_stage.py:173-232),
computed by the fused HIP criterion kernels (csrc/loss.hip).

``build_criterion(args, base, edge)`` keeps the reference signature and returns
``crit(logits, targets) -> scalar``; forward value and d/dlogits come from two launches
with no host synchronisation (the reference's ``nonzero``/index selection of positive
samples is restated with a per-sample mask, see csrc/loss.hip).  The loss classes used on
their own (``DiceLoss`` / ``TverskyLoss`` / ``ComboLoss`` / ``EdgeLoss``) are autograd nodes
over ``aau_loss_terms`` (same per-sample sums, every sample counted).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _abi, ops


def _check(l, t):
    if l.device.type != "cuda":
        raise _abi.AauError("criterion kernels need HIP tensors (no CPU fallback)")
    if l.dim() != 4 or l.shape[1] != 1 or t.shape != l.shape:
        raise _abi.AauError(f"expected logits/targets [B,1,H,W], got {tuple(l.shape)} / {tuple(t.shape)}")
    return l.float().contiguous(), t.float().contiguous()


def sample_sums(l, t, edge=True):
    """Per-sample sums [B,8]: sum t, sum p, sum p*t, sum bce, sum |grad p - grad t|, sum pbin, sum pbin*t."""
    l, t = _check(l, t)
    B, _, H, W = l.shape
    sums = torch.empty(32, B, 8, device=l.device)
    out = torch.empty(4, device=l.device)
    ops.criterion(l, t, sums, out, None, B, H, W, False, 1.0, 1.0 if edge else 0.0)
    return sums[0]


class _CritFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, t, finetune, neg_bce_w, edge_w):
        B, _, H, W = l.shape
        sums = torch.empty(32, B, 8, device=l.device)
        out = torch.empty(4, device=l.device)
        dl = torch.empty_like(l)
        ops.criterion(l, t, sums, out, dl, B, H, W, finetune, neg_bce_w, edge_w)
        ctx.save_for_backward(dl)
        ctx.parts = out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None


class _TermsFn(torch.autograd.Function):
    """One of the loss classes on its own: value and d/dlogits from aau_loss_terms (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, l, t, coef9):
        B, _, H, W = l.shape
        sums = torch.empty(32, B, 8, device=l.device)
        out = torch.empty(4, device=l.device)
        dl = torch.empty_like(l) if ctx.needs_input_grad[0] else None
        ops.loss_terms(l, t, sums, out, dl, B, H, W, coef9)
        if dl is not None:
            ctx.save_for_backward(dl)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def _terms(l, t, coef9):
    l, t = _check(l, t)
    return _TermsFn.apply(l, t, tuple(coef9))


class DiceLoss(nn.Module):
    """pipeline:173-178: mean over samples of 1 - (2*sum(p*t) + s)/(sum p + sum t + s); differentiable."""

    def __init__(self, smooth=1.):
        super().__init__()
        self.s = smooth

    def forward(self, l, t):
        return _terms(l, t, (1., 2., self.s, 0., 1., 1., self.s, 0., 0.))


class TverskyLoss(nn.Module):
    """pipeline:180-185 (dead code in the reference: LOSS_TYPE is "combo"); differentiable."""

    def __init__(self, a=0.7, b=0.3, s=1.):
        super().__init__()
        self.a, self.b, self.s = a, b, s

    def forward(self, l, t):
        return _terms(l, t, (1., 1., self.s, 1. - self.a - self.b, self.a, self.b, self.s, 0., 0.))


class ComboLoss(nn.Module):
    """pipeline:187-189: DiceLoss + BCEWithLogits (mean over all elements); differentiable."""

    def __init__(self):
        super().__init__()
        self.d = DiceLoss()

    def forward(self, l, t):
        return _terms(l, t, (1., 2., self.d.s, 0., 1., 1., self.d.s, 1., 0.))


class EdgeLoss(nn.Module):
    """pipeline:196-216: L1 distance of the Sobel magnitudes of sigmoid(logits) and targets; differentiable.
    Keeps the kx / ky buffers of the reference (they are part of its state_dict)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("kx", torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]).view(1, 1, 3, 3))
        self.register_buffer("ky", torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]).view(1, 1, 3, 3))

    def forward(self, logits, targets):
        return _terms(logits, targets, (0., 1., 1., 0., 1., 1., 1., 0., 1.))


def build_criterion(args, base, edge):
    """pipeline:219-232.  ``base`` must be a ComboLoss and ``edge`` an EdgeLoss (the only
    combination the reference builds, pipeline:308-309)."""
    if not isinstance(base, ComboLoss) or not isinstance(edge, EdgeLoss):
        raise _abi.AauError("the fused criterion implements ComboLoss + EdgeLoss (LOSS_TYPE == 'combo')")
    finetune = args.stage == "finetune"
    neg_w, edge_w = float(args.neg_bce_w), float(args.edge_w)

    def crit(l, t):
        l, t = _check(l, t)
        return _CritFn.apply(l, t, finetune, neg_w, max(edge_w, 0.0))

    return crit


def iou_score(l, t, thr=0.5):
    """pipeline:191-194."""
    return seg_metrics(l, t, thr)[1].item()


def seg_metrics(l, t, thr=0.5):
    """-> device tensor [2]: (mean soft Dice = 1 - DiceLoss, mean hard IoU); no host sync."""
    l, t = _check(l, t)
    B, _, H, W = l.shape
    sums = torch.empty(32, B, 8, device=l.device)
    out = torch.empty(2, device=l.device)
    ops.seg_metrics(l, t, sums, out, B, H, W, thr)
    return out
