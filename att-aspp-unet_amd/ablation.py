"""The ablation variant of the network (test_ablation.py:73-218) on the same HIP engine.

What differs from attention_aspp_unet_pipeline_stage.py:
  * ``AttentionGate(Fg, Fl, Fint=None)`` (:128-143): no BatchNorm, ``Fint = max(8, min(Fg, Fl) // 4)``, ``psi`` is
    ReLU -> Conv2d(Fint, 1, 1, bias=True) -> Sigmoid, and the gate is RESIDUAL: returns ``(x * a + x, a)``;
  * ``AttentionASPPUNet(in_channels, num_classes, base_c, use_att, use_aspp, att_depth)`` (:168-202): the bridge is the
    ASPP or, with ``use_aspp=False``, ``Sequential(ConvBNReLU(8c, 16c, 3), Dropout(0.1))``; gates only in ``u4`` (if
    ``att_depth >= 4``) and ``u3`` (``>= 3``), never in ``u2`` / ``u1``; with everything off it is the plain U-Net
    (BASELINE config 1);
  * ``forward`` returns ``(logits, [psi3, psi2])`` (:204-218); ``DummyAttention`` yields ``zeros(1,1,1,1)`` for psi;
  * the training script gives the attention parameters twice the backbone's learning rate (:576-586):
    ``param_groups(model, lr)`` builds those groups for ``FusedAdamW(model, groups=...)``.
The state_dict key schema is the ablation file's (e.g. ``u4.att.Wg.weight``, ``u4.att.psi.1.bias``), so its checkpoints
load with ``strict=True``.  The children are parameter containers; the engine executes the graph (engine.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _abi
from .engine import Engine
from .model import ASPP, ConvBNReLU, _NetFn


class AttentionGate(nn.Module):
    """test_ablation.py:128-143."""

    def __init__(self, Fg, Fl, Fint=None):
        super().__init__()
        if Fint is None:
            Fint = max(8, min(Fg, Fl) // 4)
        self.Wg = nn.Conv2d(Fg, Fint, 1, bias=False)
        self.Wx = nn.Conv2d(Fl, Fint, 1, bias=False)
        self.psi = nn.Sequential(nn.ReLU(True), nn.Conv2d(Fint, 1, 1, bias=True), nn.Sigmoid())


class DummyAttention(nn.Module):
    """test_ablation.py:145-147."""

    def forward(self, g, x):
        return x, torch.zeros(1, 1, 1, 1, device=x.device)


class UpBlock(nn.Module):
    """test_ablation.py:149-166."""

    def __init__(self, in_c, out_c, use_att=True):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_c, out_c, 2, 2)
        self.att = AttentionGate(out_c, out_c) if use_att else DummyAttention()
        self.conv = nn.Sequential(ConvBNReLU(in_c, out_c), ConvBNReLU(out_c, out_c))


class AttentionASPPUNet(nn.Module):
    """test_ablation.py:168-218.  ``forward(x) -> (logits [B,1,H,W], [psi3, psi2])``."""

    def __init__(self, in_channels=1, num_classes=1, base_c=32, use_att=True, use_aspp=True, att_depth=4):
        super().__init__()
        if in_channels != 1 or num_classes != 1:
            raise _abi.AauError("the HIP path implements in_channels=1, num_classes=1")
        if base_c % 8 != 0:
            raise _abi.AauError("base_c must be a multiple of 8 (16-byte channel vectors)")
        c = base_c
        prev = in_channels
        for i, w in enumerate((c, 2 * c, 4 * c, 8 * c), start=1):
            setattr(self, f"d{i}", nn.Sequential(ConvBNReLU(prev, w), ConvBNReLU(w, w)))
            setattr(self, f"p{i}", nn.MaxPool2d(2))
            prev = w
        self.bridge = ASPP(8 * c, 16 * c) if use_aspp else nn.Sequential(ConvBNReLU(8 * c, 16 * c, 3), nn.Dropout(0.1))
        self.u4 = UpBlock(16 * c, 8 * c, use_att and att_depth >= 4)
        self.u3 = UpBlock(8 * c, 4 * c, use_att and att_depth >= 3)
        self.u2 = UpBlock(4 * c, 2 * c, False)
        self.u1 = UpBlock(2 * c, c, False)
        self.out_conv = nn.Conv2d(c, num_classes, 1)
        self.base_c = base_c
        object.__setattr__(self, "_engine", Engine(self))
        object.__setattr__(self, "_trigger", None)

    @property
    def engine(self) -> Engine:
        return self._engine

    def set_precision(self, kind: str):
        """"bf16" (default) or "fp16" (IEEE half, inference only); see model.AttentionASPPUNet.set_precision."""
        self._engine.set_precision(kind)
        return self

    def _plan_for(self, x):
        if x.dim() != 4 or x.shape[1] != 1:
            raise _abi.AauError(f"expected input [B,1,H,W], got {tuple(x.shape)}")
        self._engine.ensure(x.device)
        return self._engine.plan(x.shape[0], x.shape[2], x.shape[3], self.training)

    def forward(self, x):
        plan = self._plan_for(x)
        x = x.float().contiguous()
        if self.training and torch.is_grad_enabled():
            if self._trigger is None or self._trigger.device != x.device:
                object.__setattr__(self, "_trigger", torch.zeros(1, device=x.device, requires_grad=True))
            logits = _NetFn.apply(x, self._trigger, plan)
        else:
            logits = plan.run_forward(x).clone()
        return logits, plan.psi_outputs()


def param_groups(model, lr):
    """test_ablation.py:576-586: attention parameters at ``lr``, everything else at ``lr / 2``."""
    att, bk = [], []
    for n, p in model.named_parameters():
        (att if ".att." in n or ".psi" in n else bk).append(p)
    groups = [{"params": bk, "lr": lr * 0.5}]
    if att:
        groups.append({"params": att, "lr": lr})
    return groups
