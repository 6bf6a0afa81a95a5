"""Drop-in ``nn.Module`` surface of the reference model, executed by the HIP engine.

Class names, constructor signatures, child attribute names and therefore the 196
``state_dict`` keys are those of attention_aspp_unet_pipeline_stage.py:59-127
(``ConvBNReLU``, ``ASPP``, ``AttentionGate``, ``DummyAttention``, ``UpBlock``,
``AttentionASPPUNet``).  The child ``nn.Conv2d`` / ``nn.BatchNorm2d`` /
``nn.ConvTranspose2d`` objects are parameter containers only (created in the reference's
order, so the same ``torch.manual_seed`` gives bit-identical initial weights and a
reference checkpoint loads with ``strict=True``); their ATen ``forward`` is never
called.  ``AttentionASPPUNet.forward`` replays the engine's recorded launch list; the
backward pass is the engine's own, attached to autograd as a single node.  The block
classes also run standalone (forward only, ``blocks.py``) with the reference's call
signatures.

There is no CPU / ATen fallback: calling the model on a non-HIP tensor raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _abi
from .engine import Engine


def _conv_bn(in_c, out_c, k, dilation=1, relu=True):
    layers = [nn.Conv2d(in_c, out_c, k, padding=dilation * (k // 2), dilation=dilation, bias=False),
              nn.BatchNorm2d(out_c)]
    if relu:
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


@torch.no_grad()
def _fwd_cbr(self, x):
    """Standalone forward (NCHW fp32 in / out, no autograd); see blocks.py."""
    from . import blocks
    return blocks.convbnrelu_forward(self, x)


@torch.no_grad()
def _fwd_aspp(self, x):
    from . import blocks
    return blocks.aspp_forward(self, x)


@torch.no_grad()
def _fwd_gate(self, g, x):
    from . import blocks
    return blocks.gate_forward(self, g, x)


@torch.no_grad()
def _fwd_up(self, g, x):
    from . import blocks
    return blocks.upblock_forward(self, g, x)


class ConvBNReLU(nn.Module):
    """pipeline:59-65."""

    def __init__(self, in_c, out_c, k=3):
        super().__init__()
        self.block = _conv_bn(in_c, out_c, k)

    forward = _fwd_cbr


class ASPP(nn.Module):
    """pipeline:67-83 (``rates`` may have any length here; the reference supports exactly three)."""

    def __init__(self, in_c, out_c=256, rates=(6, 12, 18)):
        super().__init__()
        self.blocks = nn.ModuleList([_conv_bn(in_c, out_c, 1)] + [_conv_bn(in_c, out_c, 3, r) for r in rates])
        self.pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_c, out_c, 1, bias=False),
                                  nn.BatchNorm2d(out_c), nn.ReLU(True))
        self.project = nn.Sequential(nn.Conv2d(out_c * (len(rates) + 2), out_c, 1, bias=False),
                                     nn.BatchNorm2d(out_c), nn.ReLU(True), nn.Dropout(0.1))

    forward = _fwd_aspp


class AttentionGate(nn.Module):
    """pipeline:85-92."""

    def __init__(self, Fg, Fl, Fint):
        super().__init__()
        self.Wg = _conv_bn(Fg, Fint, 1, relu=False)
        self.Wx = _conv_bn(Fl, Fint, 1, relu=False)
        self.psi = nn.Sequential(nn.Conv2d(Fint, 1, 1, bias=False), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(True)

    forward = _fwd_gate


class DummyAttention(nn.Module):
    """pipeline:95-96."""

    def forward(self, g, x):
        return x


class UpBlock(nn.Module):
    """pipeline:98-109."""

    def __init__(self, in_c, out_c, use_att=True):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_c, out_c, 2, 2)
        self.att = AttentionGate(out_c, out_c, out_c // 2) if use_att else DummyAttention()
        self.conv = nn.Sequential(ConvBNReLU(in_c, out_c), ConvBNReLU(out_c, out_c))

    forward = _fwd_up


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are engine plans.

    A plan owns ONE set of activation buffers, so the node supports the reference loop's pattern (pipeline:320-322) --
    one training forward, then its backward -- and, like ``torch.autograd``, gradient ACCUMULATION over several such
    forward / backward pairs: when the parameters still hold gradients at the start of a backward (no
    ``zero_grad(set_to_none=True)`` in between), the flat gradient buffer is saved, rewritten by the backward and the
    saved values are added back (one 85-MB copy and add at base_c 48; nothing when the gradients were cleared).
    A second training forward of the same shape before the backward and a second backward of the same forward
    (``retain_graph``) are NOT supported and raise instead of returning gradients of the wrong activations."""

    @staticmethod
    def forward(ctx, x, trigger, plan):
        ctx.plan = plan
        out = plan.run_forward(x)
        ctx.gen = plan.fwd_gen
        return out.clone()

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        if ctx.gen != plan.fwd_gen:
            raise _abi.AauError("backward of a forward whose activations were overwritten by a later training forward "
                                "of the same shape (one plan = one set of activation buffers)")
        if plan.bwd_gen == ctx.gen:
            raise _abi.AauError("second backward through the same forward (retain_graph is not supported: the backward "
                                "pass reuses the plan's buffers; accumulate over separate forward / backward pairs)")
        plan.bwd_gen = ctx.gen
        st = plan.eng.store
        keep = st.gflat.clone() if any(p.grad is not None for p in st.params) else None
        dp = plan.eng.dp
        if dp is not None:
            # data parallel: the bucket all-reduces run asynchronously IN PLACE on the flat buffer while the backward
            # goes on, so the kept (already reduced) gradient must not touch it before they are done --
            # DataParallel.finish() adds it behind the last of them
            dp.before_backward(keep)
        plan.run_backward(dlogits.contiguous())
        if keep is not None and dp is None:
            st.gflat.add_(keep)
        return None, None, None


class AttentionASPPUNet(nn.Module):
    """pipeline:111-127.  ``forward(x: [B,1,H,W] fp32) -> logits [B,1,H,W] fp32``; H, W multiples of 16."""

    def __init__(self, in_channels=1, num_classes=1, base_c=32, rates=(6, 12, 18)):
        super().__init__()
        if in_channels != 1 or num_classes != 1:
            raise _abi.AauError("the HIP path implements the reference configuration in_channels=1, num_classes=1")
        if base_c % 8 != 0:
            raise _abi.AauError("base_c must be a multiple of 8 (16-byte channel vectors)")
        c = base_c
        prev = in_channels
        for i, w in enumerate((c, 2 * c, 4 * c, 8 * c), start=1):
            setattr(self, f"d{i}", nn.Sequential(ConvBNReLU(prev, w), ConvBNReLU(w, w)))
            setattr(self, f"p{i}", nn.MaxPool2d(2))
            prev = w
        self.bridge = ASPP(8 * c, 16 * c, rates=rates)
        self.u4 = UpBlock(16 * c, 8 * c)
        self.u3 = UpBlock(8 * c, 4 * c)
        self.u2 = UpBlock(4 * c, 2 * c)
        self.u1 = UpBlock(2 * c, c, use_att=False)
        self.out_conv = nn.Conv2d(c, num_classes, 1)
        self.base_c = base_c
        object.__setattr__(self, "_engine", Engine(self))
        object.__setattr__(self, "_trigger", None)

    @property
    def engine(self) -> Engine:
        return self._engine

    def set_precision(self, kind: str):
        """16-bit storage type of the activations: "bf16" (default) or "fp16" -- IEEE half for inference, what the
        reference's ``torch.cuda.amp.autocast`` gives its GPU predict path (pipeline:320,437).  Returns self."""
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
            return _NetFn.apply(x, self._trigger, plan)
        return plan.run_forward(x).clone()
