"""CPU oracle: an ATen/fp32 restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``att-aspp-unet_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed CPU baseline.

Every definition cites the lines of the reference file
``attention_aspp_unet_pipeline_stage.py`` (abbreviated ``pipeline``) or
``eval_segmentation_batch.py`` (``evalseg``) that it restates.  The class
names, constructor signatures and child-attribute names are the checkpoint
schema (196 ``state_dict`` keys) and therefore have to coincide with the
reference; parameters are created in the same order so that the same
``torch.manual_seed`` gives bit-identical initial weights.

Pinned by ``oracle/make_golden.py``: the fixtures under ``tests/golden/`` were
produced by importing the reference itself in the build container and
``tests/test_oracle_golden.py`` checks this restatement against them.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Callable, Iterable, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# pipeline:29-31 -- module level constants that define "a train step"
SEED = 2025
IMG_SIZE = 512
WEIGHT_DECAY = 5e-4
GRAD_CLIP = 1.0
EARLY_STOP_PATIENCE = 15


def _conv_bn(in_c: int, out_c: int, k: int, *, dilation: int = 1, relu: bool = True) -> nn.Sequential:
    """bias-free conv -> BatchNorm2d (-> ReLU), "same" padding for odd k."""
    pad = dilation * (k // 2)
    layers = [nn.Conv2d(in_c, out_c, k, padding=pad, dilation=dilation, bias=False),
              nn.BatchNorm2d(out_c)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ConvBNReLU(nn.Module):
    """pipeline:59-65.  ``.block = [Conv2d(bias=False, pad=k//2), BatchNorm2d, ReLU]``."""

    def __init__(self, in_c, out_c, k=3):
        super().__init__()
        self.block = _conv_bn(in_c, out_c, k)

    def forward(self, x):
        return self.block(x)


class ASPP(nn.Module):
    """pipeline:67-83.  1x1 branch + one dilated 3x3 branch per rate + image-pool
    branch, concatenated in that order, then a 1x1 projection with Dropout(0.1)."""

    def __init__(self, in_c, out_c=256, rates=(6, 12, 18)):
        super().__init__()
        branches = [_conv_bn(in_c, out_c, 1)]
        branches += [_conv_bn(in_c, out_c, 3, dilation=r) for r in rates]
        self.blocks = nn.ModuleList(branches)
        self.pool = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                  nn.Conv2d(in_c, out_c, 1, bias=False),
                                  nn.BatchNorm2d(out_c),
                                  nn.ReLU(inplace=True))
        n_branch = len(rates) + 2
        self.project = nn.Sequential(nn.Conv2d(out_c * n_branch, out_c, 1, bias=False),
                                     nn.BatchNorm2d(out_c),
                                     nn.ReLU(inplace=True),
                                     nn.Dropout(0.1))

    def forward(self, x):
        size = x.shape[2:]
        feats = [branch(x) for branch in self.blocks]
        pooled = self.pool(x)
        feats.append(F.interpolate(pooled, size, mode="bilinear", align_corners=False))
        return self.project(torch.cat(feats, dim=1))


class AttentionGate(nn.Module):
    """pipeline:85-92.  ``x * sigmoid(BN(conv1x1(relu(BN(conv1x1(g)) + BN(conv1x1(x))))))``."""

    def __init__(self, Fg, Fl, Fint):
        super().__init__()
        self.Wg = _conv_bn(Fg, Fint, 1, relu=False)
        self.Wx = _conv_bn(Fl, Fint, 1, relu=False)
        self.psi = nn.Sequential(nn.Conv2d(Fint, 1, 1, bias=False), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)

    def forward(self, g, x):
        alpha = self.psi(self.relu(self.Wg(g) + self.Wx(x)))
        return x * alpha


class DummyAttention(nn.Module):
    """pipeline:95-96.  Identity on the skip tensor."""

    def forward(self, g, x):
        return x


class UpBlock(nn.Module):
    """pipeline:98-109.  ConvTranspose2d(2,2) with bias -> gate(g, skip) ->
    cat([gated skip, g]) -> two ConvBNReLU."""

    def __init__(self, in_c, out_c, use_att=True):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_c, out_c, 2, 2)
        self.att = AttentionGate(out_c, out_c, out_c // 2) if use_att else DummyAttention()
        self.conv = nn.Sequential(ConvBNReLU(in_c, out_c), ConvBNReLU(out_c, out_c))

    def forward(self, g, x):
        g = self.up(g)
        if g.shape[-2:] != x.shape[-2:]:
            g = F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False)
        x = self.att(g, x)
        return self.conv(torch.cat([x, g], dim=1))


class AttentionASPPUNet(nn.Module):
    """pipeline:111-127.  ``rates`` is a build-side extension; the
    default reproduces the reference, whose constructor does not expose it."""

    def __init__(self, in_channels=1, num_classes=1, base_c=32, rates=(6, 12, 18)):
        super().__init__()
        c = base_c
        widths = [c, 2 * c, 4 * c, 8 * c]
        prev = in_channels
        for i, w in enumerate(widths, start=1):
            setattr(self, f"d{i}", nn.Sequential(ConvBNReLU(prev, w), ConvBNReLU(w, w)))
            setattr(self, f"p{i}", nn.MaxPool2d(2))
            prev = w
        self.bridge = ASPP(8 * c, 16 * c, rates=rates)
        self.u4 = UpBlock(16 * c, 8 * c)
        self.u3 = UpBlock(8 * c, 4 * c)
        self.u2 = UpBlock(4 * c, 2 * c)
        self.u1 = UpBlock(2 * c, c, use_att=False)
        self.out_conv = nn.Conv2d(c, num_classes, 1)

    def forward(self, x):
        x1 = self.d1(x)
        x2 = self.d2(self.p1(x1))
        x3 = self.d3(self.p2(x2))
        x4 = self.d4(self.p3(x3))
        b = self.bridge(self.p4(x4))
        y = self.u4(b, x4)
        y = self.u3(y, x3)
        y = self.u2(y, x2)
        y = self.u1(y, x1)
        return self.out_conv(y)


def rename_legacy_keys(sd: dict) -> dict:
    """pipeline:134-141 -- the key rename of ``load_state_dict_compat``."""
    return {k.replace(".W_g.", ".Wg.").replace(".W_x.", ".Wx."): v for k, v in sd.items()}


# --------------------------------------------------------------------------
# losses and metrics
# --------------------------------------------------------------------------
class DiceLoss(nn.Module):
    """pipeline:173-178.  Soft Dice per sample over (H, W), mean over [B, C]."""

    def __init__(self, smooth=1.):
        super().__init__()
        self.s = smooth

    def forward(self, l, t):
        p = torch.sigmoid(l)
        inter = (p * t).sum((2, 3))
        dice = (2 * inter + self.s) / (p.sum((2, 3)) + t.sum((2, 3)) + self.s)
        return (1 - dice).mean()


class TverskyLoss(nn.Module):
    """pipeline:180-185 (dead in the reference: LOSS_TYPE == "combo")."""

    def __init__(self, a=0.7, b=0.3, s=1.):
        super().__init__()
        self.a, self.b, self.s = a, b, s

    def forward(self, l, t):
        p = torch.sigmoid(l)
        tp = (p * t).sum((2, 3))
        fp = (p * (1 - t)).sum((2, 3))
        fn = ((1 - p) * t).sum((2, 3))
        return (1 - (tp + self.s) / (tp + self.a * fp + self.b * fn + self.s)).mean()


class ComboLoss(nn.Module):
    """pipeline:187-189.  Dice + element-mean BCE-with-logits."""

    def __init__(self):
        super().__init__()
        self.d = DiceLoss()

    def forward(self, l, t):
        return self.d(l, t) + F.binary_cross_entropy_with_logits(l, t)


class EdgeLoss(nn.Module):
    """pipeline:196-216.  L1 between Sobel gradient magnitudes of sigmoid(l) and t."""

    def __init__(self):
        super().__init__()
        kx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]])
        ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]])
        self.register_buffer("kx", kx.view(1, 1, 3, 3))
        self.register_buffer("ky", ky.view(1, 1, 3, 3))

    def _mag(self, img):
        kx = self.kx.to(device=img.device, dtype=img.dtype)
        ky = self.ky.to(device=img.device, dtype=img.dtype)
        gx = F.conv2d(img, kx, padding=1)
        gy = F.conv2d(img, ky, padding=1)
        return torch.sqrt(gx ** 2 + gy ** 2 + 1e-8)

    def forward(self, logits, targets):
        p = torch.sigmoid(logits)
        return F.l1_loss(self._mag(p), self._mag(targets.to(p.dtype)))


def build_criterion(args, base, edge) -> Callable:
    """pipeline:219-232.  BCE over the whole batch (empty samples down-weighted
    in the finetune stage) + ``base`` (Dice + a second BCE) and ``edge_w * edge``
    on the positive samples only."""

    def crit(l, t):
        l, t = l.float(), t.float()
        B = t.size(0)
        is_empty = (t.sum((2, 3), keepdim=True) == 0).float()
        w = torch.ones_like(t)
        if args.stage == "finetune":
            w = torch.where(is_empty == 1, args.neg_bce_w, 1.)
        total = F.binary_cross_entropy_with_logits(l, t, weight=w)
        pos = (is_empty.view(B) == 0).nonzero(as_tuple=True)[0]
        if len(pos) > 0:
            total = total + base(l[pos], t[pos])
            if args.edge_w > 0:
                total = total + edge(l[pos], t[pos]) * args.edge_w
        return total

    return crit


def iou_score(l, t, thr=0.5):
    """pipeline:191-194.  Hard IoU per sample, eps 1e-7, mean over [B, C]."""
    p = (torch.sigmoid(l) > thr).float()
    inter = (p * t).sum((2, 3))
    union = p.sum((2, 3)) + t.sum((2, 3)) - inter
    return (inter / (union + 1e-7)).mean().item()


@torch.inference_mode()
def evaluate(model, loader, device):
    """pipeline:235-241.  Mean over batches of (1 - soft Dice loss, hard IoU)."""
    model.eval()
    d = i = 0.
    for x, y in loader:
        x, y = x.to(device), y.to(device)
        l = model(x)
        d += 1 - DiceLoss()(l, y).item()
        i += iou_score(l, y)
    return d / len(loader), i / len(loader)


def predict_prob_tta(model, x):
    """pipeline:336-338.  Average logits of x and its horizontal flip, sigmoid."""
    l = model(x)
    l_flip = torch.flip(model(torch.flip(x, [-1])), [-1])
    return torch.sigmoid((l + l_flip) / 2)[0, 0].cpu().numpy()


# evalseg:41-49 -- integer-count Dice / IoU on binarised masks
def _bin(a):
    return (np.asarray(a) > 0).astype(np.uint8)


def seg_dice(a, b, eps=1e-7):
    a, b = _bin(a), _bin(b)
    inter = int((a & b).sum())
    return (2 * inter + eps) / (int(a.sum()) + int(b.sum()) + eps)


def seg_iou(a, b, eps=1e-7):
    a, b = _bin(a), _bin(b)
    inter = int((a & b).sum())
    return (inter + eps) / (int(a.sum()) + int(b.sum()) - inter + eps)


# --------------------------------------------------------------------------
# the train step (pipeline:302-310 set-up, :316-325 inner loop) on CPU
# --------------------------------------------------------------------------
def default_args(**kw):
    """CLI defaults of pipeline:539-550 that the criterion / optimiser read."""
    base = dict(stage="main", edge_w=0.05, neg_bce_w=0.05, lr=3e-4, base_c=48,
                batch_size=8, epochs=120, seed=SEED)
    base.update(kw)
    return SimpleNamespace(**base)


def make_optimizer(model, lr=3e-4):
    """pipeline:302."""
    return torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=WEIGHT_DECAY)


def make_scheduler(opt, epochs, stage="main"):
    """pipeline:303-306.  Linear warm-up (factor 0.2 -> 1) then cosine, per epoch."""
    warm = 0 if stage == "finetune" else max(1, int(0.05 * epochs))
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=epochs - warm)
    if warm == 0:
        return cos
    lin = torch.optim.lr_scheduler.LinearLR(opt, start_factor=0.2, total_iters=warm)
    return torch.optim.lr_scheduler.SequentialLR(opt, [lin, cos], [warm])


def lr_at_epoch(ep: int, epochs: int, lr: float, stage: str = "main") -> float:
    """Closed form of the schedule above: learning rate in force during epoch
    ``ep`` (0-based), i.e. after ``ep`` calls of ``sch.step()``."""
    warm = 0 if stage == "finetune" else max(1, int(0.05 * epochs))
    if ep < warm:
        return lr * (0.2 + 0.8 * ep / warm)
    t, T = ep - warm, epochs - warm
    return lr * 0.5 * (1 + math.cos(math.pi * t / T))


def train_step(model, opt, crit, x, y) -> Tuple[float, float]:
    """pipeline:319-324 with AMP disabled (what the reference does without CUDA).
    Returns (loss, pre-clip global grad norm)."""
    opt.zero_grad(set_to_none=True)
    loss = crit(model(x), y)
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP)
    opt.step()
    return loss.item(), float(gnorm)


# --------------------------------------------------------------------------
# bf16-storage emulation (test infrastructure for the GPU parity tests)
# --------------------------------------------------------------------------
class _RoundSTE(torch.autograd.Function):
    """Round to bf16 in forward AND round the incoming gradient in backward: what happens
    when a tensor and its gradient are each stored once as bf16 in HBM."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def emulate_bf16_storage(net: nn.Module):
    """Make this fp32 CPU model round its tensors where the MI355X path stores bf16:
    conv / conv-transpose outputs (and their gradients), BN+ReLU outputs, pooled maps,
    gated skips, and the MFMA operands' weights (every conv weight except the first
    layer, out_conv and the gates' psi, which the HIP path keeps in fp32).  The random
    initial network is chaotic with respect to such rounding (ReLU / max-pool routing
    flips), so GPU gradient parity is asserted against THIS model; the plain fp32 model
    stays the reference for losses, logits and Dice."""
    handles = []
    round_out = lambda mod, inp, out: _RoundSTE.apply(out)
    for name, m in net.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            fp32_weights = m.in_channels % 8 != 0 or m.out_channels % 8 != 0
            if not fp32_weights:
                with torch.no_grad():
                    m.weight.copy_(m.weight.to(torch.bfloat16).to(torch.float32))
            if not (isinstance(m, nn.Conv2d) and m.out_channels % 8 != 0):
                handles.append(m.register_forward_hook(round_out))
        elif isinstance(m, (nn.ReLU, nn.MaxPool2d, nn.AdaptiveAvgPool2d)):
            handles.append(m.register_forward_hook(round_out))
        elif isinstance(m, nn.BatchNorm2d) and m.num_features > 1 and ".att." not in name:
            handles.append(m.register_forward_hook(round_out))
        elif isinstance(m, AttentionGate):
            handles.append(m.register_forward_hook(round_out))
    return handles
