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


def build_criterion(base, edge, edge_w: float, args):
    """test_ablation.py:346-361 (the ablation script's argument order; ``edge_w = 0`` is its ``--no_edge_loss``)."""
    from argparse import Namespace
    from .losses import build_criterion as _bc
    return _bc(Namespace(stage=args.stage, neg_bce_w=args.neg_bce_w, edge_w=float(edge_w)), base, edge)


def train(args, train_loader=None, val_loader=None):
    """test_ablation.py:540-670: the ablation family's training loop -- model flags ``no_att`` / ``no_aspp`` /
    ``att_depth``, attention parameters at twice the backbone's learning rate (:576-586), warm-up + cosine schedule,
    ``--no_edge_loss``, per-epoch train / validation loss, Dice and IoU written to ``metrics.csv`` (:605-609), best
    checkpoint by validation Dice, early stopping.  bf16 activations with fp32 master weights replace fp16 autocast +
    GradScaler; the loaders are any iterables of ``(x, y)`` batches (``--train_dir`` directories through ``dataset.py``,
    synthetic phantoms with ``--synthetic_batches N``)."""
    import csv
    from datetime import datetime
    from pathlib import Path
    from .losses import ComboLoss, DiceLoss, EdgeLoss, seg_metrics
    from .optim import FusedAdamW
    from .pipeline import (EARLY_STOP_PATIENCE, GRAD_CLIP, IMG_SIZE, WEIGHT_DECAY, SyntheticLoader, load_state_dict_compat,
                           lr_at_epoch, set_seed)
    set_seed(args.seed)
    device = torch.device("cuda", torch.cuda.current_device())
    if train_loader is None:
        n = int(getattr(args, "synthetic_batches", 0) or 0)
        size = int(getattr(args, "img_size", IMG_SIZE))
        if n > 0:
            train_loader = SyntheticLoader(n, args.batch_size, size, args.seed, device)
            val_loader = SyntheticLoader(max(1, n // 10), args.batch_size, size, args.seed + 100000, device, neg_frac=0.0)
        elif getattr(args, "train_dir", None):
            from . import dataset            # the same images/ + masks/ directories as the main script
            train_loader, val_loader = dataset.loaders_from_args(args, device)
        else:
            raise RuntimeError("train needs --train_dir (images/ + masks/), --synthetic_batches N, or loaders passed in")
    model = AttentionASPPUNet(base_c=args.base_c, use_att=not getattr(args, "no_att", False),
                              use_aspp=not getattr(args, "no_aspp", False), att_depth=getattr(args, "att_depth", 4)).to(device)
    if args.stage == "finetune":
        load_state_dict_compat(model, args.pretrained)
    opt = FusedAdamW(model, groups=param_groups(model, args.lr), weight_decay=WEIGHT_DECAY, max_grad_norm=GRAD_CLIP)
    base_lrs = [g["lr"] for g in opt.param_groups]
    crit = build_criterion(ComboLoss(), EdgeLoss(), 0.0 if getattr(args, "no_edge_loss", False) else args.edge_w, args)
    out_dir = Path(args.output_dir) / ("ckpt_main" if args.stage == "main" else "ckpt_finetune")
    out_dir.mkdir(parents=True, exist_ok=True)
    best, noimp = 0.0, 0
    best_p = out_dir / f"best_{datetime.now():%Y%m%d-%H%M%S}.pt"
    history = []
    with open(out_dir / "metrics.csv", "w", newline="") as mfp:
        writer = csv.writer(mfp)
        writer.writerow(["epoch", "train_loss", "val_loss", "train_dice", "val_dice", "train_iou", "val_iou"])
        for ep in range(1, args.epochs + 1):
            for g, lr0 in zip(opt.param_groups, base_lrs):        # SequentialLR(LinearLR, CosineAnnealingLR), per group
                g["lr"] = lr_at_epoch(ep - 1, args.epochs, lr0, args.stage)
            model.train()
            acc = torch.zeros(3, device=device)
            nb = 0
            for x, y in train_loader:
                x, y = x.to(device), y.to(device)
                opt.zero_grad(set_to_none=True)
                logits, _ = model(x)
                loss = crit(logits, y)
                loss.backward()
                opt.step()
                with torch.no_grad():
                    acc[0] += loss.detach()
                    acc[1:] += seg_metrics(logits.detach(), y)   # (1 - DiceLoss, iou_score), no host sync
                nb += 1
            model.eval()
            vacc = torch.zeros(3, device=device)
            nv = 0
            with torch.no_grad():
                for x, y in val_loader:
                    x, y = x.to(device), y.to(device)
                    logits, _ = model(x)
                    vacc[0] += crit(logits, y)
                    vacc[1:] += seg_metrics(logits, y)
                    nv += 1
            tr = (acc / max(1, nb)).tolist()
            va = (vacc / max(1, nv)).tolist()
            print(f"Dice {va[1]:.4f} | IoU {va[2]:.4f} | Loss {va[0]:.4f}")
            writer.writerow([ep, f"{tr[0]:.6f}", f"{va[0]:.6f}", f"{tr[1]:.6f}", f"{va[1]:.6f}", f"{tr[2]:.6f}", f"{va[2]:.6f}"])
            mfp.flush()
            history.append((tr[0], va[0], tr[1], va[1], tr[2], va[2]))
            if va[1] > best:
                best, noimp = va[1], 0
                torch.save({k: v.contiguous() for k, v in model.state_dict().items()}, best_p)
                print(f"best saved -> {best_p}")
            else:
                noimp += 1
                if noimp >= EARLY_STOP_PATIENCE:
                    print("Early stop")
                    break
    return model, history
