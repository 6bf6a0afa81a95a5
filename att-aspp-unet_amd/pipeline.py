"""Entry points of attention_aspp_unet_pipeline_stage.py restated over the HIP path:
``train`` (:244-333), ``evaluate`` (:235-241), ``predict_prob_tta`` (:336-338),
``load_state_dict_compat`` (:134-141), ``get_args`` (:539-550) and the module constants
(:29-31).  Dataset reading / augmentation (cv2, albumentations) and the contour
post-processing are outside the hot path; ``train``
therefore consumes any iterable of ``(x [B,1,H,W] fp32, y [B,1,H,W] {0,1})`` batches and
ships a synthetic-phantom loader for the machines that have no data.
"""
from __future__ import annotations

import argparse
import math
import os
import random
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from . import _abi, ops, synth
from .losses import ComboLoss, DiceLoss, EdgeLoss, build_criterion, seg_metrics
from .model import AttentionASPPUNet
from .optim import FusedAdamW

SEED, IMG_SIZE, WEIGHT_DECAY, GRAD_CLIP = 2025, 512, 5e-4, 1.0
EARLY_STOP_PATIENCE = 15


def set_seed(seed: int = SEED):
    """pipeline:33-52 (minus the albumentations seed, which has no counterpart here)."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def load_state_dict_compat(model, ckpt_path):
    """pipeline:134-141: legacy key rename, strict=False."""
    sd = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    new_sd = {k.replace(".W_g.", ".Wg.").replace(".W_x.", ".Wx."): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(new_sd, strict=False)
    print(f"[i] loaded with {len(missing)} missing & {len(unexpected)} unexpected keys")
    return missing, unexpected


@torch.inference_mode()
def evaluate(model, loader, device):
    """pipeline:235-241: mean over batches of (1 - soft Dice loss, hard IoU).  One host
    read at the end instead of two per batch."""
    model.eval()
    acc = torch.zeros(2, device=device)
    n = 0
    for x, y in loader:
        x, y = x.to(device), y.to(device)
        acc += seg_metrics(model(x), y, 0.5)
        n += 1
    d, i = (acc / max(n, 1)).tolist()
    return d, i


@torch.inference_mode()
def predict_prob_tta(model, x):
    """pipeline:336-338: sigmoid of the mean of logits(x) and un-flipped logits(flip(x)); numpy [H,W] of sample 0."""
    B, _, H, W = x.shape
    x = x.float().contiguous()
    xf = torch.empty_like(x)
    ops.hflip_f32(x, xf, B, H, W)
    l = model(x)
    lf = model(xf)
    prob = torch.empty_like(l)
    ops.tta_merge(l, lf, prob, B, H, W)
    return prob[0, 0].cpu().numpy()


@torch.inference_mode()
def predict_prob_tta_batch(model, x):
    """``predict_prob_tta`` for a batch, result left on the device: fp32 [B, H, W] probabilities (pipeline:336-338 per
    slice; eval-mode BatchNorm makes the batched forward equal to B single-slice forwards)."""
    B, _, H, W = x.shape
    x = x.float().contiguous()
    xf = torch.empty_like(x)
    ops.hflip_f32(x, xf, B, H, W)
    l = model(x)
    lf = model(xf)
    prob = torch.empty_like(l)
    ops.tta_merge(l, lf, prob, B, H, W)
    return prob[:, 0]


@torch.inference_mode()
def predict_masks(model, slices_u8, thr=0.48, batch=8, size=IMG_SIZE):
    """The per-slice body of ``predict`` (pipeline:449-457 / :492-498) for a stack of raw uint8 slices [N, H, W], GPU
    resident end to end: normalize -> CLAHE -> median -> resize 512 -> ToFloat | forward + h-flip TTA | resize back ->
    Gaussian 5x5 -> threshold -> refine_mask.  Returns uint8 masks [N, H, W] on the device."""
    from . import imgproc
    if isinstance(slices_u8, np.ndarray):
        slices_u8 = torch.from_numpy(np.array(slices_u8, copy=True))
    dev = next(model.parameters()).device
    sl = slices_u8.to(dev)
    if sl.dim() == 2:
        sl = sl[None]
    N, H, W = sl.shape
    x = imgproc.preprocess_frames(sl, size)
    out = torch.empty(N, H, W, dtype=torch.uint8, device=dev)
    model.eval()
    for i in range(0, N, batch):
        prob = predict_prob_tta_batch(model, x[i:i + batch])
        out[i:i + batch] = imgproc.postprocess_probability(prob, (H, W), thr)
    return out


def _read_gray(path):
    """cv2.imread(path, IMREAD_GRAYSCALE) through PIL (cv2 is not installed here): identical for single-channel files."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)


@torch.inference_mode()
def calibrate(args):
    """pipeline:376-396: sweep 17 thresholds in [0.1, 0.9] over the validation PNGs and keep the one with the best mean
    Dice.  The probability maps are computed ONCE on the GPU (the reference re-runs the network for every threshold)."""
    import json
    from . import evalseg, imgproc
    device = torch.device("cuda")
    model = AttentionASPPUNet(base_c=args.base_c).to(device)
    model.load_state_dict(torch.load(args.weights, map_location="cpu", weights_only=True))
    model.eval().set_precision(getattr(args, "precision", "fp16"))
    val_dir = Path(args.val_dir)
    imgs = sorted((val_dir / "images").glob("*.png"))
    thrs = np.linspace(0.1, 0.9, 17)
    sums = np.zeros(len(thrs))
    for p in imgs:
        sl = torch.from_numpy(_read_gray(p)).to(device)
        gt = torch.from_numpy((_read_gray(val_dir / "masks" / p.name) > 127).astype(np.uint8)).to(device)
        x = imgproc.preprocess_frames(sl[None], IMG_SIZE)
        prob = imgproc.gaussian_blur5(imgproc.resize_bilinear(predict_prob_tta_batch(model, x), tuple(sl.shape)))
        for k, thr in enumerate(thrs):
            na, nb, ni = evalseg.counts(imgproc.threshold(prob, float(thr)), gt)
            sums[k] += 2 * ni / (na + nb + 1e-7)
    best_thr = float(thrs[int(np.argmax(sums / max(len(imgs), 1)))])
    Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    json.dump({"best_thr": best_thr}, open(Path(args.output_dir) / "thr.json", "w"), indent=2)
    print(f"Calibrated thr={best_thr:.3f}")
    return best_thr


@torch.inference_mode()
def predict(args):
    """pipeline:399-523 for PNG / JPG inputs: one mask PNG per slice, computed GPU-resident (``predict_masks``), and --
    when ``--spacing_json`` names the case -- its abdominal circumference (``measure.measure_ac_mm``, :359-374) collected
    into ``ac_results.csv`` (:517-523).  ``.mha`` sweeps (:483-511) are read and written by ``mhaio.py``: every frame is
    segmented, ``measure.select_best`` picks the frame, ``output.mha`` + the frame-number JSON are written."""
    import csv
    import json
    from PIL import Image
    from . import measure
    set_seed()
    device = torch.device("cuda")
    thr = 0.48
    cfg = Path("./checkpoints/thr.json")
    if cfg.exists():
        try:
            thr = float(json.load(open(cfg))["best_thr"])
        except Exception:
            pass
    spacing_map = {}
    if getattr(args, "spacing_json", None):
        try:
            spacing_map = json.load(open(args.spacing_json, "r"))
        except Exception as e:
            print(f"cannot load spacing_json: {e}")

    def spacing_of(case_id):                     # :420-431: {"spacing": [sx, sy]} or [sx, sy]
        v = spacing_map.get(case_id)
        if isinstance(v, dict) and "spacing" in v:
            v = v["spacing"]
        if isinstance(v, (list, tuple)) and len(v) >= 2:
            return float(v[0]), float(v[1])
        return None
    model = AttentionASPPUNet(base_c=args.base_c).to(device)
    model.load_state_dict(torch.load(args.weights, map_location="cpu", weights_only=True))
    # the reference predicts in fp32 (pipeline:436-437); IEEE half is 8x closer to that than bfloat16 at the same speed
    # (tests/test_fp16_gpu.py), so inference entry points default to it; training stays bf16
    model.eval().set_precision(getattr(args, "precision", "fp16"))
    od = Path(args.out_dir)
    od.mkdir(exist_ok=True, parents=True)
    done, rows = [], []
    for p in sorted(Path(args.input_dir).iterdir()):
        if p.suffix.lower() == ".mha":
            # pipeline:483-511: every frame of the sweep, the most circular of the five largest masks, the output volume
            # (mask value 2 in that frame) + frame-number JSON, AC with the spacing of the file's own header
            from . import mhaio
            vol, hdr = mhaio.read(p)
            if vol.ndim == 2:
                vol = vol[None]
            if vol.dtype != np.uint8:
                # cv2.normalize(sl, None, 0, 255, NORM_MINMAX).astype(uint8) per slice (:491) in the file's own type; the
                # uint8 normalisation inside predict_masks is then the identity
                v = vol.astype(np.float64)
                lo, hi = v.min((1, 2), keepdims=True), v.max((1, 2), keepdims=True)
                vol = np.clip(np.rint((v - lo) * (255.0 / np.where(hi > lo, hi - lo, 1.0))), 0, 255).astype(np.uint8)
            preds = predict_masks(model, np.ascontiguousarray(vol), thr)        # uint8 [N,H,W] on the device
            bf = measure.select_best(preds, 5)
            bm = preds[bf].cpu().numpy()
            write_output_mha_and_json(bm, bf, p, od, hdr, vol.shape[0])
            sp3 = mhaio.spacing(hdr)
            ac_mm = round(measure.measure_ac_mm(bm, (float(sp3[0]), float(sp3[1]))), 1)
            rows.append((p.stem, int(bf), ac_mm))
            done.append(p.stem)
            print(f"{p.stem}: best_frame={bf}, AC={ac_mm:.1f} mm")
            continue
        if p.suffix.lower() not in {".png", ".jpg", ".jpeg"}:
            continue
        mask = predict_masks(model, _read_gray(p), thr)[0]
        Image.fromarray((mask * 255).cpu().numpy()).save(od / f"{p.stem}_mask.png")
        done.append(p.stem)
        stem = p.stem                                            # :460-469: "<case>_s<frame>"
        case_id, frame_idx = stem, -1
        if "_s" in stem:
            case_id = stem.split("_s")[0]
            try:
                frame_idx = int(stem.split("_s")[1])
            except Exception:
                frame_idx = -1
        sp = spacing_of(case_id)
        if sp is None:
            print(f"no spacing for {case_id}, skip AC")
        else:
            ac_mm = round(measure.measure_ac_mm(mask, sp), 1)
            rows.append((case_id, frame_idx, ac_mm))
            print(f"{stem}: AC={ac_mm:.1f} mm")
    if rows:
        with open(od / "ac_results.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["case_id", "frame_idx", "ac_mm"])
            w.writerows(rows)
    return done


def convert_mask_2d_to_3d(mask, frame, nf):
    """pipeline:526-529: the 2-D mask as value 2 in frame ``frame`` of an otherwise empty volume of ``nf`` frames."""
    m = (np.asarray(mask) > 0).astype(np.uint8) * 2
    vol = np.zeros((nf,) + m.shape, np.uint8)
    if 0 <= frame < nf:
        vol[frame] = m
    return vol


def write_output_mha_and_json(mask, frame, ref, od, ref_header=None, nf=None):
    """pipeline:530-536: ``<od>/<case>/images/fetal-abdomen-segmentation/output.mha`` with the reference volume's geometry
    and ``<od>/<case>/fetal-abdomen-frame-number.json``."""
    import json
    from . import mhaio
    ref = Path(ref)
    if ref_header is None or nf is None:
        ref_header, _ = mhaio.read_header(ref)
        dims = [int(t) for t in ref_header["DimSize"].split()]
        nf = dims[2] if len(dims) > 2 else 1
    cd = Path(od) / ref.stem
    (cd / "images" / "fetal-abdomen-segmentation").mkdir(parents=True, exist_ok=True)
    mhaio.write(cd / "images" / "fetal-abdomen-segmentation" / "output.mha", convert_mask_2d_to_3d(mask, int(frame), int(nf)),
                like=ref_header)
    json.dump(int(frame), open(cd / "fetal-abdomen-frame-number.json", "w"), indent=2)
    print(f"{ref.stem} frame {int(frame)}")


class GraphedForward:
    """Eval-mode forward of a fixed input shape captured once as a hipGraph and replayed (inference is
    launch-latency bound at batch 1; the recorded plan has no host synchronisation, so it captures as is)."""

    def __init__(self, model, shape, device="cuda"):
        assert not model.training, "GraphedForward captures the eval plan"
        self.model = model
        self.x = torch.zeros(*shape, device=device)
        with torch.no_grad():
            for _ in range(2):
                model(self.x)                      # builds the plan / warms the allocator outside of capture
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(self.graph, stream=s):
                self.out = model(self.x)
            torch.cuda.current_stream().wait_stream(s)

    @torch.no_grad()
    def __call__(self, x):
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.out


@torch.inference_mode()
def predict_sliding_window(model, x, window=512, stride=256, sigma_frac=0.125, forward=None):
    """Build-side extension (BASELINE config 5): tile a [1,1,H,W] frame into ``window`` x ``window`` crops on a
    ``stride`` grid, run them as ONE batch through the network and blend the logits with a Gaussian weight
    (the scheme nnU-Net uses for the reference's baseline, model.py:41-49).  Returns logits [1,1,H,W].
    ``forward`` may be a GraphedForward for the batch shape."""
    assert x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 1
    H, W = x.shape[2:]
    assert H >= window and W >= window and (H - window) % stride == 0 and (W - window) % stride == 0, \
        "frame must be covered exactly by the window grid"
    ny, nx = (H - window) // stride + 1, (W - window) // stride + 1
    crops = torch.empty(ny * nx, 1, window, window, device=x.device)
    for iy in range(ny):
        for ix in range(nx):
            crops[iy * nx + ix, 0] = x[0, 0, iy * stride:iy * stride + window, ix * stride:ix * stride + window]
    logits = (forward or model)(crops).contiguous()
    out = torch.empty(1, 1, H, W, device=x.device)
    ops.window_blend(logits, out, H, W, window, stride, ny, nx, sigma_frac * window)
    return out


def lr_at_epoch(ep, epochs, lr, stage="main"):
    """Closed form of pipeline:303-306 (LinearLR 0.2->1 for max(1, 5%) epochs, then cosine)."""
    warm = 0 if stage == "finetune" else max(1, int(0.05 * epochs))
    if ep < warm:
        return lr * (0.2 + 0.8 * ep / warm)
    return lr * 0.5 * (1 + math.cos(math.pi * (ep - warm) / (epochs - warm)))


class SyntheticLoader:
    """Deterministic phantom batches (see synth.py); stands in for FetalACDataset + DataLoader (pipeline:143-170,292-295)."""

    def __init__(self, n_batches, batch_size, size=IMG_SIZE, seed=SEED, device="cuda", neg_frac=0.2):
        self.batches = []
        for i in range(n_batches):
            x, y = synth.make_frames(batch_size, size, seed=seed + i, neg_frac=neg_frac)
            self.batches.append((x.to(device), y.to(device)))

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def train(args, train_loader=None, val_loader=None):
    """pipeline:244-333.  Same optimiser / schedule / clipping / early stopping / best-checkpoint logic;
    bf16 activations with fp32 master weights instead of fp16 autocast + GradScaler.

    Data parallel: when ``torch.distributed`` is initialised with more than one rank (one process per GPU, launched by
    ``python -m torch.distributed.run``), the model is wrapped in ``parallel.DataParallel``: every rank trains on its
    own shard (the synthetic loader is seeded per rank, the directory loader deals the shuffled frames out by rank; a
    caller-supplied loader must shard itself), gradients are
    all-reduced in buckets under the backward pass, and every rank applies the same update.  BatchNorm running
    statistics are per rank, so before validation rank 0's buffers are broadcast and rank 0's Dice / IoU
    are the ones every rank uses for the best-checkpoint and early-stop decisions -- a rank leaving the epoch loop
    alone would leave its peers waiting in the next all-reduce.  Rank 0 alone prints and writes checkpoints."""
    import torch.distributed as dist
    from .parallel import DataParallel
    set_seed(args.seed)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    device = torch.device("cuda", torch.cuda.current_device())
    if train_loader is None:
        n = int(getattr(args, "synthetic_batches", 0) or 0)
        size = int(getattr(args, "img_size", IMG_SIZE))
        if n > 0:
            train_loader = SyntheticLoader(n, args.batch_size, size, args.seed + 7919 * rank, device)
            val_loader = SyntheticLoader(max(1, n // 10), args.batch_size, size, args.seed + 100000, device, neg_frac=0.0)
        elif getattr(args, "train_dir", None):
            from . import dataset            # pipeline:248-295: images/ + masks/ directories, decoded by PIL, transformed on the GPU
            train_loader, val_loader = dataset.loaders_from_args(args, device, rank, world)
        else:
            raise RuntimeError("train needs --train_dir (images/ + masks/), --synthetic_batches N, or loaders passed in")
    model = AttentionASPPUNet(base_c=args.base_c).to(device)
    if args.stage == "finetune":
        load_state_dict_compat(model, args.pretrained)
        print(f"loaded pretrained {args.pretrained}")
    dp = DataParallel(model) if world > 1 else None
    opt = FusedAdamW(model, lr=args.lr, weight_decay=WEIGHT_DECAY, max_grad_norm=GRAD_CLIP)
    tot_ep = args.epochs
    crit = build_criterion(args, ComboLoss(), EdgeLoss())
    out_dir = Path(args.output_dir) / ("ckpt_main" if args.stage == "main" else "ckpt_finetune")
    out_dir.mkdir(parents=True, exist_ok=True)
    best, best_p, noimp = 0., out_dir / f"best_{datetime.now():%Y%m%d-%H%M%S}.pt", 0
    history = []
    for ep in range(1, tot_ep + 1):
        opt.param_groups[0]["lr"] = lr_at_epoch(ep - 1, tot_ep, args.lr, args.stage)
        model.train()
        run = torch.zeros((), device=device)
        for x, y in train_loader:
            x, y = x.to(device), y.to(device)
            opt.zero_grad(set_to_none=True)
            loss = crit(model(x), y)
            loss.backward()
            if dp is not None:
                dp.finish()                      # the bucket all-reduces were issued from inside the backward pass
                opt.step(inv_scale=dp.inv_scale)
            else:
                opt.step()
            run += loss.detach()
        if dp is not None:
            dp.sync_buffers()                    # BatchNorm running statistics diverge per shard: evaluate (and save) rank 0's
        d, i = evaluate(model, val_loader, device)
        if dp is not None:
            d, i = dp.agree(d, i)                # one decision for best-checkpoint / early stop on every rank
        history.append((float(run) / max(len(train_loader), 1), d, i))
        if rank == 0:
            print(f"Epoch {ep}/{tot_ep} loss {history[-1][0]:.4f} | Dice {d:.4f} | IoU {i:.4f}")
        if d > best:
            best, noimp = d, 0
            if rank == 0:
                torch.save({k: v.contiguous() for k, v in model.state_dict().items()}, best_p)
                print(f"best saved -> {best_p}")
        else:
            noimp += 1
            if noimp >= EARLY_STOP_PATIENCE:
                if rank == 0:
                    print("Early stop")
                break
    return model, history


def _mask_f32(y, B, H, W, device):
    """The fused criterion reads ``y`` as B*H*W fp32 values: anything else (uint8 / bool / bf16 masks, a strided view, a
    host tensor) is converted here, and a wrong shape raises, instead of being misread by the kernel."""
    if not isinstance(y, torch.Tensor) or y.numel() != B * H * W or y.dim() != 4 or tuple(y.shape) != (B, 1, H, W):
        raise _abi.AauError(f"targets must be a [{B},1,{H},{W}] tensor, got "
                            f"{tuple(y.shape) if isinstance(y, torch.Tensor) else type(y)}")
    if y.device != device or y.dtype != torch.float32 or not y.is_contiguous():
        y = y.to(device=device, dtype=torch.float32).contiguous()
    return y


class TrainStep:
    """The reference inner loop body (pipeline:319-324) as ONE fused call with no autograd graph,
    no allocator traffic and no host synchronisation: forward plan -> fused criterion (writes
    d(loss)/d(logits) straight into the plan) -> backward plan (-> bucketed all-reduce when
    data-parallel) -> clip + AdamW.  ``loss`` stays on the device."""

    def __init__(self, model, opt, args, dp=None):
        self.model, self.opt, self.dp = model, opt, dp
        self.finetune = args.stage == "finetune"
        self.neg_w, self.edge_w = float(args.neg_bce_w), max(float(args.edge_w), 0.0)
        self.sums = self.loss = None

    def __call__(self, x, y):
        m = self.model
        assert m.training, "TrainStep needs model.train()"
        plan = m._plan_for(x)
        B, _, H, W = x.shape
        y = _mask_f32(y, B, H, W, x.device)
        if self.sums is None or self.sums.shape[1] != B:
            self.sums = torch.zeros(32, B, 8, device=x.device)
            self.loss = torch.zeros(4, device=x.device)
        logits = plan.run_forward(x)
        ops.criterion(logits, y, self.sums, self.loss, plan.dlogits, B, H, W, self.finetune, self.neg_w, self.edge_w)
        plan.run_backward(None)
        inv = 1.0
        if self.dp is not None:
            self.dp.finish()
            inv = self.dp.inv_scale
        self.opt.step(inv_scale=inv)
        return self.loss[0]


class GraphedTrainStep:
    """``TrainStep`` replayed as hipGraphs.  Without data parallelism the whole step is one graph.  With it, the backward
    launch list is cut at the gradient-bucket marks: one graph per segment, and the RCCL all-reduce of a bucket is
    issued (eagerly, on its own stream) between two graph launches -- the collectives are not captured, the ~290 kernel
    launches of the step are.  (Measured on one MI355X with world size 1: the segmented form is 1 % slower than the
    eager launch list, which itself is within 1 % of the single graph, so bench.py keeps N > 1 eager.)  Inputs are copied
    into static buffers (the mask is converted to fp32 on the way), so any ``(x, y)`` of the captured shape works.
    Nothing that changes from step to step is baked into the graph: step count and dropout seed advance in device
    memory, and the learning rate / weight decay are read from the optimiser's device table, refreshed before each
    replay, so an LR scheduler keeps working (pipeline:303-306,325)."""

    def __init__(self, step: "TrainStep", x, y, warmup: int = 2):
        self.step = step
        m, dp = step.model, step.dp
        assert m.training
        B, _, H, W = x.shape
        self.x, self.y = x.float().contiguous().clone(), _mask_f32(y, B, H, W, x.device).clone()
        for _ in range(max(warmup, 1 if step.sums is None else 0)):   # plan, optimiser state, allocator pools
            step(self.x, self.y)
        torch.cuda.synchronize()
        plan = m._plan_for(self.x)
        B, _, H, W = self.x.shape
        segs = plan.bwd.segments()
        if dp is None:
            assert all(cb is None for _, cb in segs)
        inv = 1.0 if dp is None else dp.inv_scale

        def head():
            logits = plan.run_forward(self.x)
            ops.criterion(logits, self.y, step.sums, step.loss, plan.dlogits, B, H, W, step.finetune, step.neg_w,
                          step.edge_w)
            plan.begin_backward()

        def seg(k):
            plan.bwd.run_ops(segs[k][0], torch.cuda.current_stream().cuda_stream, first=(k == 0), last=(k == len(segs) - 1))

        parts = []                                   # [(callables, callback after them)]
        for k, (_, cb) in enumerate(segs):
            fns = ([head] if k == 0 else []) + [lambda k=k: seg(k)]
            parts.append((fns, cb))
        if dp is None:
            parts[-1][0].append(lambda: step.opt.step(inv_scale=inv))
            self.tail = None
        else:
            self.tail = self._capture([lambda: step.opt.step(inv_scale=inv)])
        # a segment with nothing to launch (the list ends with a mark) is not captured
        self.parts = [(self._capture(fns) if (k == 0 or segs[k][0] or len(fns) > 1) else None, cb)
                      for k, (fns, cb) in enumerate(parts)]
        m.engine.store.bind_grads()

    @staticmethod
    def _capture(fns):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            for fn in fns:
                fn()
        torch.cuda.current_stream().wait_stream(s)
        return g

    def __call__(self, x, y):
        if tuple(x.shape) != tuple(self.x.shape) or tuple(y.shape) != tuple(self.y.shape):
            raise _abi.AauError(f"GraphedTrainStep was captured for {tuple(self.x.shape)} / {tuple(self.y.shape)}, "
                                f"got {tuple(x.shape)} / {tuple(y.shape)}")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)        # copy_ converts dtype / layout / device
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y, non_blocking=True)
        self.step.opt.refresh_hyper()                 # LR schedule: a 16-byte upload when it changed, outside the graph
        for g, cb in self.parts:
            if g is not None:
                g.replay()
            if cb is not None:
                cb()
        if self.tail is not None:
            self.step.dp.finish()
            self.tail.replay()
        return self.step.loss[0]


def get_args(argv=None):
    """pipeline:539-550 (train / predict / calibrate flags), plus --synthetic_batches / --img_size."""
    p = argparse.ArgumentParser("A-ASPP-UNet unified (MI355X)")
    sp = p.add_subparsers(dest="cmd", required=True)
    t = sp.add_parser("train")
    t.add_argument("--stage", choices=["main", "finetune"], default="main")
    t.add_argument("--seed", type=int, default=SEED)
    t.add_argument("--train_dir"); t.add_argument("--neg_dir"); t.add_argument("--val_dir")
    t.add_argument("--output_dir", default="./checkpoints"); t.add_argument("--pretrained")
    t.add_argument("--epochs", type=int, default=120); t.add_argument("--batch_size", type=int, default=8)
    t.add_argument("--lr", type=float, default=3e-4); t.add_argument("--base_c", type=int, default=48)
    t.add_argument("--edge_w", type=float, default=0.05); t.add_argument("--neg_bce_w", type=float, default=0.05)
    t.add_argument("--synthetic_batches", type=int, default=0); t.add_argument("--img_size", type=int, default=IMG_SIZE)
    pr = sp.add_parser("predict")
    pr.add_argument("--weights", required=True); pr.add_argument("--input_dir", required=True)
    pr.add_argument("--out_dir", default="./preds"); pr.add_argument("--spacing_json", required=True)
    pr.add_argument("--base_c", type=int, default=48)
    pr.add_argument("--precision", choices=["fp16", "bf16"], default="fp16")
    ca = sp.add_parser("calibrate")
    ca.add_argument("--weights", required=True); ca.add_argument("--val_dir", required=True)
    ca.add_argument("--output_dir", default="./checkpoints"); ca.add_argument("--base_c", type=int, default=48)
    ca.add_argument("--precision", choices=["fp16", "bf16"], default="fp16")
    return p.parse_args(argv)


def main(argv=None):
    """pipeline:552-556: dispatch of the three sub-commands."""
    args = get_args(argv)
    if args.cmd == "train":
        return train(args)
    if args.cmd == "predict":
        return predict(args)
    if args.cmd == "calibrate":
        return calibrate(args)
