"""Directory reader of the train sub-command (reference ``attention_aspp_unet_pipeline_stage.py:143-170,248-295``):
``<dir>/images/*.png|jpg|jpeg|tif|bmp`` with the same-named ``<dir>/masks/*`` (missing mask = negative frame), the
optional ``--neg_dir``, and the 10 % validation split of the positives when no ``--val_dir`` is given.

Files are decoded on the host with PIL (cv2 is not installed here; for single-channel files IMREAD_GRAYSCALE and
``convert("L")`` agree) by a pool of threads that runs ahead of the training step, and everything after the decode runs
GPU-resident and batched: Resize(512) -> HorizontalFlip / Affine / RandomGamma / RandomBrightnessContrast /
ElasticTransform -> CLAHE(1.0, 8x8) -> MedianBlur(3) -> ToFloat(255) (``augment.py``; masks follow the geometric steps
with nearest interpolation, /255).  The shuffle and the flips come from a seeded torch generator shared by all ranks, the
other parameters from a counter-based sampler keyed by (seed, epoch, frame).  albumentations' own RNG stream is not
reproduced (parity unpinned: the library is not importable here).  ``.mha`` volumes are read by ``mhaio.py`` (middle
slice, as the reference does).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import imgproc

EXTS = {".png", ".jpg", ".jpeg", ".tif", ".bmp", ".mha"}


def collect_pair(img_dir, msk_dir=None) -> Tuple[List[Path], List[Optional[Path]]]:
    """pipeline:248-256: sorted image files and, per image, the same-named mask if it exists."""
    imgs, msks = [], []
    img_dir = Path(img_dir)
    for p in sorted(img_dir.iterdir()):
        if p.suffix.lower() not in EXTS:
            continue
        imgs.append(p)
        q = Path(msk_dir) / p.name if msk_dir else None
        msks.append(q if (q is not None and q.exists()) else None)
    return imgs, msks


def split_train_val(imgs: Sequence[Path], msks: Sequence[Optional[Path]], seed: int):
    """pipeline:270-287: hold out 10 % (at least one) of the POSITIVE frames, drawn with numpy's default_rng(seed)
    shuffle; all frames when there is no positive one.  -> (train_imgs, train_msks, val_imgs, val_msks)"""
    pos = [i for i, m in enumerate(msks) if m is not None]
    cand = pos if pos else list(range(len(imgs)))
    rng = np.random.default_rng(seed)
    rng.shuffle(cand)
    val_sel = set(cand[:max(1, int(0.1 * len(cand)))])
    tr = [i for i in range(len(imgs)) if i not in val_sel]
    va = list(val_sel)
    return [imgs[i] for i in tr], [msks[i] for i in tr], [imgs[i] for i in va], [msks[i] for i in va]


def read_gray(path) -> Optional[np.ndarray]:
    """uint8 [H, W]; of an .mha volume the middle slice (pipeline:160-162)."""
    path = Path(path)
    if path.suffix.lower() == ".mha":
        from . import mhaio
        arr, _ = mhaio.read(path)
        if arr.ndim == 3:
            arr = arr[arr.shape[0] // 2]
        return np.ascontiguousarray(arr.astype(np.uint8))
    from PIL import Image
    return np.array(Image.open(path).convert("L"), dtype=np.uint8)


class DirectoryLoader:
    """FetalACDataset + DataLoader(batch_size, shuffle=train, drop_last=train) in one object that yields device batches
    ``x [B,1,S,S] fp32 in [0,1]``, ``y [B,1,S,S] fp32 in {0,1}``.

    The reference loader is ``num_workers=0`` albumentations on the training thread (pipeline:292-295).  Here the files
    of the NEXT batches are decoded by a pool of host threads (PIL releases the GIL while it inflates a PNG) while the
    GPU works on the current one, and everything after the decode is batched GPU work: Resize -> the random transforms
    of pipeline:149-153 (``augment.py``: flip, affine, gamma, brightness / contrast, elastic; CLAHE and MedianBlur with
    albumentations' default p = 0.5 each) -> ToFloat.  ``augment=False`` gives the deterministic subset (flip only, CLAHE
    and MedianBlur always applied) that rounds 1-2 shipped."""

    def __init__(self, imgs: Sequence[Path], msks: Sequence[Optional[Path]], batch_size: int, size: int = 512, train: bool = True,
                 seed: int = 2025, device="cuda", rank: int = 0, world: int = 1, augment: bool = True, workers: int = 8,
                 prefetch: int = 3):
        self.imgs = [Path(p) for p in imgs]
        self.msks = list(msks)
        self.bs, self.size, self.train, self.device = int(batch_size), int(size), bool(train), torch.device(device)
        self.rank, self.world = int(rank), int(world)
        self.seed, self.epoch, self.augment = int(seed), 0, bool(augment)
        self.workers, self.prefetch = max(1, int(workers)), max(1, int(prefetch))
        self.gen = torch.Generator().manual_seed(int(seed))
        self.frames_decoded = 0
        if self.train and len(self.imgs) // self.world < self.bs:
            raise ValueError(f"{len(self.imgs)} training frames give rank {rank} of {world} no full batch of {self.bs}")

    def __len__(self):
        n = len(self.imgs) // self.world if self.train else len(self.imgs)
        return n // self.bs if self.train else (n + self.bs - 1) // self.bs

    def epoch_plan(self):
        """The epoch's batches as lists of (frame index, flip) -- host logic only, advances the shared generator."""
        n = len(self.imgs)
        if self.train:
            # every rank draws the SAME permutation and the SAME flips for all n frames (the shared generator stays in
            # lock-step), keeps (n // world) * world of them and takes its stride: equal shard lengths -> equal batch
            # counts on every rank (an extra backward on one rank would wait for an all-reduce no peer joins)
            order = torch.randperm(n, generator=self.gen).tolist()
            flips = (torch.rand(n, generator=self.gen) < 0.5).tolist()
            keep = (n // self.world) * self.world
            order, flips = order[:keep][self.rank::self.world], flips[:keep][self.rank::self.world]
            nb = (n // self.world) // self.bs
        else:
            order, flips = list(range(n)), [False] * n
            nb = (n + self.bs - 1) // self.bs
        return [[(order[k], flips[k]) for k in range(b * self.bs, min((b + 1) * self.bs, len(order)))] for b in range(nb)]

    # ---- host side: decode (runs on the pool) ----
    def _decode(self, i: int):
        img = read_gray(self.imgs[i])
        msk = None if self.msks[i] is None else read_gray(self.msks[i])
        return img, msk

    # ---- device side: one batch ----
    def _resize_batch(self, arrays, nearest: bool):
        """uint8 host frames of any size -> uint8 [B, S, S] on the device (frames of equal size share one launch)."""
        S = self.size
        out = torch.empty(len(arrays), S, S, dtype=torch.uint8, device=self.device)
        groups = {}
        for k, a in enumerate(arrays):
            groups.setdefault(a.shape, []).append(k)
        for shape, ks in groups.items():
            t = torch.from_numpy(np.stack([arrays[k] for k in ks])).to(self.device, non_blocking=True)
            if shape == (S, S):
                r = t
            elif nearest:
                # albumentations resizes masks with INTER_NEAREST: source index floor(dst * scale), torch's "nearest"
                r = torch.nn.functional.interpolate(t[:, None].float(), size=(S, S), mode="nearest")[:, 0].to(torch.uint8)
            else:
                r = imgproc.resize_bilinear(t, (S, S))
            out[torch.tensor(ks, device=self.device)] = r
        return out

    def _batch(self, batch, decoded, epoch):
        from . import augment as A
        S = self.size
        idx = [i for i, _ in batch]
        imgs = self._resize_batch([d[0] for d in decoded], nearest=False)
        blank = np.zeros((S, S), np.uint8)
        msks = self._resize_batch([blank if d[1] is None else d[1] for d in decoded], nearest=True)
        if self.augment:
            p = A.sample(idx, S, S, self.seed, epoch, train=self.train)
            if self.train:
                p.flip = np.asarray([f for _, f in batch], np.uint8)     # the flips of epoch_plan (shared generator)
            x, y = A.apply(imgs, msks, p, train=self.train)
        else:
            x = imgproc.to_float(imgproc.median3(imgproc.clahe(imgs))).view(len(idx), 1, S, S)
            y = (msks.float() / 255.0).view(len(idx), 1, S, S)
            fl = torch.tensor([f for _, f in batch], device=self.device)
            if bool(fl.any()):
                x = torch.where(fl[:, None, None, None], x.flip(-1), x)
                y = torch.where(fl[:, None, None, None], y.flip(-1), y)
        self.frames_decoded += len(idx)
        return x.contiguous(), y.contiguous()

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        plan = self.epoch_plan()
        epoch = self.epoch                                 # the sampler's counter: epoch 0 is the first pass
        self.epoch += 1
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            pending = []
            nxt = 0

            def top_up():
                nonlocal nxt
                while nxt < len(plan) and len(pending) < self.prefetch:
                    pending.append([pool.submit(self._decode, i) for i, _ in plan[nxt]])
                    nxt += 1
            top_up()
            for b in range(len(plan)):
                futs = pending.pop(0)
                top_up()                                   # keep the pool busy while this batch runs on the GPU
                yield self._batch(plan[b], [f.result() for f in futs], epoch)


def loaders_from_args(args, device, rank: int = 0, world: int = 1):
    """The loader pair of pipeline:258-295 from --train_dir / --neg_dir / --val_dir."""
    size = int(getattr(args, "img_size", 512))
    tr_i, tr_m = collect_pair(Path(args.train_dir) / "images", Path(args.train_dir) / "masks")
    if getattr(args, "neg_dir", None):
        ng, _ = collect_pair(Path(args.neg_dir) / "images", None)
        tr_i, tr_m = tr_i + ng, tr_m + [None] * len(ng)
    pos = sum(m is not None for m in tr_m)
    if rank == 0:
        print(f"Train samples: pos={pos}, neg={len(tr_m) - pos} (ratio={(len(tr_m) - pos) / (pos + 1e-6):.2f})")
    if getattr(args, "val_dir", None):
        va_i, va_m = collect_pair(Path(args.val_dir) / "images", Path(args.val_dir) / "masks")
    else:
        tr_i, tr_m, va_i, va_m = split_train_val(tr_i, tr_m, args.seed)
    train_ld = DirectoryLoader(tr_i, tr_m, args.batch_size, size, True, args.seed, device, rank, world)
    val_ld = DirectoryLoader(va_i, va_m, args.batch_size, size, False, args.seed, device)
    return train_ld, val_ld
