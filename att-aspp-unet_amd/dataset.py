"""Directory reader of the train sub-command (reference ``attention_aspp_unet_pipeline_stage.py:143-170,248-295``):
``<dir>/images/*.png|jpg|jpeg|tif|bmp`` with the same-named ``<dir>/masks/*`` (missing mask = negative frame), the
optional ``--neg_dir``, and the 10 % validation split of the positives when no ``--val_dir`` is given.

Files are decoded on the host with PIL (cv2 is not installed here; for single-channel files IMREAD_GRAYSCALE and
``convert("L")`` agree) and everything after the decode runs GPU-resident: the deterministic part of the reference's
transform -- Resize(512) -> CLAHE(1.0, 8x8) -> MedianBlur(3) -> ToFloat(255) (``imgproc.preprocess_frames(...,
resize_first=True)``; masks: nearest resize, /255) -- and, for training, HorizontalFlip(0.5) and the shuffle, both drawn
from a seeded torch generator.  The other random augmentations of the reference (Affine, RandomGamma,
RandomBrightnessContrast, ElasticTransform: albumentations' own samplers and RNG stream) are not reproduced.  ``.mha``
volumes are read by ``mhaio.py`` (middle slice, as the reference does).
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
    ``x [B,1,S,S] fp32 in [0,1]``, ``y [B,1,S,S] fp32 in {0,1}``."""

    def __init__(self, imgs: Sequence[Path], msks: Sequence[Optional[Path]], batch_size: int, size: int = 512, train: bool = True,
                 seed: int = 2025, device="cuda", rank: int = 0, world: int = 1):
        self.imgs = [Path(p) for p in imgs]
        self.msks = list(msks)
        self.bs, self.size, self.train, self.device = int(batch_size), int(size), bool(train), torch.device(device)
        self.rank, self.world = int(rank), int(world)
        self.gen = torch.Generator().manual_seed(int(seed))
        if self.train and len(self.imgs) // self.world < self.bs:
            raise ValueError(f"{len(self.imgs)} training frames give rank {rank} of {world} no full batch of {self.bs}")

    def __len__(self):
        n = len(self.imgs) // self.world if self.train else len(self.imgs)
        return n // self.bs if self.train else (n + self.bs - 1) // self.bs

    def _load(self, i: int, flip: bool):
        img = read_gray(self.imgs[i])
        x = imgproc.preprocess_frames(torch.from_numpy(img).to(self.device)[None], self.size, resize_first=True)[0]
        if self.msks[i] is None:
            y = torch.zeros(1, self.size, self.size, device=self.device)
        else:
            m = torch.from_numpy(read_gray(self.msks[i])).to(self.device)[None, None].float()
            # albumentations resizes masks with INTER_NEAREST: source index floor(dst * scale), torch's "nearest"
            y = torch.nn.functional.interpolate(m, size=(self.size, self.size), mode="nearest")[0] / 255.0
        if flip:
            x, y = x.flip(-1), y.flip(-1)
        return x, y

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

    def __iter__(self):
        for batch in self.epoch_plan():
            xs, ys = zip(*[self._load(i, f) for i, f in batch])
            yield torch.stack(xs).contiguous(), torch.stack(ys).contiguous()


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
