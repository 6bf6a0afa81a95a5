"""``train --train_dir`` end to end (reference pipeline:244-333 with FetalACDataset :143-170): PNG files in images/ and
masks/, decoded by a thread pool (PIL), Resize -> random augmentations -> CLAHE -> MedianBlur -> ToFloat on the GPU, one
short epoch.  The transform itself is checked against the numpy restatement in tests/test_augment_gpu.py."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_set(root, n, rng, size=(96, 120), neg_every=4):
    from PIL import Image
    (root / "images").mkdir(parents=True)
    (root / "masks").mkdir(parents=True)
    yy, xx = np.mgrid[0:size[0], 0:size[1]]
    for k in range(n):
        cy, cx, a, b = rng.integers(30, 60), rng.integers(40, 80), rng.integers(12, 25), rng.integers(15, 35)
        m = (((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2 < 1)
        neg = neg_every and k % neg_every == 0
        img = 30 + 140 * (m & (not neg)) + rng.normal(0, 15, size)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(root / "images" / f"case{k:02d}_s{k}.png")
        if not neg:
            Image.fromarray((m * 255).astype(np.uint8)).save(root / "masks" / f"case{k:02d}_s{k}.png")


def test_directory_loader_matches_the_validation_transform_and_shards(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    from att_aspp_unet_amd import dataset, imgproc
    _write_set(tmp_path / "d", 10, np.random.default_rng(0))
    imgs, msks = dataset.collect_pair(tmp_path / "d" / "images", tmp_path / "d" / "masks")
    val = dataset.DirectoryLoader(imgs, msks, 4, 64, train=False, device="cuda", augment=False)
    batches = list(val)
    assert [b[0].shape[0] for b in batches] == [4, 4, 2] and len(val) == 3
    x0, y0 = batches[0]
    assert x0.shape == (4, 1, 64, 64) and y0.shape == (4, 1, 64, 64) and x0.dtype == torch.float32
    raw = torch.from_numpy(np.array(Image.open(imgs[1]).convert("L"))).cuda()
    assert torch.equal(x0[1], imgproc.preprocess_frames(raw[None], 64, resize_first=True)[0])
    assert float(y0[0].abs().max()) == 0                      # frame 0 has no mask file: a negative
    assert set(torch.unique(y0[1]).tolist()) <= {0.0, 1.0} and float(y0[1].sum()) > 0
    # training: full batches only, a permutation per epoch, flips applied to image and mask alike, ranks disjoint
    a = dataset.DirectoryLoader(imgs, msks, 2, 64, train=True, seed=3, device="cuda", rank=0, world=2)
    b = dataset.DirectoryLoader(imgs, msks, 2, 64, train=True, seed=3, device="cuda", rank=1, world=2)
    assert len(a) == 2 and len(list(a)) == 2 and len(list(b)) == 2
    one = dataset.DirectoryLoader(imgs, msks, 5, 64, train=True, seed=3, device="cuda", augment=False)
    e1 = torch.cat([x for x, _ in one])
    e2 = torch.cat([x for x, _ in one])
    assert e1.shape[0] == 10 and not torch.equal(e1, e2)      # reshuffled
    ref = {i: imgproc.preprocess_frames(torch.from_numpy(np.array(Image.open(p).convert("L"))).cuda()[None], 64, resize_first=True)[0]
           for i, p in enumerate(imgs)}
    for x in e1:
        assert any(torch.equal(x, r) or torch.equal(x, r.flip(-1)) for r in ref.values())
    with pytest.raises(ValueError):
        dataset.DirectoryLoader(imgs, msks, 16, 64, train=True, device="cuda")
    # the reference's validation transform draws CLAHE and MedianBlur with p = 0.5 each (albumentations' default): every
    # frame is one of the four combinations of the two, and the draw is a function of (seed, epoch, frame) only
    from att_aspp_unet_amd import augment
    rv = dataset.DirectoryLoader(imgs, msks, 4, 64, train=False, seed=11, device="cuda")
    xs = torch.cat([x for x, _ in rv])
    prm = augment.sample(list(range(10)), 64, 64, 11, 0, train=False)
    for i, pth in enumerate(imgs):
        r = imgproc.resize_bilinear(torch.from_numpy(np.array(Image.open(pth).convert("L"))).cuda()[None], (64, 64))
        if prm.clahe[i]:
            r = imgproc.clahe(r)
        if prm.median[i]:
            r = imgproc.median3(r)
        assert torch.equal(xs[i, 0], imgproc.to_float(r)[0]), i
    assert 0 < prm.clahe.sum() + prm.median.sum() < 20


def test_train_from_a_directory(tmp_path, capsys):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import att_aspp_unet_amd as A
    _write_set(tmp_path / "train", 20, np.random.default_rng(1))
    _write_set(tmp_path / "neg", 3, np.random.default_rng(2), neg_every=1)
    args = A.pipeline.get_args(["train", "--train_dir", str(tmp_path / "train"), "--neg_dir", str(tmp_path / "neg"), "--epochs", "2",
                                "--batch_size", "4", "--base_c", "8", "--img_size", "64", "--output_dir", str(tmp_path / "ckpt")])
    model, hist = A.train(args)
    out = capsys.readouterr().out
    assert "Train samples: pos=15, neg=8" in out
    assert len(hist) == 2 and all(np.isfinite(h[0]) for h in hist)
    assert list((tmp_path / "ckpt" / "ckpt_main").glob("best_*.pt"))
    # with an explicit validation directory
    _write_set(tmp_path / "val", 6, np.random.default_rng(3), neg_every=0)
    args = A.pipeline.get_args(["train", "--train_dir", str(tmp_path / "train"), "--val_dir", str(tmp_path / "val"), "--epochs", "1",
                                "--batch_size", "4", "--base_c", "8", "--img_size", "64", "--output_dir", str(tmp_path / "ckpt2")])
    model, hist = A.train(args)
    assert len(hist) == 1 and 0.0 <= hist[0][1] <= 1.0
