"""The batched augmentation kernels (csrc/augment.hip, reference pipeline:149-153) against their numpy restatement
(oracle/augment_ref.py): byte results bit for bit.  Parity against albumentations / cv2 themselves is UNPINNED (neither
is importable in the build container; see the header of att-aspp-unet_amd/augment.py)."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment_ref as R
from oracle import imgproc_ref as IR


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import augment
    return augment


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def frames(rng, N, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for _ in range(N):
        cy, cx, a, b = rng.uniform(0.3, 0.7) * H, rng.uniform(0.3, 0.7) * W, rng.uniform(0.1, 0.3) * H, rng.uniform(0.1, 0.3) * W
        e = ((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2
        img = 40 + 150 * np.exp(-e) + rng.normal(0, 12, (H, W))
        out.append(np.clip(img, 0, 255).astype(np.uint8))
    return np.stack(out)


@pytest.mark.parametrize("hw", [(64, 80), (512, 512), (33, 47)])
def test_warp_lut_elastic_kernels_bit_exact(A, hw):
    H, W = hw
    rng = np.random.default_rng(H + W)
    N = 3
    img = frames(rng, N, H, W)
    msk = (img > 120).astype(np.uint8) * 255
    p = A.sample([5, 6, 7], H, W, 99, 0)
    # force every transform on frames 0 / 1, identity on frame 2
    p.inv_mats[0] = np.linalg.inv(A.affine_matrix(1.07, 0.93, 6.5, 0.02 * W, 0.01 * H, H, W))[:2].reshape(6)
    p.inv_mats[1] = np.linalg.inv(A.affine_matrix(0.92, 1.08, -7.0, 0.0, 0.02 * H, H, W))[:2].reshape(6)
    p.inv_mats[2] = [1, 0, 0, 0, 1, 0]
    inv = dev(p.inv_mats)
    w = A.warp_affine(dev(img), inv).cpu().numpy()
    wn = A.warp_affine(dev(msk), inv, nearest=True).cpu().numpy()
    for n in range(N):
        assert np.array_equal(w[n], R.warp_affine(img[n], p.inv_mats[n])), n
        assert np.array_equal(wn[n], R.warp_affine(msk[n], p.inv_mats[n], nearest=True)), n
    assert np.array_equal(w[2], img[2]) and set(np.unique(wn)) <= {0, 255}
    luts = np.stack([A.gamma_lut(0.8), A.brightness_contrast_lut(1.1, -0.1)[A.gamma_lut(1.2)], np.arange(256, dtype=np.uint8)])
    lo = A.apply_lut(dev(img), dev(luts)).cpu().numpy()
    assert all(np.array_equal(lo[n], luts[n][img[n]]) for n in range(N))
    seeds = np.array([123456789, 2**63 + 17, 5], np.uint64)
    taps = A.gaussian_taps(3.0)
    d = A.elastic_fields(dev(seeds.view(np.int64)), N, H, W).cpu().numpy()
    alpha = np.array([8.0, 40.0, 0.0], np.float32)
    ro = A.remap(dev(img), dev(d), dev(alpha)).cpu().numpy()
    rn = A.remap(dev(msk), dev(d), dev(alpha), nearest=True).cpu().numpy()
    for n in range(N):
        dn = R.elastic_fields(int(seeds[n]), H, W, taps)
        assert np.array_equal(d[n], dn), n                       # fp32 fields: same operations, same order
        assert np.array_equal(ro[n], R.remap(img[n], dn, float(alpha[n]))), n
        assert np.array_equal(rn[n], R.remap(msk[n], dn, float(alpha[n]), nearest=True)), n
    assert np.array_equal(ro[2], img[2]) and (ro[1] != img[1]).mean() > 0.2
    flags = dev(np.array([1, 0, 1], np.uint8))
    assert np.array_equal(A.hflip_frames(dev(img), flags).cpu().numpy(), np.stack([img[0][:, ::-1], img[1], img[2][:, ::-1]]))
    assert np.array_equal(A.select_frames(dev(img), dev(msk), flags).cpu().numpy(), np.stack([img[0], msk[1], img[2]]))


def test_whole_transform_matches_the_restatement_on_a_batch(A):
    """pipeline:149-155 after Resize: the composed train and validation transforms on a batch of 12 frames whose draws
    cover every branch (some frames with all transforms, some with none)."""
    H = W = 128
    rng = np.random.default_rng(4)
    img = frames(rng, 12, H, W)
    msk = (img > 110).astype(np.uint8) * 255
    for train in (True, False):
        p = A.sample(list(range(100, 112)), H, W, 2025, 3, train=train)
        x, y = A.apply(dev(img), dev(msk), p, train=train)
        xr, yr = R.apply(img, msk, p, A.gaussian_taps(A.ELASTIC_SIGMA), IR.clahe, IR.median3, train=train)
        assert x.shape == (12, 1, H, W) and y.shape == (12, 1, H, W)
        assert np.array_equal(x[:, 0].cpu().numpy(), xr) and np.array_equal(y[:, 0].cpu().numpy(), yr)
        assert set(np.unique(yr)) <= {0.0, 1.0}
    assert (p.clahe.sum() > 0) and (p.median.sum() > 0)


def test_loader_decodes_ahead_and_reports_frames_per_second(A, tmp_path):
    """FetalACDataset + DataLoader of pipeline:143-170,292-295 at the native frame size (562x744 PNGs): decode on a thread
    pool, Resize(512) + augmentations on the GPU.  Prints the feed rate beside which bench.py's step rate is to be read."""
    from PIL import Image
    from att_aspp_unet_amd import dataset
    rng = np.random.default_rng(1)
    (tmp_path / "images").mkdir(); (tmp_path / "masks").mkdir()
    fr = frames(rng, 8, 562, 744)
    n = 96
    for k in range(n):
        Image.fromarray(np.roll(fr[k % 8], k, axis=1)).save(tmp_path / "images" / f"c{k:03d}.png")
        if k % 5:
            Image.fromarray(((fr[k % 8] > 120) * 255).astype(np.uint8)).save(tmp_path / "masks" / f"c{k:03d}.png")
    imgs, msks = dataset.collect_pair(tmp_path / "images", tmp_path / "masks")
    rates = {}
    for workers in (1, 8):
        ld = dataset.DirectoryLoader(imgs, msks, 8, 512, train=True, seed=2025, device="cuda", workers=workers)
        list(ld)                                              # warm-up epoch (page cache, kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 0
        for x, y in ld:
            nb += x.shape[0]
        torch.cuda.synchronize()
        rates[workers] = nb / (time.perf_counter() - t0)
        assert nb == n and x.shape == (8, 1, 512, 512) and y.shape == (8, 1, 512, 512)
        assert 0.0 <= float(x.min()) and float(x.max()) <= 1.0 and set(torch.unique(y).tolist()) <= {0.0, 1.0}
    print(f"loader frames/s (562x744 PNG -> augmented 512x512 batch of 8): 1 decode thread {rates[1]:.0f}, 8 threads {rates[8]:.0f}")
    # the same seed reproduces the epoch stream, whatever the number of decode threads
    a = dataset.DirectoryLoader(imgs[:16], msks[:16], 8, 512, train=True, seed=7, device="cuda", workers=1)
    b = dataset.DirectoryLoader(imgs[:16], msks[:16], 8, 512, train=True, seed=7, device="cuda", workers=6)
    for (xa, ya), (xb, yb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
