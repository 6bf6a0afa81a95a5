"""GPU-resident inference tail / input pipeline / GC-wrapper steps against the
numpy / SciPy restatement in oracle/imgproc_ref.py.  Integer and byte work is compared bit for bit; the two fp32
filters follow the same operation order as the restatement and are compared exactly as well.

Parity against cv2 / skimage themselves is UNPINNED (neither is importable in the build container) EXCEPT for CLAHE +
medianBlur, which the six cv2-written frames the reference holds pin bit for bit (tests/golden/g8_clahe_frames.npz):
see the header of oracle/imgproc_ref.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import imgproc_ref as R


@pytest.fixture(scope="module")
def I():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import imgproc
    return imgproc


def blobs(rng, H, W, n=6, p_noise=0.002):
    """Synthetic mask: a few ellipses (some with holes), plus speckle."""
    yy, xx = np.mgrid[0:H, 0:W]
    m = np.zeros((H, W), bool)
    for _ in range(n):
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        a, b = rng.integers(3, max(4, H // 4)), rng.integers(3, max(4, W // 4))
        e = ((yy - cy) / a) ** 2 + ((xx - cx) / b) ** 2
        m |= e < 1
        if rng.random() < 0.6:
            m &= ~(e < 0.25)                      # a hole
    m |= rng.random((H, W)) < p_noise
    return m.astype(np.uint8)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


SIZES = [(562, 744), (64, 80), (17, 33), (224, 224)]


@pytest.mark.parametrize("hw", SIZES)
def test_resize_back_gaussian_threshold_exact(I, hw):
    rng = np.random.default_rng(hw[0])
    prob = rng.random((512, 512), dtype=np.float32)
    prob[100:300, 150:400] = np.clip(prob[100:300, 150:400] + 0.5, 0, 1)
    want_r = R.resize_linear_f32(prob, hw)
    got_r = I.resize_bilinear(dev(prob), hw).cpu().numpy()
    assert got_r.shape == hw and np.array_equal(got_r, want_r)
    want_g = R.gaussian_blur5(want_r)
    got_g = I.gaussian_blur5(dev(want_r)).cpu().numpy()
    assert np.array_equal(got_g, want_g)
    for thr in (0.3, 0.48, 0.9):
        assert np.array_equal(I.threshold(dev(want_g), thr).cpu().numpy(), (want_g > np.float32(thr)).astype(np.uint8))
    # upscaling and the identity
    small = rng.random((40, 56), dtype=np.float32)
    assert np.array_equal(I.resize_bilinear(dev(small), (97, 131)).cpu().numpy(), R.resize_linear_f32(small, (97, 131)))
    assert np.array_equal(I.resize_bilinear(dev(small), (40, 56)).cpu().numpy(), small)


@pytest.mark.parametrize("hw", SIZES)
def test_components_morphology_holes_and_refine_mask(I, hw):
    rng = np.random.default_rng(7 + hw[1])
    for trial in range(3):
        m = blobs(rng, *hw)
        md = dev(m)
        # labels: same partition as scipy's 8-connected labelling, label = first pixel in raster order
        lab = I.label(md).cpu().numpy()
        ref, n = R.ndi.label(m, structure=np.ones((3, 3)))
        assert ((lab >= 0) == (m != 0)).all()
        for k in range(1, n + 1):
            vals = np.unique(lab[ref == k])
            assert len(vals) == 1 and vals[0] == np.flatnonzero((ref == k).ravel())[0]
        for min_area in (1, 20, max(20, int(0.0015 * m.size)), m.size):
            assert np.array_equal(I.keep_largest_component(md, min_area).cpu().numpy(), R.largest_component(m, min_area))
        assert np.array_equal(I.close_ellipse7(md).cpu().numpy(), R.close_ellipse7(m))
        assert np.array_equal(I.dilate3(md).cpu().numpy(), R.ndi.binary_dilation(m, structure=np.ones((3, 3))).astype(np.uint8))
        assert np.array_equal(I.fill_holes(md).cpu().numpy(), R.ndi.binary_fill_holes(m).astype(np.uint8))
        want = R.refine_mask(m)
        assert np.array_equal(I.refine_mask(md).cpu().numpy(), want)
        assert np.array_equal(I.refine_mask(m), want)                     # numpy in -> numpy out (drop-in form)


def test_refine_mask_edge_cases_and_batches(I):
    H, W = 96, 128
    z = np.zeros((H, W), np.uint8)
    assert np.array_equal(I.refine_mask(dev(z)).cpu().numpy(), z)                     # empty (:341)
    one = z.copy(); one[5, 5] = 1
    assert I.refine_mask(dev(one)).sum().item() == 0                                 # below min_area (:344)
    full = np.ones((H, W), np.uint8)
    assert np.array_equal(I.refine_mask(dev(full)).cpu().numpy(), R.refine_mask(full))
    # two components of EQUAL size: the first in raster order wins (argmax keeps the first maximum)
    tie = z.copy(); tie[10:20, 10:20] = 1; tie[50:60, 70:80] = 1
    got = I.keep_largest_component(dev(tie), 1).cpu().numpy()
    assert got[10:20, 10:20].all() and got[50:60, 70:80].sum() == 0 and np.array_equal(got, R.largest_component(tie, 1))
    # a ring touching the border: its inside is a hole, a notch open to the border is not
    ring = z.copy(); ring[0:40, 0:40] = 1; ring[10:30, 10:30] = 0; ring[60:90, 100:128] = 1; ring[70:80, 110:128] = 0
    assert np.array_equal(I.fill_holes(dev(ring)).cpu().numpy(), R.ndi.binary_fill_holes(ring).astype(np.uint8))
    # a batch is refined frame by frame
    rng = np.random.default_rng(3)
    stack = np.stack([blobs(rng, H, W) for _ in range(5)] + [z])
    got = I.refine_mask(dev(stack)).cpu().numpy()
    for k in range(stack.shape[0]):
        assert np.array_equal(got[k], R.refine_mask(stack[k])), k


def phantom_u8(rng, H, W, lo=12, hi=201):
    yy, xx = np.mgrid[0:H, 0:W]
    img = 40 + 30 * np.sin(xx / 37.0) + 25 * np.cos(yy / 23.0)
    img += 90 * (((yy - H * 0.55) / (H * 0.22)) ** 2 + ((xx - W * 0.5) / (W * 0.18)) ** 2 < 1)
    img *= rng.gamma(4.0, 0.25, (H, W))
    img = np.clip(img, 0, None)
    img = lo + (img - img.min()) / (img.max() - img.min()) * (hi - lo)
    return img.astype(np.uint8)


@pytest.mark.parametrize("hw", [(562, 744), (512, 512), (100, 90)])
def test_input_pipeline_bit_exact(I, hw):
    rng = np.random.default_rng(hw[1])
    img = phantom_u8(rng, *hw)
    n = R.normalize_minmax(img)
    assert n.min() == 0 and n.max() == 255
    assert np.array_equal(I.normalize_minmax(dev(img)).cpu().numpy(), n)
    c = R.clahe(n)
    assert np.array_equal(I.clahe(dev(n)).cpu().numpy(), c)
    for clip, tiles in ((2.0, 8), (40.0, 4)):                                     # other limits / grids, incl. padding
        assert np.array_equal(I.clahe(dev(n), clip, tiles).cpu().numpy(), R.clahe(n, clip, tiles))
    md = R.median3(c)
    assert np.array_equal(I.median3(dev(c)).cpu().numpy(), md)
    r = R.resize_linear_u8(md, (512, 512))
    assert np.array_equal(I.resize_bilinear(dev(md), (512, 512)).cpu().numpy(), r)
    x = I.preprocess_frames(dev(np.stack([img, img[::-1].copy()])), 512)
    assert x.shape == (2, 1, 512, 512) and x.dtype == torch.float32
    assert np.array_equal(x[0, 0].cpu().numpy(), R.preprocess_frame(img))
    assert np.array_equal(x[1, 0].cpu().numpy(), R.preprocess_frame(img[::-1].copy()))
    flat = np.full(hw, 77, np.uint8)                                               # constant frame: max == min
    assert np.array_equal(I.normalize_minmax(dev(flat)).cpu().numpy(), R.normalize_minmax(flat))


def test_full_tail_on_a_stack(I):
    """postprocess_probability over a batch == the per-slice chain of pipeline:455-457."""
    rng = np.random.default_rng(11)
    probs = []
    for k in range(4):
        yy, xx = np.mgrid[0:512, 0:512]
        p = 0.9 * np.exp(-(((yy - 250 - 20 * k) / 90.0) ** 2 + ((xx - 260 + 15 * k) / 120.0) ** 2))
        p += 0.35 * (rng.random((512, 512)) < 0.01) + 0.05 * rng.random((512, 512))
        probs.append(np.clip(p, 0, 1).astype(np.float32))
    probs = np.stack(probs)
    got = I.postprocess_probability(dev(probs), (562, 744), 0.48).cpu().numpy()
    for k in range(4):
        want = R.postprocess_probability(probs[k], (562, 744), 0.48)
        assert want.sum() > 1000 and np.array_equal(got[k], want), k


def test_gc_wrapper_roi_crop_paste_and_postprocess(I):
    rng = np.random.default_rng(5)
    H, W, N = 300, 340, 6
    frames = np.stack([phantom_u8(rng, H, W).astype(np.float32) / 255 for _ in range(N)])
    frames[3] = 0.2                                                              # nothing above 1.2 x mean: frame centre
    frames[4, :, :40] += 0.7                                                     # bright edge: window clamps to the frame
    fd = dev(frames)
    org = I.roi_origin(fd, 224).cpu().numpy()
    crops = I.roi_crop(fd, dev(org), 224).cpu().numpy()
    for k in range(N):
        patch, (x0, y0) = R.crop_roi(frames[k], 224)
        assert (org[k, 0], org[k, 1]) == (x0, y0), k
        assert np.array_equal(crops[k], patch)
    logits = rng.normal(size=(N, 224, 224)).astype(np.float32) * 3
    full = I.roi_paste_sigmoid(dev(logits), dev(org), (H, W)).cpu().numpy()
    for k in range(N):
        want = np.zeros((H, W), np.float32)
        want[org[k, 1]:org[k, 1] + 224, org[k, 0]:org[k, 0] + 224] = 1 / (1 + np.exp(-logits[k].astype(np.float64)))
        assert np.abs(full[k] - want).max() < 1e-6
    areas = I.frame_areas(dev(full), 0.05).cpu().numpy()
    assert np.array_equal(areas, (full > np.float32(0.05)).sum((1, 2)))
    # postprocess of the wrapper class (:66-85) on a crafted stack, device and numpy forms
    from att_aspp_unet_amd.gc_wrapper import FetalAbdomenSegmentation, select_fetal_abdomen_mask_and_frame
    seg = FetalAbdomenSegmentation(base=8)
    prob = np.zeros((5, 96, 120), np.float32)
    prob[1, 20:40, 30:60] = 0.4
    prob[3, 10:60, 20:90] = 0.3; prob[3, 70:80, 100:110] = 0.9                     # the largest area; two components
    want = R.gc_postprocess(prob)
    assert np.array_equal(seg.postprocess(dev(prob)).cpu().numpy(), want)
    assert np.array_equal(seg.postprocess(prob), want)
    m2, idx = select_fetal_abdomen_mask_and_frame(want)
    assert idx == 3 and m2.sum() == want[3].sum()
    assert seg.postprocess(np.zeros((3, 32, 32), np.float32)).sum() == 0          # all-empty fallback (:68-69)


def test_gc_wrapper_end_to_end_shapes_and_alias_module(I):
    """128 evenly spaced frames, ROI forward in batches of 8, paste back (model_attention_aspp.py:40-64); the module name
    the reference imports exists with the keyword names it uses (:6, :36)."""
    import attention_aspp_unet as M
    net = M.AttentionASPPUNet(in_ch=1, num_classes=1, base=16)
    assert net.base_c == 16
    rng = np.random.default_rng(2)
    sweep = np.stack([phantom_u8(rng, 256, 288) for _ in range(5)])
    sweep = np.repeat(sweep, 30, axis=0)                                          # 150 frames
    torch.manual_seed(0)
    seg = M.FetalAbdomenSegmentation(base=8)
    prob, idxs = seg.predict_array(sweep, nframes=128)
    assert prob.shape == (128, 256, 288) and len(idxs) == 128 and idxs[0] == 0 and idxs[-1] == 149
    assert np.array_equal(idxs, np.linspace(0, 149, 128).astype(int))
    p = prob.cpu().numpy()
    assert np.isfinite(p).all() and p.min() >= 0 and p.max() <= 1
    # outside the ROI window the probability is exactly zero; inside it is sigmoid(logits) of the crop's forward
    vol = seg_pre = None
    from att_aspp_unet_amd.gc_wrapper import preprocess_sweep
    vol = preprocess_sweep(torch.from_numpy(sweep).cuda())[torch.from_numpy(idxs).cuda()]
    org = I.roi_origin(vol, 224).cpu().numpy()
    k = 17
    inside = np.zeros((256, 288), bool)
    inside[org[k, 1]:org[k, 1] + 224, org[k, 0]:org[k, 0] + 224] = True
    assert (p[k][~inside] == 0).all() and (p[k][inside] > 0).all()
    with torch.no_grad():
        l = seg.net(I.roi_crop(vol[k:k + 1], dev(org[k:k + 1]), 224)[:, None])[0, 0].cpu().numpy()
    # batch-of-8 vs batch-of-1 plans may pick different tilings: bf16-level differences only
    assert np.abs(p[k][inside].reshape(224, 224) - 1 / (1 + np.exp(-l))).max() < 2e-2
    mask = seg.postprocess(prob)
    assert mask.shape == prob.shape and mask.dtype == torch.uint8


def test_clahe_median_against_the_cv2_frames_the_reference_holds(I):
    """inference.py:171-183: enh = cv2.medianBlur(cv2.createCLAHE(0.8, (8, 8)).apply(orig), 3) on native 562x744
    frames, written by real cv2.  The HIP kernels reproduce all three frames with 0 mismatching pixels."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_clahe_frames.npz"))
    origs = np.stack([g[f"frame{i:03d}_orig"] for i in (0, 64, 127)])
    enhs = np.stack([g[f"frame{i:03d}_enh"] for i in (0, 64, 127)])
    out = I.median3(I.clahe(dev(origs), 0.8, 8)).cpu().numpy()                    # batched [3, 562, 744]
    assert out.shape == enhs.shape and int((out != enhs).sum()) == 0
    one = I.median3(I.clahe(dev(origs[1]), 0.8, 8)).cpu().numpy()
    assert np.array_equal(one, enhs[1])
