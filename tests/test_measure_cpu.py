"""Abdominal-circumference measurement and frame selection (reference pipeline:350-374) -- host geometry, runs without
a GPU.  cv2 is not installed, so these checks pin the restatements against closed forms (parity with cv2 itself is
unpinned, see measure.py's header): a filled ellipse's fitted axes and Ramanujan circumference, the traced contour of a
rectangle (order, area and length as a border follower that keeps every pixel reports them), the <5-point fallback, and
select_best's rule (most circular among the top-k areas)."""
import importlib
import math

import numpy as np
import pytest

measure = importlib.import_module("att-aspp-unet_amd.measure")


def ellipse_mask(H, W, cx, cy, a, b, deg):
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    t = math.radians(deg)
    u = (x - cx) * math.cos(t) + (y - cy) * math.sin(t)
    v = -(x - cx) * math.sin(t) + (y - cy) * math.cos(t)
    return ((u / a) ** 2 + (v / b) ** 2 <= 1.0).astype(np.uint8)


@pytest.mark.parametrize("a,b,deg", [(120, 80, 0), (120, 80, 30), (60, 60, 0), (150, 40, 75), (90, 70, 120)])
def test_fit_ellipse_recovers_axes_centre_and_angle(a, b, deg):
    m = ellipse_mask(512, 512, 250.0, 260.0, a, b, deg)
    cnts = measure.find_external_contours(m)
    assert len(cnts) == 1
    (cx, cy), (w, h), ang = measure.fit_ellipse(cnts[0])
    assert abs(cx - 250.0) < 0.6 and abs(cy - 260.0) < 0.6
    # boundary pixel centres sit up to one pixel inside the continuous ellipse
    assert b - 1.0 <= w / 2 <= b + 0.25
    assert a - 1.0 <= h / 2 <= a + 0.25
    assert w <= h
    if a != b:
        # OpenCV reports the angle of the first (short) axis: the long axis is at ang +- 90
        long_axis = (ang + 90.0) % 180.0
        d = abs(long_axis - deg % 180.0)
        assert min(d, 180.0 - d) < 1.0


@pytest.mark.parametrize("spacing", [(1.0, 1.0), (0.28, 0.28), (0.5, 0.5)])
def test_measure_ac_mm_matches_ramanujan_within_one_percent(spacing):
    a, b = 120, 80
    m = ellipse_mask(512, 512, 256, 256, a, b, 20)
    ac = measure.measure_ac_mm(m, spacing)
    want = measure._ellipse_circum(a * spacing[0], b * spacing[1])
    assert abs(ac - want) / want < 0.01


def test_circle_circumference():
    m = ellipse_mask(256, 256, 128, 128, 60, 60, 0)
    ac = measure.measure_ac_mm(m, (1.0, 1.0))
    assert abs(ac - 2 * math.pi * 60) / (2 * math.pi * 60) < 0.012


def test_anisotropic_spacing_scales_each_axis_as_the_reference_does():
    # pipeline:370-371 multiplies the SHORT axis by sx and the LONG axis by sy, whatever the orientation
    a, b = 100, 50
    m = ellipse_mask(384, 384, 192, 192, a, b, 0)
    (_, _), (w, h), _ = measure.fit_ellipse(measure.find_external_contours(m)[0])
    ac = measure.measure_ac_mm(m, (0.3, 0.6))
    assert ac == pytest.approx(measure._ellipse_circum(w / 2 * 0.3, h / 2 * 0.6), rel=1e-12)


def test_rectangle_contour_is_the_ordered_boundary():
    m = np.zeros((12, 14), np.uint8)
    m[3:8, 2:9] = 1                                   # 5 rows x 7 columns
    (c,) = measure.find_external_contours(m)
    assert len(c) == 2 * (7 + 5) - 4
    assert tuple(c[0]) == (2, 3)                      # topmost-leftmost pixel first
    # successive points are 8-neighbours and the contour closes
    d = np.abs(np.diff(np.vstack([c, c[:1]]), axis=0)).max(1)
    assert (d == 1).all()
    # every boundary pixel exactly once
    assert len({tuple(p) for p in c}) == len(c)
    assert measure.contour_area(c) == (7 - 1) * (5 - 1)
    assert measure.arc_length(c) == 2 * ((7 - 1) + (5 - 1))


def test_one_pixel_wide_line_is_walked_out_and_back():
    m = np.zeros((5, 9), np.uint8)
    m[2, 1:7] = 1                                     # 6 pixels in a row
    (c,) = measure.find_external_contours(m)
    assert len(c) == 2 * 6 - 2                        # a border follower visits the inner pixels twice
    assert measure.contour_area(c) == 0.0


def test_external_only_and_largest_component_wins():
    m = ellipse_mask(256, 256, 128, 128, 70, 50, 0)
    m[120:136, 120:136] = 0                           # a hole: RETR_EXTERNAL ignores it
    m[5:9, 5:9] = 1                                   # a second, small component
    cnts = measure.find_external_contours(m)
    assert len(cnts) == 2
    full = ellipse_mask(256, 256, 128, 128, 70, 50, 0)
    assert measure.measure_ac_mm(m, (1, 1)) == pytest.approx(measure.measure_ac_mm(full, (1, 1)), rel=1e-12)


def test_fewer_than_five_points_falls_back_to_the_polyline_length():
    m = np.zeros((8, 8), np.uint8)
    m[3, 3] = m[3, 4] = m[4, 3] = 1                   # 3 boundary points
    (c,) = measure.find_external_contours(m)
    assert len(c) < 5
    want = measure.arc_length(c, True) * (0.4 + 0.6) / 2
    assert measure.measure_ac_mm(m, (0.4, 0.6)) == pytest.approx(want)
    one = np.zeros((8, 8), np.uint8)
    one[3, 3] = 1
    assert measure.measure_ac_mm(one, (1, 1)) == 0.0
    assert measure.measure_ac_mm(np.zeros((8, 8), np.uint8), (1, 1)) == 0.0


def test_select_best_takes_the_most_circular_of_the_topk_areas():
    H = W = 200
    stack = np.zeros((7, H, W), np.uint8)
    stack[0] = ellipse_mask(H, W, 100, 100, 90, 30, 0)      # large, elongated
    stack[1] = ellipse_mask(H, W, 100, 100, 50, 48, 0)      # round, mid area
    stack[2] = ellipse_mask(H, W, 100, 100, 80, 40, 10)
    stack[3] = ellipse_mask(H, W, 100, 100, 12, 12, 0)      # the roundest, but tiny
    stack[4][40:160, 30:170] = 1                            # the largest: a rectangle
    stack[5] = ellipse_mask(H, W, 100, 100, 60, 35, 0)
    assert measure.select_best(stack, topk=5) == 1
    # with topk=1 only the largest area is a candidate
    assert measure.select_best(stack, topk=1) == 4
    # the empty frame never beats a non-empty one and an all-empty stack still returns an index
    assert measure.select_best(np.zeros((3, 16, 16), np.uint8)) in (0, 1, 2)


def test_circularity_of_a_disc_is_near_one():
    assert 0.88 < measure.circularity(ellipse_mask(300, 300, 150, 150, 100, 100, 0)) <= 1.0
    assert measure.circularity(ellipse_mask(300, 300, 150, 150, 120, 30, 0)) < 0.6


def test_hd95_of_the_evaluation_script():
    """eval_segmentation_batch.py:51-58 (cross erosion restated; scipy's distance transform as in the reference)."""
    evalseg = importlib.import_module("att-aspp-unet_amd.evalseg")
    a = np.zeros((64, 64), np.uint8)
    b = np.zeros((64, 64), np.uint8)
    a[20:40, 20:40] = 1
    assert evalseg.hd95(a, a) == 0.0
    b[20:40, 23:43] = 255                                   # the same square three pixels to the right
    assert evalseg.hd95(a, b) == pytest.approx(3.0)
    c = np.zeros((64, 64), np.uint8)
    c[15:45, 15:45] = 1                                     # concentric, 5 pixels larger on every side
    assert 5.0 <= evalseg.hd95(a, c) <= 5 * 2 ** 0.5 + 1e-9
    assert math.isnan(evalseg.hd95(a, np.zeros_like(a))) and math.isnan(evalseg.hd95(np.zeros_like(a), a))
    # a mask that touches the image border keeps its rim (cv2 erodes with a +inf border)
    e = np.zeros((16, 16), np.uint8)
    e[0:8, 0:8] = 1
    er = evalseg._erode_cross(e)
    assert er[0, 0] == 1 and er[7, 7] == 0 and er[0, 7] == 0 and er[3, 3] == 1
