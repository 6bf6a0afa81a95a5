"""Abdominal-circumference measurement and frame selection of the reference's predict path
(``attention_aspp_unet_pipeline_stage.py:350-374``): host geometry on ONE final mask per case (a few thousand
boundary pixels), so it is numpy on the host, not a GPU kernel -- the masks it consumes are produced GPU-resident
(``pipeline.predict_masks``).

The reference calls ``cv2.findContours`` / ``cv2.contourArea`` / ``cv2.arcLength`` / ``cv2.fitEllipse``.  cv2 is not
installed here, so these are restatements of OpenCV's published algorithms (border following with every boundary pixel
kept = CHAIN_APPROX_NONE, the shoelace area, the closed polyline length, and the least-squares conic fit of
``fitEllipseNoDirect``): **parity unpinned** against cv2 itself; the tests pin them against closed-form ellipses.
"""
from __future__ import annotations

import math

import numpy as np

# 8-neighbourhood in clockwise order starting east (x to the right, y down): E, SE, S, SW, W, NW, N, NE
_NB = ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))


def _label8(mask: np.ndarray):
    """8-connected components of a small host mask -> (labels int32, count).  Two-pass union-find, numpy only."""
    H, W = mask.shape
    lab = np.zeros((H, W), np.int32)
    parent = [0]

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    ys, xs = np.nonzero(mask)
    for y, x in zip(ys.tolist(), xs.tolist()):
        nb = []
        if y > 0:
            for dx in (-1, 0, 1):
                if 0 <= x + dx < W and lab[y - 1, x + dx]:
                    nb.append(lab[y - 1, x + dx])
        if x > 0 and lab[y, x - 1]:
            nb.append(lab[y, x - 1])
        if not nb:
            parent.append(len(parent))
            lab[y, x] = len(parent) - 1
        else:
            r = min(find(int(v)) for v in nb)
            lab[y, x] = r
            for v in nb:
                parent[find(int(v))] = r
    roots = {}
    out = np.zeros_like(lab)
    for y, x in zip(ys.tolist(), xs.tolist()):
        r = find(int(lab[y, x]))
        out[y, x] = roots.setdefault(r, len(roots) + 1)
    return out, len(roots)


def trace_contour(comp: np.ndarray) -> np.ndarray:
    """Outer boundary of ONE 8-connected component as the ordered pixel sequence (x, y) of a border follower that keeps
    every boundary pixel (cv2.findContours with RETR_EXTERNAL, CHAIN_APPROX_NONE).  Moore-neighbour tracing: at pixel p
    with backtrack position b (the last background neighbour examined), the neighbours of p are scanned clockwise from
    the one after b; the first foreground one becomes p, the one scanned before it becomes b.  Stops when the start pixel
    is about to repeat its first move (Jacob's criterion)."""
    H, W = comp.shape
    ys, xs = np.nonzero(comp)
    if ys.size == 0:
        return np.zeros((0, 2), np.int32)
    i0 = int(np.lexsort((xs, ys))[0])           # topmost, then leftmost pixel: its west neighbour is background
    start = (int(xs[i0]), int(ys[i0]))
    nb_index = {v: k for k, v in enumerate(_NB)}

    def inside(x, y):
        return 0 <= x < W and 0 <= y < H and bool(comp[y, x])

    pts = [start]
    px, py = start
    db = 4                                      # direction from p to its backtrack position: west
    first = None
    for _ in range(8 * (H * W + 8)):
        move = None
        for k in range(1, 9):
            nd = (db + k) % 8
            cx, cy = px + _NB[nd][0], py + _NB[nd][1]
            if inside(cx, cy):
                move = nd
                break
        if move is None:                        # isolated pixel
            break
        if (px, py) == start:
            if first is None:
                first = move
            elif move == first:
                break                           # closed
        prev = (move - 1) % 8                   # the position scanned just before the hit: the new backtrack position
        bx, by = px + _NB[prev][0], py + _NB[prev][1]
        px, py = cx, cy
        db = nb_index[(bx - px, by - py)]
        pts.append((px, py))
    if len(pts) > 1 and pts[-1] == start:
        pts.pop()
    return np.asarray(pts, np.int32)


def find_external_contours(mask01: np.ndarray):
    """One traced outer contour per 8-connected foreground component."""
    m = np.asarray(mask01) > 0
    lab, n = _label8(m)
    return [trace_contour(lab == k) for k in range(1, n + 1)]


def contour_area(c: np.ndarray) -> float:
    """cv2.contourArea: shoelace formula over the contour polygon."""
    if len(c) < 3:
        return 0.0
    x, y = c[:, 0].astype(np.float64), c[:, 1].astype(np.float64)
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def arc_length(c: np.ndarray, closed: bool = True) -> float:
    """cv2.arcLength: length of the (closed) polyline through the contour points."""
    if len(c) < 2:
        return 0.0
    p = c.astype(np.float64)
    d = np.diff(np.vstack([p, p[:1]]) if closed else p, axis=0)
    return float(np.sqrt((d * d).sum(1)).sum())


def fit_ellipse(c: np.ndarray):
    """cv2.fitEllipse (OpenCV imgproc/shapedescr.cpp, ``fitEllipseNoDirect``): least-squares fit of the general conic to
    the centred points (right-hand side 10000), centre from the conic's gradient, second least-squares fit of the three
    quadratic coefficients about that centre, radii and angle from them.  -> ((cx, cy), (width, height), angle_deg) with
    width <= height and the angle of the width (short) axis, as OpenCV returns the box."""
    p = np.asarray(c, np.float64).reshape(-1, 2)
    n = len(p)
    if n < 5:
        raise ValueError("fit_ellipse needs at least 5 points")
    cm = p.mean(0)
    q = p - cm
    A = np.stack([-q[:, 0] ** 2, -q[:, 1] ** 2, -q[:, 0] * q[:, 1], q[:, 0], q[:, 1]], 1)
    g = np.linalg.lstsq(A, np.full(n, 10000.0), rcond=None)[0]
    A2 = np.array([[2 * g[0], g[2]], [g[2], 2 * g[1]]])
    rp01 = np.linalg.lstsq(A2, np.array([g[3], g[4]]), rcond=None)[0]
    dx, dy = q[:, 0] - rp01[0], q[:, 1] - rp01[1]
    g3 = np.linalg.lstsq(np.stack([dx * dx, dy * dy, dx * dy], 1), np.ones(n), rcond=None)[0]
    min_eps = 1e-8
    ang = -0.5 * math.atan2(g3[2], g3[1] - g3[0])
    t = g3[2] / math.sin(-2.0 * ang) if abs(g3[2]) > min_eps else g3[1] - g3[0]
    r0 = abs(g3[0] + g3[1] - t)
    r1 = abs(g3[0] + g3[1] + t)
    r0 = math.sqrt(2.0 / r0) if r0 > min_eps else r0
    r1 = math.sqrt(2.0 / r1) if r1 > min_eps else r1
    w, h = 2 * r0, 2 * r1
    angle = 0.0
    if w > h:                                   # always, unless the ellipse is axis-aligned with its long axis vertical
        w, h = h, w
        angle = 90.0 + ang * 180.0 / math.pi
    return (float(cm[0] + rp01[0]), float(cm[1] + rp01[1])), (float(w), float(h)), float(angle)


def _ellipse_circum(a: float, b: float) -> float:
    """pipeline:356-358: Ramanujan's approximation of the ellipse perimeter (semi-axes a, b)."""
    h = ((a - b) ** 2) / ((a + b) ** 2)
    return math.pi * (a + b) * (1 + 3 * h / (10 + math.sqrt(4 - 3 * h)))


def measure_ac_mm(mask01, spacing) -> float:
    """pipeline:359-374: abdominal circumference in mm from a 0/1 mask and (sx, sy) mm per pixel: ellipse fit of the
    largest external contour; the polyline length times the mean spacing when it has fewer than 5 points."""
    m = np.asarray(mask01.cpu() if hasattr(mask01, "cpu") else mask01)
    cnts = find_external_contours(m)
    if not cnts:
        return 0.0
    c = max(cnts, key=contour_area)
    if len(c) >= 5:
        (_, _), (MA, ma), _ = fit_ellipse(c)
        return _ellipse_circum(MA / 2 * spacing[0], ma / 2 * spacing[1])
    return arc_length(c, True) * float(sum(spacing) / 2)


def circularity(mask01) -> float:
    """4 pi A / P^2 of the largest external contour (pipeline:352)."""
    cnts = find_external_contours(np.asarray(mask01.cpu() if hasattr(mask01, "cpu") else mask01))
    if not cnts:
        return 0.0
    c = max(cnts, key=contour_area)
    A, P = contour_area(c), arc_length(c, True)
    return 0.0 if P == 0 else 4 * math.pi * A / (P * P)


def select_best(pred_stack, topk: int = 5) -> int:
    """pipeline:350-353: among the ``topk`` largest masks of a sweep, the index of the most circular one.  The areas
    come from the GPU when the stack is a device tensor (one reduction); only the top-k masks travel to the host for
    their contours."""
    if hasattr(pred_stack, "is_cuda") and pred_stack.is_cuda:
        areas = pred_stack.gt(0).flatten(1).sum(1).cpu().numpy()
        get = lambda i: pred_stack[int(i)].cpu().numpy()
    else:
        arr = np.asarray(pred_stack)
        areas = np.array([(p > 0).sum() for p in arr])
        get = lambda i: arr[int(i)]
    idx = areas.argsort()[::-1][:max(1, min(topk, len(areas)))]
    return int(max(idx, key=lambda i: circularity(get(i))))
