"""Directory reader of the train sub-command (reference pipeline:248-287): file pairing and the validation split.
Host logic only -- the transforms run on the GPU (tests/test_dataset_gpu.py)."""
import importlib

import numpy as np

dataset = importlib.import_module("att-aspp-unet_amd.dataset")


def _touch(p):
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_bytes(b"x")


def test_collect_pair_sorts_filters_and_pairs_by_name(tmp_path):
    for n in ["b_s2.png", "a_s1.png", "c.JPG", "notes.txt", "d.mha", "e.bmp"]:
        _touch(tmp_path / "images" / n)
    for n in ["a_s1.png", "c.JPG", "zzz.png"]:
        _touch(tmp_path / "masks" / n)
    imgs, msks = dataset.collect_pair(tmp_path / "images", tmp_path / "masks")
    assert [p.name for p in imgs] == ["a_s1.png", "b_s2.png", "c.JPG", "d.mha", "e.bmp"]
    assert [m.name if m else None for m in msks] == ["a_s1.png", None, "c.JPG", None, None]
    imgs2, msks2 = dataset.collect_pair(tmp_path / "images", None)
    assert imgs2 == imgs and msks2 == [None] * 5


def test_split_holds_out_ten_percent_of_the_positives_with_the_reference_rng(tmp_path):
    imgs = [tmp_path / f"i{k}.png" for k in range(25)]
    msks = [(tmp_path / f"m{k}.png") if k % 5 else None for k in range(25)]      # 20 positives, 5 negatives
    ti, tm, vi, vm = dataset.split_train_val(imgs, msks, seed=2025)
    assert len(vi) == 2 and all(m is not None for m in vm)                        # int(0.1 * 20) positives
    assert len(ti) == 23 and set(ti) | set(vi) == set(imgs) and not set(ti) & set(vi)
    # the draw is numpy's default_rng(seed).shuffle of the positive indices, as pipeline:275-279 does it
    pos = [k for k in range(25) if k % 5]
    rng = np.random.default_rng(2025)
    rng.shuffle(pos)
    assert {p.name for p in vi} == {f"i{k}.png" for k in pos[:2]}
    # the order of the training list is the directory order
    assert [p.name for p in ti] == [f"i{k}.png" for k in range(25) if k not in set(pos[:2])]
    # no positive frame at all: the candidates are all frames, at least one is held out
    ti, tm, vi, vm = dataset.split_train_val(imgs[:4], [None] * 4, seed=1)
    assert len(vi) == 1 and len(ti) == 3


def test_rank_shards_are_disjoint_and_give_every_rank_the_same_number_of_batches(tmp_path):
    """n = 17 frames, 2 ranks, batches of 3: both ranks must iterate 2 batches (8 // 3), over disjoint frames, and leave
    the shared generator in the same state (the next epoch's permutation agrees)."""
    imgs = [tmp_path / f"i{k}.png" for k in range(17)]
    lds = [dataset.DirectoryLoader(imgs, [None] * 17, 3, 64, True, 2025, "cpu", r, 2) for r in range(2)]
    for epoch in range(3):
        plans = [ld.epoch_plan() for ld in lds]
        assert len(plans[0]) == len(plans[1]) == len(lds[0]) == len(lds[1]) == 2
        assert all(len(b) == 3 for p in plans for b in p)
        seen = [{i for b in p for i, _ in b} for p in plans]
        assert not seen[0] & seen[1]
    assert lds[0].gen.get_state().equal(lds[1].gen.get_state())
    one = dataset.DirectoryLoader(imgs, [None] * 17, 3, 64, True, 2025, "cpu")
    assert len(one.epoch_plan()) == len(one) == 5
    val = dataset.DirectoryLoader(imgs, [None] * 17, 3, 64, False, 2025, "cpu")
    assert [len(b) for b in val.epoch_plan()] == [3, 3, 3, 3, 3, 2] and len(val) == 6
