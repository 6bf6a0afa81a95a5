"""The predict sub-command end to end on PNG inputs (reference pipeline:399-523): one mask per slice, computed
GPU-resident, plus the abdominal-circumference table for the cases the spacing file names."""
import csv
import json
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_predict_writes_masks_and_ac_results(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from PIL import Image
    import att_aspp_unet_amd as A
    from att_aspp_unet_amd import measure, pipeline
    rng = np.random.default_rng(5)
    inp = tmp_path / "in"
    inp.mkdir()
    yy, xx = np.mgrid[0:300, 0:360]
    names = ["caseA_s3", "caseA_s17", "caseB_s0", "loose", "caseC_sx"]
    for k, n in enumerate(names):
        img = 40 + 150 * (((yy - 150) / (70 + 5 * k)) ** 2 + ((xx - 180) / (100 - 4 * k)) ** 2 < 1) + rng.normal(0, 12, yy.shape)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(inp / f"{n}.png")
    (inp / "notes.txt").write_text("ignored")
    torch.manual_seed(0)
    net = A.AttentionASPPUNet(base_c=8).cuda()
    torch.save(net.state_dict(), tmp_path / "w.pth")
    sp = {"caseA": {"spacing": [0.3, 0.4]}, "caseB": [0.5, 0.5], "caseC": [0.2, 0.2, 1.0]}
    json.dump(sp, open(tmp_path / "sp.json", "w"))
    args = types.SimpleNamespace(weights=str(tmp_path / "w.pth"), input_dir=str(inp), out_dir=str(tmp_path / "out"),
                                 spacing_json=str(tmp_path / "sp.json"), base_c=8, precision="fp16")
    done = pipeline.predict(args)
    assert sorted(done) == sorted(names)
    rows = list(csv.reader(open(tmp_path / "out" / "ac_results.csv")))
    assert rows[0] == ["case_id", "frame_idx", "ac_mm"]
    got = {(r[0], int(r[1])): float(r[2]) for r in rows[1:]}
    assert set(got) == {("caseA", 3), ("caseA", 17), ("caseB", 0), ("caseC", -1)}      # "loose" has no spacing
    for n in names:
        m = np.asarray(Image.open(tmp_path / "out" / f"{n}_mask.png"))
        assert m.shape == (300, 360) and set(np.unique(m)) <= {0, 255}
    spacing = {"caseA": (0.3, 0.4), "caseB": (0.5, 0.5), "caseC": (0.2, 0.2)}
    for (case, fi), ac in got.items():
        stem = f"{case}_s{fi}" if fi >= 0 else "caseC_sx"
        m = (np.asarray(Image.open(tmp_path / "out" / f"{stem}_mask.png")) > 0).astype(np.uint8)
        assert ac == round(measure.measure_ac_mm(m, spacing[case]), 1)


def test_select_best_on_a_device_stack_matches_the_host():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import measure
    yy, xx = np.mgrid[0:160, 0:160]
    stack = np.stack([(((yy - 80) / a) ** 2 + ((xx - 80) / b) ** 2 < 1).astype(np.uint8)
                      for a, b in [(70, 20), (40, 38), (60, 30), (10, 10), (50, 45), (66, 33)]])
    assert measure.select_best(torch.from_numpy(stack).cuda(), topk=4) == measure.select_best(stack, topk=4)


def test_predict_on_an_mha_sweep_writes_the_output_volume(tmp_path):
    """pipeline:483-511,526-536: every frame segmented, select_best, output.mha (value 2 in the chosen frame, geometry of
    the input), the frame-number JSON and the AC row with the spacing of the file's header."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import att_aspp_unet_amd as A
    from att_aspp_unet_amd import measure, mhaio, pipeline
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:112, 0:144]
    vol = np.stack([np.clip(40 + 150 * (((yy - 56) / (20 + 4 * k)) ** 2 + ((xx - 72) / (30 + 3 * k)) ** 2 < 1) + rng.normal(0, 10, yy.shape),
                            0, 255).astype(np.uint8) for k in range(6)])
    inp = tmp_path / "in"
    inp.mkdir()
    like = {"ElementSpacing": "0.3 0.45 1", "Offset": "5 6 7"}
    mhaio.write(inp / "sweepA.mha", vol, like=like)
    mhaio.write(inp / "sweepB.mha", vol.astype(np.int16) * 3 - 100, like=like, compress=False)   # another element type
    torch.manual_seed(0)
    net = A.AttentionASPPUNet(base_c=8).cuda()
    torch.save(net.state_dict(), tmp_path / "w.pth")
    json.dump({}, open(tmp_path / "sp.json", "w"))
    args = types.SimpleNamespace(weights=str(tmp_path / "w.pth"), input_dir=str(inp), out_dir=str(tmp_path / "out"),
                                 spacing_json=str(tmp_path / "sp.json"), base_c=8, precision="fp16")
    done = pipeline.predict(args)
    assert sorted(done) == ["sweepA", "sweepB"]
    rows = {r[0]: r for r in list(csv.reader(open(tmp_path / "out" / "ac_results.csv")))[1:]}
    for case in ("sweepA", "sweepB"):
        out, h = mhaio.read(tmp_path / "out" / case / "images" / "fetal-abdomen-segmentation" / "output.mha")
        bf = json.load(open(tmp_path / "out" / case / "fetal-abdomen-frame-number.json"))
        assert out.shape == vol.shape and out.dtype == np.uint8 and set(np.unique(out)) <= {0, 2}
        assert mhaio.spacing(h) == (0.3, 0.45, 1.0) and h["Offset"] == "5 6 7"
        assert all(out[k].max() == 0 for k in range(6) if k != bf)
        assert int(rows[case][1]) == bf
        assert float(rows[case][2]) == round(measure.measure_ac_mm((out[bf] > 0).astype(np.uint8), (0.3, 0.45)), 1)
    # the int16 sweep is an affine map of the uint8 one: per-slice min-max normalisation makes the inputs identical
    a, _ = mhaio.read(tmp_path / "out" / "sweepA" / "images" / "fetal-abdomen-segmentation" / "output.mha")
    b, _ = mhaio.read(tmp_path / "out" / "sweepB" / "images" / "fetal-abdomen-segmentation" / "output.mha")
    assert np.array_equal(a, b)
    # the frame choice is select_best over the per-frame masks the device path produces
    masks = pipeline.predict_masks(net.eval().set_precision("fp16"), vol, 0.48)
    assert measure.select_best(masks, 5) == json.load(open(tmp_path / "out" / "sweepA" / "fetal-abdomen-frame-number.json"))


def test_gc_wrapper_reads_mha(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import gc_wrapper, mhaio
    rng = np.random.default_rng(1)
    vol = (rng.random((10, 256, 256)) * 255).astype(np.uint8)
    mhaio.write(tmp_path / "s.mha", vol)
    x = gc_wrapper.load_image_file_as_array(location=tmp_path / "s.mha")
    assert x.shape == (1, 10, 256, 256) and x.dtype == torch.float32 and x.is_cuda
    assert torch.equal(x[0], gc_wrapper.preprocess_sweep(torch.from_numpy(vol).cuda()))
    torch.manual_seed(0)
    seg = gc_wrapper.FetalAbdomenSegmentation(base=8)
    prob = seg.predict([tmp_path / "s.mha"])
    assert prob.shape == (128, 256, 256) and seg.case_id == "s"


def test_gc_entry_point_writes_the_challenge_outputs(tmp_path):
    """inference.py:50-133: <input>/images/stacked-fetal-ultrasound/*.mha -> <output>/images/fetal-abdomen-segmentation/
    <case>.mha (uint8 {0,1}, one frame set, as many frames as the sweep, 0.28 mm spacing) + the frame-number JSON."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import gc_wrapper, mhaio
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:240, 0:256]
    vol = np.stack([np.clip(30 + 160 * (((yy - 120) / (30 + k % 40)) ** 2 + ((xx - 128) / (45 + k % 30)) ** 2 < 1) + rng.normal(0, 12, yy.shape),
                            0, 255).astype(np.uint8) for k in range(140)])
    loc = tmp_path / "in" / "images" / "stacked-fetal-ultrasound"
    loc.mkdir(parents=True)
    mhaio.write(loc / "sweep.mha", vol)
    torch.manual_seed(0)
    assert gc_wrapper.run(tmp_path / "in", tmp_path / "out", case_id="caseX", base=8) == 0
    out, h = mhaio.read(tmp_path / "out" / "images" / "fetal-abdomen-segmentation" / "caseX.mha")
    frame = json.load(open(tmp_path / "out" / "fetal-abdomen-frame-number.json"))
    assert out.shape == vol.shape and out.dtype == np.uint8 and set(np.unique(out)) <= {0, 1}
    assert mhaio.spacing(h) == (0.28, 0.28, 0.28) and h["CompressedData"] == "True"
    assert -1 <= frame < 128
    assert all(out[k].max() == 0 for k in range(vol.shape[0]) if k != frame)
    # the helpers on their own
    v = gc_wrapper.convert_2d_mask_to_3d(mask_2d=np.eye(4), frame_number=2, number_of_frames=5)
    assert v.shape == (5, 4, 4) and v[2].max() == 2 and v.sum() == 8
    assert gc_wrapper.convert_2d_mask_to_3d(mask_2d=np.eye(4), frame_number=-1, number_of_frames=3).sum() == 0
    with pytest.raises(ValueError):
        gc_wrapper.convert_2d_mask_to_3d(mask_2d=np.eye(4), frame_number=7, number_of_frames=3)
    with pytest.raises(FileNotFoundError):
        gc_wrapper.run(tmp_path / "nothing", tmp_path / "out2")
