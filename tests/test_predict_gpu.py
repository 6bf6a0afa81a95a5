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
