"""CPU-side checks of the nn.Module surface (no kernels run): checkpoint schema, seeded
initialisation, reference-checkpoint loading, and loud failure without a GPU."""
import pytest
import torch

import att_aspp_unet_amd as A
from oracle import ref_cpu as O


def test_state_dict_schema_and_seeded_init_equal_reference(golden):
    g = golden("g1_step_c8_128.npz")
    torch.manual_seed(2025)
    m = A.AttentionASPPUNet(base_c=8)
    sd = m.state_dict()
    ref = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init/")}
    assert list(sd) == list(ref) and len(sd) == 196
    for k in sd:
        assert sd[k].shape == ref[k].shape and torch.equal(sd[k], ref[k]), k
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in O.AttentionASPPUNet(base_c=8).named_parameters()]


def test_reference_checkpoint_loads_strict_and_roundtrips(golden, tmp_path):
    g = golden("g4_trained_c8_128.npz")
    ref_sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(ref_sd, strict=True)
    o = O.AttentionASPPUNet(base_c=8)
    o.load_state_dict(m.state_dict(), strict=True)      # and back into the reference-shaped model
    for k, v in o.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k
    # legacy key names (pipeline:134-141)
    legacy = {k.replace(".Wg.", ".W_g.").replace(".Wx.", ".W_x."): v for k, v in ref_sd.items()}
    path = tmp_path / "legacy.pt"
    torch.save(legacy, path)
    m2 = A.AttentionASPPUNet(base_c=8)
    missing, unexpected = A.load_state_dict_compat(m2, path)
    assert not missing and not unexpected
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), ref_sd.values()))


def test_constructor_surface():
    assert A.ConvBNReLU(8, 16).block[0].weight.shape == (16, 8, 3, 3)
    a = A.ASPP(16, 32)
    assert len(a.blocks) == 4 and a.project[0].weight.shape == (32, 160, 1, 1) and a.project[3].p == 0.1
    gte = A.AttentionGate(16, 16, 8)
    assert gte.psi[0].weight.shape == (1, 8, 1, 1) and isinstance(gte.psi[2], torch.nn.Sigmoid)
    u = A.UpBlock(32, 16)
    assert u.up.weight.shape == (32, 16, 2, 2) and u.up.bias.shape == (16,)
    assert isinstance(A.UpBlock(16, 8, use_att=False).att, A.DummyAttention)
    with pytest.raises(A._abi.AauError):
        A.AttentionASPPUNet(base_c=12)


def test_no_silent_cpu_fallback():
    m = A.AttentionASPPUNet(base_c=8)
    with pytest.raises(A._abi.AauError):
        m(torch.zeros(2, 1, 32, 32))
    with pytest.raises(A._abi.AauError):
        A.build_criterion(O.default_args(), A.ComboLoss(), A.EdgeLoss())(torch.zeros(2, 1, 16, 16), torch.zeros(2, 1, 16, 16))


def test_lr_schedule_matches_oracle_closed_form():
    for stage, ep in (("main", 120), ("finetune", 40)):
        for e in range(ep):
            assert A.lr_at_epoch(e, ep, 3e-4, stage) == pytest.approx(O.lr_at_epoch(e, ep, 3e-4, stage), rel=1e-12)


def test_command_line_of_the_reference_script():
    """attention_aspp_unet_pipeline_stage.py:538-556: train / predict / calibrate sub-commands with the reference's flags
    (``python -m att_aspp_unet_amd <cmd> ...``).  Parsing only: running them needs a GPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cmd, flags in (("train", ["--stage", "--train_dir", "--neg_dir", "--val_dir", "--output_dir", "--pretrained", "--epochs",
                                  "--batch_size", "--lr", "--base_c", "--edge_w", "--neg_bce_w", "--seed"]),
                       ("predict", ["--weights", "--input_dir", "--out_dir", "--spacing_json", "--base_c"]),
                       ("calibrate", ["--weights", "--val_dir", "--output_dir", "--base_c"])):
        r = subprocess.run([sys.executable, "-m", "att_aspp_unet_amd", cmd, "--help"], cwd=root, capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-400:]
        for f in flags:
            assert f in r.stdout, (cmd, f)
    r = subprocess.run([sys.executable, "-m", "att_aspp_unet_amd", "predict"], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "required" in r.stderr
