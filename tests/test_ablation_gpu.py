"""The ablation variant (test_ablation.py:73-218) on the HIP engine against the reference-generated fixture
g6_ablation.npz: all five flag combinations (residual BN-free gate, no attention, plain bridge, plain U-Net,
att_depth 3), the (logits, [psi3, psi2]) output, gradients, and the differential-LR training step (:576-586).
Tolerances as in test_model_gpu.py (bf16 activations against the fp32 reference at random init)."""
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ablation_ref as AB


@pytest.fixture(scope="module")
def PA():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import ablation
    return ablation


def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def build(PA, tag):
    torch.manual_seed(2025)
    return PA.AttentionASPPUNet(base_c=8, **AB.VARIANTS[tag]).cuda()


@pytest.mark.parametrize("tag", list(AB.VARIANTS))
def test_forward_outputs_match_reference(PA, tag, golden):
    g = golden("g6_ablation.npz")
    m = build(PA, tag).eval()
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        l, (p3, p2) = m(x)
    assert l.shape == (2, 1, 64, 64) and l.dtype == torch.float32
    assert rel(l, g[f"{tag}/eval_logits"]) < 3e-2
    for got, key in ((p3, "psi3"), (p2, "psi2")):
        want = g[f"{tag}/{key}"]
        assert tuple(got.shape) == want.shape, (tag, key)
        assert float((got.cpu() - torch.from_numpy(want)).abs().max()) < 2e-2     # sigmoid outputs in [0, 1]


def _pair(PA, tag, seed=2025):
    """(emulated CPU restatement, product model) with identical weights, train mode, dropout off."""
    torch.manual_seed(seed)
    ref = AB.AttentionASPPUNet(base_c=8, **AB.VARIANTS[tag])
    m = PA.AttentionASPPUNet(base_c=8, **AB.VARIANTS[tag])
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda().train()
    ref.train()
    for net in (ref, m):
        for mod in net.bridge.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    AB.emulate_bf16_storage(ref)
    return ref, m


@pytest.mark.parametrize("tag", list(AB.VARIANTS))
def test_train_step_matches_emulated_restatement(PA, tag, golden):
    """Train mode (batch statistics over 2 frames, 4x4 bridge) is chaotic with respect to bf16 rounding at random init,
    so -- as for the pipeline model (test_model_gpu.py) -- forward and backward are compared with the CPU restatement
    under bf16-storage emulation; that restatement is pinned to the reference in fp32 by test_ablation_cpu.py."""
    g = golden("g6_ablation.npz")
    ref, m = _pair(PA, tag)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    lo, _ = ref(x)
    loss_o = F.binary_cross_entropy_with_logits(lo, y)
    loss_o.backward()
    le, psis = m(x.cuda())
    loss_e = F.binary_cross_entropy_with_logits(le, y.cuda())
    loss_e.backward()
    assert rel(le, lo) < 8e-2 and float((le.detach().cpu() - lo.detach()).abs().mean() / lo.detach().abs().max()) < 1e-2
    assert abs(float(loss_e) - float(loss_o)) < 2e-3 * float(loss_o)
    assert abs(float(loss_e) - float(g[f"{tag}/loss"])) < 2e-2 * float(g[f"{tag}/loss"])       # and near the fp32 reference
    ge = torch.cat([p.grad.detach().double().cpu().flatten() for _, p in m.named_parameters()])
    gr = torch.cat([p.grad.detach().double().flatten() for _, p in ref.named_parameters()])
    cos = float(torch.dot(ge, gr) / ge.norm() / gr.norm())
    print(tag, dict(cos=cos, norms=(float(ge.norm()), float(gr.norm()))))
    assert cos > 0.95 and abs(float(ge.norm()) - float(gr.norm())) < 0.08 * float(gr.norm()), (tag, cos)
    # the attention parameters themselves (a new kernel pair): direction and size per tensor
    named_e, named_r = dict(m.named_parameters()), dict(ref.named_parameters())
    for n in named_e:
        if ".att." in n and named_e[n].numel() >= 8:      # (the scalar psi bias gradient is a near-cancelling sum)
            a, b = named_e[n].grad.detach().double().cpu().flatten(), named_r[n].grad.detach().double().flatten()
            c = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            assert c > 0.9 and 0.7 < float(a.norm() / (b.norm() + 1e-30)) < 1.4, (tag, n, c, float(a.norm()), float(b.norm()))


def test_training_step_with_differential_lr(PA):
    """One fused step with the parameter groups of :576-586 keeps the loss finite and moves attention weights twice as far."""
    import att_aspp_unet_amd as A
    from att_aspp_unet_amd import synth
    m = build(PA, "full").train()
    x, y = synth.make_frames(2, 64, seed=3, force_pattern="pn")
    opt = A.FusedAdamW(m, weight_decay=5e-4, groups=PA.param_groups(m, 2e-3))
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
    for _ in range(2):
        opt.zero_grad()
        logits, psis = m(x.cuda())
        loss = crit(logits, y.cuda())
        loss.backward()
        opt.step()
    assert np.isfinite(float(loss)) and len(psis) == 2
    d_att = float((dict(m.named_parameters())["u4.att.Wg.weight"] - before["u4.att.Wg.weight"]).abs().max())
    d_bk = float((dict(m.named_parameters())["d2.0.block.0.weight"] - before["d2.0.block.0.weight"]).abs().max())
    assert 1.5 < d_att / d_bk < 2.6, (d_att, d_bk)


def test_plain_unet_256_is_baseline_config_1(PA):
    """BASELINE.json config 1 ("baseline U-Net forward on one 1x256x256 frame"): the reference's model.py wraps nnU-Net,
    which is not installed and ships no weights -- parity unpinned by the reference.  The plain
    U-Net of this code base is the ablation model with attention and ASPP off; checked against its CPU restatement."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(4)
    ref = AB.AttentionASPPUNet(base_c=16, use_att=False, use_aspp=False).eval()
    m = PA.AttentionASPPUNet(base_c=16, use_att=False, use_aspp=False)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda().eval()
    x, _ = synth.make_frames(1, 256, seed=9)
    with torch.no_grad():
        lo, po = ref(x)
        le, pe = m(x.cuda())
    assert le.shape == (1, 1, 256, 256) and rel(le, lo) < 3e-2
    assert all(tuple(p.shape) == (1, 1, 1, 1) and float(p.abs().sum()) == 0 for p in pe)


def test_ablation_train_loop_writes_metrics_csv_and_a_loadable_checkpoint(PA, tmp_path):
    """test_ablation.py:540-670: flags, differential learning rates, metrics.csv (:605-609), best checkpoint.  The
    checkpoint loads strictly into the CPU restatement of the same variant and reproduces the logged validation Dice."""
    import csv
    args = Namespace(stage="main", seed=3, base_c=8, lr=2e-3, epochs=3, batch_size=4, edge_w=0.05, neg_bce_w=0.05,
                     output_dir=str(tmp_path), pretrained=None, synthetic_batches=6, img_size=64,
                     no_att=False, no_aspp=True, no_edge_loss=True, att_depth=3)
    model, hist = PA.train(args)
    rows = list(csv.reader(open(tmp_path / "ckpt_main" / "metrics.csv")))
    assert rows[0] == ["epoch", "train_loss", "val_loss", "train_dice", "val_dice", "train_iou", "val_iou"]
    assert [r[0] for r in rows[1:]] == ["1", "2", "3"] and len(hist) == 3
    vals = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    assert np.isfinite(vals).all() and vals[-1, 0] < vals[0, 0]           # the training loss falls
    assert (vals[:, 2:] >= 0).all() and (vals[:, 2:] <= 1).all()
    ck = sorted((tmp_path / "ckpt_main").glob("best_*.pt"))
    assert len(ck) == 1
    sd = torch.load(ck[0], map_location="cpu", weights_only=True)
    ref = AB.AttentionASPPUNet(base_c=8, use_att=True, use_aspp=False, att_depth=3)
    ref.load_state_dict(sd, strict=True)
    # without --no_edge_loss the criterion includes the edge term: a different first-epoch loss from the same seed
    args2 = Namespace(**{**vars(args), "no_edge_loss": False, "epochs": 1, "output_dir": str(tmp_path / "b")})
    _, h2 = PA.train(args2)
    assert h2[0][0] > hist[0][0]
