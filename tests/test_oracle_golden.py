"""Pins the CPU oracle (oracle/ref_cpu.py) to the fixtures produced by executing the
reference itself (oracle/make_golden.py).  CPU only."""
import argparse

import numpy as np
import torch

from oracle import ref_cpu as O


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def test_seeded_init_matches_reference(golden):
    g = golden("g1_step_c8_128.npz")
    torch.manual_seed(2025)
    net = O.AttentionASPPUNet(base_c=8)
    ref_sd = _sd(g, "init/")
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())  # 196 keys, same order
    assert len(sd) == 196
    for k in sd:
        assert torch.equal(sd[k], ref_sd[k]), k


def test_full_step_matches_reference(golden):
    g = golden("g1_step_c8_128.npz")
    net = O.AttentionASPPUNet(base_c=8)
    net.load_state_dict(_sd(g, "init/"), strict=True)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(x).numpy(), g["eval_logits"], rtol=0, atol=2e-6)
    net.train()
    net.bridge.project[3].p = 0.0
    crit_m = O.build_criterion(O.default_args(stage="main"), O.ComboLoss(), O.EdgeLoss())
    crit_f = O.build_criterion(O.default_args(stage="finetune"), O.ComboLoss(), O.EdgeLoss())
    opt = O.make_optimizer(net, 3e-4)
    opt.zero_grad(set_to_none=True)
    logits = net(x)
    np.testing.assert_allclose(logits.detach().numpy(), g["train_logits"], rtol=0, atol=2e-6)
    loss = crit_m(logits, y)
    assert abs(loss.item() - float(g["loss_main"])) < 2e-6
    assert abs(crit_f(logits.detach(), y).item() - float(g["loss_finetune"])) < 2e-6
    for k, v in _sd(g, "after_fwd/").items():
        np.testing.assert_allclose(net.state_dict()[k].numpy(), v.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    loss.backward()
    for k, p in net.named_parameters():
        ref = g["grad/" + k]
        tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(p.grad.numpy(), ref, rtol=0, atol=tol, err_msg=k)
    gn = torch.nn.utils.clip_grad_norm_(net.parameters(), O.GRAD_CLIP)
    assert abs(float(gn) - float(g["grad_norm"])) < 1e-5
    opt.step()
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), g["after_step/" + k], rtol=0, atol=2e-6, err_msg=k)


def test_criterion_matches_reference(golden):
    g = golden("g2_loss.npz")
    for tag in ("mixed", "allneg", "allpos"):
        l0, t = torch.from_numpy(g[f"{tag}/logits"]), torch.from_numpy(g[f"{tag}/targets"])
        for stage in ("main", "finetune"):
            crit = O.build_criterion(O.default_args(stage=stage), O.ComboLoss(), O.EdgeLoss())
            l = l0.clone().requires_grad_(True)
            v = crit(l, t)
            v.backward()
            assert abs(v.item() - float(g[f"{tag}/{stage}/loss"])) < 1e-6, (tag, stage)
            np.testing.assert_allclose(l.grad.numpy(), g[f"{tag}/{stage}/dlogits"], rtol=0, atol=1e-9)
        assert abs((1 - O.DiceLoss()(l0, t).item()) - float(g[f"{tag}/dice_eval"])) < 1e-6
        assert abs(O.iou_score(l0, t) - float(g[f"{tag}/iou"])) < 1e-6


def test_aspp_nondefault_rates(golden):
    g = golden("g3_aspp_rates.npz")
    aspp = O.ASPP(16, 32, rates=(2, 5, 9))
    aspp.load_state_dict(_sd(g, "init/"), strict=True)
    x = torch.from_numpy(g["x"])
    aspp.eval()
    with torch.no_grad():
        np.testing.assert_allclose(aspp(x).numpy(), g["eval_out"], rtol=0, atol=2e-6)
    aspp.train()
    aspp.project[3].p = 0.0
    with torch.no_grad():
        np.testing.assert_allclose(aspp(x).numpy(), g["train_out"], rtol=0, atol=5e-6)


def test_trained_weights_eval_and_metrics(golden):
    g = golden("g4_trained_c8_128.npz")
    net = O.AttentionASPPUNet(base_c=8)
    net.load_state_dict(_sd(g, "sd/"), strict=True)
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    d, i = O.evaluate(net, [(x[:4], y[:4]), (x[4:], y[4:])], torch.device("cpu"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-5
    assert abs(i - float(g["evaluate_iou"])) < 1e-5
    with torch.no_grad():
        lv = net(x)
    np.testing.assert_allclose(lv.numpy(), g["eval_logits"], rtol=0, atol=5e-5)
    masks = (torch.sigmoid(lv) > 0.5).numpy().astype(np.uint8)[:, 0] * 255
    gts = (y.numpy()[:, 0] > 0).astype(np.uint8) * 255
    dice = np.array([O.seg_dice(m, t) for m, t in zip(masks, gts)])
    iou = np.array([O.seg_iou(m, t) for m, t in zip(masks, gts)])
    np.testing.assert_allclose(dice, g["seg_dice"], atol=1e-3)
    np.testing.assert_allclose(iou, g["seg_iou"], atol=1e-3)
    assert dice.min() > 0.95  # the fixture has decisive masks
    with torch.inference_mode():
        np.testing.assert_allclose(O.predict_prob_tta(net, x[:1]), g["tta_prob0"], rtol=0, atol=1e-5)


def test_realistic_width_fixture_pins_oracle(golden):
    """base_c 16 / 256x256, weights trained with the reference (oracle/make_golden_c16.py): eval logits, evaluate() and the
    integer-count Dice of eval_segmentation_batch.py:41-49 from the restatement on the stored (bf16-representable) weights."""
    g = golden("g7_trained_c16_256.npz")
    sd = {k[8:]: torch.from_numpy(v.copy()).view(torch.bfloat16).float() for k, v in g.items() if k.startswith("sd_bf16/")}
    sd.update({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd_raw/")})
    net = O.AttentionASPPUNet(base_c=16)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    with torch.no_grad():
        lv = net(x[:4])
    ref = g["eval_logits"][:4].astype(np.float32)
    assert np.abs(lv.numpy() - ref).max() < 2e-3 * np.abs(ref).max() + 1e-3       # fixture logits are stored as fp16
    d, i = O.evaluate(net, [(x[:4], y[:4]), (x[4:], y[4:])], torch.device("cpu"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-5 and abs(i - float(g["evaluate_iou"])) < 1e-5


def test_metric_width_fixture_pins_oracle(golden):
    """base_c 48 / 512x512 -- the configuration bench.py measures -- with weights the reference trained briefly on synthetic
    phantoms (oracle/make_golden_c48.py): eval logits and evaluate() from the restatement on the stored weights."""
    g = golden("g9_trained_c48_512.npz")
    sd = {k[8:]: torch.from_numpy(v.copy()).view(torch.bfloat16).float() for k, v in g.items() if k.startswith("sd_bf16/")}
    sd.update({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd_raw/")})
    net = O.AttentionASPPUNet(base_c=48)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"]).float()
    with torch.no_grad():
        lv = net(x[:2])
    ref = g["eval_logits"][:2].astype(np.float32)
    assert np.abs(lv.numpy() - ref).max() < 2e-3 * np.abs(ref).max() + 1e-3       # fixture logits are stored as fp16
    assert float(g["undecided_share"]) < 0.01 and g["seg_dice"].min() > 0.9        # decisive masks
    d, i = O.evaluate(net, [(x[:2], y[:2]), (x[2:], y[2:])], torch.device("cpu"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-5 and abs(i - float(g["evaluate_iou"])) < 1e-5


def test_legacy_key_rename():
    sd = {"u4.att.W_g.0.weight": 1, "u4.att.W_x.1.bias": 2, "d1.0.block.0.weight": 3}
    out = O.rename_legacy_keys(sd)
    assert set(out) == {"u4.att.Wg.0.weight", "u4.att.Wx.1.bias", "d1.0.block.0.weight"}


def test_lr_schedule_closed_form():
    net = torch.nn.Linear(2, 2)
    for stage, epochs in (("main", 120), ("main", 10), ("finetune", 30)):
        opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
        sch = O.make_scheduler(opt, epochs, stage)
        for ep in range(epochs):
            assert abs(opt.param_groups[0]["lr"] - O.lr_at_epoch(ep, epochs, 3e-4, stage)) < 1e-10, (stage, ep)
            opt.step()
            sch.step()
