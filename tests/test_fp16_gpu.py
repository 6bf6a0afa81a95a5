"""fp16 inference (BASELINE config 5 names fp16; the reference's GPU predict path runs under torch.cuda.amp.autocast,
pipeline:320,437): the same kernels built for IEEE half (libaau_f16.so), selected per model with set_precision."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import att_aspp_unet_amd as A
    return A


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _g7(A, golden):
    g = golden("g7_trained_c16_256.npz")
    sd = {k[8:]: torch.from_numpy(v.copy()).view(torch.bfloat16).float() for k, v in g.items() if k.startswith("sd_bf16/")}
    sd.update({k[7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith("sd_raw/")})
    m = A.AttentionASPPUNet(base_c=16)
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), g


def test_fp16_eval_forward_is_closer_to_the_fp32_reference_than_bf16(A, golden):
    """Trained base_c 16 / 256x256 fixture (reference logits, fp32): half precision has 3 more mantissa bits than
    bfloat16, so its logits sit closer to the fp32 reference; Dice / IoU agree within 1e-3 either way."""
    m, g = _g7(A, golden)
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    ref = g["eval_logits"].astype(np.float32)
    with torch.no_grad():
        l_bf = m(x)
        m.set_precision("fp16")
        l_h = m(x)
        l_h2 = m(x)
    assert torch.equal(l_h, l_h2)
    e_bf, e_h = rel(l_bf, ref), rel(l_h, ref)
    print(dict(bf16=e_bf, fp16=e_h))
    assert e_h < 4e-3 and e_h < 0.5 * e_bf, (e_h, e_bf)
    d, i = A.evaluate(m, [(x[:4], y[:4]), (x[4:], y[4:])], torch.device("cuda"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-3 and abs(i - float(g["evaluate_iou"])) < 1e-3, (d, i)
    masks = (torch.sigmoid(l_h) > 0.5).to(torch.uint8)[:, 0]
    for k in range(8):
        assert abs(int(masks[k].sum()) - int(g["mask_counts"][k])) <= max(4, 0.001 * int(g["mask_counts"][k])), k
    tta = A.predict_prob_tta(m, x[:1])
    assert np.abs(tta - g["tta_prob0"].astype(np.float32)).max() < 5e-3
    # back to bf16: the first result again, bit for bit (plans of both precisions coexist)
    m.set_precision("bf16")
    with torch.no_grad():
        assert torch.equal(m(x), l_bf)


def test_fp16_is_inference_only_and_fails_loudly_for_training(A):
    m = A.AttentionASPPUNet(base_c=8).cuda().set_precision("fp16")
    m.train()
    with pytest.raises(A.AauError, match="inference precision"):
        m(torch.rand(2, 1, 32, 32, device="cuda"))
    with pytest.raises(A.AauError, match="unknown precision"):
        m.set_precision("fp8")
    # training in bf16 keeps its optimizer state across an fp16 evaluation in between
    m.set_precision("bf16")
    opt = A.FusedAdamW(m, lr=1e-3)
    crit = A.build_criterion(__import__("argparse").Namespace(stage="main", neg_bce_w=1.0, edge_w=0.05), A.ComboLoss(), A.EdgeLoss())
    from att_aspp_unet_amd import synth
    x, y = synth.make_frames(4, 32, seed=3)
    x, y = x.cuda(), y.cuda()

    def one():
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        return float(loss)
    l0 = one()
    m.eval().set_precision("fp16")
    with torch.no_grad():
        assert bool(torch.isfinite(m(x)).all())
    m.train().set_precision("bf16")
    l1 = one()
    assert int(m.engine.store.step_dev.item()) == 2 and l1 < l0 * 1.05


def test_config5_sliding_window_in_fp16_matches_the_fp32_oracle(A):
    """1 x 1 x 1024 x 1024, 9 windows of 512 at stride 256, rates (6, 12, 18, 24), hipGraph -- in fp16 as BASELINE.json
    states it; the blended logits against the CPU oracle's fp32 forward of the same windows (base_c 16 keeps the oracle
    to seconds)."""
    torch.manual_seed(11)
    rates = (6, 12, 18, 24)
    ref = O.AttentionASPPUNet(base_c=16, rates=rates)
    ref.train()
    with torch.no_grad():
        for s in range(2):
            ref(torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(s)))
    ref.eval()
    m = A.AttentionASPPUNet(base_c=16, rates=rates)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda().eval().set_precision("fp16")
    big = torch.rand(1, 1, 1024, 1024, generator=torch.Generator().manual_seed(4))
    yy, xx = torch.meshgrid(torch.arange(1024.), torch.arange(1024.), indexing="ij")
    for cy, cx, r in ((300, 280, 120), (700, 760, 180)):
        big[0, 0] += 0.5 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))
    big = big.clamp(0, 1)
    win, stride = 512, 256
    gf9 = A.GraphedForward(m, (9, 1, win, win))
    out = A.predict_sliding_window(m, big.cuda(), win, stride, forward=gf9).cpu().numpy()[0, 0]
    acc, wsum = np.zeros((1024, 1024)), np.zeros((1024, 1024))
    c = 0.5 * (win - 1)
    y2, x2 = np.mgrid[0:win, 0:win]
    gw = np.exp(-((y2 - c) ** 2 + (x2 - c) ** 2) / (2 * (0.125 * win) ** 2))
    with torch.no_grad():
        for iy in range(3):
            for ix in range(3):
                crop = big[:, :, iy * stride:iy * stride + win, ix * stride:ix * stride + win].contiguous()
                acc[iy * stride:iy * stride + win, ix * stride:ix * stride + win] += gw * ref(crop).numpy()[0, 0]
                wsum[iy * stride:iy * stride + win, ix * stride:ix * stride + win] += gw
    want = acc / wsum
    err = float(np.abs(out - want).max() / np.abs(want).max())
    print("config5 fp16 vs fp32 oracle", err)
    assert err < 5e-3, err
