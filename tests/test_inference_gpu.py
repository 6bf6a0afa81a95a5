"""Inference-side extensions and edge cases on the GPU: sliding-window blend, hipGraph replay, all-negative
batches, batch-size / shape changes on one model instance, eval determinism."""
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as O


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import att_aspp_unet_amd as a
    return a


@pytest.fixture(scope="module")
def trained(A, golden):
    g = golden("g4_trained_c8_128.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}, strict=True)
    return m.cuda().eval(), g


def test_sliding_window_equals_windowed_forward_plus_cpu_blend(A, trained):
    m, g = trained
    x = torch.from_numpy(g["x"][:1]).cuda()                       # [1,1,128,128]
    win, stride = 64, 32
    got = A.predict_sliding_window(m, x, window=win, stride=stride).cpu().numpy()[0, 0]
    # CPU restatement of the blend over the SAME per-window logits (per-window forward == plain forward)
    n = (128 - win) // stride + 1
    acc, wsum = np.zeros((128, 128)), np.zeros((128, 128))
    c = 0.5 * (win - 1)
    yy, xx = np.mgrid[0:win, 0:win]
    gw = np.exp(-((yy - c) ** 2 + (xx - c) ** 2) / (2 * (0.125 * win) ** 2))
    with torch.no_grad():
        for iy in range(n):
            for ix in range(n):
                crop = x[:, :, iy * stride:iy * stride + win, ix * stride:ix * stride + win].contiguous()
                l = m(crop).cpu().numpy()[0, 0]
                acc[iy * stride:iy * stride + win, ix * stride:ix * stride + win] += gw * l
                wsum[iy * stride:iy * stride + win, ix * stride:ix * stride + win] += gw
    ref = acc / wsum
    assert np.abs(got - ref).max() < 2e-2 * np.abs(ref).max()      # batch-of-9 vs batch-of-1 windows: same weights, eval BN
    # one window covering the frame == plain forward
    with torch.no_grad():
        full = m(x)
    one = A.predict_sliding_window(m, x, window=128, stride=128)
    assert torch.allclose(one, full, atol=1e-6)


def test_graphed_forward_replays_identically(A, trained):
    m, g = trained
    x = torch.from_numpy(g["x"][:4]).cuda()
    with torch.no_grad():
        ref = m(x).clone()
    gf = A.GraphedForward(m, tuple(x.shape))
    for _ in range(3):
        out = gf(x)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # a different input through the same graph
    x2 = torch.flip(x, [0])
    with torch.no_grad():
        ref2 = m(x2).clone()
    assert torch.equal(gf(x2), ref2)


def test_all_negative_batch_and_finetune_weights_full_step(A):
    """No positive sample: dice / edge terms vanish (pipeline:227-231), the step must still be finite and match the oracle."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(3)
    ref = O.AttentionASPPUNet(base_c=8)
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().train()
    ref.train()
    ref.bridge.project[3].p = 0.0
    m.bridge.project[3].p = 0.0
    x, y = synth.make_frames(2, 64, seed=8, force_pattern="nn")
    assert float(y.sum()) == 0
    for stage in ("main", "finetune"):
        args = Namespace(stage=stage, edge_w=0.05, neg_bce_w=0.05)
        lr_ = O.build_criterion(args, O.ComboLoss(), O.EdgeLoss())(ref(x), y)
        le_ = A.build_criterion(args, A.ComboLoss(), A.EdgeLoss())(m(x.cuda()), y.cuda())
        assert abs(le_.item() - lr_.item()) < 2e-3 * abs(lr_.item()), stage
        le_.backward()
        gsum = sum(float(p.grad.abs().sum()) for p in m.parameters())
        assert np.isfinite(gsum) and gsum > 0


def test_one_model_many_shapes_and_modes(A, trained):
    m, g = trained
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        a8 = m(x)
        a3 = m(x[:3])                      # new batch size -> new plan, same weights
        a1 = m(x[:1, :, :64, :96])         # new spatial shape (fully convolutional)
        again = m(x)
    assert torch.equal(a8, again)                         # eval forward is deterministic (no atomics on this path)
    assert torch.allclose(a8[:3], a3, atol=1e-5)
    assert a1.shape == (1, 1, 64, 96) and bool(torch.isfinite(a1).all())
    from att_aspp_unet_amd._abi import AauError
    with pytest.raises(AauError):
        m(x[:, :, :60, :60])               # H, W must be multiples of 16: the documented error, not a bare assert
    m.train()
    with pytest.raises(AauError):
        m(x[:1])                           # training batch of 1: same restriction as the reference's pooled-branch BN
    m.eval()
