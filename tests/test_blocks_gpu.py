"""Standalone block execution (reference call signatures) against the golden ASPP fixture and the CPU oracle."""
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


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_aspp_nondefault_rates_matches_reference_golden(A, golden):
    g = golden("g3_aspp_rates.npz")
    m = A.ASPP(16, 32, rates=(2, 5, 9))
    m.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init/")}, strict=True)
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    m.eval()
    assert rel(m(x), g["eval_out"]) < 2e-2
    m.train()
    m.project[3].p = 0.0
    assert rel(m(x), g["train_out"]) < 3e-2


def test_convbnrelu_gate_upblock_match_oracle(A):
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(6)
    # ConvBNReLU
    ro, m = O.ConvBNReLU(16, 24), A.ConvBNReLU(16, 24)
    m.load_state_dict(ro.state_dict())
    x = torch.randn(3, 16, 24, 40, generator=g)
    for mode in ("train", "eval"):
        getattr(ro, mode)(); getattr(m, mode)()
        with torch.no_grad():
            ref = ro(x)
        assert rel(m.cuda()(x.cuda()), ref) < 2e-2, mode
    assert rel(m.block[1].running_mean, ro.block[1].running_mean) < 2e-2
    # AttentionGate
    ro, m = O.AttentionGate(32, 32, 16), A.AttentionGate(32, 32, 16)
    m.load_state_dict(ro.state_dict())
    gg, xx = torch.randn(2, 32, 16, 16, generator=g), torch.randn(2, 32, 16, 16, generator=g)
    for mode in ("train", "eval"):
        getattr(ro, mode)(); getattr(m, mode)()
        with torch.no_grad():
            ref = ro(gg, xx)
        assert rel(m.cuda()(gg.cuda(), xx.cuda()), ref) < 2e-2, mode
    # UpBlock with and without attention
    for use_att in (True, False):
        ro, m = O.UpBlock(32, 16, use_att), A.UpBlock(32, 16, use_att)
        m.load_state_dict(ro.state_dict())
        gg, xx = torch.randn(2, 32, 8, 8, generator=g), torch.randn(2, 16, 16, 16, generator=g)
        ro.eval(); m.eval()
        with torch.no_grad():
            ref = ro(gg, xx)
        assert rel(m.cuda()(gg.cuda(), xx.cuda()), ref) < 3e-2, use_att
