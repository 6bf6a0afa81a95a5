"""GPU parity of single HIP kernels (through the C ABI) against CPU fp32 references.

Inputs are rounded to bf16 first, so the only differences left are the fp32
accumulation order and the final bf16 store (relative 2^-9 per element).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kernels_ref as R


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def pack_fwd(w_oihw, cpad):
    """[O][I][kh][kw] fp32 -> bf16 [O][T][Cpad] (what aau_pack_weights produces for the forward GEMM)."""
    O, I, kh, kw = w_oihw.shape
    out = torch.zeros(O, kh * kw, cpad)
    out[:, :, :I] = w_oihw.permute(0, 2, 3, 1).reshape(O, kh * kw, I)
    return out.to(torch.bfloat16)


def pack_dgrad(w_oihw, cpad):
    """-> bf16 [I][T flipped][Cpad(O)]"""
    O, I, kh, kw = w_oihw.shape
    out = torch.zeros(I, kh * kw, cpad)
    out[:, :, :O] = w_oihw.flip(2, 3).permute(1, 2, 3, 0).reshape(I, kh * kw, O)
    return out.to(torch.bfloat16)


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, dil
    (2, 16, 16, 48, 48, 3, 1),
    (1, 24, 40, 96, 96, 3, 1),
    (2, 32, 32, 64, 128, 3, 6),
    (2, 8, 8, 192, 96, 1, 1),
    (1, 16, 16, 8, 16, 3, 1),
    (1, 32, 32, 128, 200, 3, 18),
    (3, 12, 20, 104, 56, 3, 2),
    # halo-tiled 3x3 kernel (H, W multiples of 16): narrow / wide tiles, channel tails, many chunks
    (2, 32, 48, 96, 96, 3, 1),
    (1, 16, 32, 200, 104, 3, 1),
    (2, 16, 16, 64, 48, 3, 1),
    (1, 48, 16, 8, 24, 3, 1),
    (1, 16, 16, 384, 192, 3, 1),
]


@pytest.mark.parametrize("case", [(5, 256, 256, 48, 48), (4, 256, 272, 96, 48), (4, 272, 256, 48, 96), (5, 256, 256, 8, 56),
                                  # >= 2048 patches, Cin <= 64, <= 48 output channels: two patch streams per workgroup
                                  (8, 256, 256, 48, 48), (8, 256, 256, 64, 40), (16, 256, 128, 8, 48)])
def test_resident_weight_3x3_path(ops, case):
    """>= 1024 patches of 16x16 and a small weight matrix select the persistent resident-weight kernels."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    ref = torch.relu(R.conv_fwd(x, w) + bias)
    raw = R.conv_fwd(x, w).reshape(-1, Cout)
    cpad = ops.cpad_of(Cin)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout + 8, 3, 3, 1, 1, 1, cpad, relu=1)
    out = torch.zeros(N, H, W, Cout + 8, dtype=torch.bfloat16, device="cuda")
    stats = ops.stats_buffer(Cout)
    ops.conv_igemm(d, dev(x.to(torch.bfloat16)), dev(pack_fwd(w, cpad)), out, bias=dev(bias), stats=stats)
    torch.cuda.synchronize()
    got = out.cpu()
    assert rel_err(got[..., :Cout], ref) < 6e-3
    assert float(got[..., Cout:].abs().max()) == 0
    s = ops.stats_totals(stats, Cout).float().cpu()
    assert float((s[0] - raw.sum(0)).abs().max()) < 2e-3 * float(raw.abs().sum(0).max())
    assert torch.allclose(s[1], (raw ** 2).sum(0), rtol=2e-3)


@pytest.mark.parametrize("case", CONV_CASES)
def test_igemm_forward_and_stats(ops, case):
    N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    ref = R.conv_fwd(x, w, dil)
    cpad = ops.cpad_of(Cin)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil, cpad)
    xd, wd = dev(x.to(torch.bfloat16)), dev(pack_fwd(w, cpad))
    out = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = ops.stats_buffer(Cout)
    ops.conv_igemm(d, xd, wd, out, stats=stats)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 6e-3
    s = ops.stats_totals(stats, Cout).float().cpu()
    flat = ref.reshape(-1, Cout)
    assert float((s[0] - flat.sum(0)).abs().max()) < 2e-3 * float(flat.abs().sum(0).max())
    assert torch.allclose(s[1], (flat ** 2).sum(0), rtol=2e-3)


def test_igemm_epilogue_bias_affine_relu_accumulate_pitch(ops):
    N, H, W, Cin, Cout = 2, 8, 8, 32, 48
    g = torch.Generator().manual_seed(5)
    xw = R.bf16_round(torch.randn(N, H, W, Cin + 16, generator=g))  # source = channel slice [8:40] of a wider tensor
    x = xw[..., 8:8 + Cin]
    w = R.bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5)
    bias, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    prev = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = torch.relu((R.conv_fwd(x, w) + bias) * scale + shift + prev)
    cpad = ops.cpad_of(Cin)
    xd = dev(xw.to(torch.bfloat16))
    wide = torch.zeros(N, H, W, Cout + 24, dtype=torch.bfloat16, device="cuda")
    wide[..., 16:16 + Cout] = dev(prev.to(torch.bfloat16))
    d = ops.conv_desc(N, H, W, Cin, Cin + 16, H, W, Cout, Cout + 24, Cpad=cpad, accumulate=1, relu=1)
    ops.conv_igemm(d, xd[..., 8:], dev(pack_fwd(w, cpad)), wide[..., 16:], bias=dev(bias), scale=dev(scale), shift=dev(shift))
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., 16:16 + Cout], ref) < 8e-3
    assert float(got[..., :16].abs().max()) == 0 and float(got[..., 16 + Cout:].abs().max()) == 0  # neighbours untouched


@pytest.mark.parametrize("case", [(4, 256, 256, 96, 48), (4, 256, 256, 40, 104), (5, 256, 240, 192, 96)])
def test_resident_weight_1x1_path(ops, case):
    """>= 1024 patches of 16x16 and a small weight matrix: 1x1 convs take the persistent resident-weight kernel.
    Forward with BN statistics, strided source / destination pitches, then bias + affine + ReLU + accumulate."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    xw = R.bf16_round(torch.randn(N, H, W, Cin + 8, generator=g))
    x = xw[..., 8:]
    w = R.bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5)
    raw = R.conv_fwd(x, w)
    cpad = ops.cpad_of(Cin)
    xd, wd = dev(xw.to(torch.bfloat16)), dev(pack_fwd(w, cpad))
    wide = torch.zeros(N, H, W, Cout + 16, dtype=torch.bfloat16, device="cuda")
    d = ops.conv_desc(N, H, W, Cin, Cin + 8, H, W, Cout, Cout + 16, Cpad=cpad)
    stats = ops.stats_buffer(Cout)
    ops.conv_igemm(d, xd[..., 8:], wd, wide[..., 8:], stats=stats)
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., 8:8 + Cout], raw) < 6e-3
    assert float(got[..., :8].abs().max()) == 0 and float(got[..., 8 + Cout:].abs().max()) == 0
    flat = raw.reshape(-1, Cout)
    st = ops.stats_totals(stats, Cout).float().cpu()
    assert float((st[0] - flat.sum(0)).abs().max()) < 2e-3 * float(flat.abs().sum(0).max())
    assert torch.allclose(st[1], (flat ** 2).sum(0), rtol=2e-3)
    # epilogue: bias, affine, ReLU, accumulate into the existing destination
    bias, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    prev = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = torch.relu((raw + bias) * scale + shift + prev)
    wide.zero_()
    wide[..., 8:8 + Cout] = dev(prev.to(torch.bfloat16))
    d2 = ops.conv_desc(N, H, W, Cin, Cin + 8, H, W, Cout, Cout + 16, Cpad=cpad, accumulate=1, relu=1)
    ops.conv_igemm(d2, xd[..., 8:], wd, wide[..., 8:], bias=dev(bias), scale=dev(scale), shift=dev(shift))
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., 8:8 + Cout], ref) < 8e-3
    assert float(got[..., :8].abs().max()) == 0 and float(got[..., 8 + Cout:].abs().max()) == 0


def test_resident_weight_convT_shuffle_path(ops):
    """ConvTranspose2d(2,2) forward at a resolution that takes the resident-weight 1x1 kernel (pixel-shuffle store)."""
    N, H, W, Ci, Co = 4, 256, 256, 96, 48
    g = torch.Generator().manual_seed(31)
    x = R.bf16_round(torch.randn(N, H, W, Ci, generator=g))
    w = R.bf16_round(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)  # IOHW
    b = torch.randn(Co, generator=g)
    ref = R.convT_fwd(x, w, b)
    cpad = ops.cpad_of(Ci)
    wp = torch.zeros(4 * Co, 1, cpad)
    wp[:, 0, :Ci] = w.permute(2, 3, 1, 0).reshape(4 * Co, Ci)
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, 4 * Co, 2 * Co, Cpad=cpad, shuffle2x2=1)
    wide = torch.zeros(N, 2 * H, 2 * W, 2 * Co, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(d, dev(x.to(torch.bfloat16)), dev(wp.to(torch.bfloat16)), wide[..., Co:], bias=dev(b))
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., Co:], ref) < 6e-3
    assert float(got[..., :Co].abs().max()) == 0


def test_halo3x3_epilogue_bias_affine_relu_accumulate_pitch(ops):
    N, H, W, Cin, Cout = 2, 16, 32, 40, 56
    g = torch.Generator().manual_seed(55)
    xw = R.bf16_round(torch.randn(N, H, W, Cin + 8, generator=g))
    x = xw[..., 8:]
    w = R.bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
    bias, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    prev = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = torch.relu((R.conv_fwd(x, w) + bias) * scale + shift + prev)
    cpad = ops.cpad_of(Cin)
    wide = torch.zeros(N, H, W, Cout + 16, dtype=torch.bfloat16, device="cuda")
    wide[..., 8:8 + Cout] = dev(prev.to(torch.bfloat16))
    d = ops.conv_desc(N, H, W, Cin, Cin + 8, H, W, Cout, Cout + 16, 3, 3, 1, 1, 1, cpad, accumulate=1, relu=1)
    stats = ops.stats_buffer(Cout)
    ops.conv_igemm(d, dev(xw.to(torch.bfloat16))[..., 8:], dev(pack_fwd(w, cpad)), wide[..., 8:], bias=dev(bias),
                   scale=dev(scale), shift=dev(shift), stats=stats)
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., 8:8 + Cout], ref) < 8e-3
    assert float(got[..., :8].abs().max()) == 0 and float(got[..., 8 + Cout:].abs().max()) == 0
    raw = R.conv_fwd(x, w).reshape(-1, Cout)
    assert torch.allclose(ops.stats_totals(stats, Cout).float().cpu()[1], (raw ** 2).sum(0), rtol=2e-3)


@pytest.mark.parametrize("case", [(2, 16, 16, 48, 48, 3, 1), (1, 32, 32, 64, 96, 3, 6), (2, 8, 8, 96, 192, 1, 1),
                                  (2, 32, 16, 96, 48, 3, 1), (1, 16, 48, 40, 136, 3, 1)])
def test_igemm_dgrad(ops, case):
    N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator().manual_seed(17)
    dy = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5)
    ref = R.conv_dgrad(dy, w, (H, W), dil)
    cpad = ops.cpad_of(Cout)
    d = ops.conv_desc(N, H, W, Cout, Cout, H, W, Cin, Cin, k, k, 1, dil * (k // 2), dil, cpad)
    out = torch.empty(N, H, W, Cin, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(d, dev(dy.to(torch.bfloat16)), dev(pack_dgrad(w, cpad)), out)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 6e-3


def test_convT_forward_shuffle_and_dgrad(ops):
    N, H, W, Ci, Co = 2, 8, 12, 64, 32
    g = torch.Generator().manual_seed(23)
    x = R.bf16_round(torch.randn(N, H, W, Ci, generator=g))
    w = R.bf16_round(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)  # IOHW
    b = torch.randn(Co, generator=g)
    ref = R.convT_fwd(x, w, b)
    cpad = ops.cpad_of(Ci)
    wp = torch.zeros(4 * Co, 1, cpad)  # forward GEMM rows: (pos, co) x cin
    wp[:, 0, :Ci] = w.permute(2, 3, 1, 0).reshape(4 * Co, Ci)
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, 4 * Co, Co + 8, Cpad=cpad, shuffle2x2=1)
    wide = torch.zeros(N, 2 * H, 2 * W, Co + 8, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(d, dev(x.to(torch.bfloat16)), dev(wp.to(torch.bfloat16)), wide[..., 8:], bias=dev(b))
    torch.cuda.synchronize()
    assert rel_err(wide.cpu()[..., 8:], ref) < 6e-3
    # data gradient: stride-2 2x2 gather of dy, rows = cin, K = (pos, co)
    dy = R.bf16_round(torch.randn(N, 2 * H, 2 * W, Co, generator=g))
    refd = R.convT_dgrad(dy, w)
    cpd = ops.cpad_of(Co)
    wdg = torch.zeros(Ci, 4, cpd)
    wdg[:, :, :Co] = w.permute(0, 2, 3, 1).reshape(Ci, 4, Co)
    dd = ops.conv_desc(N, 2 * H, 2 * W, Co, Co, H, W, Ci, Ci, 2, 2, 2, 0, 1, cpd)
    out = torch.empty(N, H, W, Ci, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(dd, dev(dy.to(torch.bfloat16)), dev(wdg.to(torch.bfloat16)), out)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), refd) < 6e-3


WGRAD_CASES = [
    (2, 16, 16, 48, 48, 3, 1),
    (1, 24, 40, 96, 48, 3, 1),
    (2, 16, 16, 48, 96, 3, 1),
    (2, 32, 32, 96, 192, 3, 6),
    (2, 8, 8, 192, 96, 1, 1),
    (1, 16, 16, 8, 16, 3, 1),
    (3, 12, 20, 104, 56, 3, 2),
    # all-taps 3x3 kernel (H % 8 == 0, W % 16 == 0)
    (2, 32, 48, 96, 96, 3, 1),
    (1, 16, 32, 200, 104, 3, 1),
    (3, 8, 16, 48, 144, 3, 1),
    (1, 64, 64, 24, 40, 3, 1),
    # row-reuse 3x3 kernel (csrc/wgrad3x3r.hip): <3,4> = 48 q x 64 c tiles, <6,2> = 96 q x 32 c tiles, ragged tails
    (2, 16, 32, 64, 48, 3, 1),
    (1, 32, 32, 192, 96, 3, 1),
    (2, 16, 16, 128, 144, 3, 1),
    (1, 24, 48, 320, 40, 3, 1),
    (2, 16, 32, 32, 96, 3, 1),
    (1, 32, 16, 96, 192, 3, 1),
    (1, 16, 16, 160, 288, 3, 1),
    (5, 8, 16, 384, 384, 3, 1),
]


def _wgrad_ws(ops, d, mode):
    """mode 'atomic': fp32-atomic split-K; 'slab': deterministic split-K through a NaN-poisoned workspace."""
    if mode == "atomic":
        return None
    n = ops.conv_wgrad_ws_bytes(d)
    assert n > 0 and n % 16 == 0
    return torch.full((n // 4,), float("nan"), device="cuda")


@pytest.mark.parametrize("mode", ["slab", "atomic"])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad(ops, case, mode):
    N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    dy = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = R.conv_wgrad(x, dy, (Cout, Cin, k, k), dil)  # OIHW
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil)
    dw = torch.zeros(Cout, k * k, Cin, device="cuda")
    ws = _wgrad_ws(ops, d, mode)
    xd, dyd = dev(x.to(torch.bfloat16)), dev(dy.to(torch.bfloat16))
    ops.conv_wgrad(d, xd, dyd, dw, ws)
    torch.cuda.synchronize()
    got = dw.cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-3
    if mode == "slab":
        # accumulates into dw (like the atomic form) and is bitwise reproducible
        dw2 = dw.clone()
        ops.conv_wgrad(d, xd, dyd, dw2, ws)
        dw3 = torch.zeros_like(dw)
        ops.conv_wgrad(d, xd, dyd, dw3, ws)
        assert torch.equal(dw3, dw)
        assert rel_err(dw2.cpu(), 2 * dw.cpu()) < 1e-6


@pytest.mark.parametrize("mode", ["slab", "atomic"])
def test_wgrad_convT(ops, mode):
    N, H, W, Ci, Co = 2, 8, 12, 64, 32
    g = torch.Generator().manual_seed(29)
    x = R.bf16_round(torch.randn(N, H, W, Ci, generator=g))
    dy = R.bf16_round(torch.randn(N, 2 * H, 2 * W, Co, generator=g))
    w = torch.zeros(Ci, Co, 2, 2)
    ref = R.convT_wgrad(x, dy, w)  # [Ci][Co][2][2]
    d = ops.conv_desc(N, 2 * H, 2 * W, Co, Co, H, W, Ci, Ci, 2, 2, 2, 0, 1)
    dw = torch.zeros(Ci, 4, Co, device="cuda")
    ops.conv_wgrad(d, dev(dy.to(torch.bfloat16)), dev(x.to(torch.bfloat16)), dw, _wgrad_ws(ops, d, mode))
    torch.cuda.synchronize()
    got = dw.cpu().reshape(Ci, 2, 2, Co).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-3


# ---- grouped big-tile weight gradient (aau_conv_wgrad_group, csrc/wgradL.hip) ----
def _group_problem(ops, N, H, W, Cin, Cout, k, dil, seed, x=None, xpitch=None, zpitch=None):
    g = torch.Generator().manual_seed(seed)
    if x is None:
        x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    dy = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = R.conv_wgrad(x, dy, (Cout, Cin, k, k), dil)
    xp, zp = xpitch or Cin, zpitch or Cout
    xd = torch.zeros(N, H, W, xp, dtype=torch.bfloat16, device="cuda")
    xd[..., xp - Cin:] = x.to(torch.bfloat16).cuda()                 # the tensor is a channel slice of a wider buffer
    zd = torch.zeros(N, H, W, zp, dtype=torch.bfloat16, device="cuda")
    zd[..., :Cout] = dy.to(torch.bfloat16).cuda()
    d = ops.conv_desc(N, H, W, Cin, xp, H, W, Cout, zp, k, k, 1, dil * (k // 2), dil)
    dw = torch.zeros(Cout, k * k, Cin, device="cuda")
    return dict(d=d, src=xd[..., xp - Cin:], dz=zd[..., :Cout], dw=dw, ref=ref, x=x, k=k)


def _check_group(ops, probs, tol=2e-3):
    ops.conv_wgrad_group([p["d"] for p in probs], [p["src"] for p in probs], [p["dz"] for p in probs], [p["dw"] for p in probs])
    torch.cuda.synchronize()
    for p in probs:
        Cout, T, Cin = p["dw"].shape
        got = p["dw"].cpu().reshape(Cout, p["k"], p["k"], Cin).permute(0, 3, 1, 2)
        assert rel_err(got, p["ref"]) < tol, (Cout, Cin, p["k"], rel_err(got, p["ref"]))
    # dw += : a second call doubles the result, bitwise reproducibly
    first = [p["dw"].clone() for p in probs]
    ops.conv_wgrad_group([p["d"] for p in probs], [p["src"] for p in probs], [p["dz"] for p in probs], [p["dw"] for p in probs])
    torch.cuda.synchronize()
    for p, f in zip(probs, first):
        assert rel_err(p["dw"].cpu(), 2 * f.cpu()) < 1e-6
        z = torch.zeros_like(f)
        ops.conv_wgrad_group([p["d"]], [p["src"]], [p["dz"]], [z])
        assert torch.equal(z, f)


def test_wgrad_group_small_ragged_shapes(ops):
    """Tile tails in both channel dimensions, dilation skipping, pitched operands, a strided 2x2 problem."""
    probs = [
        _group_problem(ops, 2, 16, 32, 104, 200, 3, 2, 1, xpitch=136, zpitch=208),
        _group_problem(ops, 2, 16, 32, 200, 104, 1, 1, 2),
        _group_problem(ops, 1, 32, 64, 96, 96, 3, 6, 3),                       # two 32-pixel segments per row
        _group_problem(ops, 4, 16, 16, 192, 384, 1, 1, 4, zpitch=768),         # linear: W need not be a multiple of 32
    ]
    _check_group(ops, probs)
    # ConvTranspose2d(2,2) weight gradient as the engine states it: "x" = fine-grid gradient, "dz" = coarse input
    N, H, W, Ci, Co = 1, 32, 32, 128, 96
    g = torch.Generator().manual_seed(31)
    x = R.bf16_round(torch.randn(N, H, W, Ci, generator=g))
    dy = R.bf16_round(torch.randn(N, 2 * H, 2 * W, Co, generator=g))
    ref = R.convT_wgrad(x, dy, torch.zeros(Ci, Co, 2, 2))
    d = ops.conv_desc(N, 2 * H, 2 * W, Co, Co, H, W, Ci, Ci, 2, 2, 2, 0, 1)
    dw = torch.zeros(Ci, 4, Co, device="cuda")
    ops.conv_wgrad_group([d], [dev(dy.to(torch.bfloat16))], [dev(x.to(torch.bfloat16))], [dw])
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().reshape(Ci, 2, 2, Co).permute(0, 3, 1, 2), ref) < 2e-3


def test_wgrad_group_at_the_real_bridge_shape(ops):
    """8 x 32 x 32, 384 -> 768: 1x1 + dilation 6 / 12 / 18 on ONE input, and the 3840 -> 768 projection, as one launch
    (what engine.py records for the ASPP bridge at base_c 48)."""
    N, H, W, Ci, Co = 8, 32, 32, 384, 768
    first = _group_problem(ops, N, H, W, Ci, Co, 1, 1, 11, zpitch=5 * Co)
    probs = [first]
    for i, dil in enumerate((6, 12, 18)):
        probs.append(_group_problem(ops, N, H, W, Ci, Co, 3, dil, 12 + i, x=first["x"], zpitch=5 * Co))
        probs[-1]["src"] = first["src"]                                       # the branches share their input tensor
    probs.append(_group_problem(ops, N, H, W, 5 * Co, Co, 1, 1, 20))
    assert ops.conv_wgrad_group_ok([p["d"] for p in probs])
    _check_group(ops, probs)


# ---- two-plane ("planar concat") operands (aau.h: src_split_c / dst_split_c) ----
def test_two_plane_operands_equal_the_interleaved_form(ops):
    """u1.conv.0 of the decoder at level 1 (pipeline:108-109): its input cat([skip, up]) and the gradient of that cat are
    kept as two dense [M][48] planes.  Forward, data-gradient (two-plane destination) and weight-gradient (two-plane
    source) give bit-identical results to the interleaved [M][96] layout."""
    N, H, W, Cs, Co = 4, 256, 256, 48, 48
    M = N * H * W
    g = torch.Generator().manual_seed(5)
    cat = torch.randn(M, 2 * Cs, generator=g).to(torch.bfloat16).cuda()
    planes = torch.stack([cat[:, :Cs], cat[:, Cs:]]).contiguous()                 # [2][M][Cs]
    w = R.bf16_round(torch.randn(Co, 2 * Cs, 3, 3, generator=g) / (2 * Cs * 9) ** 0.5)
    cp = ops.cpad_of(2 * Cs)
    wp = dev(pack_fwd(w, cp))
    # forward
    d_i = ops.conv_desc(N, H, W, 2 * Cs, 2 * Cs, H, W, Co, Co, 3, 3, 1, 1, 1, cp)
    d_p = ops.conv_desc(N, H, W, 2 * Cs, Cs, H, W, Co, Co, 3, 3, 1, 1, 1, cp, src_split=(Cs, M * Cs))
    assert ops.conv_split_ok(d_p, 0)
    out_i, out_p = (torch.empty(M, Co, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    st_i, st_p = ops.stats_buffer(Co), ops.stats_buffer(Co)
    ops.conv_igemm(d_i, cat, wp, out_i, stats=st_i)
    ops.conv_igemm(d_p, planes, wp, out_p, stats=st_p)
    torch.cuda.synchronize()
    assert torch.equal(out_i, out_p) and torch.equal(st_i, st_p)
    # data gradient: 48 -> 96 channels, destination in two planes
    dz = torch.randn(M, Co, generator=g).to(torch.bfloat16).cuda()
    cpd = ops.cpad_of(Co)
    wd = dev(pack_dgrad(w, cpd))
    dd_i = ops.conv_desc(N, H, W, Co, Co, H, W, 2 * Cs, 2 * Cs, 3, 3, 1, 1, 1, cpd)
    dd_p = ops.conv_desc(N, H, W, Co, Co, H, W, 2 * Cs, Cs, 3, 3, 1, 1, 1, cpd, dst_split=(Cs, M * Cs))
    assert ops.conv_split_ok(dd_p, 0)
    din_i = torch.empty(M, 2 * Cs, dtype=torch.bfloat16, device="cuda")
    din_p = torch.empty(2, M, Cs, dtype=torch.bfloat16, device="cuda")
    sa, sb = ops.stats_buffer(2 * Cs), ops.stats_buffer(2 * Cs)
    ops.conv_igemm(dd_i, dz, wd, din_i, stats=sa)
    ops.conv_igemm(dd_p, dz, wd, din_p, stats=sb)
    torch.cuda.synchronize()
    assert torch.equal(din_p[0], din_i[:, :Cs]) and torch.equal(din_p[1], din_i[:, Cs:]) and torch.equal(sa, sb)
    # weight gradient with a two-plane source
    dw_i = torch.zeros(Co, 9, 2 * Cs, device="cuda")
    dw_p = torch.zeros_like(dw_i)
    wi = ops.conv_desc(N, H, W, 2 * Cs, 2 * Cs, H, W, Co, Co, 3, 3, 1, 1, 1)
    wpd = ops.conv_desc(N, H, W, 2 * Cs, Cs, H, W, Co, Co, 3, 3, 1, 1, 1, src_split=(Cs, M * Cs))
    assert ops.conv_split_ok(wpd, 1)
    ops.conv_wgrad(wi, cat, dz, dw_i, _wgrad_ws(ops, wi, "slab"))
    ops.conv_wgrad(wpd, planes, dz, dw_p, _wgrad_ws(ops, wpd, "slab"))
    torch.cuda.synchronize()
    assert torch.equal(dw_i, dw_p)
    # a descriptor whose kernel cannot serve two planes is refused, not silently misread
    small = ops.conv_desc(1, 32, 32, 2 * Cs, Cs, 32, 32, Co, Co, 3, 3, 1, 1, 1, cp, src_split=(Cs, 32 * 32 * Cs))
    assert not ops.conv_split_ok(small, 0)
    with pytest.raises(Exception, match="two-plane"):
        ops.conv_igemm(small, planes, wp, out_p)


class launch_tags:
    """Kernel-variant tags of the launches made inside the block (the library's own per-launch records)."""

    def __enter__(self):
        from att_aspp_unet_amd import _abi
        self.abi = _abi
        _abi.prof_collect_launches()
        _abi.prof_enable(True)
        self.tags = []
        return self.tags

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.abi.prof_enable(False)
        self.tags.extend(r["tag"] for r in self.abi.prof_collect_launches())
        return False


WIDE_CASES = [
    # N, H, W, Cin, Cout, k, dil -- forced onto the 128 x 192 three-stage tile (AAU_IGEMM_WIDE=1)
    (2, 16, 16, 64, 192, 1, 1),       # one K-step
    (1, 8, 8, 320, 192, 1, 1),        # five chunks, M below one tile
    (1, 12, 20, 128, 384, 3, 2),      # ragged M (240 = 128 + 112), 18 steps, two channel tiles
    (2, 32, 32, 64, 192, 3, 18),      # tile-level tap skipping, one chunk per tap
    (1, 20, 12, 192, 576, 1, 1),      # three channel tiles
    # the bridge at its real shape (pipeline:67-83 at base_c 48): chosen by the dispatch rule itself
    (8, 32, 32, 384, 768, 3, 6),
    (8, 32, 32, 384, 768, 3, 12),
    (8, 32, 32, 384, 768, 3, 18),
    (8, 32, 32, 384, 768, 1, 1),
    (8, 32, 32, 3840, 768, 1, 1),
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_igemm_wide_tile_forward_and_stats(ops, case, monkeypatch):
    N, H, W, Cin, Cout, k, dil = case
    if N * H * W < 8192:
        monkeypatch.setenv("AAU_IGEMM_WIDE", "1")
    g = torch.Generator().manual_seed(sum(case))
    x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    ref = R.conv_fwd(x, w, dil)
    cpad = ops.cpad_of(Cin)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil, cpad)
    out = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = ops.stats_buffer(Cout)
    with launch_tags() as tags:
        ops.conv_igemm(d, dev(x.to(torch.bfloat16)), dev(pack_fwd(w, cpad)), out, stats=stats)
    assert any(t.startswith("igemm<64,192,0>") for t in tags), tags
    assert rel_err(out.cpu(), ref) < 6e-3
    s = ops.stats_totals(stats, Cout).float().cpu()
    flat = ref.reshape(-1, Cout)
    assert float((s[0] - flat.sum(0)).abs().max()) < 2e-3 * float(flat.abs().sum(0).max())
    assert torch.allclose(s[1], (flat ** 2).sum(0), rtol=2e-3)


def test_igemm_wide_tile_epilogues(ops, monkeypatch):
    """bias + affine + ReLU + accumulate into a channel slice, and the ConvTranspose pixel-shuffle store, on the wide tile."""
    monkeypatch.setenv("AAU_IGEMM_WIDE", "1")
    N, H, W, Cin, Cout = 2, 12, 12, 96, 192
    g = torch.Generator().manual_seed(77)
    xw = R.bf16_round(torch.randn(N, H, W, Cin + 16, generator=g))
    x = xw[..., 8:8 + Cin]
    w = R.bf16_round(torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5)
    bias, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    prev = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = torch.relu((R.conv_fwd(x, w) + bias) * scale + shift + prev)
    cpad = ops.cpad_of(Cin)
    wide = torch.zeros(N, H, W, Cout + 24, dtype=torch.bfloat16, device="cuda")
    wide[..., 16:16 + Cout] = dev(prev.to(torch.bfloat16))
    d = ops.conv_desc(N, H, W, Cin, Cin + 16, H, W, Cout, Cout + 24, Cpad=cpad, accumulate=1, relu=1)
    ops.conv_igemm(d, dev(xw.to(torch.bfloat16))[..., 8:], dev(pack_fwd(w, cpad)), wide[..., 16:], bias=dev(bias),
                   scale=dev(scale), shift=dev(shift))
    torch.cuda.synchronize()
    got = wide.cpu()
    assert rel_err(got[..., 16:16 + Cout], ref) < 8e-3
    assert float(got[..., :16].abs().max()) == 0 and float(got[..., 16 + Cout:].abs().max()) == 0
    # ConvTranspose2d(2,2): N = 4*Co = 192 columns, pixel-shuffle store into the upper half of a concat buffer
    N, H, W, Ci, Co = 2, 10, 14, 128, 48
    x = R.bf16_round(torch.randn(N, H, W, Ci, generator=g))
    wt = R.bf16_round(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)
    b = torch.randn(Co, generator=g)
    ref = R.convT_fwd(x, wt, b)
    cpad = ops.cpad_of(Ci)
    wp = torch.zeros(4 * Co, 1, cpad)
    wp[:, 0, :Ci] = wt.permute(2, 3, 1, 0).reshape(4 * Co, Ci)
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, 4 * Co, 2 * Co, Cpad=cpad, shuffle2x2=1)
    cat = torch.zeros(N, 2 * H, 2 * W, 2 * Co, dtype=torch.bfloat16, device="cuda")
    with launch_tags() as tags:
        ops.conv_igemm(d, dev(x.to(torch.bfloat16)), dev(wp.to(torch.bfloat16)), cat[..., Co:], bias=dev(b))
    assert any(t.startswith("igemm<64,192,0>") for t in tags), tags
    got = cat.cpu()
    assert rel_err(got[..., Co:], ref) < 6e-3
    assert float(got[..., :Co].abs().max()) == 0


# ---- grouped data gradient (aau_conv_igemm_group, csrc/igemm_group.hip) ----
def _dgrad_group_problem(ops, N, H, W, Cin, Cout, segs, seed, pitch_extra=0):
    g = torch.Generator().manual_seed(seed)
    descs, srcs, wpks, ref = [], [], [], 0
    cpad = ops.cpad_of(Cin)
    assert cpad == Cin
    for i, (k, dil) in enumerate(segs):
        xw = R.bf16_round(torch.randn(N, H, W, Cin + pitch_extra, generator=g))
        w = R.bf16_round(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
        ref = ref + R.conv_fwd(xw[..., :Cin], w, dil)
        descs.append(ops.conv_desc(N, H, W, Cin, Cin + pitch_extra, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil, cpad,
                                   accumulate=1 if i > 0 else 0))
        srcs.append(dev(xw.to(torch.bfloat16)))
        wpks.append(dev(pack_fwd(w, cpad)))
    return descs, srcs, wpks, ref


@pytest.mark.parametrize("nsplit", ["1", "2", "3", "4"])
@pytest.mark.parametrize("case", [
    (1, 12, 20, 64, 192, [(1, 1), (3, 2), (3, 5)]),         # ragged M (240), one chunk per tap
    (2, 16, 16, 128, 384, [(3, 6), (1, 1), (3, 12), (3, 18)]),  # taps skipped per tile, two channel tiles
    (1, 8, 8, 64, 192, [(1, 1), (1, 1)]),                   # two steps in all: some K-ranges are empty
])
def test_igemm_group_matches_the_sum_of_convolutions(ops, case, nsplit, monkeypatch):
    monkeypatch.setenv("AAU_GROUP_NSPLIT", nsplit)
    N, H, W, Cin, Cout, segs = case
    descs, srcs, wpks, ref = _dgrad_group_problem(ops, N, H, W, Cin, Cout, segs, seed=N + H + Cin + len(segs), pitch_extra=8)
    assert ops.conv_igemm_group_ok(descs)
    nws = ops.conv_igemm_group_ws_bytes(descs) // 4
    assert (nws == 0) == (nsplit == "1")
    ws = torch.full((max(nws, 4),), float("nan"), device="cuda")
    out = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm_group(descs, srcs, wpks, out, ws)
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), ref) < 6e-3
    # accumulate into an existing gradient; twice the same bits
    prev = R.bf16_round(torch.randn(N, H, W, Cout, generator=torch.Generator().manual_seed(3)))
    descs[0].accumulate = 1
    outs = []
    for _ in range(2):
        o = dev(prev.to(torch.bfloat16)).clone()
        ops.conv_igemm_group(descs, srcs, wpks, o, ws)
        torch.cuda.synchronize()
        outs.append(o.cpu())
    assert rel_err(outs[0], ref + prev) < 6e-3
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_igemm_group_at_the_real_bridge_shape_and_its_limits(ops):
    """dL/dx of the ASPP bridge at base_c 48, batch 8, 512x512: 8x32x32, 768 -> 384, 1x1 + d6 + d12 + d18."""
    N, H, W, Cin, Cout = 8, 32, 32, 768, 384
    segs = [(1, 1), (3, 6), (3, 12), (3, 18)]
    descs, srcs, wpks, ref = _dgrad_group_problem(ops, N, H, W, Cin, Cout, segs, seed=11)
    assert ops.conv_igemm_group_ok(descs)
    ws = torch.empty(ops.conv_igemm_group_ws_bytes(descs) // 4, device="cuda")
    assert ws.numel() == 2 * N * H * W * Cout          # 128 tiles -> two K-ranges
    out = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    with launch_tags() as tags:
        ops.conv_igemm_group(descs, srcs, wpks, out, ws)
    assert tags == ["igemm_group<128,192>"], tags
    assert rel_err(out.cpu(), ref) < 6e-3
    # out of range: one segment, a later segment that does not accumulate, mismatched shapes, missing workspace
    assert not ops.conv_igemm_group_ok(descs[:1])
    bad = [ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, 3, 3, 1, 6, 6, Cin) for _ in range(2)]
    assert not ops.conv_igemm_group_ok(bad)
    bad[1].accumulate = 1
    assert ops.conv_igemm_group_ok(bad)
    bad[1].Cout = 192
    assert not ops.conv_igemm_group_ok(bad)
    with pytest.raises(Exception):
        ops.conv_igemm_group(descs, srcs, wpks, out, None)


# ---- strips with register-resident weights (conv3x3s.hip): 48 / 96 channels in and out ----
STRIP_CASES = [
    # N, H, W, Cin, Cout   (units = N * W/16 * vertical segments; a segment restarts the 16-row block stream)
    (1, 16, 16, 48, 48),       # one patch: the initial two-row block + one block
    (2, 48, 32, 96, 48),       # three patches per strip, four units
    (1, 64, 16, 48, 96),       # one strip cut into vertical segments (two row quads per wave)
    (3, 32, 80, 96, 96),
    (2, 256, 16, 48, 48),      # long strips: the ring wraps several times
    (9, 16, 464, 96, 48),      # more units than CUs (persistent workgroups take a second unit)
]


@pytest.mark.parametrize("case", STRIP_CASES)
def test_strip_kernel_forward_stats_and_eval_epilogue(ops, case):
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = R.bf16_round(torch.randn(N, H, W, Cin, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    raw = R.conv_fwd(x, w)
    cpad = ops.cpad_of(Cin)
    xd, wd = dev(x.to(torch.bfloat16)), dev(pack_fwd(w, cpad))
    # training form: raw output + statistics, destination NaN-poisoned (every element must be written exactly once)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, 3, 3, 1, 1, 1, cpad)
    out = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = ops.stats_buffer(Cout)
    with launch_tags() as lt:
        ops.conv_igemm(d, xd, wd, out, stats=stats)
    assert [t.split(" ")[0] for t in lt] == [f"conv3x3s<{Cin},{Cout}>"], lt
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), raw) < 6e-3
    s = ops.stats_totals(stats, Cout).float().cpu()
    flat = raw.reshape(-1, Cout)
    assert float((s[0] - flat.sum(0)).abs().max()) < 2e-3 * float(flat.abs().sum(0).max())
    assert torch.allclose(s[1], (flat ** 2).sum(0), rtol=2e-3)
    # inference form: bias, folded-BN affine, ReLU, padded destination pitch (neighbours untouched)
    bias, scale, shift = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = torch.relu((raw + bias) * scale + shift)
    d2 = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout + 8, 3, 3, 1, 1, 1, cpad, relu=1)
    out2 = torch.zeros(N, H, W, Cout + 8, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(d2, xd, wd, out2, bias=dev(bias), scale=dev(scale), shift=dev(shift))
    torch.cuda.synchronize()
    got = out2.cpu()
    assert rel_err(got[..., :Cout], ref) < 8e-3
    assert float(got[..., Cout:].abs().max()) == 0
    # the launch is bitwise reproducible, and identical to the LDS-resident-weight / halo kernels it replaces
    out3 = torch.empty_like(out)
    st3 = ops.stats_buffer(Cout)
    ops.conv_igemm(d, xd, wd, out3, stats=st3)
    torch.cuda.synchronize()
    assert torch.equal(out, out3) and torch.equal(stats, st3)


@pytest.mark.parametrize("case", STRIP_CASES)
def test_strip_kernel_with_batchnorm_relu_applied_on_the_input(ops, case):
    """aau_conv_igemm_bnin: the producing layer's y = relu(z * scale + shift) is applied on the operand in LDS.  Same bits
    -- output and statistics -- as aau_bn_act into a buffer followed by aau_conv_igemm (pipeline:59-65 twice), including
    the zero padding at the image border (the padding of y is 0, not relu(shift)) and channels with negative scale; and
    against the fp32 reference of conv(relu(bn(z)))."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(1000 + sum(case))
    z = R.bf16_round(torch.randn(N, H, W, Cin, generator=g) * 1.5)
    w = R.bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    scale = torch.randn(Cin, generator=g)                        # both signs
    shift = torch.randn(Cin, generator=g) * 0.5 + 0.4            # mostly positive: relu(shift) != 0 would show in the border
    M = N * H * W
    cpad = ops.cpad_of(Cin)
    zd, wd = dev(z.to(torch.bfloat16)), dev(pack_fwd(w, cpad))
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, 3, 3, 1, 1, 1, cpad)
    assert ops.conv_bnin_ok(d)
    # the two-kernel path
    y = torch.empty(M, Cin, dtype=torch.bfloat16, device="cuda")
    ops.bn_act(zd, Cin, y, Cin, dev(scale), dev(shift), M, Cin)
    ref = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    st_ref = ops.stats_buffer(Cout)
    ops.conv_igemm(d, y, wd, ref, stats=st_ref)
    # the fused path (NaN-poisoned destination)
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = ops.stats_buffer(Cout)
    with launch_tags() as lt:
        ops.conv_igemm_bnin(d, zd, dev(scale), dev(shift), wd, out, stats=st)
    assert [t.split(" ")[0] for t in lt] == [f"conv3x3s<{Cin},{Cout}>"], lt
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    # the statistics: per-wave fp32 partial sums of the SAME accumulator values, added exactly across waves and workgroups;
    # the two instantiations of the kernel may contract v * v + s differently, so the totals agree to fp32 rounding of the
    # partial sums, not to the last bit (which replica a workgroup adds into also follows the alternating traversal)
    ta, tb = ops.stats_totals(st, Cout), ops.stats_totals(st_ref, Cout)
    assert float((ta - tb).abs().max()) <= 1e-5 * float(tb.abs().max())
    assert torch.equal(zd.cpu(), z.to(torch.bfloat16))           # the source is not written
    yr = R.bf16_round(torch.relu(z * scale + shift))
    assert rel_err(out.float().cpu().view(N, H, W, Cout), R.conv_fwd(yr, w)) < 6e-3
    # without statistics, and repeated: same bits
    out2 = torch.empty_like(out)
    ops.conv_igemm_bnin(d, zd, dev(scale), dev(shift), wd, out2)
    torch.cuda.synchronize()
    assert torch.equal(out2.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("case", [(2, 32, 48, 48, 48), (1, 64, 16, 48, 48), (3, 16, 80, 48, 48),
                                  # the 96 x 32 row-reuse tiling (wgrad3x3r<6,2>): two ping-pong groups, an odd patch count per
                                  # group, and the one-group form (90 tiles)
                                  (2, 32, 48, 96, 96), (3, 24, 16, 64, 96), (1, 16, 16, 960, 288)])
def test_weight_gradient_with_batchnorm_relu_applied_on_the_input(ops, case):
    """aau_conv_wgrad_bnin: same bits as aau_bn_act into a buffer followed by aau_conv_wgrad (deterministic split-K
    workspace), zero padding of the ACTIVATION at the border included."""
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(2000 + sum(case))
    z = R.bf16_round(torch.randn(N, H, W, Cin, generator=g) * 1.5)
    dz = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    scale = torch.randn(Cin, generator=g)
    shift = torch.randn(Cin, generator=g) * 0.5 + 0.4
    M = N * H * W
    zd, dzd = dev(z.to(torch.bfloat16)), dev(dz.to(torch.bfloat16))
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, 3, 3, 1, 1, 1)
    assert ops.conv_wgrad_bnin_ok(d)
    ws = torch.empty(ops.conv_wgrad_ws_bytes(d) // 4, device="cuda")
    y = torch.empty(M, Cin, dtype=torch.bfloat16, device="cuda")
    ops.bn_act(zd, Cin, y, Cin, dev(scale), dev(shift), M, Cin)
    ref = torch.zeros(Cout, 9, Cin, device="cuda")
    ops.conv_wgrad(d, y, dzd, ref, ws)
    got = torch.zeros(Cout, 9, Cin, device="cuda")
    ops.conv_wgrad_bnin(d, zd, dev(scale), dev(shift), dzd, got, ws)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    yr = R.bf16_round(torch.relu(z * scale + shift))
    want = R.conv_wgrad(yr, dz, (Cout, Cin, 3, 3), 1)
    assert rel_err(got.cpu().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2), want) < 2e-3


def test_strip_kernel_data_gradient_matches_conv2d_input(ops):
    N, H, W, Cin, Cout = 2, 64, 48, 96, 48          # forward 96 -> 48; its data gradient is a 48 -> 96 convolution
    g = torch.Generator().manual_seed(11)
    w = R.bf16_round(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    dz = R.bf16_round(torch.randn(N, H, W, Cout, generator=g))
    ref = R.conv_dgrad(dz, w, (H, W))
    cpd = ops.cpad_of(Cout)
    dd = ops.conv_desc(N, H, W, Cout, Cout, H, W, Cin, Cin, 3, 3, 1, 1, 1, cpd)
    din = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    with launch_tags() as lt:
        ops.conv_igemm(dd, dev(dz.to(torch.bfloat16)), dev(pack_dgrad(w, cpd)), din)
    assert [t.split(" ")[0] for t in lt] == ["conv3x3s<48,96>"], lt
    torch.cuda.synchronize()
    assert rel_err(din.cpu(), ref) < 6e-3


def test_strip_data_gradient_with_fused_batchnorm_backward_sums(ops):
    """aau_conv_igemm_bnred (48 -> 48): the data gradient is bit-identical to aau_conv_igemm's, and the sums its epilogue
    accumulates for the consuming [BatchNorm -> ReLU] layer equal those of the separate aau_bn_bwd_reduce pass over (z, dy)."""
    N, H, W, C_ = 2, 64, 80, 48
    M = N * H * W
    g = torch.Generator().manual_seed(21)
    w = R.bf16_round(torch.randn(C_, C_, 3, 3, generator=g) / (C_ * 9) ** 0.5)
    dz = R.bf16_round(torch.randn(N, H, W, C_, generator=g))
    z = torch.randn(M, C_, generator=g).to(torch.bfloat16).cuda()            # raw conv output of the consuming layer
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.3
    zf = z.float().cpu()
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    cpd = ops.cpad_of(C_)
    dd = ops.conv_desc(N, H, W, C_, C_, H, W, C_, C_, 3, 3, 1, 1, 1, cpd)
    assert ops.conv_bnred_ok(dd)
    dzd, wd = dev(dz.to(torch.bfloat16)), dev(pack_dgrad(w, cpd))
    ref = torch.empty(M, C_, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(dd, dzd, wd, ref)
    out = torch.full((M, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
    sums = ops.stats_buffer(C_)
    with launch_tags() as tags:
        ops.conv_igemm_bnred(dd, dzd, wd, out, z, C_, dev(scale), dev(shift), dev(mean), dev(invstd), sums)
    assert [t.split(" ")[0] for t in tags] == ["conv3x3s<48,48>"]
    red = torch.zeros(2, C_, device="cuda")
    ops.stats_to_red(sums, C_, red)
    red_ref = torch.zeros(2, C_, device="cuda")
    ops.bn_bwd_reduce(z, C_, ref, C_, None, 0, None, C_, dev(scale), dev(shift), dev(mean), dev(invstd), red_ref, N, H, W, C_)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    tol = 2e-4 * float(red_ref.abs().max()) + 1e-3
    assert float((red - red_ref).abs().max()) < tol, (red - red_ref).abs().max()
    # an independent fp64 check of the sums themselves
    gz = ref.float().cpu().double() * ((zf * scale + shift) > 0)
    s1 = gz.sum(0)
    s2 = (gz * ((zf.double() - mean.double()) * invstd.double())).sum(0)
    assert float((red[0].cpu().double() - s1).abs().max()) < 1e-3 * float(s1.abs().max()) + 1e-2
    assert float((red[1].cpu().double() - s2).abs().max()) < 1e-3 * float(s2.abs().max()) + 1e-2
    # run-to-run bitwise reproducible
    sums2 = ops.stats_buffer(C_)
    ops.conv_igemm_bnred(dd, dzd, wd, out, z, C_, dev(scale), dev(shift), dev(mean), dev(invstd), sums2)
    torch.cuda.synchronize()
    assert torch.equal(sums, sums2)


@pytest.mark.parametrize("case", [(8, 384, 768), (2, 64, 128), (16, 40, 24), (3, 8, 8)])
def test_image_pool_branch_kernels(ops, case):
    """aau_poolbranch_fwd / _bwd / _dx against torch autograd of Conv2d(Cin, Cout, 1, bias=False) -> BatchNorm2d (training)
    -> ReLU on [B, Cin, 1, 1] (pipeline:75-77) in fp32 from the same bf16-rounded operands."""
    B, Cin, Cout = case
    g = torch.Generator().manual_seed(5 + B)
    x = R.bf16_round(torch.randn(B, Cin, generator=g))
    w = R.bf16_round(torch.randn(Cout, Cin, generator=g) / Cin ** 0.5)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    dy = R.bf16_round(torch.randn(B, Cout, generator=g))
    rm0, rv0 = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    # reference
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    zr = xr @ wr.t()
    yr = torch.relu(torch.nn.functional.batch_norm(zr, rm, rv, gr, br, True, 0.1, 1e-5))
    yr.backward(dy)
    cpf, cpd = ops.cpad_of(Cin), ops.cpad_of(Cout)
    xp = Cin + 8                                               # a pitch
    xd = torch.zeros(B, xp, dtype=torch.bfloat16, device="cuda")
    xd[:, :Cin] = x.to(torch.bfloat16).cuda()
    wf = dev(pack_fwd(w[:, :, None, None], cpf))
    wd = dev(pack_dgrad(w[:, :, None, None], cpd))
    z = torch.empty(B, Cout, dtype=torch.bfloat16, device="cuda")
    f32 = lambda *s: torch.zeros(*s, device="cuda")
    scale, shift, mean, invstd = f32(Cout), f32(Cout), f32(Cout), f32(Cout)
    rmd, rvd, nbt = dev(rm0.clone()), dev(rv0.clone()), torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.poolbranch_fwd(xd, xp, wf, cpf, z, dev(gamma), dev(beta), rmd, rvd, nbt, scale, shift, mean, invstd, B, Cin, Cout)
    torch.cuda.synchronize()
    assert int(nbt) == 1
    assert rel_err(z.cpu(), zr.detach()) < 6e-3
    assert float((mean.cpu() - zr.detach().mean(0)).abs().max()) < 1e-4
    var = zr.detach().var(0, unbiased=False)
    assert float((invstd.cpu() - 1 / torch.sqrt(var + 1e-5)).abs().max()) < 2e-3 * float((1 / torch.sqrt(var + 1e-5)).max())
    assert float((rmd.cpu() - rm).abs().max()) < 1e-4 and float((rvd.cpu() - rv).abs().max()) < 1e-3
    y = torch.relu(z.float() * scale + shift).cpu()
    # z is stored in bf16: its rounding reaches y multiplied by the channel's scale (large where B = 2 values lie close)
    assert bool(((y - yr.detach()).abs() <= 2.0 ** -8 * (z.float().abs() * scale.abs()).cpu() + 1e-2).all())
    # backward
    dz = torch.empty(B, Cout, dtype=torch.bfloat16, device="cuda")
    dgam, dbet = f32(Cout) + 1.0, f32(Cout) + 2.0             # accumulated into
    dw = f32(Cout, Cin) + 0.5
    dyd = torch.zeros(B, Cout + 16, dtype=torch.bfloat16, device="cuda")
    dyd[:, :Cout] = dy.to(torch.bfloat16).cuda()
    ops.poolbranch_bwd(dyd, Cout + 16, z, xd, xp, dev(gamma), scale, shift, mean, invstd, dz, dgam, dbet, dw, B, Cin, Cout)
    dx = torch.full((B, xp), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.poolbranch_dx(dz, wd, cpd, dx, xp, B, Cin, Cout)
    torch.cuda.synchronize()
    tol = lambda t: 2e-2 * float(t.abs().max()) + 1e-3
    # (1) the BatchNorm-backward formulas in fp64 from the kernel's own z and saved statistics
    zc, sc, sh, mu, isd = (t.cpu().double() for t in (z.float(), scale, shift, mean, invstd))
    g_ = dy.double() * ((zc.float() * sc.float() + sh.float()) > 0)
    zh = (zc - mu) * isd
    s1, s2 = g_.sum(0), (g_ * zh).sum(0)
    dze = gamma.double() * isd * (g_ - s1 / B - zh * s2 / B)
    assert float((dbet.cpu() - 2.0 - s1).abs().max()) < 1e-4 * float(s1.abs().max()) + 1e-5
    assert float((dgam.cpu() - 1.0 - s2).abs().max()) < 1e-4 * float(s2.abs().max()) + 1e-5
    assert float((dz.float().cpu() - dze).abs().max()) < 2.0 ** -7 * float(dze.abs().max()) + 1e-6
    dwe = dz.float().cpu().double().t() @ x.double()
    assert float((dw.cpu() - 0.5 - dwe).abs().max()) < 1e-4 * float(dwe.abs().max()) + 1e-5
    # (2) torch autograd in fp32 (B >= 8: with fewer samples the rounding of z moves zhat by O(1) in channels whose
    # values lie close together); masks can differ where y is within rounding of zero: channels whose mask agrees
    if B >= 8:
        okc = (((z.float() * scale + shift).cpu() > 0) == (yr.detach() > 0)).all(0)
        assert okc.float().mean() > 0.9
        assert float(((dbet.cpu() - 2.0) - br.grad)[okc].abs().max()) < tol(br.grad)
        assert float(((dgam.cpu() - 1.0) - gr.grad)[okc].abs().max()) < tol(gr.grad)
        assert float(((dw.cpu() - 0.5) - wr.grad)[okc].abs().max()) < tol(wr.grad)
        if bool(okc.all()):
            assert float((dx[:, :Cin].float().cpu() - xr.grad).abs().max()) < tol(xr.grad)
    assert float(dx[:, Cin:].float().min()) == 7.0            # the pitch tail is not written
    # dx from the kernel's own dz, exactly
    dxe = dz.float().cpu() @ w
    assert float((dx[:, :Cin].float().cpu() - dxe).abs().max()) < 1e-2 * float(dxe.abs().max()) + 1e-4
    with pytest.raises(RuntimeError):
        ops.poolbranch_dx(dz, wd, cpd, dx, xp, 17, Cin, Cout)
    # one sample per channel: the reference's BatchNorm2d raises in training mode (pipeline:75-77), so does the library
    with pytest.raises(RuntimeError):
        ops.poolbranch_fwd(xd, xp, wf, cpf, z, dev(gamma), dev(beta), rmd, rvd, nbt, scale, shift, mean, invstd, 1, Cin, Cout)
    with pytest.raises(RuntimeError):
        ops.poolbranch_bwd(dyd, Cout + 16, z, xd, xp, dev(gamma), scale, shift, mean, invstd, dz, dgam, dbet, dw, 1, Cin, Cout)


def test_igemm_multi_equals_the_separate_launches(ops):
    """aau_conv_igemm_multi at the bridge shape (8 x 32 x 32, 384 -> 768: 1x1 + dilation 6 / 12 / 18 on one input, each into
    its slice of a [.., 4 * 768]-pitch buffer with its own statistics): same bits as four aau_conv_igemm launches."""
    N, H, W, Ci, Co = 8, 32, 32, 384, 768
    g = torch.Generator().manual_seed(77)
    x = dev(R.bf16_round(torch.randn(N, H, W, Ci, generator=g)).to(torch.bfloat16))
    cp = ops.cpad_of(Ci)
    probs = []
    for k, dil in ((3, 6), (3, 12), (3, 18), (1, 1)):
        w = R.bf16_round(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
        probs.append((ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, k, k, 1, dil * (k // 2), dil, cp), dev(pack_fwd(w, cp))))
    descs = [p[0] for p in probs]
    assert ops.conv_igemm_multi_ok(descs) and not ops.conv_igemm_multi_ok(descs[:1])
    ref, refst = [], []
    for d, wd in probs:
        o = torch.full((N, H, W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        st = ops.stats_buffer(Co)
        ops.conv_igemm(d, x, wd, o, stats=st)
        ref.append(o)
        refst.append(st)
    outs = [torch.full((N, H, W, Co), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in probs]
    sts = [ops.stats_buffer(Co) for _ in probs]
    with launch_tags() as lt:
        ops.conv_igemm_multi(descs, [x] * 4, [p[1] for p in probs], outs, sts)
    assert [t.split("|")[-1] for t in lt] == ["igemm<64,192,0> multi"], lt
    torch.cuda.synchronize()
    for o, r_, st, rs in zip(outs, ref, sts, refst):
        assert torch.equal(o, r_) and torch.equal(st, rs)
    # two problems, one without statistics
    outs2 = [torch.empty_like(outs[0]) for _ in range(2)]
    ops.conv_igemm_multi(descs[2:], [x] * 2, [p[1] for p in probs[2:]], outs2, [None, ops.stats_buffer(Co)])
    torch.cuda.synchronize()
    assert torch.equal(outs2[0], ref[2]) and torch.equal(outs2[1], ref[3])
    # a problem the wide tile does not serve is refused
    small = ops.conv_desc(1, 16, 16, 64, 64, 16, 16, 96, 96, 1, 1, 1, 0, 1, 64)
    assert not ops.conv_igemm_multi_ok([descs[0], small])


@pytest.mark.parametrize("case", [(4, 256, 256, 96, 48, 0, 0), (4, 256, 256, 40, 104, 0, 1), (4, 256, 256, 48, 96, 0, 1),
                                  (5, 256, 240, 192, 96, 0, 0), (8, 128, 128, 192, 384, 1, 0), (4, 256, 256, 96, 192, 1, 0)])
def test_round4_1x1_kernel_equals_the_ring_kernel_it_replaces(ops, case, monkeypatch):
    """conv1x1_rs_kernel (scalar-offset fills and stores, prefetched read-modify-write with hand-counted waits) against
    conv1x1_resw_kernel (AAU_PW_OLD=1) on the same operands: outputs bit for bit (statistics to fp32 rounding) -- plain, accumulating
    (bias + affine + ReLU on top of an existing destination), pixel-shuffle stores, channel tails in both operands."""
    N, H, W, Cin, Cout, shuffle, acc = case
    g = torch.Generator().manual_seed(900 + sum(case))
    x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
    cp = ops.cpad_of(Cin)
    w = torch.zeros(Cout, 1, cp)
    w[:, 0, :Cin] = torch.randn(Cout, Cin, generator=g) / Cin ** 0.5
    w = w.to(torch.bfloat16).cuda()
    Co = Cout // 4 if shuffle else Cout
    bias = torch.randn(Co, generator=g).cuda()
    scale, shift = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    prev = torch.randn(N * H * W * (4 if shuffle else 1), Co + 8, generator=g).to(torch.bfloat16).cuda()
    if shuffle:
        d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Co + 8, Cpad=cp, shuffle2x2=1)
    else:
        d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Co + 8, Cpad=cp, accumulate=acc, relu=acc)

    def run():
        out = prev.clone()
        st = None if acc else ops.stats_buffer(Cout if not shuffle else Cout)
        with launch_tags() as lt:
            if acc:
                ops.conv_igemm(d, x, w, out, bias=bias, scale=scale, shift=shift)
            elif shuffle:
                ops.conv_igemm(d, x, w, out, bias=bias)
            else:
                ops.conv_igemm(d, x, w, out, stats=st)
        torch.cuda.synchronize()
        return out, (None if st is None or shuffle else st.clone()), [t.split(" ")[0] for t in lt]

    new, st_new, tag_new = run()
    monkeypatch.setenv("AAU_PW_OLD", "1")
    old, st_old, tag_old = run()
    assert tag_new[0].startswith("conv1x1_rs<") and tag_old[0].startswith("conv1x1_resw<"), (tag_new, tag_old)
    assert torch.equal(new.view(torch.int16), old.view(torch.int16))
    assert float(new[:, Co:].float().abs().sum()) == float(prev[:, Co:].float().abs().sum())     # the pad columns are untouched
    if st_new is not None:      # (the per-workgroup fp32 partial sums contract differently in the two instantiations: last bits)
        a, b = ops.stats_totals(st_new, Cout), ops.stats_totals(st_old, Cout)
        assert float(((a - b).abs() / (b.abs() + 1.0)).max()) < 1e-5


@pytest.mark.parametrize("case", [(4, 256, 256, 96, 48, 0), (4, 256, 256, 40, 104, 0), (4, 256, 256, 96, 192, 1),
                                  (8, 128, 128, 192, 384, 1), (2, 256, 256, 64, 192, 2)])
def test_1x1_and_convT_forward_with_batchnorm_relu_applied_on_the_input(ops, case):
    """aau_conv_igemm_bnin on the resident-weight 1x1 kernel (conv1x1_rs): the same bits as aau_bn_act into a buffer followed
    by aau_conv_igemm -- plain 1x1 with statistics, ConvTranspose forward (pixel shuffle + bias) into a dense plane (mode 1)
    and into the upper half of an interleaved concat buffer (mode 2), a channel tail in the last 32-channel chunk."""
    N, H, W, Cin, Cout, mode = case
    g = torch.Generator().manual_seed(7000 + sum(case))
    z = (torch.randn(N * H * W, Cin, generator=g) * 1.5).to(torch.bfloat16).cuda()
    scale, shift = torch.randn(Cin, generator=g).cuda(), (torch.randn(Cin, generator=g) * 0.5 + 0.3).cuda()
    cp = ops.cpad_of(Cin)
    w = torch.zeros(Cout, 1, cp)
    w[:, 0, :Cin] = torch.randn(Cout, Cin, generator=g) / Cin ** 0.5
    w = w.to(torch.bfloat16).cuda()
    if mode == 0:
        d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, Cpad=cp)
        shape, view, bias = (N * H * W, Cout), (lambda t: t), None
    else:
        Co = Cout // 4
        pitch = Co if mode == 1 else 2 * Co
        d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, pitch, Cpad=cp, shuffle2x2=1)
        shape, view = (N * 4 * H * W, pitch), (lambda t: t if mode == 1 else t[:, Co:])
        bias = torch.randn(Co, generator=g).cuda()
    assert ops.conv_bnin_ok(d)
    y = torch.empty_like(z)
    ops.bn_act(z, Cin, y, Cin, scale, shift, N * H * W, Cin)
    ref = torch.full(shape, 2.0, dtype=torch.bfloat16, device="cuda")
    st_ref = ops.stats_buffer(Cout) if mode == 0 else None
    ops.conv_igemm(d, y, w, view(ref), bias=bias, stats=st_ref)
    got = torch.full(shape, 2.0, dtype=torch.bfloat16, device="cuda")
    st = ops.stats_buffer(Cout) if mode == 0 else None
    with launch_tags() as lt:
        ops.conv_igemm_bnin(d, z, scale, shift, w, view(got), stats=st, bias=bias)
    torch.cuda.synchronize()
    assert lt[0].startswith("conv1x1_rs<"), lt
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    if st is not None:
        a, b = ops.stats_totals(st, Cout), ops.stats_totals(st_ref, Cout)
        assert float(((a - b).abs() / (b.abs() + 1.0)).max()) < 1e-5


@pytest.mark.parametrize("case", [("taps", 1, 128, 128, 48, 96), ("taps", 2, 64, 128, 96, 192), ("lin", 2, 64, 64, 96, 48),
                                  ("lin", 1, 128, 128, 192, 96)])
def test_weight_gradient_with_batchnorm_relu_applied_on_the_dz_operand(ops, case):
    """aau_conv_wgrad_bnin_dz (the ConvTranspose2d weight gradient, whose `dz` operand is the coarse input activation): the
    same bits as aau_bn_act on that operand followed by aau_conv_wgrad, through the deterministic split-K workspace."""
    kind, N, Ho, Wo, Cs, Cq = case            # Cs: channels of `src`, Cq: channels of the `dz` operand
    g = torch.Generator().manual_seed(8000 + N + Ho + Cs + Cq)
    k = 2 if kind == "taps" else 1
    H, W = Ho * k, Wo * k
    src = torch.randn(N * H * W, Cs, generator=g).to(torch.bfloat16).cuda()
    zq = (torch.randn(N * Ho * Wo, Cq, generator=g) * 1.5).to(torch.bfloat16).cuda()
    scale, shift = torch.randn(Cq, generator=g).cuda(), (torch.randn(Cq, generator=g) * 0.5 + 0.3).cuda()
    if kind == "taps":
        d = ops.conv_desc(N, H, W, Cs, Cs, Ho, Wo, Cq, Cq, 2, 2, 2, 0, 1)
    else:
        d = ops.conv_desc(N, H, W, Cs, Cs, Ho, Wo, Cq, Cq, 1, 1, 1, 0, 1)
    assert ops.conv_wgrad_bnin_dz_ok(d)
    ws = torch.full((ops.conv_wgrad_ws_bytes(d) // 4,), float("nan"), device="cuda")
    act = torch.empty_like(zq)
    ops.bn_act(zq, Cq, act, Cq, scale, shift, N * Ho * Wo, Cq)
    ref = torch.zeros(Cq, k * k, Cs, device="cuda")
    ops.conv_wgrad(d, src, act, ref, ws)
    got = torch.zeros(Cq, k * k, Cs, device="cuda")
    ops.conv_wgrad_bnin_dz(d, src, zq, scale, shift, got, ws)
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0 and torch.equal(got, ref)
    with pytest.raises(Exception, match="not served"):        # the all-taps 3x3 kernel has its own form (aau_conv_wgrad_bnin)
        d3 = ops.conv_desc(N, Ho, Wo, Cs, Cs, Ho, Wo, Cq, Cq, 3, 3, 1, 1, 1)
        ops.conv_wgrad_bnin_dz(d3, src[: N * Ho * Wo], zq, scale, shift, torch.zeros(Cq, 9, Cs, device="cuda"), ws)
