"""GPU parity of the bandwidth-bound kernels (BN / pool / gate / loss / optimiser / packing)
against CPU fp32 references and the reference-generated golden vectors."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import kernels_ref as R
from oracle import ref_cpu as O


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from att_aspp_unet_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def zeros(*s, dtype=torch.float32):
    return torch.zeros(*s, dtype=dtype, device="cuda")


def test_conv1_bn_relu_chain_and_running_stats(ops):
    N, H, W, C = 2, 24, 40, 48
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 1, H, W, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    conv = F.conv2d(x, w, padding=1)
    bn = torch.nn.BatchNorm2d(C)
    bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
    bn.train()
    ref = torch.relu(bn(conv)).permute(0, 2, 3, 1)
    z = zeros(N, H, W, C, dtype=torch.bfloat16)
    stats = ops.stats_buffer(C)
    ops.conv1_fwd(dev(x), dev(w.reshape(C, 9).contiguous()), z, stats, N, H, W, C)
    scale, shift, sm, si = zeros(C), zeros(C), zeros(C), zeros(C)
    rm, rv, nbt = zeros(C), torch.ones(C, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.bn_finalize(stats, dev(gamma), dev(beta), rm, rv, nbt, scale, shift, sm, si, C, N * H * W)
    y = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_act(z, C, y, C, scale, shift, N * H * W, C, relu=1)
    torch.cuda.synchronize()
    assert rel_err(z.cpu(), conv.permute(0, 2, 3, 1)) < 5e-3
    assert rel_err(y.cpu(), ref) < 1.5e-2
    assert torch.allclose(rm.cpu(), bn.running_mean, rtol=1e-3, atol=1e-5)
    assert torch.allclose(rv.cpu(), bn.running_var, rtol=1e-3, atol=1e-5)
    assert int(nbt.item()) == 1
    # eval-mode fold
    sc2, sh2 = zeros(C), zeros(C)
    ops.bn_fold_eval(dev(gamma), dev(beta), rm, rv, sc2, sh2, C)
    torch.cuda.synchronize()
    assert torch.allclose(sc2.cpu(), gamma / torch.sqrt(bn.running_var + 1e-5), rtol=1e-3)
    # weight gradient of the first layer
    dz = R.bf16_round(torch.randn(N, H, W, C, generator=g))
    refw = torch.nn.grad.conv2d_weight(x, (C, 1, 3, 3), dz.permute(0, 3, 1, 2), padding=1)
    dw = zeros(C, 9)
    ops.conv1_wgrad(dev(x), dev(bf(dz)), dw, N, H, W, C)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu().reshape(C, 1, 3, 3), refw) < 1e-3


@pytest.mark.parametrize("C", [48, 96, 8, 200])
def test_maxpool_and_bn_backward_with_pool_routing(ops, C):
    N, H, W = 2, 12, 20
    g = torch.Generator().manual_seed(C)
    z = R.bf16_round(torch.randn(N, H, W, C, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gy = R.bf16_round(torch.randn(N, H, W, C, generator=g))
    gp = R.bf16_round(torch.randn(N, H // 2, W // 2, C, generator=g))
    # CPU reference through autograd; y is rounded to bf16 before pooling like the stored activation
    zc = z.clone().requires_grad_(True)
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    flat = zc.reshape(-1, C)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    ypre = (zc - mean) * invstd * gam + bet
    y = torch.relu(ypre)
    yq = y + (R.bf16_round(y.detach()) - y.detach())  # straight-through bf16 rounding
    p = F.max_pool2d(yq.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    ((y * gy).sum() + (p * gp).sum()).backward()
    # device
    zd = dev(bf(z))
    scale, shift = dev((gamma * invstd).detach()), dev((beta - mean * gamma * invstd).detach())
    yd = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_act(zd, C, yd, C, scale, shift, N * H * W, C, relu=1)
    pd = zeros(N, H // 2, W // 2, C, dtype=torch.bfloat16)
    ops.maxpool2(yd, C, pd, C, N, H, W, C)
    torch.cuda.synchronize()
    assert rel_err(pd.cpu(), p.detach()) < 1e-2
    red = zeros(ops.STAT_REPLICAS, 2, C)
    dz = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_bwd_reduce(zd, C, dev(bf(gy)), C, dev(bf(gp)), C, dz, C, scale, shift, dev(mean.detach()), dev(invstd.detach()),
                      red, N, H, W, C, relu=1)
    dgam, dbet = zeros(C), zeros(C)
    ops.bn_bwd_apply(zd, C, dz, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red, dgam, dbet, N * H * W, C)
    torch.cuda.synchronize()
    assert rel_err(dz.cpu(), zc.grad) < 2e-2
    assert rel_err(dgam.cpu(), gam.grad) < 1e-2
    assert rel_err(dbet.cpu(), bet.grad) < 1e-2
    # the form the engine runs: neither pass stores the routed gradient, the apply pass redoes the routing -- same bits
    red2 = zeros(ops.STAT_REPLICAS, 2, C)
    ops.bn_bwd_reduce(zd, C, dev(bf(gy)), C, dev(bf(gp)), C, None, C, scale, shift, dev(mean.detach()), dev(invstd.detach()),
                      red2, N, H, W, C, relu=1)
    dz2 = torch.full((N, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    dgam2, dbet2 = zeros(C), zeros(C)
    ops.bn_bwd_apply_pool(zd, C, dz2, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red2, dgam2, dbet2, N, H, W, C,
                          dev(bf(gy)), C, dev(bf(gp)), C, scale, shift, relu=1)
    torch.cuda.synchronize()
    assert torch.equal(red2, red)
    assert torch.equal(dz2.view(torch.int16), dz.view(torch.int16))
    assert torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet)
    # no skip path (dy = NULL): only the pooled gradient arrives
    red3 = zeros(ops.STAT_REPLICAS, 2, C)
    dz3a = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_bwd_reduce(zd, C, None, 0, dev(bf(gp)), C, dz3a, C, scale, shift, dev(mean.detach()), dev(invstd.detach()),
                      red3, N, H, W, C, relu=1)
    ops.bn_bwd_apply(zd, C, dz3a, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red3, zeros(C), zeros(C), N * H * W, C)
    dz3b = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_bwd_apply_pool(zd, C, dz3b, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red3, zeros(C), zeros(C), N, H, W,
                          C, None, 0, dev(bf(gp)), C, scale, shift, relu=1)
    torch.cuda.synchronize()
    assert torch.equal(dz3a.view(torch.int16), dz3b.view(torch.int16))
    # no-pool variant (plain BN + ReLU backward)
    zc2 = z.clone().requires_grad_(True)
    flat2 = zc2.reshape(-1, C)
    m2, v2 = flat2.mean(0), flat2.var(0, unbiased=False)
    y2 = torch.relu((zc2 - m2) / torch.sqrt(v2 + 1e-5) * gamma + beta)
    (y2 * gy).sum().backward()
    red.zero_()
    ops.bn_bwd_reduce(zd, C, dev(bf(gy)), C, None, 0, dz, C, scale, shift, dev(mean.detach()), dev(invstd.detach()),
                      red, N, H, W, C, relu=1)
    ops.bn_bwd_apply(zd, C, dz, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red, None, None, N * H * W, C)
    torch.cuda.synchronize()
    assert rel_err(dz.cpu(), zc2.grad) < 2e-2
    # same, without storing the masked gradient: reduce(dz=None) + apply(dy=...) recomputes the mask
    red.zero_()
    dz2 = zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_bwd_reduce(zd, C, dev(bf(gy)), C, None, 0, None, C, scale, shift, dev(mean.detach()), dev(invstd.detach()),
                      red, N, H, W, C, relu=1)
    ops.bn_bwd_apply(zd, C, dz2, C, dev(gamma), dev(mean.detach()), dev(invstd.detach()), red, None, None, N * H * W, C,
                     dy=dev(bf(gy)), dyp=C, scale=scale, shift=shift, relu=1)
    torch.cuda.synchronize()
    assert torch.equal(dz2.cpu(), dz.cpu())


@pytest.mark.parametrize("C", [48, 8, 104])
def test_first_layer_fused_bn_backward_and_weight_gradient(ops, C):
    """aau_bn_bwd_apply_conv1 == aau_bn_bwd_apply (dz stored in bf16) followed by aau_conv1_wgrad, and both match
    the CPU weight gradient of Conv2d(1, C, 3, pad 1) (pipeline:113)."""
    N, H, W = 2, 24, 40
    g = torch.Generator().manual_seed(100 + C)
    x = torch.randn(N, H, W, generator=g)
    z = R.bf16_round(torch.randn(N, H, W, C, generator=g))
    gy = R.bf16_round(torch.randn(N, H, W, C, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    flat = z.reshape(-1, C)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    zd, gyd, xd = dev(bf(z)), dev(bf(gy)), dev(x)
    scale, shift = dev(gamma * invstd), dev(beta - mean * gamma * invstd)
    red = zeros(ops.STAT_REPLICAS, 2, C)
    ops.bn_bwd_reduce(zd, C, gyd, C, None, 0, None, C, scale, shift, dev(mean), dev(invstd), red, N, H, W, C, relu=1)
    # unfused pair
    dz = zeros(N, H, W, C, dtype=torch.bfloat16)
    dg0, db0, dw0 = zeros(C), zeros(C), zeros(C, 9)
    ops.bn_bwd_apply(zd, C, dz, C, dev(gamma), dev(mean), dev(invstd), red, dg0, db0, N * H * W, C, dy=gyd, dyp=C,
                     scale=scale, shift=shift, relu=1)
    ops.conv1_wgrad(xd, dz, dw0, N, H, W, C)
    # fused
    dg1, db1, dw1 = zeros(C), zeros(C), zeros(C, 9)
    ws = torch.full((ops.STAT_REPLICAS * C * 9,), float("nan"), device="cuda")
    ops.bn_bwd_apply_conv1(zd, C, dev(gamma), dev(mean), dev(invstd), red, dg1, db1, N, H, W, C, gyd, C, scale, shift,
                           xd, dw1, ws)
    torch.cuda.synchronize()
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
    assert rel_err(dw1.cpu(), dw0.cpu()) < 1e-4               # same bf16-rounded dz, different summation order
    # CPU: weight gradient from the stored dz
    ref = torch.nn.grad.conv2d_weight(x[:, None], (C, 1, 3, 3), dz.float().cpu().permute(0, 3, 1, 2), padding=1)
    assert rel_err(dw1.cpu().reshape(C, 1, 3, 3), ref) < 1e-4


def test_dropout_mask_is_consistent_between_forward_and_backward(ops):
    M, C, p = 4096, 64, 0.1
    z = torch.ones(M, C, dtype=torch.bfloat16, device="cuda")
    one, zero = torch.ones(C, device="cuda"), zeros(C)
    y = zeros(M, C, dtype=torch.bfloat16)
    ops.bn_act(z, C, y, C, one, zero, M, C, relu=1, drop_p=p, drop_seed=1234)
    torch.cuda.synchronize()
    yc = y.float().cpu()
    kept = yc > 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    assert torch.allclose(yc[kept], torch.full_like(yc[kept], 1 / (1 - p)), rtol=1e-2)
    dz, red = zeros(M, C, dtype=torch.bfloat16), zeros(ops.STAT_REPLICAS, 2, C)
    ops.bn_bwd_reduce(z, C, torch.ones_like(z), C, None, 0, dz, C, one, zero, zero, one, red, 1, 1, M, C, relu=1,
                      drop_p=p, drop_seed=1234)
    torch.cuda.synchronize()
    assert torch.equal(dz.float().cpu() > 0, kept)


def test_gap_spatial_sum_colsum_broadcast(ops):
    N, HW, C = 3, 100, 72
    g = torch.Generator().manual_seed(3)
    x = R.bf16_round(torch.randn(N, HW, C, generator=g))
    xd = dev(bf(x))
    pooled, ws = zeros(N, C, dtype=torch.bfloat16), torch.full((ops.GAP_WS_ROWS, N, C), float("nan"), device="cuda")
    ops.gap_fwd(xd, C, pooled, ws, N, HW, C)
    ssum = zeros(N, C, dtype=torch.bfloat16)
    ops.spatial_sum(xd, C, ssum, ws, N, HW, C)
    cs = zeros(C)
    ops.colsum(xd, C, cs, zeros(ops.STAT_REPLICAS, C + 8), N * HW, C)
    torch.cuda.synchronize()
    assert rel_err(pooled.cpu(), x.mean(1)) < 6e-3
    assert rel_err(ssum.cpu(), x.sum(1)) < 6e-3
    assert rel_err(cs.cpu(), x.reshape(-1, C).sum(0)) < 1e-4
    # broadcast of a per-image row through bn_act, and the GAP backward accumulate
    out = zeros(N, HW, C, dtype=torch.bfloat16)
    ops.bn_act(pooled, C, out, C, torch.ones(C, device="cuda"), zeros(C), N * HW, C, relu=0, bcast_hw=HW)
    dx = dev(bf(x)).clone()
    ops.gap_bwd(pooled, dx, C, N, HW, C)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), pooled.cpu()[:, None, :].expand(N, HW, C))
    assert rel_err(dx.cpu(), x + pooled.float().cpu()[:, None, :] / HW) < 6e-3


@pytest.mark.parametrize("Fi,C", [(48, 96), (24, 48), (96, 192)])
def test_attention_gate_elementwise_forward_backward(ops, Fi, C):
    M = 2 * 16 * 16
    g = torch.Generator().manual_seed(Fi)
    zg, zx = R.bf16_round(torch.randn(M, Fi, generator=g)), R.bf16_round(torch.randn(M, Fi, generator=g) * 1.5 + 0.3)
    x = R.bf16_round(torch.randn(M, C, generator=g))
    dout = R.bf16_round(torch.randn(M, C, generator=g))
    P = {k: (torch.rand(n, generator=g) + 0.5) for k, n in (("gg", Fi), ("gx", Fi), ("g1", 1))}
    P.update({k: torch.randn(n, generator=g) * 0.2 for k, n in (("bg", Fi), ("bx", Fi), ("b1", 1))})
    P["w"] = torch.randn(Fi, generator=g) / Fi ** 0.5
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    zgc, zxc, xc = zg.clone().requires_grad_(True), zx.clone().requires_grad_(True), x.clone().requires_grad_(True)

    def bn(t, gam, bet):
        m, v = t.mean(0), t.var(0, unbiased=False)
        return (t - m) / torch.sqrt(v + 1e-5) * gam + bet, m, 1 / torch.sqrt(v + 1e-5)

    ng, mg, ig = bn(zgc, leaves["gg"], leaves["bg"])
    nx, mx, ix = bn(zxc, leaves["gx"], leaves["bx"])
    s = torch.relu(ng + nx)
    psi = (s * leaves["w"]).sum(1, keepdim=True)
    n1, m1, i1 = bn(psi, leaves["g1"], leaves["b1"])
    alpha = torch.sigmoid(n1)
    out = xc * alpha
    (out * dout).sum().backward()

    d = lambda t: dev(t.detach().contiguous())
    sg, hg = d(P["gg"] * ig), d(P["bg"] - mg * P["gg"] * ig)
    sx, hx = d(P["gx"] * ix), d(P["bx"] - mx * P["gx"] * ix)
    zgd, zxd, xd = dev(bf(zg)), dev(bf(zx)), dev(bf(x))
    psi_pre, st1 = zeros(M), ops.stats_buffer(1)
    ops.gate_psi(zgd, zxd, sg, hg, sx, hx, d(P["w"]), psi_pre, st1, M, Fi)
    sc1, sh1, sm1, si1 = zeros(1), zeros(1), zeros(1), zeros(1)
    ops.bn_finalize(st1, d(P["g1"]), d(P["b1"]), None, None, None, sc1, sh1, sm1, si1, 1, M)
    al = zeros(M)
    wide = zeros(M, 2 * C, dtype=torch.bfloat16)
    ops.gate_apply(xd, C, psi_pre, sc1, sh1, al, wide, 2 * C, M, C)
    torch.cuda.synchronize()
    assert rel_err(psi_pre.cpu(), psi.detach()[:, 0]) < 1e-3
    assert abs(float(sm1.item()) - float(m1.item())) < 1e-3 and abs(float(si1.item()) / float(i1.item()) - 1) < 1e-3
    assert rel_err(wide.cpu()[:, :C], out.detach()) < 1e-2
    assert float(wide.cpu()[:, C:].abs().max()) == 0
    # backward
    dx, dq, red1 = zeros(M, C, dtype=torch.bfloat16), zeros(M), torch.full((4,), float("nan"), device="cuda")
    ops.gate_bwd1(dev(bf(dout)), C, xd, C, al, psi_pre, sm1, si1, dx, C, dq, red1, M, C)
    ds = zeros(M, Fi, dtype=torch.bfloat16)
    dw, tot = zeros(Fi), torch.full((4, Fi), float("nan"), device="cuda")
    dg1, db1 = zeros(1), zeros(1)
    ops.gate_bwd2(dq, psi_pre, red1, d(P["g1"]), sm1, si1, zgd, zxd, sg, hg, sx, hx, d(mg), d(ig), d(mx), d(ix),
                  d(P["w"]), ds, tot, dg1, db1, M, Fi)
    dzg, dzx = zeros(M, Fi, dtype=torch.bfloat16), zeros(M, Fi, dtype=torch.bfloat16)
    dgg, dbg, dgx, dbx = zeros(Fi), zeros(Fi), zeros(Fi), zeros(Fi)
    ops.gate_bwd3(ds, zgd, zxd, d(P["gg"]), d(mg), d(ig), d(P["gx"]), d(mx), d(ix), tot, dzg, dzx,
                  dgg, dbg, dgx, dbx, dw, M, Fi)
    torch.cuda.synchronize()
    # the x gradient has two parts; this kernel chain produces the direct one (dout*alpha)
    assert rel_err(dx.cpu(), (dout * alpha.detach())) < 1e-2
    assert rel_err(dzg.cpu(), zgc.grad) < 3e-2
    assert rel_err(dzx.cpu(), zxc.grad) < 3e-2
    assert rel_err(dw.cpu(), leaves["w"].grad) < 1e-2
    for got, key in ((dgg, "gg"), (dbg, "bg"), (dgx, "gx"), (dbx, "bx"), (dg1, "g1"), (db1, "b1")):
        assert rel_err(got.cpu(), leaves[key].grad) < 2e-2, key


def test_outconv_forward_backward(ops):
    M, C = 3000, 48
    g = torch.Generator().manual_seed(9)
    y = R.bf16_round(torch.randn(M, C, generator=g))
    w, b = torch.randn(C, generator=g) / 7, torch.randn(1, generator=g)
    dl = torch.randn(M, generator=g)
    logits = zeros(M)
    ops.outconv_fwd(dev(bf(y)), C, dev(w), dev(b), logits, M, C)
    dy, dw, db = zeros(M, C, dtype=torch.bfloat16), zeros(C), zeros(1)
    ops.outconv_bwd(dev(bf(y)), C, dev(dl), dev(w), dy, C, dw, db, zeros(ops.STAT_REPLICAS, C + 8), M, C)
    torch.cuda.synchronize()
    assert rel_err(logits.cpu(), y @ w + b) < 1e-5
    assert rel_err(dy.cpu(), dl[:, None] * w[None, :]) < 6e-3
    assert rel_err(dw.cpu(), (dl[:, None] * y).sum(0)) < 1e-4
    assert abs(float(db.item()) - float(dl.sum())) < 1e-2


def test_criterion_and_metrics_against_reference_golden(ops, golden):
    g = golden("g2_loss.npz")
    for tag in ("mixed", "allneg", "allpos"):
        l, t = dev(torch.from_numpy(g[f"{tag}/logits"])), dev(torch.from_numpy(g[f"{tag}/targets"]))
        B, _, H, W = l.shape
        for stage in ("main", "finetune"):
            sums, out, dl = zeros(32, B, 8), zeros(4), zeros(B, 1, H, W)
            ops.criterion(l, t, sums, out, dl, B, H, W, finetune=(stage == "finetune"))
            torch.cuda.synchronize()
            ref_loss = float(g[f"{tag}/{stage}/loss"])
            assert abs(float(out[0].item()) - ref_loss) < 2e-5 * max(1, abs(ref_loss)), (tag, stage)
            ref_d = torch.from_numpy(g[f"{tag}/{stage}/dlogits"])
            assert float((dl.cpu() - ref_d).abs().max()) < 2e-4 * float(ref_d.abs().max()) + 1e-9, (tag, stage)
        sums, m = zeros(32, B, 8), zeros(2)
        ops.seg_metrics(l, t, sums, m, B, H, W, 0.5)
        torch.cuda.synchronize()
        assert abs(float(m[0].item()) - float(g[f"{tag}/dice_eval"])) < 1e-5
        assert abs(float(m[1].item()) - float(g[f"{tag}/iou"])) < 1e-5


def test_clip_and_adamw_match_torch(ops):
    n = 100003
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(n, generator=g)
    pc = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pc], lr=3e-4, weight_decay=5e-4)
    pd, md, vd = dev(p0.clone()), zeros(n), zeros(n)
    step, ws = torch.zeros(1, dtype=torch.int64, device="cuda"), zeros(ops.SQNORM_WS)
    for it in range(3):
        grad = torch.randn(n, generator=g) * (0.01 if it == 1 else 1.0)  # step 1 is below the clip threshold
        pc.grad = grad.clone()
        gn = torch.nn.utils.clip_grad_norm_([pc], 1.0)
        opt.step()
        gd = dev(grad)
        ops.grad_sqnorm(gd, n, 1.0, ws)
        ops.adamw_step(pd, md, vd, gd, n, ws, step, 3e-4)
        torch.cuda.synchronize()
        assert abs(float(ws[0].item()) ** 0.5 - float(gn)) < 1e-3 * float(gn)
        assert float((pd.cpu() - pc.detach()).abs().max()) < 2e-6
    assert int(step.item()) == 3
    # non-finite gradients skip the step
    gd = dev(torch.full((n,), float("inf")))
    before = pd.clone()
    ops.grad_sqnorm(gd, n, 1.0, ws)
    ops.adamw_step(pd, md, vd, gd, n, ws, step, 3e-4)
    torch.cuda.synchronize()
    assert torch.equal(pd, before) and int(step.item()) == 3


def test_pack_weights_table(ops):
    from att_aspp_unet_amd._abi import PackEntry
    O_, I_, k = 40, 24, 3
    g = torch.Generator().manual_seed(6)
    w = torch.randn(O_, I_, k, k, generator=g)
    w_cl = w.permute(0, 2, 3, 1).contiguous()          # physical [O][kh][kw][I]
    wt = torch.randn(16, 8, 2, 2, generator=g)         # convT IOHW
    wt_cl = wt.permute(0, 2, 3, 1).contiguous()        # physical [I][kh][kw][O]
    flat = torch.cat([w_cl.reshape(-1), wt_cl.reshape(-1)])
    T = k * k
    entries, dst, blk = [], 0, 0

    def add(**kw):
        nonlocal dst, blk
        e = PackEntry(**kw, dst_off=dst, blk_begin=blk)
        entries.append(e)
        tot = e.R * e.T * e.Cpad
        dst += tot
        blk += (tot // 8 + 255) // 256
        return dst - tot, tot

    o1, n1 = add(src_off=0, R=O_, T=T, C=I_, Cpad=32, s_r=T * I_, s_t=I_, s_c=1, t_flip=0, R2=0, s_r2=0)
    o2, n2 = add(src_off=0, R=I_, T=T, C=O_, Cpad=64, s_r=1, s_t=I_, s_c=T * I_, t_flip=1, R2=0, s_r2=0)
    base = w_cl.numel()
    o3, n3 = add(src_off=base, R=4 * 8, T=1, C=16, Cpad=32, s_r=8, s_t=0, s_c=4 * 8, t_flip=0, R2=8, s_r2=1)
    o4, n4 = add(src_off=base, R=16, T=4, C=8, Cpad=32, s_r=4 * 8, s_t=8, s_c=1, t_flip=0, R2=0, s_r2=0)
    arr = (PackEntry * len(entries))(*entries)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    packed = zeros(dst, dtype=torch.bfloat16)
    ops.pack_weights(dev(flat), packed, table, len(entries), blk)
    torch.cuda.synchronize()
    pk = packed.float().cpu()
    exp1 = torch.zeros(O_, T, 32); exp1[:, :, :I_] = w.permute(0, 2, 3, 1).reshape(O_, T, I_)
    exp2 = torch.zeros(I_, T, 64); exp2[:, :, :O_] = w.flip(2, 3).permute(1, 2, 3, 0).reshape(I_, T, O_)
    exp3 = torch.zeros(32, 1, 32); exp3[:, 0, :16] = wt.permute(2, 3, 1, 0).reshape(32, 16)
    exp4 = torch.zeros(16, 4, 32); exp4[:, :, :8] = wt.permute(0, 2, 3, 1).reshape(16, 4, 8)
    for off, n, exp in ((o1, n1, exp1), (o2, n2, exp2), (o3, n3, exp3), (o4, n4, exp4)):
        assert torch.equal(pk[off:off + n], R.bf16_round(exp).reshape(-1))


def test_traversal_hint_does_not_change_results(ops):
    """aau_traverse(1): launches alternate the direction in which they walk their tensors (Infinity-Cache hint);
    outputs must be identical to the default direction."""
    from att_aspp_unet_amd import _abi
    N, H, W, C = 2, 32, 32, 48
    g = torch.Generator().manual_seed(77)
    z = dev(bf(torch.randn(N, H, W, C, generator=g)))
    x = dev(bf(torch.randn(N, H, W, C, generator=g)))
    w = R.bf16_round(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    scale, shift = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g) * 0.2)
    cpad = ops.cpad_of(C)
    wp = torch.zeros(C, 9, cpad)
    wp[:, :, :C] = w.permute(0, 2, 3, 1).reshape(C, 9, C)
    wp = dev(wp.to(torch.bfloat16))
    d = ops.conv_desc(N, H, W, C, C, H, W, C, C, 3, 3, 1, 1, 1, cpad)

    def run():
        y = zeros(N, H, W, C, dtype=torch.bfloat16)
        p = zeros(N, H // 2, W // 2, C, dtype=torch.bfloat16)
        out = zeros(N, H, W, C, dtype=torch.bfloat16)
        dw = zeros(C, 9, C)
        ws = torch.empty(ops.conv_wgrad_ws_bytes(d) // 4, device="cuda")
        for _ in range(2):                         # both directions of every kernel are exercised
            ops.bn_act(z, C, y, C, scale, shift, N * H * W, C, relu=1)
            ops.bn_act_pool(z, C, y, C, p, C, scale, shift, N, H, W, C)
            ops.conv_igemm(d, x, wp, out)
            dw.zero_()
            ops.conv_wgrad(d, x, z, dw, ws)
        torch.cuda.synchronize()
        return y.clone(), p.clone(), out.clone(), dw.clone()

    ref = run()
    _abi.fn("aau_traverse")(1)
    try:
        got = run()
    finally:
        _abi.fn("aau_traverse")(0)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("C", [48, 8, 104])
def test_network_head_fused_forward_and_backward(ops, C):
    """aau_bn_act_outconv / aau_bn_bwd_reduce_outconv / aau_bn_bwd_apply_rank1 against the unfused chain
    bn_act -> outconv_fwd and outconv_bwd -> bn_bwd_reduce -> bn_bwd_apply (pipeline:121-122,126)."""
    M = 2 * 24 * 40
    g = torch.Generator().manual_seed(200 + C)
    z = dev(bf(torch.randn(M, C, generator=g)))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    zf = z.float().cpu()
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    scale, shift = dev(gamma * invstd), dev(beta - mean * gamma * invstd)
    w, b = dev(torch.randn(C, generator=g) * 0.3), dev(torch.randn(1, generator=g))
    dl = dev(torch.randn(M, generator=g))
    # unfused
    y = zeros(M, C, dtype=torch.bfloat16)
    ops.bn_act(z, C, y, C, scale, shift, M, C, relu=1)
    lg0 = zeros(M)
    ops.outconv_fwd(y, C, w, b, lg0, M, C)
    dy = zeros(M, C, dtype=torch.bfloat16)
    dw0, db0 = zeros(C), zeros(1)
    ws = torch.full((ops.STAT_REPLICAS * (C + 8),), float("nan"), device="cuda")
    ops.outconv_bwd(y, C, dl, w, dy, C, dw0, db0, ws, M, C)
    red0 = zeros(ops.STAT_REPLICAS, 2, C)
    ops.bn_bwd_reduce(z, C, dy, C, None, 0, None, C, scale, shift, dev(mean), dev(invstd), red0, 1, 1, M, C, relu=1)
    dz0, dg0, dbt0 = zeros(M, C, dtype=torch.bfloat16), zeros(C), zeros(C)
    ops.bn_bwd_apply(z, C, dz0, C, dev(gamma), dev(mean), dev(invstd), red0, dg0, dbt0, M, C, dy=dy, dyp=C, scale=scale,
                     shift=shift, relu=1)
    # fused
    lg1 = zeros(M)
    ops.bn_act_outconv(z, C, scale, shift, w, b, lg1, M, C)
    red1, dw1, db1 = zeros(ops.STAT_REPLICAS, 2, C), zeros(C), zeros(1)
    ops.bn_bwd_reduce_outconv(z, C, dl, w, scale, shift, dev(mean), dev(invstd), red1, dw1, db1, None, M, C)
    dz1, dg1, dbt1 = zeros(M, C, dtype=torch.bfloat16), zeros(C), zeros(C)
    ops.bn_bwd_apply_rank1(z, C, dz1, C, dev(gamma), dev(mean), dev(invstd), red1, dg1, dbt1, M, C, dl, w, scale, shift)
    torch.cuda.synchronize()
    assert rel_err(lg1.cpu(), lg0.cpu()) < 1e-5               # same products, different summation order
    assert rel_err(dw1.cpu(), dw0.cpu()) < 1e-5 and rel_err(db1.cpu(), db0.cpu()) < 1e-5
    assert rel_err(red1.sum(0).cpu(), red0.sum(0).cpu()) < 1e-5
    assert rel_err(dg1.cpu(), dg0.cpu()) < 1e-5 and rel_err(dbt1.cpu(), dbt0.cpu()) < 1e-5
    assert rel_err(dz1.cpu(), dz0.cpu()) < 1e-2 and float((dz1.float() != dz0.float()).float().mean()) < 1e-2


@pytest.mark.parametrize("C", [48, 8, 104])
def test_first_layer_z_recomputed_instead_of_stored(ops, C):
    """aau_conv1_bn_act / aau_conv1_bn_bwd_reduce / aau_bn_bwd_apply_conv1(z=NULL) recompute the first layer's z from
    the frame with the fma chain of aau_conv1_fwd: same bits as the stored-z path."""
    N, H, W = 2, 24, 40
    M = N * H * W
    g = torch.Generator().manual_seed(300 + C)
    x = dev(torch.randn(N, H, W, generator=g))
    w = dev(torch.randn(C, 9, generator=g) * 0.4)
    gy = dev(bf(torch.randn(N, H, W, C, generator=g)))
    gamma = dev(torch.rand(C, generator=g) + 0.5)
    # stored-z path
    z = zeros(N, H, W, C, dtype=torch.bfloat16)
    st0 = ops.stats_buffer(C)
    ops.conv1_fwd(x, w, z, st0, N, H, W, C)
    st1 = ops.stats_buffer(C)
    ops.conv1_fwd(x, w, None, st1, N, H, W, C)                       # statistics only
    zf = z.float().reshape(-1, C)
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, torch.randn(C, generator=g).cuda() * 0.2 - mean * gamma * invstd
    y0, y1 = zeros(N, H, W, C, dtype=torch.bfloat16), zeros(N, H, W, C, dtype=torch.bfloat16)
    ops.bn_act(z, C, y0, C, scale, shift, M, C, relu=1)
    ops.conv1_bn_act(x, w, y1, C, scale, shift, N, H, W, C)
    red0, red1 = zeros(ops.STAT_REPLICAS, 2, C), zeros(ops.STAT_REPLICAS, 2, C)
    ops.bn_bwd_reduce(z, C, gy, C, None, 0, None, C, scale, shift, mean, invstd, red0, N, H, W, C, relu=1)
    ops.conv1_bn_bwd_reduce(x, w, gy, C, scale, shift, mean, invstd, red1, N, H, W, C)
    outs = []
    for zz, ww in ((z, None), (None, w)):
        dg, db, dw = zeros(C), zeros(C), zeros(C, 9)
        ws = torch.full((ops.STAT_REPLICAS * C * 9,), float("nan"), device="cuda")
        ops.bn_bwd_apply_conv1(zz, C, gamma, mean, invstd, red0, dg, db, N, H, W, C, gy, C, scale, shift, x, dw, ws, w=ww)
        outs.append((dg, db, dw))
    torch.cuda.synchronize()
    assert torch.equal(ops.stats_totals(st1, C), ops.stats_totals(st0, C))     # same sums, same bits (fixed-point statistics)
    assert torch.equal(y1, y0)
    assert rel_err(red1.sum(0).cpu(), red0.sum(0).cpu()) < 1e-5
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel_err(outs[1][2].cpu(), outs[0][2].cpu()) < 1e-5


@pytest.mark.parametrize("case", [(8, 32, 32, 768), (2, 64, 64, 384), (8, 128, 128, 48), (1, 8, 8, 104), (3, 16, 24, 2048)])
def test_bn_backward_sums_are_order_independent_and_exact(ops, case):
    """The reduce pass adds the workgroups' rows in a fixed order (csrc/common.h: red_fold_launch): the totals are
    bitwise reproducible, equal a float64 reference to fp32 rounding, and one workspace serves launches of different
    grids one after another (pooled and not) without being cleared."""
    N, H, W, C = case
    g = torch.Generator().manual_seed(sum(case))
    z = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    gy = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    gp = torch.randn(N, H // 2, W // 2, C, generator=g).to(torch.bfloat16)
    flat = z.float().reshape(-1, C)
    mean, invstd = flat.mean(0), 1 / torch.sqrt(flat.var(0, unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    zd, gyd, gpd = dev(z), dev(gy), dev(gp)
    args = (dev(scale), dev(shift), dev(mean), dev(invstd))
    ws = torch.full((ops.bn_red_ws_bytes(C) // 4,), float("nan"), device="cuda")
    t = z.double() * scale.double() + shift.double()
    gm = torch.where(t.float() > 0, gy.double(), torch.zeros((), dtype=torch.float64))     # the mask is taken in fp32
    zh = (z.double() - mean.double()) * invstd.double()
    ref = torch.stack([gm.sum((0, 1, 2)), (gm * zh).sum((0, 1, 2))])
    outs = []
    for rep in range(3):
        red = torch.full((2, C), float("nan"), device="cuda")
        ops.bn_bwd_reduce(zd, C, gyd, C, None, 0, None, C, *args, red, N, H, W, C, relu=1, ws=ws)
        outs.append(red.clone())
        # the pooled form on the same workspace in between (a different grid: other group sizes)
        dz = torch.empty(N, H, W, C, dtype=torch.bfloat16, device="cuda")
        redp = torch.full((2, C), float("nan"), device="cuda")
        ops.bn_bwd_reduce(zd, C, gyd, C, gpd, C, dz, C, *args, redp, N, H, W, C, relu=1, ws=ws)
        outs.append(redp.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[4])
    assert torch.equal(outs[1], outs[3]) and torch.equal(outs[1], outs[5])
    assert not torch.isnan(outs[1]).any()
    err = (outs[0].double().cpu() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2e-5, float(err)


def test_multi_layer_batchnorm_launches_equal_the_single_layer_ones(ops):
    """aau_bn_finalize_multi / aau_bn_act_multi / aau_bn_bwd_reduce_multi / aau_bn_bwd_apply_multi (the ASPP branches,
    pipeline:80-83): each layer of a multi launch gives the bits of its single-layer launch -- statistics, running statistics,
    activations (written into channel slices of one concat buffer), dz, dgamma / dbeta; the backward sums agree to fp32
    rounding (the multi launch splits the pixels over fewer workgroups per layer)."""
    from att_aspp_unet_amd._abi import fn, check
    n, N, H, W, Cc = 3, 2, 16, 24, 64
    M = N * H * W
    g = torch.Generator().manual_seed(4)
    zs = [dev((torch.randn(M, Cc, generator=g) * (1 + i)).to(torch.bfloat16)) for i in range(n)]
    gam = [dev(torch.rand(Cc, generator=g) + 0.5) for _ in range(n)]
    bet = [dev(torch.randn(Cc, generator=g) * 0.3) for _ in range(n)]
    dcat = dev(torch.randn(M, n * Cc + 16, generator=g).to(torch.bfloat16))       # the gradients: slices of one wider buffer
    stream = torch.cuda.current_stream().cuda_stream

    def stats_of(z):
        st = ops.stats_buffer(Cc)
        d = ops.conv_desc(N, H, W, Cc, Cc, H, W, Cc, Cc, Cpad=ops.cpad_of(Cc))
        # identity 1x1 conv: its epilogue accumulates (sum, sum of squares) of z
        wid = torch.zeros(Cc, 1, ops.cpad_of(Cc), dtype=torch.bfloat16)
        wid[torch.arange(Cc), 0, torch.arange(Cc)] = 1
        out = torch.empty(M, Cc, dtype=torch.bfloat16, device="cuda")
        ops.conv_igemm(d, z, dev(wid), out, stats=st)
        return st

    def run(multi):
        f32 = lambda *s: torch.zeros(*s, device="cuda")
        sts = [stats_of(z) for z in zs]
        sc, sh, mu, isd = ([f32(Cc) for _ in range(n)] for _ in range(4))
        rm, rv = [f32(Cc) + 0.1 for _ in range(n)], [f32(Cc) + 1.0 for _ in range(n)]
        nbt = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(n)]
        cat = torch.zeros(M, n * Cc + 16, dtype=torch.bfloat16, device="cuda")
        ys = [cat[:, i * Cc:] for i in range(n)]
        dys = [dcat[:, i * Cc:] for i in range(n)]
        red = [f32(2 * Cc) for _ in range(n)]
        dzs = [torch.zeros(M, Cc, dtype=torch.bfloat16, device="cuda") for _ in range(n)]
        dgam, dbet = [f32(Cc) + 1 for _ in range(n)], [f32(Cc) + 2 for _ in range(n)]
        if multi:
            tabF = ops.ptr_table([[sts[i], gam[i], bet[i], rm[i], rv[i], nbt[i], sc[i], sh[i], mu[i], isd[i]] for i in range(n)])
            check(fn("aau_bn_finalize_multi")(n, tabF, ops.stat_words(Cc) * 8, Cc, M, 1e-5, 0.1, stream))
            tabA = ops.ptr_table([[zs[i], ys[i], sc[i], sh[i]] for i in range(n)])
            check(fn("aau_bn_act_multi")(n, tabA, Cc, n * Cc + 16, M, Cc, 1, stream))
            ws = f32(n, 2 * Cc * 1024)
            tabR = ops.ptr_table([[zs[i], dys[i], sc[i], sh[i], mu[i], isd[i], red[i], ws[i]] for i in range(n)])
            check(fn("aau_bn_bwd_reduce_multi")(n, tabR, Cc, n * Cc + 16, N, H, W, Cc, 1, stream))
            tabP = ops.ptr_table([[zs[i], dzs[i], gam[i], mu[i], isd[i], red[i], dgam[i], dbet[i], dys[i], sc[i], sh[i]]
                                  for i in range(n)])
            check(fn("aau_bn_bwd_apply_multi")(n, tabP, Cc, Cc, n * Cc + 16, M, Cc, 1, stream))
        else:
            for i in range(n):
                ops.bn_finalize(sts[i], gam[i], bet[i], rm[i], rv[i], nbt[i], sc[i], sh[i], mu[i], isd[i], Cc, M)
                ops.bn_act(zs[i], Cc, ys[i], n * Cc + 16, sc[i], sh[i], M, Cc)
                ops.bn_bwd_reduce(zs[i], Cc, dys[i], n * Cc + 16, None, 0, None, Cc, sc[i], sh[i], mu[i], isd[i], red[i], N, H, W, Cc)
                ops.bn_bwd_apply(zs[i], Cc, dzs[i], Cc, gam[i], mu[i], isd[i], red[i], dgam[i], dbet[i], M, Cc, dys[i],
                                 n * Cc + 16, sc[i], sh[i])
        torch.cuda.synchronize()
        return dict(sc=sc, sh=sh, mu=mu, isd=isd, rm=rm, rv=rv, nbt=nbt, cat=cat, red=red, dz=dzs, dgam=dgam, dbet=dbet)

    a, b = run(False), run(True)
    for k in ("sc", "sh", "mu", "isd", "rm", "rv", "nbt"):
        for i in range(n):
            assert torch.equal(a[k][i], b[k][i]), (k, i)
    assert torch.equal(a["cat"], b["cat"]) and float(b["cat"][:, n * Cc:].abs().max()) == 0
    for i in range(n):
        assert float((a["red"][i] - b["red"][i]).abs().max()) <= 1e-5 * float(a["red"][i].abs().max())
        assert rel_err(b["dz"][i], a["dz"][i]) < 1e-2
        assert float((a["dgam"][i] - b["dgam"][i]).abs().max()) <= 1e-5 * float(a["dgam"][i].abs().max()) + 1e-6
        assert float((a["dbet"][i] - b["dbet"][i]).abs().max()) <= 1e-5 * float(a["dbet"][i].abs().max()) + 1e-6
    # with the single-layer sums handed to the multi apply the gradients are the same bits
    assert int(b["nbt"][0]) == 1


def test_zero_multi_clears_buffers_and_bumps_the_counter(ops):
    """aau_zero_multi: any number of buffers (groups of eight per launch), the 64-bit counter bumped once, mod 2^64."""
    from att_aspp_unet_amd import _abi
    bufs = [torch.full((4 * (i + 1) * 13,), float(i + 1), device="cuda") for i in range(11)]
    bufs.append(torch.full((6,), 7, dtype=torch.int64, device="cuda"))           # 48 bytes
    guard = torch.full((64,), 3.0, device="cuda")
    cnt = torch.tensor([-5], dtype=torch.int64, device="cuda")
    ops.zero_multi(bufs, cnt, 0x9E3779B97F4A7C15)
    torch.cuda.synchronize()
    assert all(int((b != 0).sum()) == 0 for b in bufs) and bool((guard == 3.0).all())
    assert int(cnt) == ((-5 + 0x9E3779B97F4A7C15) + (1 << 63)) % (1 << 64) - (1 << 63)
    ops.zero_multi([], cnt, 5)                                               # counter only
    torch.cuda.synchronize()
    assert int(cnt) == ((0x9E3779B97F4A7C15) + (1 << 63)) % (1 << 64) - (1 << 63)
    odd = torch.ones(9, device="cuda")                                        # 36 bytes: not a multiple of 16
    with pytest.raises(_abi.AauError):
        ops.zero_multi([odd])
    with pytest.raises(_abi.AauError):
        ops.zero_multi([torch.ones(16, device="cuda")[1:13]])                # misaligned start
