"""Race screen: repeat deterministic launches (no atomics / no stats) and compare outputs bitwise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops
torch.manual_seed(0)
def run_case(name, N, H, W, Ci, Co, k, dil, reps=40, extra=None):
    x = torch.randn(N, H, W, Ci, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, k * k, cp, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    if cp != Ci: w[:, :, Ci:] = 0
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, k, k, 1, dil * (k // 2), dil, cp)
    outs = []
    other = torch.randn(64 << 20, device="cuda")  # background traffic between launches changes timing
    for r in range(reps):
        out = torch.full((N, H, W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm(d, x, w, out)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out)
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) else 1 for o in outs[1:])
    nan = int(torch.isnan(outs[0].float()).sum())
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {nan}", flush=True)
    return bad
tot = 0
tot += run_case("conv3x3g<96> 64x64 768->384", 8, 64, 64, 768, 384, 3, 1)
tot += run_case("conv3x3g<96> 128x128 96->192", 8, 128, 128, 96, 192, 3, 1)
tot += run_case("conv3x3g<48> small 48->48", 2, 64, 64, 48, 48, 3, 1)
tot += run_case("conv3x3s<48,48> 512 48->48", 8, 512, 512, 48, 48, 3, 1, reps=12)
tot += run_case("conv3x3s<96,48> 512 96->48", 8, 512, 512, 96, 48, 3, 1, reps=12)
tot += run_case("conv3x3s<48,96> 256 48->96", 8, 256, 256, 48, 96, 3, 1, reps=12)
tot += run_case("conv3x3s<96,96> 256 96->96", 8, 256, 256, 96, 96, 3, 1, reps=12)
tot += run_case("resw<48> 256 64->48 (round-2 kernel)", 8, 256, 256, 64, 48, 3, 1, reps=12)
tot += run_case("igemm dil6 32x32 384->768", 8, 32, 32, 384, 768, 3, 6)
tot += run_case("igemm SMALL dgrad-like 768->384 d12", 8, 32, 32, 768, 384, 3, 12)
tot += run_case("igemm 1x1 3840->768", 8, 32, 32, 3840, 768, 1, 1)
tot += run_case("igemm 1x1 96->48 256", 8, 256, 256, 96, 48, 1, 1, reps=12)


def run_wgrad(name, N, H, W, Ci, Co, k, dil, reps=12):
    """split-K through slabs: bitwise reproducible by construction; the screen repeats it under changing traffic"""
    x = torch.randn(N, H, W, Ci, device="cuda").to(torch.bfloat16)
    dz = torch.randn(N, H, W, Co, device="cuda").to(torch.bfloat16)
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, k, k, 1, dil * (k // 2), dil)
    ws = torch.full((ops.conv_wgrad_ws_bytes(d) // 4,), float("nan"), device="cuda")
    other = torch.randn(64 << 20, device="cuda")
    outs = []
    for r in range(reps):
        dw = torch.zeros(Co, k * k, Ci, device="cuda")
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_wgrad(d, x, dz, dw, ws)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(dw)
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:])
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0]).sum())}", flush=True)
    return bad


tot += run_wgrad("wgrad3x3 512 48->48", 8, 512, 512, 48, 48, 3, 1)
tot += run_wgrad("wgrad3x3 64 768->384", 8, 64, 64, 768, 384, 3, 1)
tot += run_wgrad("wgrad 1x1 256 96->48", 8, 256, 256, 96, 48, 1, 1)
tot += run_wgrad("wgrad dil6 32 384->768", 8, 32, 32, 384, 768, 3, 6)


def run_group(name, reps=30):
    """grouped bridge input gradient (igemm_group.hip): register-staged ring, two K-ranges, fixed-order slab sum"""
    B, H, Ci, Co = 8, 32, 768, 384
    segs = [(1, 1), (3, 6), (3, 12), (3, 18)]
    descs, srcs, wpks = [], [], []
    for i, (k, dil) in enumerate(segs):
        srcs.append(torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16))
        wpks.append((torch.randn(Co, k * k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16))
        descs.append(ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, dil * (k // 2), dil, Ci, accumulate=1 if i else 0))
    ws = torch.full((ops.conv_igemm_group_ws_bytes(descs) // 4,), float("nan"), device="cuda")
    other = torch.randn(64 << 20, device="cuda")
    outs = []
    for r in range(reps):
        out = torch.full((B, H, H, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm_group(descs, srcs, wpks, out, ws)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out)
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) else 1 for o in outs[1:])
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
    return bad


tot += run_group("igemm_group bridge dgrad")
tot += run_case("igemm wide convT-like 768->1536", 8, 32, 32, 768, 1536, 1, 1)
tot += run_case("igemm wide dil18 384->768", 8, 32, 32, 384, 768, 3, 18)


def run_bnred(name, reps=12):
    """strip input gradient with the fused BatchNorm-backward sums (hand-counted vmcnt for the z loads): outputs AND sums"""
    N, H, W, C_ = 8, 512, 512, 48
    dz = torch.randn(N, H, W, C_, device="cuda").to(torch.bfloat16)
    z = torch.randn(N * H * W, C_, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C_, 9, 64, device="cuda") / (C_ * 9) ** 0.5).to(torch.bfloat16)
    w[:, :, C_:] = 0
    sc, sh = torch.rand(C_, device="cuda") + 0.5, torch.randn(C_, device="cuda") * 0.3
    mu, istd = torch.zeros(C_, device="cuda"), torch.ones(C_, device="cuda")
    d = ops.conv_desc(N, H, W, C_, C_, H, W, C_, C_, 3, 3, 1, 1, 1, 64)
    other = torch.randn(64 << 20, device="cuda")
    outs, sums = [], []
    for r in range(reps):
        out = torch.full((N * H * W, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
        st = ops.stats_buffer(C_)
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm_bnred(d, dz, w, out, z, C_, sc, sh, mu, istd, st)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out); sums.append(st)
    torch.cuda.synchronize()
    bad = sum(0 if (torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) and torch.equal(s_, sums[0])) else 1
              for o, s_ in zip(outs[1:], sums[1:]))
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
    return bad


tot += run_bnred("conv3x3s bnred 512 48->48")
tot += run_case("conv1x1_resw 256 96->48", 8, 256, 256, 96, 48, 1, 1, reps=12)


# ---- round 4: the MFMA kernels the screen did not cover yet ------------------------------------------
def run_acc(name, N, H, W, Ci, Co, k, reps=12, shuffle=0):
    """a conv that ADDS into its destination (read-modify-write epilogues): every repetition starts from the same contents"""
    x = torch.randn(N, H, W, Ci, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, k * k, cp, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    if cp != Ci: w[:, :, Ci:] = 0
    if shuffle:
        d = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co // 4, Cpad=cp, shuffle2x2=1)
        base = torch.full((N * 4 * H * W, Co // 4), float("nan"), dtype=torch.bfloat16, device="cuda")
    else:
        d = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, k, k, 1, k // 2, 1, cp, accumulate=1)
        base = torch.randn(N * H * W, Co, device="cuda").to(torch.bfloat16)
    other = torch.randn(64 << 20, device="cuda")
    outs = []
    for r in range(reps):
        out = base.clone()
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm(d, x, w, out)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out)
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) else 1 for o in outs[1:])
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
    return bad


tot += run_acc("conv3x3p dgrad accumulate 128 192->384", 8, 128, 128, 192, 384, 3)
tot += run_acc("conv3x3p dgrad accumulate 64 384->768", 8, 64, 64, 384, 768, 3)
tot += run_acc("conv1x1_rs accumulate 256 48->96", 8, 256, 256, 48, 96, 1)
tot += run_acc("conv1x1_rs accumulate 128 96->192", 8, 128, 128, 96, 192, 1)
tot += run_acc("conv1x1_rs ConvT 256 96->4x48", 8, 256, 256, 96, 192, 1, shuffle=1)
tot += run_acc("conv1x1_rs ConvT 128 192->4x96", 8, 128, 128, 192, 384, 1, shuffle=1)


def run_bnin(name, N, H, W, Ci, Co, reps=12):
    """strip conv and 3x3 weight gradient with BatchNorm + ReLU applied on the operand in LDS (own-piece rewrites behind
    counted vmcnt waits)"""
    z = (torch.randn(N, H, W, Ci, device="cuda") * 1.5).to(torch.bfloat16)
    dz = torch.randn(N, H, W, Co, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 9, cp, device="cuda") / (Ci * 9) ** 0.5).to(torch.bfloat16)
    if cp != Ci: w[:, :, Ci:] = 0
    sc, sh = torch.randn(Ci, device="cuda"), torch.randn(Ci, device="cuda") * 0.5 + 0.4
    df = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, 3, 3, 1, 1, 1, cp)
    dw_ = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, 3, 3, 1, 1, 1)
    assert ops.conv_bnin_ok(df) and ops.conv_wgrad_bnin_ok(dw_)
    ws = torch.full((ops.conv_wgrad_ws_bytes(dw_) // 4,), float("nan"), device="cuda")
    other = torch.randn(64 << 20, device="cuda")
    outs, dws, sts = [], [], []
    for r in range(reps):
        out = torch.full((N * H * W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        dw = torch.zeros(Co, 9, Ci, device="cuda")
        st = ops.stats_buffer(Co)
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm_bnin(df, z, sc, sh, w, out, st)
        ops.conv_wgrad_bnin(dw_, z, sc, sh, dz, dw, ws)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out); dws.append(dw); sts.append(ops.stats_totals(st, Co))
    torch.cuda.synchronize()
    bad = sum(0 if (torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) and torch.equal(g, dws[0]) and torch.equal(t, sts[0]))
              else 1 for o, g, t in zip(outs[1:], dws[1:], sts[1:]))
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
    return bad


tot += run_bnin("bnin conv3x3s + wgrad3x3 512 48->48", 8, 512, 512, 48, 48)
tot += run_bnin("bnin conv3x3s + wgrad3x3r 256 96->96", 8, 256, 256, 96, 96)


def run_wgrad_group(name, reps=12):
    """grouped bridge weight gradient (wgradL_pp: per-XCD work queues, ping-pong groups, ordered read-modify-write of dw)"""
    B, H, Ci, Co = 8, 32, 768, 384
    segs = [(1, 1), (3, 6), (3, 12), (3, 18)]
    descs, srcs, dzs = [], [], []
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    for k, dil in segs:
        srcs.append(x)
        dzs.append(torch.randn(B, H, H, Co, device="cuda").to(torch.bfloat16))
        descs.append(ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, dil * (k // 2), dil))
    assert ops.conv_wgrad_group_ok(descs)
    other = torch.randn(64 << 20, device="cuda")
    outs = []
    for r in range(reps):
        dws = [torch.zeros(Co, k * k, Ci, device="cuda") for k, _ in segs]
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_wgrad_group(descs, srcs, dzs, dws)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(torch.cat([t.flatten() for t in dws]))
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:])
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0]).sum())}", flush=True)
    return bad


tot += run_wgrad_group("wgradL grouped bridge wgrad")


def run_multi(name, reps=20):
    """the four spatial ASPP branches as one launch (igemm_multi), outputs and statistics"""
    B, H, Ci, Co = 8, 32, 384, 768
    segs = [(1, 1), (3, 6), (3, 12), (3, 18)]
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    descs, wpks = [], []
    for k, dil in segs:
        wpks.append((torch.randn(Co, k * k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16))
        descs.append(ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, dil * (k // 2), dil, Ci))
    assert ops.conv_igemm_multi_ok(descs)
    other = torch.randn(64 << 20, device="cuda")
    outs, sts = [], []
    for r in range(reps):
        dsts = [torch.full((B * H * H, Co), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in segs]
        st = [ops.stats_buffer(Co) for _ in segs]
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm_multi(descs, [x] * 4, wpks, dsts, st)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(torch.cat([t.flatten() for t in dsts])); sts.append(torch.cat([ops.stats_totals(t, Co).flatten() for t in st]))
    torch.cuda.synchronize()
    bad = sum(0 if (torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) and torch.equal(t, sts[0])) else 1
              for o, t in zip(outs[1:], sts[1:]))
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
    return bad


tot += run_multi("igemm_multi ASPP branches")


def run_poolbranch(name, reps=20):
    """image-pool branch kernels (one wave per channel, batch statistics in registers)"""
    B, Ci, Co = 8, 384, 768
    g = torch.randn(B, Ci, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 1, cp, device="cuda") / Ci ** 0.5).to(torch.bfloat16)
    gam, bet = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1
    dy = torch.randn(B, Co, device="cuda").to(torch.bfloat16)
    cpd = ops.cpad_of(Co)
    wd = (torch.randn(Ci, 1, cpd, device="cuda") / Co ** 0.5).to(torch.bfloat16)
    other = torch.randn(64 << 20, device="cuda")
    outs = []
    for r in range(reps):
        z = torch.empty(B, Co, dtype=torch.bfloat16, device="cuda")
        rm, rv, nbt = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
        sc, sh, mu, istd = (torch.empty(Co, device="cuda") for _ in range(4))
        dz = torch.empty(B, Co, dtype=torch.bfloat16, device="cuda")
        dga, dbe, dw = torch.zeros(Co, device="cuda"), torch.zeros(Co, device="cuda"), torch.zeros(Co, 1, Ci, device="cuda")
        dx = torch.empty(B, Ci, dtype=torch.bfloat16, device="cuda")
        if r % 3 == 1: other.mul_(1.0001)
        ops.poolbranch_fwd(g, Ci, w, cp, z, gam, bet, rm, rv, nbt, sc, sh, mu, istd, B, Ci, Co)
        ops.poolbranch_bwd(dy, Co, z, g, Ci, gam, sc, sh, mu, istd, dz, dga, dbe, dw, B, Ci, Co)
        ops.poolbranch_dx(dz, wd, cpd, dx, Ci, B, Ci, Co)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(torch.cat([z.float().flatten(), sc, sh, dz.float().flatten(), dga, dbe, dw.flatten(), dx.float().flatten()]))
    torch.cuda.synchronize()
    bad = sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:])
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0]).sum())}", flush=True)
    return bad


tot += run_poolbranch("poolbranch fwd / bwd / dx")


def run_bnin_up(name, N, H, Ci, Co, reps=12):
    """ConvTranspose2d(2,2) with BatchNorm + ReLU applied on its operand: forward (conv1x1_rs, own-piece rewrite of the pixel
    tiles) and weight gradient (wgrad_kernel, the dz operand)"""
    z = (torch.randn(N * H * H, Ci, device="cuda") * 1.5).to(torch.bfloat16)
    dfine = torch.randn(N * 4 * H * H, Co, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(4 * Co, 1, cp, device="cuda") / Ci ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Co, device="cuda")
    sc, sh = torch.randn(Ci, device="cuda"), torch.randn(Ci, device="cuda") * 0.5 + 0.3
    df = ops.conv_desc(N, H, H, Ci, Ci, H, H, 4 * Co, Co, Cpad=cp, shuffle2x2=1)
    dw_ = ops.conv_desc(N, 2 * H, 2 * H, Co, Co, H, H, Ci, Ci, 2, 2, 2, 0, 1)
    assert ops.conv_bnin_ok(df) and ops.conv_wgrad_bnin_dz_ok(dw_)
    ws = torch.full((ops.conv_wgrad_ws_bytes(dw_) // 4,), float("nan"), device="cuda")
    other = torch.randn(64 << 20, device="cuda")
    outs, dws = [], []
    for r in range(reps):
        out = torch.full((N * 4 * H * H, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        dw = torch.zeros(Ci, 4, Co, device="cuda")
        if r % 3 == 1: other.mul_(1.0001)
        ops.conv_igemm_bnin(df, z, sc, sh, w, out, bias=bias)
        ops.conv_wgrad_bnin_dz(dw_, dfine, z, sc, sh, dw, ws)
        if r % 3 == 2: other.add_(1e-3)
        outs.append(out); dws.append(dw)
    torch.cuda.synchronize()
    bad = sum(0 if (torch.equal(o.view(torch.int16), outs[0].view(torch.int16)) and torch.equal(g, dws[0])) else 1
              for o, g in zip(outs[1:], dws[1:]))
    print(f"{name:28s} mismatching repeats {bad}/{reps-1}  nan {int(torch.isnan(outs[0].float()).sum()) + int(torch.isnan(dws[0]).sum())}", flush=True)
    return bad


tot += run_bnin_up("bnin ConvT fwd + wgrad 256 96->48", 8, 256, 96, 48)
tot += run_bnin_up("bnin ConvT fwd + wgrad 128 192->96", 8, 128, 192, 96)
print("TOTAL mismatches", tot)
