"""Round-2 API behaviour on the GPU: standalone loss classes with autograd (pipeline:173-216), integer-count Dice / IoU
(evalseg:41-49, exact), the autograd-node guards, mask dtype handling of the fused step, LR schedule under a replayed
hipGraph, parameter groups (ablation:576-586) and the dropout-seed chain."""
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


def _lt(seed=0, B=3, H=48, W=64, neg=True):
    g = torch.Generator().manual_seed(seed)
    l = torch.randn(B, 1, H, W, generator=g) * 2
    t = (torch.rand(B, 1, H, W, generator=g) > 0.6).float()
    if neg:
        t[1] = 0
    return l, t


@pytest.mark.parametrize("name", ["DiceLoss", "TverskyLoss", "ComboLoss", "EdgeLoss"])
def test_standalone_losses_value_and_gradient_match_reference_classes(A, name):
    l, t = _lt(3)
    ref = getattr(O, name)()
    lr = l.clone().requires_grad_(True)
    vr = ref(lr, t)
    vr.backward()
    mine = getattr(A, name)().cuda()
    lg = l.cuda().requires_grad_(True)
    vg = mine(lg, t.cuda())
    (vg * 2.0).backward()              # an upstream factor must scale the gradient
    assert abs(float(vg) - float(vr)) < 2e-6 * max(1.0, abs(float(vr))), name
    gr = lr.grad
    err = float((lg.grad.cpu() / 2.0 - gr).abs().max())
    assert err < 1e-5 * float(gr.abs().max()) + 1e-9, (name, err)


def test_edge_loss_keeps_reference_buffers(A):
    sd = A.EdgeLoss().state_dict()
    rd = O.EdgeLoss().state_dict()
    assert set(sd) == set(rd) == {"kx", "ky"}
    for k in sd:
        assert torch.equal(sd[k].cpu(), rd[k])


def test_integer_dice_iou_are_exact_on_device_and_host(A):
    rng = np.random.default_rng(5)
    for shape in [(562, 744), (7, 13), (1, 1), (3, 512, 512)]:
        a = (rng.random(shape) > 0.55).astype(np.uint8) * 255
        b = (rng.random(shape) > 0.45).astype(np.uint8) * 255
        want = (O.seg_dice(a, b), O.seg_iou(a, b))
        assert (A.evalseg.dice(a, b), A.evalseg.iou(a, b)) == want                       # host arrays
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        assert (A.evalseg.dice(ta, tb), A.evalseg.iou(ta, tb)) == want                   # uint8 on the device
        assert A.evalseg.dice(ta.float() / 255, tb > 0) == want[0]                       # fp32 x bool
        assert A.evalseg.counts(ta.long(), tb.to(torch.bfloat16)) == A.evalseg.counts(a, b)
    z = np.zeros((16, 16), np.uint8)
    assert A.evalseg.dice(torch.from_numpy(z).cuda(), torch.from_numpy(z).cuda()) == O.seg_dice(z, z) == 1.0


def _tiny(A, seed=2025, p_drop=None):
    torch.manual_seed(seed)
    m = A.AttentionASPPUNet(base_c=8).cuda().train()
    if p_drop is not None:
        m.bridge.project[3].p = p_drop
    return m


def test_autograd_node_rejects_unsupported_patterns(A):
    from att_aspp_unet_amd._abi import AauError
    from att_aspp_unet_amd import synth
    m = _tiny(A, p_drop=0.0)
    x, y = synth.make_frames(2, 64, seed=3, force_pattern="pn")
    x, y = x.cuda(), y.cuda()
    crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
    # (1) two training forwards of one shape, then backward through the first: its activations are gone
    l1 = crit(m(x), y)
    l2 = crit(m(x), y)
    with pytest.raises(AauError, match="overwritten"):
        l1.backward()
    l2.backward()                      # the latest forward is fine
    g_first = m.engine.store.gflat.clone()
    # (2) a second backward through the same forward
    l3 = crit(m(x), y)
    l3.backward(retain_graph=True)
    with pytest.raises(AauError, match="second backward"):
        l3.backward()
    assert torch.isfinite(g_first).all()


def test_fused_step_accepts_any_mask_dtype_and_rejects_wrong_shapes(A):
    from att_aspp_unet_amd._abi import AauError
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(2, 64, seed=4, force_pattern="pn")
    x = x.cuda()
    losses = []
    for conv in (lambda t: t.cuda(), lambda t: t.bool().cuda(), lambda t: (t * 255).to(torch.uint8).cuda().clamp(max=1),
                 lambda t: t.to(torch.bfloat16).cuda(), lambda t: t, lambda t: t.cuda().expand(2, 1, 64, 64).transpose(2, 3).transpose(2, 3)):
        m = _tiny(A, p_drop=0.0)
        step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args)
        losses.append(float(step(x, conv(y))))
    assert max(losses) - min(losses) < 5e-5 * abs(losses[0]), losses   # BN-statistics ordering noise between runs
    m = _tiny(A, p_drop=0.0)
    step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args)
    with pytest.raises(AauError, match="targets"):
        step(x, y.cuda()[:, :, :32])
    with pytest.raises(AauError, match="targets"):
        step(x, y.cuda().reshape(2, 64, 64))


def test_graphed_step_follows_the_lr_schedule(A):
    """Three steps with lr 1e-3, 5e-4, 0 -- eagerly, and as ONE captured graph replayed three times: the replay must
    read the learning rate from device memory (first Adam steps move every weight by ~lr)."""
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(2, 64, seed=5, force_pattern="pn")
    x, y = x.cuda(), y.cuda()
    lrs = (1e-3, 5e-4, 0.0)

    def run(graphed):
        m = _tiny(A, p_drop=0.0)
        opt = A.FusedAdamW(m, lr=lrs[0])
        step = A.TrainStep(m, opt, args)
        fn = step
        if graphed:
            st0 = {k: v.clone() for k, v in m.state_dict().items()}
            fn = A.GraphedTrainStep(step, x, y, warmup=1)   # the warm-up step trains: put the start state back
            m.load_state_dict(st0)
            eng = m.engine.store
            eng.m.zero_(); eng.v.zero_(); eng.step_dev.zero_()
        m._plan_for(x)                                       # the flat parameter store exists from the first plan on
        snaps = [m.engine.store.flat.clone()]
        for lr in lrs:
            opt.param_groups[0]["lr"] = lr
            fn(x, y)
            snaps.append(m.engine.store.flat.clone())
        return snaps

    e, g = run(False), run(True)
    assert torch.equal(e[0], g[0])
    for s in (e, g):
        d1, d2 = (s[1] - s[0]).abs(), (s[2] - s[1]).abs()
        moved = d1 > 1e-4                                  # weights with a real gradient
        r = float(d2[moved].median() / d1[moved].median())
        assert 0.3 < r < 0.7, r                            # the second step ran at half the rate ...
        assert 0.8e-3 < float(d1[moved].median()) < 1.2e-3
        assert torch.equal(s[3], s[2])                     # ... and the third at lr 0 moved nothing
    # every sum of the step is order-independent (fixed-point statistics, rows + fold, slabs): the captured graph and the
    # eager launch list produce the same weights bit for bit
    for a, b in zip(e, g):
        assert torch.equal(a, b)


def test_parameter_groups_give_attention_its_own_rate(A):
    """ablation:576-586: attention parameters at lr, the rest at lr/2 -- against torch.optim.AdamW with the same groups."""
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(2, 64, seed=6, force_pattern="pn")
    m = _tiny(A, p_drop=0.0)
    att = [p for n, p in m.named_parameters() if ".att." in n or ".psi" in n]
    bk = [p for n, p in m.named_parameters() if not (".att." in n or ".psi" in n)]
    opt = A.FusedAdamW(m, weight_decay=5e-4, groups=[{"params": bk, "lr": 1e-3}, {"params": att, "lr": 2e-3}])
    step = A.TrainStep(m, opt, args)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    step(x.cuda(), y.cuda())
    grads = {n: p.grad.detach().clone().contiguous() for n, p in m.named_parameters()}
    after = {n: p.detach().clone() for n, p in m.named_parameters()}
    # reference update from the SAME gradients
    ref = {n: torch.nn.Parameter(v.cpu().contiguous().clone()) for n, v in before.items()}
    for n in ref:
        ref[n].grad = grads[n].cpu().clone()
    torch.nn.utils.clip_grad_norm_(list(ref.values()), 1.0)
    ropt = torch.optim.AdamW([{"params": [ref[n] for n in ref if not (".att." in n or ".psi" in n)], "lr": 1e-3},
                              {"params": [ref[n] for n in ref if (".att." in n or ".psi" in n)], "lr": 2e-3}], weight_decay=5e-4)
    ropt.step()
    worst = max(float((after[n].cpu() - ref[n].detach()).abs().max()) for n in ref)
    assert worst < 2e-6, worst
    # the two groups really moved at different rates (first Adam step: |delta| ~ lr)
    d_att = max(float((after[n] - before[n]).abs().max()) for n in before if ".att." in n)
    d_bk = max(float((after[n] - before[n]).abs().max()) for n in before if n.startswith("d1."))
    assert 1.6 < d_att / d_bk < 2.4, (d_att, d_bk)
    with pytest.raises(Exception, match="no group"):
        A.FusedAdamW(m, groups=[{"params": att, "lr": 1e-3}])


def test_dropout_seed_follows_torch_seed(A):
    from att_aspp_unet_amd import synth
    x, _ = synth.make_frames(2, 64, seed=7)
    x = x.cuda()

    def logits(seed):
        m = _tiny(A, seed=seed, p_drop=0.5)
        ref = _tiny(A, seed=2025)                   # identical weights for every call
        m.load_state_dict(ref.state_dict())
        torch.manual_seed(seed)                     # the engine reads torch's seed when it builds its first plan
        with torch.no_grad():
            return m(x).clone()

    a, b, c = logits(11), logits(11), logits(12)
    # BatchNorm statistics are summed with fp32 atomics only where noted in DESIGN.md; the mask is the big effect
    assert float((a - b).abs().mean()) < 0.2 * float((a - c).abs().mean())
    assert float((a - c).abs().max()) > 1e-3


def test_forward_is_bitwise_reproducible_in_training_mode(A):
    """BatchNorm batch statistics are accumulated with order-independent fixed-point atomics (include/aau.h: aau_stat):
    two training forwards of the same input give the same logits bit for bit (round 1: fp32 atomics, logits moved by
    ~1e-3 between runs and the gradient cosine dropped to 0.99)."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(2025)
    m = A.AttentionASPPUNet(base_c=16).cuda().train()
    m.bridge.project[3].p = 0.0
    x, _ = synth.make_frames(4, 128, seed=21)
    x = x.cuda()
    with torch.no_grad():
        outs = [m(x).clone() for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # a non-finite activation poisons the statistics (NaN batch mean, as in torch) instead of turning into finite garbage
    assert bool(torch.isfinite(m.state_dict()["d1.0.block.1.running_mean"]).all())
    xb = x.clone(); xb[0, 0, 5, 5] = float("inf")
    with torch.no_grad():
        m(xb)
    assert bool(torch.isnan(m.state_dict()["d1.0.block.1.running_mean"]).any())


def test_training_step_is_bitwise_reproducible(A):
    """Three full steps (forward, criterion, backward, clip, AdamW; dropout on, p = 0.1) from the same state and seed, twice:
    identical weights, gradients, Adam moments and running statistics bit for bit.  Nothing in the step is summed with
    float atomics: BatchNorm statistics and criterion sums are fixed-point integer adds, every other cross-workgroup
    sum is per-workgroup rows added in a fixed order (csrc/common.h: red_fold_launch, wg_reduce_kernel)."""
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(4, 128, seed=9, force_pattern="ppnp")
    x, y = x.cuda(), y.cuda()

    def run():
        torch.manual_seed(77)
        m = A.AttentionASPPUNet(base_c=16).cuda().train()
        opt = A.FusedAdamW(m, lr=1e-3)
        step = A.TrainStep(m, opt, args)
        losses = [float(step(x, y)) for _ in range(3)]
        st = m.engine.store
        return losses, st.flat.clone(), st.gflat.clone(), st.m.clone(), st.v.clone(), {k: v.clone() for k, v in m.state_dict().items()}

    a, b = run(), run()
    assert a[0] == b[0], (a[0], b[0])
    for u, v in zip(a[1:5], b[1:5]):
        assert torch.equal(u, v)
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k
    assert a[0][2] < a[0][0]


@pytest.mark.parametrize("switch", ["AAU_POOL_APPLY_ROUTES", "AAU_NO_IGEMM_GROUP", "AAU_NO_WIDE_STORE", "AAU_RESW_NOPAIR"])
def test_experiment_switches_do_not_change_the_result(A, switch, monkeypatch):
    """The opt-in / ablation forms of this round's kernels compute the same step: the pooled apply pass that redoes the
    max-pool routing is bitwise identical; the ungrouped bridge input gradient, the 8-byte epilogue stores and the
    unpaired Cin % 32 <= 16 chunk only change the order of fp32 additions (or nothing at all)."""
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(4, 128, seed=5, force_pattern="ppnp")
    x, y = x.cuda(), y.cuda()

    def run():
        torch.manual_seed(3)
        m = A.AttentionASPPUNet(base_c=24).cuda().train()     # bridge 192 -> 384: the grouped input gradient applies
        m.engine.set_drop_p(0.0) if hasattr(m.engine, "set_drop_p") else None
        opt = A.FusedAdamW(m, lr=1e-3)
        step = A.TrainStep(m, opt, args)
        loss = float(step(x, y))
        return loss, m.engine.store.gflat.clone()

    la, ga = run()
    monkeypatch.setenv(switch, "1")
    lb, gb = run()
    if switch in ("AAU_POOL_APPLY_ROUTES", "AAU_NO_WIDE_STORE"):
        assert la == lb and torch.equal(ga, gb)
    else:
        assert abs(la - lb) <= 1e-5 * abs(la)
        cos = torch.nn.functional.cosine_similarity(ga, gb, dim=0)
        assert float(cos) > 0.9999 and abs(float(ga.norm() / gb.norm()) - 1) < 1e-3


@pytest.mark.parametrize("switch", ["AAU_NO_BNRED", "AAU_NO_IGEMM_MULTI", "AAU_NO_POOLBRANCH", "AAU_BRIDGE_WG_SIDE", "AAU_NO_BNIN",
                                    "AAU_NO_BN_MULTI", "AAU_NO_BNIN_UP"])
def test_default_on_fused_paths_against_their_off_switches_at_the_metric_shape(A, switch, monkeypatch):
    """Whole-step A/B of the fused paths that are ON by default, at the shape where they engage (base_c 48, 8 x 512 x 512:
    48-channel strip levels -> aau_conv_igemm_bnred; bridge 384 -> 768 on 8192 pixels = 256 wide tiles per branch ->
    aau_conv_igemm_multi; batch <= 16 -> poolbranch kernels).  The multi-problem launch and the side-stream placement of
    the grouped weight gradient run the same arithmetic (bitwise); the fused BatchNorm-backward sums and the image-pool
    branch kernels add in another order; the BatchNorm + ReLU applied on a conv's operand (aau_conv_igemm_bnin /
    aau_conv_wgrad_bnin) gives the same conv outputs bit for bit, its statistics' fp32 partial sums may round differently."""
    from att_aspp_unet_amd import synth
    args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    x, y = synth.make_frames(8, 512, seed=5)
    x, y = x.cuda(), y.cuda()

    def run():
        torch.manual_seed(3)
        m = A.AttentionASPPUNet(base_c=48).cuda().train()
        m.bridge.project[3].p = 0.0
        step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args)
        loss = float(step(x, y))
        names = [op[2] for op in m._plan_for(x).fwd.ops + m._plan_for(x).bwd.ops]
        g = m.engine.store.gflat.clone()
        del step, m
        torch.cuda.empty_cache()
        return loss, g, names

    la, ga, na = run()
    assert "aau_conv_igemm_bnred" in na and "aau_conv_igemm_multi" in na and "aau_poolbranch_fwd" in na
    # d1.1, d2.1, u2.conv.1, u1.conv.1; u2.up and u1.up (the activation in front of a transposed conv)
    assert na.count("aau_conv_igemm_bnin") == 6 and na.count("aau_conv_wgrad_bnin") == 4 and na.count("aau_conv_wgrad_bnin_dz") == 2
    monkeypatch.setenv(switch, "1")
    lb, gb, nb = run()
    assert "aau_bn_finalize_multi" in na and "aau_bn_bwd_apply_multi" in na
    off = {"AAU_NO_BNRED": "aau_conv_igemm_bnred", "AAU_NO_POOLBRANCH": "aau_poolbranch_fwd",
           "AAU_NO_BNIN": "aau_conv_igemm_bnin", "AAU_NO_BN_MULTI": "aau_bn_finalize_multi",
           "AAU_NO_BNIN_UP": "aau_conv_wgrad_bnin_dz"}.get(switch)
    if off:
        assert off not in nb
    if switch == "AAU_NO_IGEMM_MULTI":
        assert nb.count("aau_conv_igemm_multi") < na.count("aau_conv_igemm_multi")
    if switch in ("AAU_NO_IGEMM_MULTI", "AAU_BRIDGE_WG_SIDE"):
        assert la == lb and torch.equal(ga, gb)
    else:
        assert abs(la - lb) <= 1e-5 * abs(la)
        cos = torch.nn.functional.cosine_similarity(ga, gb, dim=0)
        assert float(cos) > 0.9999 and abs(float(ga.norm() / gb.norm()) - 1) < 1e-3


def test_gradient_accumulation_over_forward_backward_pairs(A):
    """torch semantics of ``.grad``: two forward / backward pairs without clearing the gradients in between ADD; clearing
    them (set_to_none, the default of optimisers and of Module.zero_grad) starts afresh.  Bitwise: g(a) + g(b)."""
    from att_aspp_unet_amd import synth
    m = _tiny(A, p_drop=0.0)
    crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
    xa, ya = (t.cuda() for t in synth.make_frames(2, 64, seed=3, force_pattern="pn"))
    xb, yb = (t.cuda() for t in synth.make_frames(2, 64, seed=4, force_pattern="pp"))
    opt = A.FusedAdamW(m, lr=0.0)

    def grads(x, y):
        opt.zero_grad()
        crit(m(x), y).backward()
        return m.engine.store.gflat.clone()
    ga, gb = grads(xa, ya), grads(xb, yb)
    assert not torch.equal(ga, gb)
    opt.zero_grad()
    crit(m(xa), ya).backward()
    crit(m(xb), yb).backward()                      # no zero_grad in between: accumulates
    acc = m.engine.store.gflat
    assert torch.equal(acc, ga + gb)
    p = dict(m.named_parameters())["d2.0.block.0.weight"]
    assert p.grad is not None and torch.equal(p.grad, m.engine.store.gviews["d2.0.block.0.weight"])
    m.zero_grad()                                   # Module.zero_grad: set_to_none -> the next backward starts afresh
    crit(m(xa), ya).backward()
    assert torch.equal(m.engine.store.gflat, ga)
