"""End-to-end GPU parity of the HIP engine (through the nn.Module surface) against the
reference-generated golden vectors and the CPU oracle.

Tolerances: the HIP path stores activations and GEMM operands in bf16 (fp32 accumulate,
fp32 master weights / statistics / loss).  Against the fp32 reference that gives
  logits:      max |diff| <= 2 % of max |logit|   (random init: logits are O(1e-2), errors are relative to that)
  loss:        |diff| <= 1e-3 relative
  Dice / IoU:  |diff| <= 1e-3 absolute (the BASELINE.json bar)
  gradients:   asserted at the TRAINED fixture weights; at the random initial weights the
               reference's own gradients move by a median 26 % per tensor under bf16
               rounding (ReLU / max-pool routing flips; see oracle.ref_cpu.emulate_bf16_storage),
               so there only global statistics (cosine, norm) are asserted.
"""
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


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def main_args(**kw):
    d = dict(stage="main", edge_w=0.05, neg_bce_w=0.05)
    d.update(kw)
    return Namespace(**d)


def flat_grads(named, ref=None):
    return torch.cat([(p.grad if ref is None else torch.from_numpy(ref["grad/" + k])).detach().float().cpu().flatten()
                      for k, p in named])


def test_eval_forward_random_init_matches_reference(A, golden):
    g = golden("g1_step_c8_128.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(_sd(g, "init/"), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]).cuda())
    assert out.shape == (2, 1, 128, 128) and out.dtype == torch.float32
    assert rel(out, g["eval_logits"]) < 2e-2


def test_trained_weights_logits_dice_iou_tta(A, golden):
    g = golden("g4_trained_c8_128.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(_sd(g, "sd/"), strict=True)
    m = m.cuda().eval()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    with torch.no_grad():
        lv = m(x)
    assert rel(lv, g["eval_logits"]) < 1.5e-2
    # evaluate(): soft Dice / hard IoU, mean of per-batch means (pipeline:235-241)
    d, i = A.evaluate(m, [(x[:4], y[:4]), (x[4:], y[4:])], torch.device("cuda"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-3
    assert abs(i - float(g["evaluate_iou"])) < 1e-3
    # integer-count Dice / IoU of the binarised masks (eval_segmentation_batch.py:41-49)
    masks = (torch.sigmoid(lv) > 0.5).cpu().numpy().astype(np.uint8)[:, 0] * 255
    gts = (g["y"][:, 0] > 0).astype(np.uint8) * 255
    dice = np.array([O.seg_dice(a, b) for a, b in zip(masks, gts)])
    iou = np.array([O.seg_iou(a, b) for a, b in zip(masks, gts)])
    assert np.abs(dice - g["seg_dice"]).max() < 1e-3 and abs(dice.mean() - g["seg_dice"].mean()) < 1e-3
    assert np.abs(iou - g["seg_iou"]).max() < 2e-3
    prob = A.predict_prob_tta(m, x[:1])
    assert prob.shape == (128, 128) and np.abs(prob - g["tta_prob0"]).max() < 2e-2
    assert A.iou_score(lv, y) == pytest.approx(O.iou_score(torch.from_numpy(g["eval_logits"]), torch.from_numpy(g["y"])), abs=1e-3)


def test_train_forward_loss_and_running_stats_random_init(A, golden):
    g = golden("g1_step_c8_128.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(_sd(g, "init/"), strict=True)
    m = m.cuda().train()
    m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    lt = m(x)
    assert rel(lt, g["train_logits"]) < 3e-2
    crit = A.build_criterion(main_args(), A.ComboLoss(), A.EdgeLoss())
    critf = A.build_criterion(main_args(stage="finetune"), A.ComboLoss(), A.EdgeLoss())
    loss = crit(lt, y)
    assert abs(loss.item() - float(g["loss_main"])) < 1e-3 * float(g["loss_main"])
    assert abs(critf(lt.detach(), y).item() - float(g["loss_finetune"])) < 1e-3 * float(g["loss_finetune"])
    sd = m.state_dict()
    for k, v in _sd(g, "after_fwd/").items():
        if "num_batches" in k:
            assert int(sd[k].item()) == int(v.item()), k
        else:
            assert rel(sd[k], v) < 4e-2, k
    # global statistics of the gradient at the (chaotic) random initial point
    loss.backward()
    named = list(m.named_parameters())
    ge, gr = flat_grads(named), flat_grads(named, g)
    cos = float(torch.dot(ge, gr) / ge.norm() / gr.norm())
    assert cos > 0.99, cos
    assert abs(float(ge.norm()) - float(g["grad_norm"])) < 0.02 * float(g["grad_norm"])


def test_trained_step_gradients_match_reference(A, golden):
    g4, g5 = golden("g4_trained_c8_128.npz"), golden("g5_trained_step.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(_sd(g4, "sd/"), strict=True)
    m = m.cuda().train()
    m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g4["x"]).cuda(), torch.from_numpy(g4["y"]).cuda()
    crit = A.build_criterion(main_args(), A.ComboLoss(), A.EdgeLoss())
    lt = m(x)
    loss = crit(lt, y)
    loss.backward()
    assert rel(lt, g5["train_logits"]) < 2e-2
    assert abs(loss.item() - float(g5["loss_main"])) < 2e-3 * float(g5["loss_main"])
    named = list(m.named_parameters())
    ge, gr = flat_grads(named), flat_grads(named, g5)
    cos = float(torch.dot(ge, gr) / ge.norm() / gr.norm())
    assert cos > 0.9995, cos
    assert abs(float(ge.norm()) - float(g5["grad_norm"])) < 0.01 * float(g5["grad_norm"])
    # Per-tensor table (scripts/grad_table.py prints it; round 2, deterministic BatchNorm statistics): median max-norm
    # error 1.7 %, 90th percentile 6.4 %.  Every tensor above 10 % carries < 0.05 % of the gradient norm and is a
    # near-cancelling sum: the scalar psi bias of u3 (40 %), the BatchNorm bias of the ASPP projection (26 %) and of
    # d4.0 (13 %), and three weight tensors of the 8x8 / 16x16 levels (d4.1, bridge.project.0, bridge.blocks.2.0: 11-13 %
    # in max norm, 3-6 % in L2).  So: tight L2 bounds where the gradient lives, an absolute bound (relative to the global
    # norm) everywhere, and the max-norm statistics as a regression guard.
    G = float(gr.double().norm())
    rows = []
    for k, p in named:
        r = torch.from_numpy(g5["grad/" + k]).double()
        e = p.grad.detach().double().cpu()
        rows.append((rel(e, r), float((e - r).norm() / (r.norm() + 1e-30)), float((e - r).norm() / G), float(r.norm() / G), k))
    errs = sorted(r[0] for r in rows)
    med, p90, worst = errs[len(errs) // 2], errs[int(len(errs) * 0.9)], max(rows)
    # measured (round 4): median 1.7 %, 90th percentile 6.4 %, worst 31.7 % (u3.att.psi.1.bias)
    assert med < 0.025, (med, p90, worst)
    assert p90 < 0.08, (med, p90, worst)
    assert worst[0] < 0.45, worst
    # what sits above 10 % must be one of the six near-cancelling sums named above, carrying next to nothing of the gradient
    loud = [r for r in rows if r[0] > 0.10]
    known = {"u3.att.psi.1.bias", "bridge.project.1.bias", "d4.0.block.1.bias", "bridge.project.0.weight",
             "d4.1.block.0.weight", "bridge.blocks.2.0.weight"}
    assert {r[4] for r in loud} <= known and all(r[3] < 2e-3 for r in loud), loud
    heavy = [r for r in rows if r[3] >= 1e-3]                # the tensors that carry 99.99 % of the gradient
    assert len(heavy) >= 30 and max(r[1] for r in heavy) < 0.07, sorted(heavy, key=lambda r: -r[1])[:3]
    assert max(r[2] for r in rows) < 5e-3, sorted(rows, key=lambda r: -r[2])[:3]
    for k, p in named:  # gradients are views of the engine's flat buffer with the parameter's own strides
        assert p.grad.shape == p.shape and p.grad.stride() == p.stride()


def test_backward_is_consistent_with_forward_directional_derivative(A, golden):
    """(L(theta + e*d) - L(theta - e*d)) / 2e  ==  <grad, d>  in the engine's own arithmetic."""
    g4 = golden("g4_trained_c8_128.npz")
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(_sd(g4, "sd/"), strict=True)
    m = m.cuda().train()
    m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g4["x"]).cuda(), torch.from_numpy(g4["y"]).cuda()
    crit = A.build_criterion(main_args(), A.ComboLoss(), A.EdgeLoss())
    crit(m(x), y).backward()
    params = [p for p in m.parameters()]
    grads = [p.grad.detach().clone() for p in params]
    gn = torch.sqrt(sum((g_ ** 2).sum() for g_ in grads))
    for seed, use_grad in ((0, True), (1, False), (2, False)):
        if use_grad:
            d = [g_ / gn for g_ in grads]
        else:
            gen = torch.Generator(device="cuda").manual_seed(seed)
            d = [torch.randn(p.shape, device="cuda", generator=gen) * p.detach().abs().mean() for p in params]
            # keep the random direction where the loss is actually sensitive: mix with the gradient direction
            dn = torch.sqrt(sum((t ** 2).sum() for t in d))
            d = [t / dn * 0.5 + g_ / gn * 0.5 for t, g_ in zip(d, grads)]
        pred = float(sum((g_ * t).sum() for g_, t in zip(grads, d)))
        eps = 2e-2
        vals = []
        with torch.no_grad():
            for sgn in (+1, -1):
                for p, t in zip(params, d):
                    p.add_(t, alpha=sgn * eps)
                vals.append(crit(m(x), y).item())
                for p, t in zip(params, d):
                    p.add_(t, alpha=-sgn * eps)
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - pred) < 0.05 * abs(pred) + 1e-4, (seed, fd, pred)


def test_optimizer_step_and_short_training_tracks_oracle(A):
    """20 steps of the full step (fwd + criterion + bwd + clip + AdamW) on the same data / init as
    the CPU oracle (dropout off): the loss curves agree and the loss decreases."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(7)
    ref = O.AttentionASPPUNet(base_c=8)
    m = A.AttentionASPPUNet(base_c=8)
    m.load_state_dict(ref.state_dict(), strict=True)
    init = torch.cat([p.detach().flatten().clone() for p in ref.parameters()])
    m = m.cuda().train()
    ref.train()
    ref.bridge.project[3].p = 0.0
    m.bridge.project[3].p = 0.0
    opt_r = O.make_optimizer(ref, 2e-3)
    opt = A.FusedAdamW(m, lr=2e-3)
    crit_r = O.build_criterion(O.default_args(), O.ComboLoss(), O.EdgeLoss())
    crit = A.build_criterion(main_args(), A.ComboLoss(), A.EdgeLoss())
    lr_, le_ = [], []
    for step in range(20):
        x, y = synth.make_frames(4, 64, seed=500 + step)
        l_r, _ = O.train_step(ref, opt_r, crit_r, x, y)
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x.cuda()), y.cuda())
        loss.backward()
        opt.step()
        lr_.append(l_r)
        le_.append(loss.item())
    lr_, le_ = np.array(lr_), np.array(le_)
    assert np.abs(le_ - lr_).max() < 0.05 * lr_.max(), (lr_, le_)
    assert le_[-5:].mean() < 0.9 * le_[:5].mean()
    # the accumulated parameter updates point the same way (Adam steps are +-lr per element, so
    # noisy small gradients flip individual signs; the update as a whole must agree)
    ue = torch.cat([p.detach().cpu().flatten() for p in m.parameters()]) - init
    ur = torch.cat([p.detach().flatten() for p in ref.parameters()]) - init
    cos = float(torch.dot(ue, ur) / ue.norm() / ur.norm())
    print(dict(curve_max_dev=float(np.abs(le_ - lr_).max() / lr_.max()), first3=float(np.abs(le_[:3] - lr_[:3]).max()), cos=cos))
    # 0.46 measured: after 20 sign-like Adam steps on a 64x64 problem the two trajectories have separated element by
    # element while the losses still agree to a few percent - this is a guard against a wrong update rule, not a metric
    assert cos > 0.4, cos


@pytest.mark.parametrize("cfg", [dict(base_c=16, B=3, H=64, W=96), dict(base_c=8, B=2, H=48, W=32, rates=(2, 5, 9)),
                                 dict(base_c=8, B=2, H=64, W=64, rates=(6, 12, 18, 24))])
def test_other_shapes_and_rates_against_emulated_oracle(A, cfg):
    """Non-square frames, odd batch, wider model and the ``rates`` extension (4 rates = BASELINE
    config 5's ASPP; parity unpinned by the reference, which cannot run 4 rates) -- against the
    CPU oracle with bf16 storage emulation."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(11)
    rates = cfg.get("rates", (6, 12, 18))
    ref = O.AttentionASPPUNet(base_c=cfg["base_c"], rates=rates)
    m = A.AttentionASPPUNet(base_c=cfg["base_c"], rates=rates)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(cfg["B"], 1, cfg["H"], cfg["W"], generator=g)
    ref.train(); m.train()
    ref.bridge.project[3].p = 0.0; m.bridge.project[3].p = 0.0
    O.emulate_bf16_storage(ref)
    with torch.no_grad():
        lo = ref(x)
        le = m(x.cuda())
    # train mode with tiny batches is the noisiest setting (batch statistics over 2-3 frames)
    # (measured, round 4: max 4.1-4.3 %, mean 0.49-0.55 % over the three configurations; eval 0.03-0.24 %)
    assert rel(le, lo) < 6e-2
    assert float((le.cpu() - lo).abs().mean() / lo.abs().max()) < 8e-3
    ref.eval(); m.eval()
    with torch.no_grad():
        assert rel(m(x.cuda()), ref(x)) < 8e-3


def test_benchmark_configuration_step_matches_oracle(A):
    """The configuration bench.py measures (base_c 48, 1x512x512; batch 4 here to keep the CPU oracle at a few seconds):
    this is where the resident-weight / two-stream 3x3 kernels, the slab split-K weight gradients, the fused network
    head and the fused first-layer backward are actually selected.  One full training step against the CPU oracle
    with bf16 storage emulation (random init, so gradients are compared by global statistics, see module docstring)."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(5)
    ref = O.AttentionASPPUNet(base_c=48)
    m = A.AttentionASPPUNet(base_c=48)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.cuda().train()
    ref.train()
    ref.bridge.project[3].p = 0.0
    m.bridge.project[3].p = 0.0
    x, y = synth.make_frames(4, 512, seed=21)
    O.emulate_bf16_storage(ref)
    args = main_args()
    lo = ref(x)
    loss_o = O.build_criterion(args, O.ComboLoss(), O.EdgeLoss())(lo, y)
    loss_o.backward()
    step = A.TrainStep(m, A.FusedAdamW(m, lr=0.0), args, None)      # lr 0: gradients stay inspectable, weights fixed
    loss_e = float(step(x.cuda(), y.cuda()).item())
    plan = m._plan_for(x.cuda())
    le = plan.logits.clone()
    named = list(m.named_parameters())
    ge = torch.cat([p.grad.detach().double().cpu().flatten() for _, p in named])
    gr = torch.cat([p.grad.detach().double().flatten() for _, p in ref.named_parameters()])
    cos = float(torch.dot(ge, gr) / ge.norm() / gr.norm())
    stats = dict(logit_max=rel(le, lo), logit_mean=float((le.cpu() - lo.detach()).abs().mean() / lo.detach().abs().max()),
                 loss=(loss_e, float(loss_o)), cos=cos, gnorm=(float(ge.norm()), float(gr.norm())))
    print(stats)
    # 1M logits: the maximum is a tail statistic of the bf16 path (3 % at the 65k-pixel fixtures)
    assert stats["logit_max"] < 6e-2 and stats["logit_mean"] < 4e-3, stats
    assert abs(loss_e - float(loss_o)) < 2e-3 * float(loss_o), stats
    # measured: cosine 0.9999, norm within 0.02 % -- the gates sit one order of magnitude above that
    assert cos > 0.999, stats
    assert abs(float(ge.norm()) - float(gr.norm())) < 0.005 * float(gr.norm()), stats
    # running statistics of the first and the last BatchNorm
    sd_e, sd_o = m.state_dict(), ref.state_dict()
    for k in ("d1.0.block.1.running_mean", "d1.0.block.1.running_var", "u1.conv.1.block.1.running_mean",
              "u1.conv.1.block.1.running_var"):
        assert rel(sd_e[k], sd_o[k]) < 4e-2, k


def test_dropout_draws_a_new_mask_every_step_also_under_graph_replay(A):
    """The Dropout(0.1) of the ASPP projection (pipeline:78) is counter based with a DEVICE-resident seed that the
    forward advances on the stream, so a training step captured as a hipGraph does not replay one frozen mask."""
    from att_aspp_unet_amd import synth
    torch.manual_seed(3)
    m = A.AttentionASPPUNet(base_c=8).cuda().train()
    assert m.bridge.project[3].p == 0.1
    x, _ = synth.make_frames(2, 64, seed=4)
    x = x.cuda()
    with torch.no_grad():
        a, b = m(x).clone(), m(x).clone()
        assert not torch.equal(a, b)                           # eager: two steps, two masks
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            out = m(x)
        torch.cuda.current_stream().wait_stream(s)
        g.replay(); r1 = out.clone()
        g.replay(); r2 = out.clone()
        torch.cuda.synchronize()
        assert not torch.equal(r1, r2)
        # and dropout is still dropout: ~10 % effect, not garbage
        m.eval()
        e = m(x)
        assert float((r1 - r2).abs().max()) < 2.0 * float(e.abs().max()) + 1.0


def _g7_model(A, golden):
    g = golden("g7_trained_c16_256.npz")
    sd = {}
    for k, v in g.items():
        if k.startswith("sd_bf16/"):
            sd[k[8:]] = torch.from_numpy(v.copy()).view(torch.bfloat16).float()
        elif k.startswith("sd_raw/"):
            sd[k[7:]] = torch.from_numpy(v.copy())
    m = A.AttentionASPPUNet(base_c=16)
    m.load_state_dict(sd, strict=True)
    return m.cuda(), g


def test_realistic_width_dice_within_1e3_of_reference(A, golden):
    """base_c 16, 1x256x256, weights trained with the REFERENCE (oracle/make_golden_c16.py): the BASELINE.json bar
    |Dice_build - Dice_ref| <= 1e-3 at a realistic width, for evaluate() (pipeline:235-241) and for the integer-count
    Dice / IoU of eval_segmentation_batch.py:41-49 on the thresholded masks."""
    m, g = _g7_model(A, golden)
    m.eval()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    with torch.no_grad():
        l = m(x)
    ref_l = torch.from_numpy(g["eval_logits"].astype(np.float32))
    assert rel(l, ref_l) < 1.5e-2, rel(l, ref_l)
    d, i = A.evaluate(m, [(x[:4], y[:4]), (x[4:], y[4:])], torch.device("cuda"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-3 and abs(i - float(g["evaluate_iou"])) < 1e-3, (d, i)
    masks = (torch.sigmoid(l) > 0.5).to(torch.uint8)[:, 0]
    gts = (y[:, 0] > 0).to(torch.uint8)
    for k in range(8):
        assert abs(A.evalseg.dice(masks[k], gts[k]) - float(g["seg_dice"][k])) < 1e-3, k
        assert abs(A.evalseg.iou(masks[k], gts[k]) - float(g["seg_iou"][k])) < 2e-3, k
        assert abs(int(masks[k].sum()) - int(g["mask_counts"][k])) <= max(8, 0.002 * int(g["mask_counts"][k])), k
    tta = A.predict_prob_tta(m, x[:1])
    assert np.abs(tta - g["tta_prob0"].astype(np.float32)).max() < 2e-2
    assert float(np.mean((tta > 0.5) != (g["tta_prob0"].astype(np.float32) > 0.5))) < 1e-3


def test_metric_width_dice_within_1e3_of_reference(A, golden):
    """base_c 48, 1x512x512 (the benchmark configuration) with reference-trained weights (oracle/make_golden_c48.py): the
    BASELINE.json bar |Dice_build - Dice_ref| <= 1e-3 at the metric width, for evaluate() (pipeline:235-241) and for the
    integer-count Dice / IoU of eval_segmentation_batch.py:41-49 on the thresholded masks."""
    g = golden("g9_trained_c48_512.npz")
    sd = {}
    for k, v in g.items():
        if k.startswith("sd_bf16/"):
            sd[k[8:]] = torch.from_numpy(v.copy()).view(torch.bfloat16).float()
        elif k.startswith("sd_raw/"):
            sd[k[7:]] = torch.from_numpy(v.copy())
    m = A.AttentionASPPUNet(base_c=48)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).float().cuda()
    with torch.no_grad():
        l = m(x)
    ref_l = torch.from_numpy(g["eval_logits"].astype(np.float32))
    assert rel(l, ref_l) < 2e-2, rel(l, ref_l)
    d, i = A.evaluate(m, [(x[:2], y[:2]), (x[2:], y[2:])], torch.device("cuda"))
    assert abs(d - float(g["evaluate_dice"])) < 1e-3 and abs(i - float(g["evaluate_iou"])) < 1e-3, (d, i)
    masks = (torch.sigmoid(l) > 0.5).to(torch.uint8)[:, 0]
    gts = (y[:, 0] > 0).to(torch.uint8)
    for k in range(4):
        assert abs(A.evalseg.dice(masks[k], gts[k]) - float(g["seg_dice"][k])) < 1e-3, k
        assert abs(A.evalseg.iou(masks[k], gts[k]) - float(g["seg_iou"][k])) < 2e-3, k
        assert abs(int(masks[k].sum()) - int(g["mask_counts"][k])) <= max(8, 0.002 * int(g["mask_counts"][k])), k


def test_realistic_width_train_step_matches_reference(A, golden):
    m, g = _g7_model(A, golden)
    m.train()
    m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    crit = A.build_criterion(main_args(), A.ComboLoss(), A.EdgeLoss())
    loss = crit(m(x), y)
    loss.backward()
    assert abs(loss.item() - float(g["train_loss"])) < 2e-3 * float(g["train_loss"])
    named = dict(m.named_parameters())
    assert list(named) == list(g["grad_names"])
    got = np.array([float(p.grad.double().norm()) for p in named.values()])
    want = g["grad_norms"]
    assert abs(np.linalg.norm(got) - float(g["grad_norm"])) < 0.01 * float(g["grad_norm"])
    heavy = want >= 1e-3 * np.linalg.norm(want)
    assert np.all(np.abs(got[heavy] - want[heavy]) < 0.05 * want[heavy]), \
        sorted(zip(np.abs(got - want)[heavy] / want[heavy], np.array(list(named))[heavy]))[-3:]
