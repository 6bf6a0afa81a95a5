"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE.  Run as ``python oracle/make_golden.py`` in the container that
has ``/root/reference`` mounted.  The reference module
``attention_aspp_unet_pipeline_stage.py`` imports cv2 / albumentations / skimage at
the top for its data-augmentation and post-processing code; none of them is
touched by the hot path (model ``:59-127``, losses ``:173-232``, evaluate
``:235-241``), so empty placeholder modules are registered for them before the
import (recipe: SURVEY.md Appendix A).  Nothing of the reference is written into
the repository: the fixtures hold inputs and outputs only.

Fixtures
  g1_step_c8_128.npz   seed-2025 init (base_c 8), x/y 2x1x128x128 ("pn"), eval
                       logits, train logits (dropout p=0), BN running stats after
                       the train forward, main/finetune losses, all parameter
                       gradients, pre-clip grad norm, parameters after one
                       clip + AdamW step.
  g2_loss.npz          criterion values and d(loss)/d(logits) for fixed logits /
                       targets: mixed batch and all-negative batch, both stages.
  g3_aspp_rates.npz    ASPP(16, 32, rates=(2,5,9)) forward, eval and train.  (The
                       reference's ASPP hard-codes ``out_c*5`` input channels for
                       its projection, pipeline:78, so it raises for any number
                       of rates other than three: the 4-rate set of BASELINE
                       config 5 cannot be executed by the reference and stays
                       "parity unpinned".)
  g5_trained_step.npz  train-mode forward/backward (dropout p=0) of the g4 weights on the g4
                       validation frames: logits, loss, all parameter gradients, grad norm.
  g4_trained_c8_128.npz briefly trained weights (base_c 8) with decisive masks,
                       an 8-frame validation set, logits, evaluate() Dice/IoU and
                       the evalseg integer-count Dice/IoU per frame.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def import_reference():
    sys.dont_write_bytecode = True  # /root/reference is read-only
    sys.path.insert(0, "/root/reference")

    def placeholder(name, attrs=()):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
        return m

    placeholder("cv2")
    alb = placeholder("albumentations", ["CLAHE", "Compose", "HorizontalFlip", "MedianBlur",
                                         "RandomBrightnessContrast", "RandomGamma", "Resize", "ToFloat"])
    alb.pytorch = placeholder("albumentations.pytorch", ["ToTensorV2"])
    placeholder("skimage").measure = placeholder("skimage.measure", ["label"])
    import attention_aspp_unet_pipeline_stage as ref
    import eval_segmentation_batch as evalseg
    return ref, evalseg


def sd_to_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=400)
    a = ap.parse_args()
    torch.set_num_threads(8)
    ref, evalseg = import_reference()
    from att_aspp_unet_amd import synth
    os.makedirs(OUT, exist_ok=True)
    ns = lambda **k: argparse.Namespace(**k)
    main_args = ns(stage="main", edge_w=0.05, neg_bce_w=0.05)
    fine_args = ns(stage="finetune", edge_w=0.05, neg_bce_w=0.05)

    # ---------------- g1: one full step ----------------
    torch.manual_seed(2025)
    net = ref.AttentionASPPUNet(base_c=8)
    x, y = synth.make_frames(2, 128, seed=11, force_pattern="pn")
    out = {"x": x.numpy(), "y": y.numpy()}
    out.update(sd_to_np(net.state_dict(), "init/"))
    net.eval()
    with torch.no_grad():
        out["eval_logits"] = net(x).numpy()
    net.train()
    net.bridge.project[3].p = 0.0
    crit_m = ref.build_criterion(main_args, ref.ComboLoss(), ref.EdgeLoss())
    crit_f = ref.build_criterion(fine_args, ref.ComboLoss(), ref.EdgeLoss())
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4, weight_decay=ref.WEIGHT_DECAY)
    opt.zero_grad(set_to_none=True)
    logits = net(x)
    logits.retain_grad()
    loss = crit_m(logits, y)
    out["train_logits"] = logits.detach().numpy()
    out["loss_main"] = np.float64(loss.item())
    out["loss_finetune"] = np.float64(crit_f(logits.detach(), y).item())
    out.update(sd_to_np({k: v for k, v in net.state_dict().items() if "running" in k or "num_batches" in k},
                        "after_fwd/"))
    loss.backward()
    out["dlogits"] = logits.grad.numpy()
    for k, p in net.named_parameters():
        out["grad/" + k] = p.grad.detach().numpy().copy()
    gn = torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)
    out["grad_norm"] = np.float64(float(gn))
    opt.step()
    for k, p in net.named_parameters():
        out["after_step/" + k] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g1_step_c8_128.npz"), **out)
    print("g1: loss", out["loss_main"], "gnorm", out["grad_norm"])

    # ---------------- g2: criterion alone ----------------
    g = torch.Generator().manual_seed(7)
    out = {}
    for tag, pattern in (("mixed", "pnpn"), ("allneg", "nnnn"), ("allpos", "pppp")):
        _, t = synth.make_frames(4, 64, seed=5, force_pattern=pattern)
        l = (torch.randn(4, 1, 64, 64, generator=g) * 3.0 + (t - 0.5) * 2.0)
        out[f"{tag}/logits"], out[f"{tag}/targets"] = l.numpy(), t.numpy()
        for stage, crit in (("main", crit_m), ("finetune", crit_f)):
            lg = l.clone().requires_grad_(True)
            v = crit(lg, t)
            v.backward()
            out[f"{tag}/{stage}/loss"] = np.float64(v.item())
            out[f"{tag}/{stage}/dlogits"] = lg.grad.numpy()
        with torch.no_grad():
            out[f"{tag}/dice_eval"] = np.float64(1 - ref.DiceLoss()(l, t).item())
            out[f"{tag}/iou"] = np.float64(ref.iou_score(l, t))
    np.savez_compressed(os.path.join(OUT, "g2_loss.npz"), **out)
    print("g2:", {k: float(v) for k, v in out.items() if k.endswith("loss")})

    # ---------------- g3: ASPP with non-default rates ----------------
    torch.manual_seed(3)
    aspp = ref.ASPP(16, 32, rates=(2, 5, 9))
    xa = torch.randn(2, 16, 32, 32, generator=torch.Generator().manual_seed(4))
    out = {"x": xa.numpy()}
    out.update(sd_to_np(aspp.state_dict(), "init/"))
    aspp.eval()
    with torch.no_grad():
        out["eval_out"] = aspp(xa).numpy()
    aspp.train()
    aspp.project[3].p = 0.0
    with torch.no_grad():
        out["train_out"] = aspp(xa).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_aspp_rates.npz"), **out)

    # ---------------- g4: briefly trained weights ----------------
    torch.manual_seed(2025)
    net = ref.AttentionASPPUNet(base_c=8)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=ref.WEIGHT_DECAY)
    net.train()
    for step in range(a.train_steps):
        xb, yb = synth.make_frames(4, 128, seed=1000 + step)
        opt.zero_grad(set_to_none=True)
        loss = crit_m(net(xb), yb)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)
        opt.step()
        if step % 40 == 0 or step == a.train_steps - 1:
            print(f"g4 train step {step}: loss {loss.item():.4f}", flush=True)
    xv, yv = synth.make_frames(8, 128, seed=99, force_pattern="pppnpppn")
    out = {"x": xv.numpy(), "y": yv.numpy()}
    out.update(sd_to_np(net.state_dict(), "sd/"))
    d, i = ref.evaluate(net, [(xv[:4], yv[:4]), (xv[4:], yv[4:])], torch.device("cpu"))
    out["evaluate_dice"], out["evaluate_iou"] = np.float64(d), np.float64(i)
    with torch.no_grad():
        lv = net(xv)  # evaluate() left the model in eval mode
    out["eval_logits"] = lv.numpy()
    masks = (torch.sigmoid(lv) > 0.5).numpy().astype(np.uint8)[:, 0] * 255
    gts = (yv.numpy()[:, 0] > 0).astype(np.uint8) * 255
    out["seg_dice"] = np.array([evalseg.dice(m, t) for m, t in zip(masks, gts)])
    out["seg_iou"] = np.array([evalseg.iou(m, t) for m, t in zip(masks, gts)])
    with torch.inference_mode():  # as predict() does, pipeline:398
        out["tta_prob0"] = ref.predict_prob_tta(net, xv[:1])
    np.savez_compressed(os.path.join(OUT, "g4_trained_c8_128.npz"), **out)
    print("g4: evaluate dice/iou", d, i, "seg dice", out["seg_dice"])

    # ---------------- g5: one train-mode step at the trained weights (well conditioned gradients) ----------------
    net.train()
    net.bridge.project[3].p = 0.0
    for p_ in net.parameters():
        p_.grad = None
    lt = net(xv)
    loss = crit_m(lt, yv)
    loss.backward()
    out5 = {"train_logits": lt.detach().numpy().copy(), "loss_main": np.float64(loss.item())}
    for k, p_ in net.named_parameters():
        out5["grad/" + k] = p_.grad.detach().numpy().copy()
    out5["grad_norm"] = np.float64(float(torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)))
    np.savez_compressed(os.path.join(OUT, "g5_trained_step.npz"), **out5)
    print("g5: loss", out5["loss_main"], "gnorm", out5["grad_norm"])


if __name__ == "__main__":
    main()
