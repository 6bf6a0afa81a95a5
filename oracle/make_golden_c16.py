"""Generate tests/golden/g7_trained_c16_256.npz by EXECUTING THE REFERENCE (build container only; recipe of make_golden.py).

TEST INFRASTRUCTURE.  A realistic-width fixture: base_c 16, 1x256x256 frames, weights trained for a
few hundred reference steps so that the masks are decisive, then ROUNDED TO bf16 (halves the fixture; the rounded weights
are loaded back into the reference before anything is recorded, so every stored number belongs to exactly these weights).
Holds: the weights (uint16 bf16 bit patterns), an 8-frame validation set, eval logits, evaluate() Dice / IoU
(pipeline:235-241), per-frame integer-count Dice / IoU (eval_segmentation_batch.py:41-49), the TTA probability of frame 0,
and one train-mode step at these weights (loss, gradient norm, per-tensor gradient norms)."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=220)
    a = ap.parse_args()
    torch.set_num_threads(8)
    ref, evalseg = import_reference()
    from att_aspp_unet_amd import synth
    args = argparse.Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    crit = ref.build_criterion(args, ref.ComboLoss(), ref.EdgeLoss())
    torch.manual_seed(2025)
    net = ref.AttentionASPPUNet(base_c=16)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-3, weight_decay=ref.WEIGHT_DECAY)
    net.train()
    for step in range(a.train_steps):
        xb, yb = synth.make_frames(4, 256, seed=5000 + step)
        opt.zero_grad(set_to_none=True)
        loss = crit(net(xb), yb)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)
        opt.step()
        if step % 20 == 0 or step == a.train_steps - 1:
            print(f"train step {step}: loss {loss.item():.4f}", flush=True)
    # round to bf16 and reload
    sd = {k: (v.to(torch.bfloat16).to(torch.float32) if v.dtype == torch.float32 else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.float32:
            out["sd_bf16/" + k] = v.to(torch.bfloat16).view(torch.int16).numpy().copy()
        else:
            out["sd_raw/" + k] = v.numpy().copy()
    xv, yv = synth.make_frames(8, 256, seed=199, force_pattern="pppnpppp")
    out["x"], out["y"] = xv.numpy(), yv.numpy()
    d, i = ref.evaluate(net, [(xv[:4], yv[:4]), (xv[4:], yv[4:])], torch.device("cpu"))
    out["evaluate_dice"], out["evaluate_iou"] = np.float64(d), np.float64(i)
    with torch.no_grad():
        lv = net(xv)
    out["eval_logits"] = lv.numpy().astype(np.float16)          # logits O(10): fp16 keeps 3 decimals, plenty for a 1e-2 check
    masks = (torch.sigmoid(lv) > 0.5).numpy().astype(np.uint8)[:, 0] * 255
    gts = (yv.numpy()[:, 0] > 0).astype(np.uint8) * 255
    out["seg_dice"] = np.array([evalseg.dice(m, t) for m, t in zip(masks, gts)])
    out["seg_iou"] = np.array([evalseg.iou(m, t) for m, t in zip(masks, gts)])
    out["mask_counts"] = np.array([int((m > 0).sum()) for m in masks])
    with torch.inference_mode():
        out["tta_prob0"] = ref.predict_prob_tta(net, xv[:1]).astype(np.float16)
    net.train()
    net.bridge.project[3].p = 0.0
    for p_ in net.parameters():
        p_.grad = None
    lt = net(xv)
    loss = crit(lt, yv)
    loss.backward()
    out["train_loss"] = np.float64(loss.item())
    named = dict(net.named_parameters())
    out["grad_names"] = np.array(list(named.keys()))
    out["grad_norms"] = np.array([float(p.grad.double().norm()) for p in named.values()])
    out["grad_norm"] = np.float64(float(torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)))
    path = os.path.join(ROOT, "tests", "golden", "g7_trained_c16_256.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; evaluate dice/iou", d, i, "seg dice", out["seg_dice"])


if __name__ == "__main__":
    main()
