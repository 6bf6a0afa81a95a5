"""Generate tests/golden/g9_trained_c48_512.npz by EXECUTING THE REFERENCE (build container only; recipe of make_golden.py).

TEST INFRASTRUCTURE.  The metric-width fixture: base_c 48, 1x512x512 frames -- the configuration
bench.py measures -- with weights the reference trained briefly on synthetic phantoms so that the masks are decisive, then
ROUNDED TO bf16 (the rounded weights are loaded back into the reference before anything is recorded).  Holds: the weights
(uint16 bf16 bit patterns, ~42 MB), a 4-frame validation set, eval logits (fp16), evaluate() Dice / IoU (pipeline:235-241),
per-frame integer-count Dice / IoU (eval_segmentation_batch.py:41-49) and the foreground pixel counts."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=90)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g9_trained_c48_512.npz"))
    a = ap.parse_args()
    torch.set_num_threads(8)
    ref, evalseg = import_reference()
    from att_aspp_unet_amd import synth
    args = argparse.Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
    crit = ref.build_criterion(args, ref.ComboLoss(), ref.EdgeLoss())
    torch.manual_seed(2026)
    net = ref.AttentionASPPUNet(base_c=48)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=ref.WEIGHT_DECAY)
    net.train()
    t0 = time.time()
    for step in range(a.train_steps):
        xb, yb = synth.make_frames(2, 512, seed=7000 + step)
        opt.zero_grad(set_to_none=True)
        loss = crit(net(xb), yb)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), ref.GRAD_CLIP)
        opt.step()
        if step % 5 == 0 or step == a.train_steps - 1:
            print(f"train step {step}: loss {loss.item():.4f}  ({time.time() - t0:.0f} s)", flush=True)
    sd = {k: (v.to(torch.bfloat16).to(torch.float32) if v.dtype == torch.float32 else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.float32:
            out["sd_bf16/" + k] = v.to(torch.bfloat16).view(torch.int16).numpy().copy()
        else:
            out["sd_raw/" + k] = v.numpy().copy()
    xv, yv = synth.make_frames(4, 512, seed=299, force_pattern="ppnp")
    out["x"], out["y"] = xv.numpy(), (yv.numpy() > 0).astype(np.uint8)
    net.eval()
    d, i = ref.evaluate(net, [(xv[:2], yv[:2]), (xv[2:], yv[2:])], torch.device("cpu"))
    out["evaluate_dice"], out["evaluate_iou"] = np.float64(d), np.float64(i)
    with torch.no_grad():
        lv = net(xv)
    out["eval_logits"] = lv.numpy().astype(np.float16)
    masks = (torch.sigmoid(lv) > 0.5).numpy().astype(np.uint8)[:, 0] * 255
    gts = (yv.numpy()[:, 0] > 0).astype(np.uint8) * 255
    out["seg_dice"] = np.array([evalseg.dice(m, t) for m, t in zip(masks, gts)])
    out["seg_iou"] = np.array([evalseg.iou(m, t) for m, t in zip(masks, gts)])
    out["mask_counts"] = np.array([int((m > 0).sum()) for m in masks])
    # how decisive: share of pixels whose |logit| is below 0.5 (a bf16 activation path moves logits by a few 1e-2)
    out["undecided_share"] = np.float64(float((lv.abs() < 0.5).float().mean()))
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, os.path.getsize(a.out), "bytes; evaluate dice/iou", d, i, "seg dice", out["seg_dice"], "counts",
          out["mask_counts"], "undecided", out["undecided_share"])


if __name__ == "__main__":
    main()
