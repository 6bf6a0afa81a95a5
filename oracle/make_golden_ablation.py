"""Generate tests/golden/g6_ablation.npz by EXECUTING THE REFERENCE's test_ablation.py (build container only).

TEST INFRASTRUCTURE.  Same recipe as make_golden.py: the file imports cv2 / albumentations / skimage at the top for data
and visualisation code that the model classes (:73-218) never touch, so placeholder modules are registered first (cv2
needs a placeholder that answers any attribute: a colour-map constant is read at import time).  Nothing of the reference
is written into the repository: the fixture holds inputs, outputs and per-tensor checksums only.

For every variant of VARIANTS (full / no attention / no ASPP / plain U-Net / att_depth 3) at base_c 8, seed 2025:
  key list and shapes of the state_dict, per-tensor fp64 sums of the seed-2025 initial weights (the restatement must
  reproduce them bit for bit from the same seed), x / y (2 x 1 x 64 x 64), eval-mode logits and attention maps,
  train-mode logits (dropout p = 0), the loss BCEWithLogits(logits, y), per-tensor gradient norms and a few full
  gradient tensors (attention, first and last layer)."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")

    class AnyAttr(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return 0

    def placeholder(name, attrs=()):
        m = AnyAttr(name)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
        return m

    placeholder("cv2")
    alb = placeholder("albumentations", ["CLAHE", "Compose", "HorizontalFlip", "MedianBlur", "RandomBrightnessContrast",
                                         "RandomGamma", "Resize", "ToFloat"])
    alb.pytorch = placeholder("albumentations.pytorch", ["ToTensorV2"])
    placeholder("skimage").measure = placeholder("skimage.measure", ["label"])
    import test_ablation as ab
    return ab


def main():
    ab = import_reference()
    from oracle.ablation_ref import VARIANTS
    from att_aspp_unet_amd import synth
    x, y = synth.make_frames(2, 64, seed=17, force_pattern="pn")
    out = {"x": x.numpy(), "y": y.numpy()}
    full_keys = ("u4.att.Wg.weight", "u4.att.psi.1.weight", "u4.att.psi.1.bias", "u3.att.Wx.weight", "d1.0.block.0.weight",
                 "out_conv.weight", "out_conv.bias", "u4.up.bias")
    for tag, kw in VARIANTS.items():
        torch.manual_seed(2025)
        net = ab.AttentionASPPUNet(base_c=8, **kw)
        sd = net.state_dict()
        out[f"{tag}/keys"] = np.array(list(sd.keys()))
        out[f"{tag}/shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[f"{tag}/init_sums"] = np.array([float(v.double().sum()) for v in sd.values()])
        net.eval()
        with torch.no_grad():
            l, (p3, p2) = net(x)
        out[f"{tag}/eval_logits"], out[f"{tag}/psi3"], out[f"{tag}/psi2"] = l.numpy(), p3.numpy(), p2.numpy()
        net.train()
        for m in net.bridge.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        l, _ = net(x)
        loss = F.binary_cross_entropy_with_logits(l, y)
        loss.backward()
        out[f"{tag}/train_logits"] = l.detach().numpy()
        out[f"{tag}/loss"] = np.float64(loss.item())
        named = dict(net.named_parameters())
        out[f"{tag}/grad_names"] = np.array(list(named.keys()))
        out[f"{tag}/grad_norms"] = np.array([float(p.grad.double().norm()) for p in named.values()])
        for k in full_keys:
            if k in named:
                out[f"{tag}/grad/{k}"] = named[k].grad.numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "g6_ablation.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
