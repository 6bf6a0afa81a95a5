"""Probe: the 1x1 / ConvTranspose weight gradients (wgrad_kernel) against their split targets (AAU_WG_TARGET)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

B = 8


def run(name, H, Ci, Co, k=1, stride=1):
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    Ho = H * 2 if k == 2 else H
    dz = torch.randn(B, Ho, Ho, Co, device="cuda").to(torch.bfloat16)
    if k == 2:   # ConvTranspose(2, 2) weight gradient as the reference engine states it: taps over the 2x2 outputs
        d = ops.conv_desc(B, Ho, Ho, Co, Co, H, H, Ci, Ci, 2, 2, 2, 0, 1)
        src, dzz, dw = dz, x, torch.zeros(Ci, 4, Co, device="cuda")
    else:
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 1, 1, 1, 0, 1)
        src, dzz, dw = x, dz, torch.zeros(Co, 1, Ci, device="cuda")
    byts = (x.numel() + dz.numel()) * 2
    res = []
    for tgt in (None, 128, 256, 512, 1024, 2048):
        if tgt is None:
            os.environ.pop("AAU_WG_TARGET", None)
        else:
            os.environ["AAU_WG_TARGET"] = str(tgt)
        ws = torch.empty(ops.conv_wgrad_ws_bytes(d) // 4, device="cuda")
        for _ in range(3):
            ops.conv_wgrad(d, src, dzz, dw, ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_wgrad(d, src, dzz, dw, ws)
        e1.record()
        torch.cuda.synchronize()
        res.append((tgt, e0.elapsed_time(e1) * 1e3 / 20))
    print(f"{name:40s} {byts / 1e6:6.0f} MB  " + "  ".join(f"{t}: {u:6.1f}" for t, u in res), flush=True)


run("u2.att wgrad 96 -> 48 @256", 256, 96, 48)
run("u2.att wgrad 48 -> 96 @256 (Wg side)", 256, 48, 96)
run("u3.att wgrad 192 -> 96 @128", 128, 192, 96)
run("u4.att wgrad 384 -> 192 @64", 64, 384, 192)
run("u1.att-like 48 -> 24 @512", 512, 48, 24)
