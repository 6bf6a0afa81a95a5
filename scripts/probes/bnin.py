"""Timing probe: the strip conv with BatchNorm + ReLU applied on its input (aau_conv_igemm_bnin) against bn_act + conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

B = 8


def timeit(fn, n=20, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, H, Ci, Co in (("d1.1", 512, 48, 48), ("d2.1", 256, 96, 96), ("d2.0-like", 256, 48, 96), ("u1.c0-like", 512, 96, 48)):
    M = B * H * H
    z = torch.randn(M, Ci, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(z)
    scale, shift = torch.randn(Ci, device="cuda"), torch.randn(Ci, device="cuda")
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 9, cp, device="cuda") / (Ci * 9) ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
    stats = ops.stats_buffer(Co)
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, cp)
    t_act = timeit(lambda: ops.bn_act(z, Ci, y, Ci, scale, shift, M, Ci))
    t_conv = timeit(lambda: ops.conv_igemm(d, y, w, out, stats=stats))
    t_both = timeit(lambda: (ops.bn_act(z, Ci, y, Ci, scale, shift, M, Ci), ops.conv_igemm(d, y, w, out, stats=stats)))
    t_bnin = timeit(lambda: ops.conv_igemm_bnin(d, z, scale, shift, w, out, stats=stats))
    print(f"{name:10s} {Ci:3d}->{Co:3d} @{H}: bn_act {t_act:6.1f}  conv {t_conv:6.1f}  bn_act+conv {t_both:6.1f}  bnin {t_bnin:6.1f} us", flush=True)

print("weight gradients:")
for name, H, Ci, Co in (("d1.1", 512, 48, 48),):
    M = B * H * H
    z = torch.randn(M, Ci, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(z)
    dz = torch.randn(M, Co, device="cuda").to(torch.bfloat16)
    scale, shift = torch.randn(Ci, device="cuda"), torch.randn(Ci, device="cuda")
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1)
    ws = torch.empty(ops.conv_wgrad_ws_bytes(d) // 4, device="cuda")
    dw = torch.zeros(Co, 9, Ci, device="cuda")
    t_w = timeit(lambda: ops.conv_wgrad(d, y, dz, dw, ws))
    t_wb = timeit(lambda: ops.conv_wgrad_bnin(d, z, scale, shift, dz, dw, ws))
    print(f"{name:10s} {Ci:3d}->{Co:3d} @{H}: wgrad {t_w:6.1f}  wgrad_bnin {t_wb:6.1f} us", flush=True)
