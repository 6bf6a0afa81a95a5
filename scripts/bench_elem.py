"""Micro-benchmark of the first-layer backward tail at the headline shape (bs 8, 512x512, c=48):
aau_bn_bwd_apply + aau_conv1_wgrad against the fused aau_bn_bwd_apply_conv1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops

N, H, W, C = 8, 512, 512, 48
M = N * H * W


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(N, H, W, device="cuda")
z = torch.randn(M, C, device="cuda").to(torch.bfloat16)
dy = torch.randn(M, C, device="cuda").to(torch.bfloat16)
dz = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
gamma, mean, invstd = torch.rand(C, device="cuda") + 0.5, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
scale, shift = gamma.clone(), torch.zeros(C, device="cuda")
red = torch.zeros(ops.STAT_REPLICAS, 2, C, device="cuda")
dg, db, dw = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, 9, device="cuda")
ws = torch.empty(ops.STAT_REPLICAS * C * 9, device="cuda")
ta = timeit(lambda: ops.bn_bwd_apply(z, C, dz, C, gamma, mean, invstd, red, dg, db, M, C, dy=dy, dyp=C, scale=scale, shift=shift, relu=1))
tw = timeit(lambda: ops.conv1_wgrad(x, dz, dw, N, H, W, C))
tf = timeit(lambda: ops.bn_bwd_apply_conv1(z, C, gamma, mean, invstd, red, dg, db, N, H, W, C, dy, C, scale, shift, x, dw, ws))
print(f"bn_bwd_apply {ta:.1f} us + conv1_wgrad {tw:.1f} us = {ta + tw:.1f} us;  fused {tf:.1f} us")
