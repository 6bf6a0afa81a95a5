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

# ---- attention-gate backward, middle pass (u2 / u3 / u4 shapes) ----
for name, Mg, Fi in (("u2", 8 * 256 * 256, 48), ("u3", 8 * 128 * 128, 96), ("u4", 8 * 64 * 64, 192)):
    R = ops.STAT_REPLICAS
    f32 = lambda *s: torch.randn(*s, device="cuda")
    dq, psi_pre = f32(Mg), f32(Mg)
    zg, zx = f32(Mg, Fi).to(torch.bfloat16), f32(Mg, Fi).to(torch.bfloat16)
    dsb = torch.empty(Mg, Fi, device="cuda", dtype=torch.bfloat16)
    one = lambda: torch.ones(1, device="cuda")
    vec = lambda: torch.rand(Fi, device="cuda") + 0.5
    red1, redg, redx = torch.zeros(R, 2, device="cuda"), torch.zeros(R, 2, Fi, device="cuda"), torch.zeros(R, 2, Fi, device="cuda")
    args = (dq, psi_pre, red1, one(), one(), one(), zg, zx, vec(), vec(), vec(), vec(), vec(), vec(), vec(), vec(), vec(),
            dsb, torch.zeros(R, Fi, device="cuda"), redg, redx, one(), one(), Mg, Fi)
    t = timeit(lambda: ops.gate_bwd2(*args))
    gb = (Mg * Fi * 2 * 3 + Mg * 8) / 1e9
    print(f"gate_bwd2 {name}: {t:.1f} us  ({gb / t * 1e3:.2f} TB/s)")

# ---- reduce -> apply pair at level-1 size (Infinity Cache residency experiment: AAU_REV_APPLY=1) ----
red.zero_()
def pair():
    ops.bn_bwd_reduce(z, C, dy, C, None, 0, None, C, scale, shift, mean, invstd, red, N, H, W, C, relu=1)
    ops.bn_bwd_apply(z, C, dz, C, gamma, mean, invstd, red, dg, db, M, C, dy=dy, dyp=C, scale=scale, shift=shift, relu=1)
print(f"reduce+apply pair: {timeit(pair):.1f} us")
