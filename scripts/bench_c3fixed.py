"""Fixed cost per workgroup of the halo kernel: the same 256 x 256 x 8 grid (2048 tiles of 96 channels) at growing Cin."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, H, Co = 8, 256, 96
for Ci in (32, 64, 96, 192, 384):
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 9, cp, device="cuda") / (Ci * 9) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    st = ops.stats_buffer(Co)
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, cp)
    t1 = timeit(lambda: ops.conv_igemm(d, x, w, out, stats=st))
    t0 = timeit(lambda: ops.conv_igemm(d, x, w, out))
    gf = 2.0 * B * H * H * Ci * Co * 9 / 1e9
    print(f"Cin {Ci:4d}: {t1:7.1f} us with stats, {t0:7.1f} us without  ({gf / t0 * 1e3:6.0f} TF/s)", flush=True)
