"""Fixed cost of the wide igemm tile: 8x32x32, Cout 768, K = 64 ... 3840, with / without BN statistics."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops


def timeit(fn, n=20):
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


B, H, Co = 8, 32, 768
for Ci in (64, 384, 768, 1536, 3840):
    cp = ops.cpad_of(Ci)
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 1, cp, device="cuda") / Ci ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    stats = ops.stats_buffer(Co)
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 1, 1, 1, 0, 1, cp)
    t1 = timeit(lambda: ops.conv_igemm(d, x, w, out, stats=stats))
    t0 = timeit(lambda: ops.conv_igemm(d, x, w, out))
    print(f"K={Ci:5d} steps={cp // 64:3d}  with stats {t1:6.1f} us   without {t0:6.1f} us")
