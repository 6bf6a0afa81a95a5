"""Main-loop rate of the grouped weight-gradient kernel on synthetic 1x1 problems with an exact tile count."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops


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


B, H = 8, 32
M = B * H * H
for Ci, Co, k, dil in ((3072, 3072, 1, 1), (3072, 1536, 1, 1), (6144, 3072, 1, 1), (384, 768, 3, 1), (384, 768, 3, 6), (384, 768, 3, 18), (3840, 768, 1, 1)):
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    dz = torch.randn(B, H, H, Co, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(Co, k * k, Ci, device="cuda")
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, dil * (k // 2), dil)
    tiles = ((Ci + 191) // 192) * ((Co + 191) // 192) * k * k
    gf = 2.0 * M * Ci * Co * k * k / 1e9
    t = timeit(lambda: ops.conv_wgrad_group([d], [x], [dz], [dw]))
    print(f"Cin {Ci:5d} Cout {Co:5d} k{k} d{dil:2d}: {tiles:4d} tiles {gf:7.1f} GF {t:8.1f} us {gf / t * 1e3:7.0f} TF/s", flush=True)
