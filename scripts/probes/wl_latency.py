"""Timing probe: one 192 x 192 tile of the grouped weight gradient (wgradL) on operands that fit in one XCD's L2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)


def run(N, H, W, Ci, Co, reps=50):
    x = (torch.randn(N, H, W, Ci, device=dev, generator=g)).to(torch.bfloat16)
    dz = (torch.randn(N, H, W, Co, device=dev, generator=g)).to(torch.bfloat16)
    dw = torch.zeros(Co, 1, Ci, device=dev)
    d = ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, Co, 1, 1, 1, 0, 1)
    for _ in range(5):
        ops.conv_wgrad_group([d], [x], [dz], [dw])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_wgrad_group([d], [x], [dz], [dw])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    M = N * H * W
    tiles = ((Ci + 191) // 192) * ((Co + 191) // 192)
    print(f"M {M:6d}  {Ci:4d} -> {Co:4d}  tiles {tiles:3d}  operands {M * (Ci + Co) * 2 / 1e6:6.1f} MB   {us:7.1f} us   {us / (M / 32):.3f} us per K-step", flush=True)


run(2, 32, 32, 192, 192)     # 1 tile, 1.6 MB
run(4, 32, 32, 192, 192)     # 1 tile, 3.1 MB
run(8, 32, 32, 192, 192)     # 1 tile, 6.3 MB
run(8, 32, 32, 384, 384)     # 4 tiles
run(8, 32, 32, 768, 768)     # 16 tiles, 25 MB
run(8, 32, 32, 1536, 1536)   # 64 tiles
run(8, 32, 32, 3072, 1536)   # 128 tiles
run(8, 32, 32, 3072, 3072)   # 256 tiles, 100 MB
