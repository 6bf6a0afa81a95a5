"""Diagnostic (libaau_STAMP.so): per-workgroup s_memtime stamps of the halo conv kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops
B = 8
for name, H, Ci, Co in (("d1.1", 512, 48, 48), ("u1.c0", 512, 96, 48), ("d2.1", 256, 96, 96), ("u4.c0", 64, 768, 384)):
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 9, cp, device="cuda") / (Ci * 9) ** 0.5).to(torch.bfloat16)
    out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(32, 2, Co, device="cuda")
    BQ = 48 if Co <= 48 else 96
    grid = ((Co + BQ - 1) // BQ) * (H // 16) ** 2 * B
    dbg = torch.zeros(grid, 4, dtype=torch.int64, device="cuda")
    d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, 3, 3, 1, 1, 1, cp)
    for _ in range(3):
        ops.conv_igemm(d, x, w, out, shift=dbg, stats=stats)
    torch.cuda.synchronize()
    t = dbg.cpu().double()
    setup = t[:, 1] - t[:, 0]; loop = t[:, 2] - t[:, 1]; epi = t[:, 3] - t[:, 2]; life = t[:, 3] - t[:, 0]
    span = (t[:, 3].max() - t[:, 0].min())
    print(f"{name}: blocks {grid}  first-barrier wait {setup.median():.0f}  loop {loop.median():.0f}  epilogue {epi.median():.0f}  "
          f"lifetime {life.median():.0f} (p90 {life.quantile(0.9):.0f})  kernel span {span:.0f} ticks; steps {cp // 32 * 9}")
