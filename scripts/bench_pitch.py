"""What do half-row (interleaved concat) stores / loads cost?  bn_act on 8 x 512 x 512 x 48 with dense and 2x pitches."""
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

for C, M in ((48, 8 * 512 * 512), (96, 8 * 256 * 256), (192, 8 * 128 * 128)):
    z = torch.randn(M, 2 * C, device="cuda").to(torch.bfloat16)
    y = torch.empty(M, 2 * C, device="cuda", dtype=torch.bfloat16)
    zd = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    yd = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    sc, sh = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mb = M * C * 2 / 1e6
    t_dd = timeit(lambda: ops.bn_act(zd, C, yd, C, sc, sh, M, C, relu=1))
    t_di = timeit(lambda: ops.bn_act(zd, C, y[:, C:], 2 * C, sc, sh, M, C, relu=1))
    t_id = timeit(lambda: ops.bn_act(z[:, C:], 2 * C, yd, C, sc, sh, M, C, relu=1))
    print(f"C {C:3d} ({mb:6.1f} MB per tensor): dense->dense {t_dd:6.1f} us ({2 * mb / t_dd / 1e3 * 1e3:5.0f} GB/s) | dense->half rows {t_di:6.1f} us | half rows->dense {t_id:6.1f} us", flush=True)
