"""Timing probe: the wide-tile implicit GEMM (igemm<64,192>) at the bridge shapes; AAU_IGEMM_ABL=4 / 8 / 12 drop the
activation / weight / both operand streams (results wrong, timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

N, H, W = 8, 32, 32
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.1).to(torch.bfloat16)


def run(name, Cin, Cout, k, dil, reps=30):
    x = rnd(N, H, W, Cin)
    cp = ops.cpad_of(Cin)
    w = rnd(Cout, k * k, cp)
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=dev)
    st = ops.stats_buffer(Cout)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil, cp)
    for _ in range(3):
        ops.conv_igemm(d, x, w, out, stats=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_igemm(d, x, w, out, stats=st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gf = 2.0 * N * H * W * Cin * Cout * k * k / 1e9
    print(f"{name:28s} {us:7.1f} us   {gf / us * 1e-3:6.3f} PFLOP/s (nominal taps)", flush=True)


print("AAU_IGEMM_ABL =", os.environ.get("AAU_IGEMM_ABL", "0"))
run("1x1 384->768", 384, 768, 1, 1)
run("3x3 dil 1 384->768", 384, 768, 3, 1)
run("3x3 dil 6 384->768", 384, 768, 3, 6)
run("3x3 dil 18 384->768", 384, 768, 3, 18)
run("1x1 3840->768", 3840, 768, 1, 1)
