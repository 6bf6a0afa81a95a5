"""Diagnostic: where a K-step of the wide implicit-GEMM tile spends its time (s_memtime stamps; needs the stamp build:
python att-aspp-unet_amd/build.py -DAAU_IGEMM_STAMP --tag=stamp, run with AAU_LIB=.../lib/libaau_stamp.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

N, H, W = 8, 32, 32
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.1).to(torch.bfloat16)


def run(name, Cin, Cout, k, dil):
    x = rnd(N, H, W, Cin)
    cp = ops.cpad_of(Cin)
    w = rnd(Cout, k * k, cp)
    out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=dev)
    d = ops.conv_desc(N, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, dil * (k // 2), dil, cp)
    dbg = torch.zeros(64 * 8 * 10, dtype=torch.int64, device=dev)
    for _ in range(20):
        ops.conv_igemm(d, x, w, out, shift=dbg)
    torch.cuda.synchronize()
    t = dbg.view(64, 8, 10).double().cpu()
    steps = t[..., 4]
    per = lambda i: float((t[..., i] / steps).mean())
    clk = float((t[..., 3] / t[..., 5].clamp(min=1)).mean()) * 100.0      # MHz: s_memtime ticks per 100-MHz s_memrealtime tick
    print(f"   prologue {float(t[..., 6].mean()):8.0f} cycles   loop {float(t[..., 3].mean()):8.0f}   epilogue {float(t[..., 8].mean()):8.0f}   [stamp2: top->fetch issued + fragments landed {float((t[..., 7] / steps).mean()):7.1f} per step]")
    print(f"   s_memtime runs at {clk:7.1f} MHz (loop spans {float(t[..., 5].mean()) / 100.0:6.2f} us)")
    print(f"{name:24s} steps/tile {float(steps.mean()):5.1f}   per step [s_memtime ticks]: top->barrier {per(2):7.1f}  barrier wait {per(0):7.1f}  "
          f"barrier->top {per(1):7.1f}   loop total {per(3):7.1f}", flush=True)
    # per-wave spread of the barrier wait in workgroup 0
    print("   wg0 waves  top->barrier:", [round(float(v), 1) for v in (t[0, :, 2] / steps[0])], " barrier wait:", [round(float(v), 1) for v in (t[0, :, 0] / steps[0])])


run("3x3 dil 6 384->768", 384, 768, 3, 6)
run("1x1 3840->768", 3840, 768, 1, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
