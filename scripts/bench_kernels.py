"""Per-layer micro-benchmark of the MFMA conv kernels at the headline shapes (bs 8, 512x512, base_c 48)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops

c, B = 48, 8
LAYERS = [  # name, H(=W), Cin, Cout, k, dil
    ("d1.1", 512, 48, 48, 3, 1), ("d2.0", 256, 48, 96, 3, 1), ("d2.1", 256, 96, 96, 3, 1),
    ("d3.0", 128, 96, 192, 3, 1), ("d3.1", 128, 192, 192, 3, 1), ("d4.0", 64, 192, 384, 3, 1),
    ("d4.1", 64, 384, 384, 3, 1), ("br.1x1", 32, 384, 768, 1, 1), ("br.d6", 32, 384, 768, 3, 6),
    ("br.d18", 32, 384, 768, 3, 18), ("br.proj", 32, 3840, 768, 1, 1), ("u4.c0", 64, 768, 384, 3, 1),
    ("u3.c0", 128, 384, 192, 3, 1), ("u2.c0", 256, 192, 96, 3, 1), ("u1.c0", 512, 96, 48, 3, 1),
    ("u2.Wg", 256, 96, 48, 1, 1),
]
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--modes", default="fwd,dgrad,wgrad")
a = ap.parse_args()


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
    return e0.elapsed_time(e1) / n * 1e3  # us


tot = {m: 0.0 for m in a.modes.split(",")}
for name, H, Ci, Co, k, dil in LAYERS:
    if a.only and a.only not in name:
        continue
    M = B * H * H
    gf = 2.0 * M * Ci * Co * k * k / 1e9
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    dz = torch.randn(B, H, H, Co, device="cuda").to(torch.bfloat16)
    pad = dil * (k // 2)
    line = f"{name:8s} {gf:7.1f} GF"
    if "fwd" in tot:
        cp = ops.cpad_of(Ci)
        w = (torch.randn(Co, k * k, cp, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16)
        out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
        stats = torch.zeros(32, 2, Co, device="cuda")
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, pad, dil, cp)
        t = timeit(lambda: ops.conv_igemm(d, x, w, out, stats=stats))
        tot["fwd"] += t
        line += f" | fwd {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "dgrad" in tot:
        cp = ops.cpad_of(Co)
        w = (torch.randn(Ci, k * k, cp, device="cuda") / (Co * k * k) ** 0.5).to(torch.bfloat16)
        out = torch.empty(B, H, H, Ci, device="cuda", dtype=torch.bfloat16)
        d = ops.conv_desc(B, H, H, Co, Co, H, H, Ci, Ci, k, k, 1, pad, dil, cp)
        t = timeit(lambda: ops.conv_igemm(d, dz, w, out))
        tot["dgrad"] += t
        line += f" | dgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "wgrad" in tot:
        dw = torch.zeros(Co, k * k, Ci, device="cuda")
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, pad, dil)
        t = timeit(lambda: ops.conv_wgrad(d, x, dz, dw))
        tot["wgrad"] += t
        line += f" | wgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    print(line, flush=True)
print("totals (us):", {k: round(v, 1) for k, v in tot.items()})
