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
    ("br.d12", 32, 384, 768, 3, 12), ("br.d18", 32, 384, 768, 3, 18), ("br.proj", 32, 3840, 768, 1, 1),
    ("br.u4T", 32, 768, 1536, 1, 1), ("br.u3T", 64, 384, 768, 1, 1), ("br.u4g", 64, 384, 192, 1, 1),   # ConvT as GEMMs, a gate conv
    ("u4.c0", 64, 768, 384, 3, 1),
    ("u3.c0", 128, 384, 192, 3, 1), ("u2.c0", 256, 192, 96, 3, 1), ("u1.c0", 512, 96, 48, 3, 1),
    ("u2.Wg", 256, 96, 48, 1, 1),
]
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--modes", default="fwd,dgrad,wgrad")
ap.add_argument("--atomic", action="store_true", help="fp32-atomic split-K in the weight gradients (no workspace)")
a = ap.parse_args()


_ws = None


def wgrad(d, src, dz, dw):
    """aau_conv_wgrad with a shared deterministic split-K workspace (as the engine runs it)."""
    global _ws
    if a.atomic:
        return ops.conv_wgrad(d, src, dz, dw)
    n = ops.conv_wgrad_ws_bytes(d) // 4
    if _ws is None or _ws.numel() < n:
        _ws = torch.empty(n, device="cuda")
    ops.conv_wgrad(d, src, dz, dw, _ws)


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
    if a.only and not any(o in name for o in a.only.split(",")):
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
        stats = ops.stats_buffer(Co)
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
        t = timeit(lambda: wgrad(d, x, dz, dw))
        tot["wgrad"] += t
        line += f" | wgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    print(line, flush=True)
# ---- ConvTranspose2d(2,2) of the decoder (engine.py: forward = GEMM + pixel-shuffle store into the cat buffer's
# upper half; backward = 2x2 stride-2 conv over the fine gradient) and the gate 1x1 convs with their strided /
# accumulating operands ----
UPS = [("u1.up", 256, 96, 48), ("u2.up", 128, 192, 96), ("u3.up", 64, 384, 192), ("u4.up", 32, 768, 384)]
for name, hi, gc, Co in UPS:
    if a.only and not any(o in name for o in a.only.split(",")):
        continue
    ho = 2 * hi
    Mi, Mo = B * hi * hi, B * ho * ho
    gf = 2.0 * Mi * gc * 4 * Co / 1e9
    g = torch.randn(Mi, gc, device="cuda").to(torch.bfloat16)
    # level 1 keeps the concatenation as two dense planes (engine.py: the strip kernels read a two-plane source), the deeper
    # levels interleave skip | up in one buffer of pitch 2 Co
    planar = name == "u1.up"
    cp2 = Co if planar else 2 * Co                      # pitch of the buffer the up half lives in
    catbuf = torch.randn(Mo, cp2, device="cuda").to(torch.bfloat16)
    cat = catbuf if planar else None
    up = catbuf if planar else catbuf[:, Co:]
    line = f"{name:8s} {gf:7.1f} GF"
    if "fwd" in tot:
        cp = ops.cpad_of(gc)
        w = (torch.randn(4 * Co, 1, cp, device="cuda") / gc ** 0.5).to(torch.bfloat16)
        bias = torch.zeros(Co, device="cuda")
        d = ops.conv_desc(B, hi, hi, gc, gc, hi, hi, 4 * Co, cp2, Cpad=cp, shuffle2x2=1)
        t = timeit(lambda: ops.conv_igemm(d, g, w, up, bias=bias))
        tot["fwd"] += t
        line += f" | fwd {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "dgrad" in tot:
        cp = ops.cpad_of(Co)
        w = (torch.randn(gc, 4, cp, device="cuda") / (4 * Co) ** 0.5).to(torch.bfloat16)
        out = torch.empty(Mi, gc, device="cuda", dtype=torch.bfloat16)
        d = ops.conv_desc(B, ho, ho, Co, cp2, hi, hi, gc, gc, 2, 2, 2, 0, 1, cp)
        t = timeit(lambda: ops.conv_igemm(d, up, w, out))
        tot["dgrad"] += t
        line += f" | dgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "wgrad" in tot:
        dw = torch.zeros(gc, 4, Co, device="cuda")
        d = ops.conv_desc(B, ho, ho, Co, cp2, hi, hi, gc, gc, 2, 2, 2, 0, 1)
        t = timeit(lambda: wgrad(d, up, g, dw))
        tot["wgrad"] += t
        line += f" | wgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    print(line, flush=True)
GATES = [("u2.gate", 256, 96), ("u3.gate", 128, 192), ("u4.gate", 64, 384)]
for name, ho, Co in GATES:
    if a.only and not any(o in name for o in a.only.split(",")):
        continue
    Mo, Fi = B * ho * ho, Co // 2
    gf = 2 * 2.0 * Mo * Co * Fi / 1e9          # Wg and Wx together
    cat = torch.randn(Mo, 2 * Co, device="cuda").to(torch.bfloat16)
    skip = torch.randn(Mo, Co, device="cuda").to(torch.bfloat16)
    zg, zx = torch.randn(Mo, Fi, device="cuda").to(torch.bfloat16), torch.randn(Mo, Fi, device="cuda").to(torch.bfloat16)
    line = f"{name:8s} {gf:7.1f} GF"
    if "fwd" in tot:
        cp = ops.cpad_of(Co)
        w = (torch.randn(Fi, 1, cp, device="cuda") / Co ** 0.5).to(torch.bfloat16)
        st = ops.stats_buffer(Fi)
        d1 = ops.conv_desc(B, ho, ho, Co, 2 * Co, ho, ho, Fi, Fi, Cpad=cp)
        d2 = ops.conv_desc(B, ho, ho, Co, Co, ho, ho, Fi, Fi, Cpad=cp)
        t = timeit(lambda: (ops.conv_igemm(d1, cat[:, Co:], w, zg, stats=st), ops.conv_igemm(d2, skip, w, zx, stats=st)))
        tot["fwd"] += t
        line += f" | fwd {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "dgrad" in tot:
        cp = ops.cpad_of(Fi)
        w = (torch.randn(Co, 1, cp, device="cuda") / Fi ** 0.5).to(torch.bfloat16)
        d1 = ops.conv_desc(B, ho, ho, Fi, Fi, ho, ho, Co, 2 * Co, Cpad=cp, accumulate=1)
        d2 = ops.conv_desc(B, ho, ho, Fi, Fi, ho, ho, Co, Co, Cpad=cp, accumulate=1)
        t = timeit(lambda: (ops.conv_igemm(d1, zg, w, cat[:, Co:]), ops.conv_igemm(d2, zx, w, skip)))
        tot["dgrad"] += t
        line += f" | dgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    if "wgrad" in tot:
        dw = torch.zeros(Fi, 1, Co, device="cuda")
        d1 = ops.conv_desc(B, ho, ho, Co, 2 * Co, ho, ho, Fi, Fi)
        d2 = ops.conv_desc(B, ho, ho, Co, Co, ho, ho, Fi, Fi)
        t = timeit(lambda: (wgrad(d1, cat[:, Co:], zg, dw), wgrad(d2, skip, zx, dw)))
        tot["wgrad"] += t
        line += f" | wgrad {t:7.1f} us {gf / t * 1e3:6.0f} TF"
    print(line, flush=True)
print("totals (us):", {k: round(v, 1) for k, v in tot.items()})
# ---- the ASPP bridge weight gradients as ONE grouped launch (csrc/wgradL.hip) vs the per-problem kernels ----
if not a.only or "bridge" in a.only or "br." in a.only:
    H, Ci, Co = 32, 384, 768
    M = B * H * H
    x = torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16)
    cat = torch.randn(B, H, H, 5 * Co, device="cuda").to(torch.bfloat16)
    dcat = torch.randn(B, H, H, 5 * Co, device="cuda").to(torch.bfloat16)
    dzp = torch.randn(B, H, H, Co, device="cuda").to(torch.bfloat16)
    descs, srcs, dzs, dws, gf = [], [], [], [], 0.0
    for i, (k, dil) in enumerate(((1, 1), (3, 6), (3, 12), (3, 18))):
        descs.append(ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, 5 * Co, k, k, 1, dil * (k // 2), dil))
        srcs.append(x); dzs.append(dcat[..., i * Co:]); dws.append(torch.zeros(Co, k * k, Ci, device="cuda"))
        gf += 2.0 * M * Ci * Co * k * k / 1e9
    descs.append(ops.conv_desc(B, H, H, 5 * Co, 5 * Co, H, H, Co, Co))
    srcs.append(cat); dzs.append(dzp); dws.append(torch.zeros(Co, 1, 5 * Co, device="cuda"))
    gf += 2.0 * M * 5 * Co * Co / 1e9
    if ops.conv_wgrad_group_ok(descs):
        t = timeit(lambda: ops.conv_wgrad_group(descs, srcs, dzs, dws))
        t0 = timeit(lambda: [wgrad(d_, s_, z_, w_) for d_, s_, z_, w_ in zip(descs, srcs, dzs, dws)])
        print(f"bridge wgrad (1x1 + d6 + d12 + d18 + proj) {gf:7.1f} GF | grouped {t:7.1f} us {gf / t * 1e3:6.0f} TF (dense-equivalent) | "
              f"per-problem kernels {t0:7.1f} us {gf / t0 * 1e3:6.0f} TF", flush=True)
