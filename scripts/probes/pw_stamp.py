"""Diagnostic: where a tile step of conv1x1_resw spends its time.  Needs the stamp build
    python att-aspp-unet_amd/build.py -DAAU_PW_STAMP --tag=pwstamp
and runs with AAU_LIB=att-aspp-unet_amd/lib/libaau_pwstamp.so (the `shift` argument carries the debug buffer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

B = 8
dev = "cuda"


def run(name, H, Ci, Co, shuffle, accumulate=0):
    M = B * H * H
    x = torch.randn(M, Ci, device=dev).to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 1, cp, device=dev) / Ci ** 0.5).to(torch.bfloat16)
    if shuffle:
        out = torch.zeros(B * 2 * H * 2 * H, Co // 4, device=dev, dtype=torch.bfloat16)
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co // 4, Cpad=cp, shuffle2x2=1)
    else:
        out = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, Cpad=cp, accumulate=accumulate)
    dbg = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=dev)
    for _ in range(5):
        ops.conv_igemm(d, x, w, out, shift=dbg)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.conv_igemm(d, x, w, out, shift=dbg)
    e1.record()
    torch.cuda.synchronize()
    t = dbg.view(-1, 8, 8).double().cpu()
    t = t[t[:, 0, 5] > 0]
    steps = t[..., 5].mean()
    tot = t[..., 6].mean()
    names = ["wait for the tile (vmcnt)", "barrier", "issue of the next fill", "LDS reads + MFMA", "epilogue (per patch)"]
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:6.1f} us (stamp build), {t.shape[0]} workgroups, {steps:5.1f} tile steps per wave, "
          f"{tot:9.0f} s_memtime ticks per wave in the loop")
    for i, nm in enumerate(names):
        print(f"     {nm:28s} {float(t[..., i].mean()):10.0f} ticks = {float(t[..., i].mean()) / tot * 100:5.1f} %   per step {float(t[..., i].mean()) / steps:7.1f}")
    rest = tot - float(sum(t[..., i].mean() for i in range(5)))
    print(f"     {'(unstamped)':28s} {rest:10.0f} ticks = {rest / tot * 100:5.1f} %")
    w0 = t[0]
    print("     workgroup 0, per wave: wait", [int(v) for v in w0[:, 0]], " barrier", [int(v) for v in w0[:, 1]])


run("u1.up forward  (96 -> 4x48, 256^2 -> 512^2)", 256, 96, 192, True)
run("u2.up forward  (192 -> 4x96, 128^2 -> 256^2)", 128, 192, 384, True)
run("u2 gate Wx     (96 -> 48, 256^2)", 256, 96, 48, False)
run("u2 gate dgrad  (48 -> 96, 256^2, accumulate)", 256, 48, 96, False, accumulate=1)
