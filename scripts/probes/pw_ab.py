"""Same-process A/B of the two resident-weight 1x1 kernels (conv1x1_rs_kernel against conv1x1_resw_kernel, AAU_PW_OLD=1)
on the metric model's ConvT / gate shapes: average of 50 launches each, events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

B = 8
dev = "cuda"


def run(name, H, Ci, Co, shuffle, accumulate=0, relu=0):
    M = B * H * H
    x = torch.randn(M, Ci, device=dev).to(torch.bfloat16)
    cp = ops.cpad_of(Ci)
    w = (torch.randn(Co, 1, cp, device=dev) / Ci ** 0.5).to(torch.bfloat16)
    if shuffle:
        out = torch.zeros(B * 2 * H * 2 * H, Co // 4, device=dev, dtype=torch.bfloat16)
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co // 4, Cpad=cp, shuffle2x2=1)
        byts = M * Ci * 2 + M * Co * 2
    else:
        out = torch.zeros(M, Co, device=dev, dtype=torch.bfloat16)
        d = ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, Cpad=cp, accumulate=accumulate)
        byts = M * Ci * 2 + M * Co * 2 * (2 if accumulate else 1)
    res = []
    for old in (0, 1):
        if old:
            os.environ["AAU_PW_OLD"] = "1"
        else:
            os.environ.pop("AAU_PW_OLD", None)
        for _ in range(5):
            ops.conv_igemm(d, x, w, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.conv_igemm(d, x, w, out)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 50)
    print(f"{name:46s} new {res[0]:6.1f} us  old {res[1]:6.1f} us   {byts / 1e6:6.0f} MB -> {byts / res[0] / 1e6:5.2f} TB/s")


run("u1.up forward  (96 -> 4x48, 256^2)", 256, 96, 192, True)
run("u2.up forward  (192 -> 4x96, 128^2)", 128, 192, 384, True)
run("u3.up forward  (384 -> 4x192, 64^2)", 64, 384, 768, True)
run("u1 gate Wx (48 -> 24, 512^2)", 512, 48, 24, False)
run("u2 gate Wx (96 -> 48, 256^2)", 256, 96, 48, False)
run("u3 gate Wx (192 -> 96, 128^2)", 128, 192, 96, False)
run("u1 gate dgrad (24 -> 48, 512^2, accumulate)", 512, 24, 48, False, accumulate=1)
run("u2 gate dgrad (48 -> 96, 256^2, accumulate)", 256, 48, 96, False, accumulate=1)
run("u3 gate dgrad (96 -> 192, 128^2, accumulate)", 128, 96, 192, False, accumulate=1)
run("u4 gate dgrad (192 -> 384, 64^2, accumulate)", 64, 192, 384, False, accumulate=1)
run("u2 gate dgrad, plain store", 256, 48, 96, False, accumulate=0)
run("u1.up bytes as a plain 1x1 (96 -> 192, 256^2)", 256, 96, 192, False)
run("u2.up bytes as a plain 1x1 (192 -> 384, 128^2)", 128, 192, 384, False)
