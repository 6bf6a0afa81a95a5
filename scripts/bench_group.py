"""Bridge input gradient (8x32x32, 768 -> 384, 1x1 + d6 + d12 + d18): four accumulate launches vs one grouped launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from att_aspp_unet_amd import ops


def timeit(fn, n=20):
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


B, H, Ci, Co = 8, 32, 768, 384
segs = [(1, 1), (3, 6), (3, 12), (3, 18)]
descs, srcs, wpks = [], [], []
for i, (k, dil) in enumerate(segs):
    srcs.append(torch.randn(B, H, H, Ci, device="cuda").to(torch.bfloat16))
    wpks.append((torch.randn(Co, k * k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(torch.bfloat16))
    descs.append(ops.conv_desc(B, H, H, Ci, Ci, H, H, Co, Co, k, k, 1, dil * (k // 2), dil, Ci, accumulate=1 if i else 0))
out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16)
gf = sum(2.0 * B * H * H * Ci * Co * k * k for k, _ in segs) / 1e9


def separate():
    for d, s, w in zip(descs, srcs, wpks):
        ops.conv_igemm(d, s, w, out)


t = timeit(separate)
print(f"separate launches      {t:7.1f} us  {gf / t * 1e3:6.0f} TF (nominal)")
for ns in ("1", "2", "3", "4"):
    os.environ["AAU_GROUP_NSPLIT"] = ns
    ws = torch.empty(max(ops.conv_igemm_group_ws_bytes(descs) // 4, 4), device="cuda")
    t = timeit(lambda: ops.conv_igemm_group(descs, srcs, wpks, out, ws))
    print(f"grouped, {ns} K-range(s)   {t:7.1f} us  {gf / t * 1e3:6.0f} TF (nominal)")
