"""Inference-side numbers for BASELINE configs 2 and 5 (not the headline metric; reported in DESIGN.md).
  config 2: forward only, 1x512x512, batch 4, eval mode (BN folded into the conv epilogue), hipGraph replay
  config 5: 1x1024x1024 frame, 512-windows on a 256 grid (9 windows as one batch), ASPP rates (6,12,18,24), hipGraph
"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import att_aspp_unet_amd as A
from att_aspp_unet_amd import synth

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

torch.manual_seed(2025)
m = A.AttentionASPPUNet(base_c=48).cuda().eval()
x, _ = synth.make_frames(4, 512, seed=1)
x = x.cuda()
gf = A.GraphedForward(m, (4, 1, 512, 512))
t_eager = timeit(lambda: m(x)); t_graph = timeit(lambda: gf(x))
res = {"config2_fwd_bs4_512": {"eager_ms": t_eager * 1e3, "graph_ms": t_graph * 1e3, "images_per_s": 4 / t_graph,
                                "conv_TFLOPs": 4 * 226.76e9 / t_graph / 1e12}}
gf1 = A.GraphedForward(m, (1, 1, 512, 512))
t1 = timeit(lambda: gf1(x[:1]))
res["fwd_bs1_512_graph_ms"] = t1 * 1e3
m5 = A.AttentionASPPUNet(base_c=48, rates=(6, 12, 18, 24)).cuda().eval()
big = torch.rand(1, 1, 1024, 1024, device="cuda")
gf9 = A.GraphedForward(m5, (9, 1, 512, 512))
t5 = timeit(lambda: A.predict_sliding_window(m5, big, 512, 256, forward=gf9), n=10)
res["config5_1024_sliding_window_9x512"] = {"ms_per_frame": t5 * 1e3, "frames_per_s": 1 / t5}
print(json.dumps(res))
