"""Per-tensor gradient error table of the trained fixture step (tests/test_model_gpu.py::test_trained_step_gradients_match_reference)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import att_aspp_unet_amd as A
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g4, g5 = dict(np.load(os.path.join(G, "g4_trained_c8_128.npz"))), dict(np.load(os.path.join(G, "g5_trained_step.npz")))
m = A.AttentionASPPUNet(base_c=8)
m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g4.items() if k.startswith("sd/")}, strict=True)
m = m.cuda().train(); m.bridge.project[3].p = 0.0
x, y = torch.from_numpy(g4["x"]).cuda(), torch.from_numpy(g4["y"]).cuda()
crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
runs = []
for _ in range(2):
    for p in m.parameters(): p.grad = None
    loss = crit(m(x), y); loss.backward()
    runs.append({k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
tot = np.sqrt(sum(float((torch.from_numpy(g5["grad/" + k]).double() ** 2).sum()) for k in runs[0]))
rows = []
for k, ge in runs[0].items():
    gr = torch.from_numpy(g5["grad/" + k])
    mx = float((ge - gr).abs().max() / (gr.abs().max() + 1e-30))
    l2 = float((ge - gr).norm() / (gr.norm() + 1e-30))
    glob = float((ge - gr).norm() / tot)
    rr = float((ge - runs[1][k]).norm() / (gr.norm() + 1e-30))
    rows.append((mx, l2, glob, float(gr.norm() / tot), rr, k, tuple(gr.shape)))
rows.sort(reverse=True)
print(f"{'max-rel':>8s} {'L2-rel':>8s} {'err/|G|':>9s} {'|g|/|G|':>8s} {'run2run':>8s}  tensor")
for r in rows: print(f"{r[0]:8.4f} {r[1]:8.4f} {r[2]:9.2e} {r[3]:8.4f} {r[4]:8.1e}  {r[5]} {r[6]}")
