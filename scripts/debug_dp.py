import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import att_aspp_unet_amd as A
import torch.distributed as dist
from argparse import Namespace
from att_aspp_unet_amd import synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
x, y = synth.make_frames(2, 64, seed=5); x, y = x.cuda(), y.cuda()
def run(use_dp, nsteps=3):
    torch.manual_seed(1)
    m = A.AttentionASPPUNet(base_c=8).cuda().train(); m.bridge.project[3].p = 0.0
    dp = A.DataParallel(m) if use_dp else None
    step = A.TrainStep(m, A.FusedAdamW(m, lr=1e-3), args, dp)
    losses = [float(step(x, y).item()) for _ in range(nsteps)]
    g = m.engine.store.gflat.clone()
    return losses, torch.cat([p.detach().flatten() for p in m.parameters()]).cpu(), g.cpu()
for a, b, n in ((False, False, 1), (False, True, 1), (True, True, 1), (False, False, 3), (False, True, 3)):
    r1, r2 = run(a, n), run(b, n)
    d = (r1[1] - r2[1]).abs(); dg = (r1[2] - r2[2]).abs()
    print(f"dp {a} vs {b}, {n} steps: losses {r1[0]} {r2[0]} | weight maxdiff {float(d.max()):.2e} frac>1e-4 {float((d>1e-4).float().mean()):.3f} | grad maxdiff {float(dg.max()):.2e} rel {float(dg.max()/r1[2].abs().max()):.2e}")
dist.destroy_process_group()
