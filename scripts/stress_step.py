"""Run-to-run spread of one training step (same weights, same batch): how far apart are two executions, and in
which plan buffer does the spread start?  fp32 atomics in the BN statistics make the step non-bitwise-reproducible;
anything much larger than a few bf16 ulps in early buffers would point at a race."""
import sys
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, ".")
import att_aspp_unet_amd as A
from att_aspp_unet_amd import engine as E

mode = sys.argv[1] if len(sys.argv) > 1 else "trained"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bufs = []
orig_new = E.Plan.new


def new(self, *shape, dtype=E.BF16):
    t = orig_new(self, *shape, dtype=dtype)
    bufs.append((len(bufs), tuple(shape), t))
    return t


E.Plan.new = new
args = Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05)
if mode == "trained":
    g = np.load("tests/golden/g4_trained_c8_128.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    x, y = torch.from_numpy(g["x"][:4]).cuda(), torch.from_numpy(g["y"][:4]).cuda()
    c = 8
else:
    from att_aspp_unet_amd import synth
    c = int(mode[1:]) if mode.startswith("c") else 8
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    torch.manual_seed(1)
    sd = {k: v.clone() for k, v in A.AttentionASPPUNet(base_c=c).state_dict().items()}
    x, y = synth.make_frames(2, size, seed=5)
    x, y = x.cuda(), y.cuda()

m = A.AttentionASPPUNet(base_c=c)
m.load_state_dict(sd)
m = m.cuda().train()
m.bridge.project[3].p = 0.0
opt = A.FusedAdamW(m, lr=0.0)            # lr 0, weight decay irrelevant: weights stay put
for gpar in opt.param_groups:
    gpar["weight_decay"] = 0.0
step = A.TrainStep(m, opt, args, None)
m.engine.ensure(x.device)
st = m.engine.store
w0 = st.flat.clone()
res = []
for r in range(runs):
    st.flat.copy_(w0)
    loss = float(step(x, y).item())
    torch.cuda.synchronize()
    plan = m._plan_for(x)
    extra = [plan.stats_arena.buf, plan.vec_arena.buf, plan.red_arena.buf]
    res.append((loss, st.gflat.clone(), [t.clone().float() for _, _, t in bufs] + [t.clone() for t in extra]))
l0, g0, s0 = res[0]
print("weights unchanged:", torch.equal(st.flat, w0))
worst, wr = 2.0, 0
for r in range(1, runs):
    l, g_, _ = res[r]
    cos = float(torch.dot(g0, g_) / g0.norm() / g_.norm())
    print(f"run {r}: loss {l:.7f} (d {l - l0:+.2e})  cos(g0,g) {cos:.6f}  |g| {float(g_.norm()):.5f}")
    if cos < worst:
        worst, wr = cos, r
print("worst run", wr, "cos", worst)
names = [f"buf{i}{shp}" for i, shp, _ in bufs] + ["stats_arena", "vec_arena", "red_arena"]
for nm, a, b in zip(names, s0, res[wr][2]):
    fin = torch.isfinite(a) & torch.isfinite(b)
    d = (a - b).abs()[fin]
    nd = int((d > 0).sum())
    if nd:
        print(f"  {nm}: {nd}/{a.numel()} differ, max abs {float(d.max()):.3e}, ref absmax {float(a[fin].abs().max()):.3e}")
