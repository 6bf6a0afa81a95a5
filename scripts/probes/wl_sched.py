"""Timing probe: how the grouped weight-gradient launch (wgradL) scales with the number and the mix of its tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from att_aspp_unet_amd import ops

N, H, W, Ci, Co = 8, 32, 32, 384, 768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
x = rnd(N, H, W, Ci)
dcat = rnd(N, H, W, 5 * Co)
cat5 = rnd(N, H, W, 5 * Co)
dzp = rnd(N, H, W, Co)
up_in = rnd(N, H, W, Co)
up_dy = rnd(N, 2 * H, 2 * W, Ci)


def prob(kind, i=0):
    if kind == "1x1":
        return ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, 5 * Co, 1, 1, 1, 0, 1), x, dcat[..., :Co], torch.zeros(Co, 1, Ci, device=dev)
    if kind.startswith("dil"):
        d = int(kind[3:])
        return ops.conv_desc(N, H, W, Ci, Ci, H, W, Co, 5 * Co, 3, 3, 1, d, d), x, dcat[..., i * Co:(i + 1) * Co], torch.zeros(Co, 9, Ci, device=dev)
    if kind == "proj":
        return ops.conv_desc(N, H, W, 5 * Co, 5 * Co, H, W, Co, Co, 1, 1, 1, 0, 1), cat5, dzp, torch.zeros(Co, 1, 5 * Co, device=dev)
    if kind == "up":
        return ops.conv_desc(N, 2 * H, 2 * W, Ci, Ci, H, W, Co, Co, 2, 2, 2, 0, 1), up_dy, up_in, torch.zeros(Co, 4, Ci, device=dev)
    raise ValueError(kind)


def run(names, reps=20):
    ps = [prob(k, i) for i, k in enumerate(names)]
    descs, srcs, dzs, dws = zip(*ps)
    tiles = sum(((d.Cout + 191) // 192) * ((d.Cin + 191) // 192) * d.KH * d.KW for d in descs)
    q = torch.zeros(ops.wgrad_group_queue_words(), dtype=torch.int32, device=dev)
    out = []
    for form in FORMS:
        os.environ["AAU_WL_FORM"] = str(form)
        ts = []
        for rep in range(3):
            for _ in range(3):
                q.zero_()
                ops.conv_wgrad_group(list(descs), list(srcs), list(dzs), list(dws), q)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tot = 0.0
            for _ in range(reps):
                q.zero_()
                e0.record()
                ops.conv_wgrad_group(list(descs), list(srcs), list(dzs), list(dws), q)
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            ts.append(tot / reps * 1e3)
        out.append(f"form {form}: {min(ts):7.1f}")
    print(f"{'+'.join(names):46s} tiles {tiles:4d}  " + "   ".join(out) + "  us", flush=True)


FORMS = [int(f) for f in os.environ.get("WL_FORMS", "0,1,2").split(",")]
run(["proj"])
run(["proj"] * 4)
run(["proj"] * 6)
run(["dil18"])
run(["dil6", "dil12", "dil18"])
run(["1x1", "dil6", "dil12", "dil18", "proj"])
run(["up", "1x1", "dil6", "dil12", "dil18", "proj"])
