import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import att_aspp_unet_amd as A
from oracle import ref_cpu as O
from argparse import Namespace
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = dict(np.load(os.path.join(G, "g1_step_c8_128.npz")))
sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init/")}
x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
# emulated oracle
net = O.AttentionASPPUNet(base_c=8); net.load_state_dict(sd); net.train(); net.bridge.project[3].p = 0
O.emulate_bf16_storage(net)
crit = O.build_criterion(O.default_args(), O.ComboLoss(), O.EdgeLoss())
lo = net(x); loss = crit(lo, y); loss.backward()
ge = {k: p.grad.clone() for k, p in net.named_parameters()}
m = A.AttentionASPPUNet(base_c=8); m.load_state_dict(sd, strict=True); m = m.cuda(); m.train(); m.bridge.project[3].p = 0.0
critg = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
lt = m(x.cuda()); lg = critg(lt, y.cuda()); lg.backward()
print("logits: engine vs emu", rel(lt, lo), "engine vs fp32", rel(lt, g["train_logits"]), "emu vs fp32", rel(lo, g["train_logits"]))
print("loss", lg.item(), loss.item(), float(g["loss_main"]))
rows = []
for k, p in m.named_parameters():
    rows.append((rel(p.grad, ge[k]), rel(p.grad, g["grad/"+k]), rel(ge[k], g["grad/"+k]), k))
rows.sort(reverse=True)
for r in rows[:15]: print("eng-emu %.4f eng-fp32 %.4f emu-fp32 %.4f %s" % r)
print("median eng-emu", sorted(r[0] for r in rows)[len(rows)//2])
fe = torch.cat([ge[k].flatten() for k, _ in m.named_parameters()]); fm = torch.cat([p.grad.detach().cpu().flatten() for _, p in m.named_parameters()])
ff = torch.cat([torch.from_numpy(g["grad/"+k]).flatten() for k, _ in m.named_parameters()])
cos = lambda a, b: float(torch.dot(a, b)/a.norm()/b.norm())
print("cos eng-emu", cos(fm, fe), "eng-fp32", cos(fm, ff), "emu-fp32", cos(fe, ff))
