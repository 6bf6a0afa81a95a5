import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import att_aspp_unet_amd as A
from argparse import Namespace
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = dict(np.load(os.path.join(G, "g1_step_c8_128.npz")))
sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init/")}
m = A.AttentionASPPUNet(base_c=8)
m.load_state_dict(sd, strict=True)
m = m.cuda()
x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
m.eval()
with torch.no_grad():
    le = m(x)
print("eval logits rel err", rel(le, g["eval_logits"]), "max|ref|", np.abs(g["eval_logits"]).max())
m.train(); m.bridge.project[3].p = 0.0
crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
opt = A.FusedAdamW(m, lr=3e-4)
lt = m(x)
print("train logits rel err", rel(lt, g["train_logits"]))
loss = crit(lt, y)
print("loss", loss.item(), "ref", float(g["loss_main"]))
loss.backward()
worst = []
for k, p in m.named_parameters():
    r = rel(p.grad, g["grad/" + k])
    worst.append((r, k, float(np.abs(g["grad/" + k]).max())))
worst.sort(reverse=True)
for w in worst[:12]: print("grad", w)
print("median grad rel err", sorted(w[0] for w in worst)[len(worst)//2])
for k, v in m.state_dict().items():
    if "running" in k or "num_batches" in k:
        pass
rs = [(rel(m.state_dict()[k[10:]], v), k) for k, v in g.items() if k.startswith("after_fwd/") and "num_batches" not in k]
print("running stats worst", sorted(rs, reverse=True)[:3])
opt.step()
torch.cuda.synchronize()
print("grad norm", float(opt.grad_norm()), "ref", float(g["grad_norm"]))
ps = [(float((p.detach().cpu() - torch.from_numpy(g["after_step/" + k])).abs().max()), k) for k, p in m.named_parameters()]
print("post-step worst abs", sorted(ps, reverse=True)[:3])
# trained weights / dice
g4 = dict(np.load(os.path.join(G, "g4_trained_c8_128.npz")))
m2 = A.AttentionASPPUNet(base_c=8); m2.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g4.items() if k.startswith("sd/")}); m2 = m2.cuda().eval()
xv, yv = torch.from_numpy(g4["x"]).cuda(), torch.from_numpy(g4["y"]).cuda()
with torch.no_grad(): lv = m2(xv)
print("trained eval logits rel", rel(lv, g4["eval_logits"]), "max", np.abs(g4["eval_logits"]).max())
d, i = A.evaluate(m2, [(xv[:4], yv[:4]), (xv[4:], yv[4:])], torch.device("cuda"))
print("evaluate", d, i, "ref", float(g4["evaluate_dice"]), float(g4["evaluate_iou"]))
print("tta", rel(A.predict_prob_tta(m2, xv[:1]), g4["tta_prob0"]))
