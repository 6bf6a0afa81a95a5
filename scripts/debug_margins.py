import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import att_aspp_unet_amd as A
from argparse import Namespace
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g4 = dict(np.load(os.path.join(G, "g4_trained_c8_128.npz"))); g5 = dict(np.load(os.path.join(G, "g5_trained_step.npz")))
g1 = dict(np.load(os.path.join(G, "g1_step_c8_128.npz")))
def rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
crit = A.build_criterion(Namespace(stage="main", edge_w=0.05, neg_bce_w=0.05), A.ComboLoss(), A.EdgeLoss())
for it in range(4):
    m = A.AttentionASPPUNet(base_c=8); m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g4.items() if k.startswith("sd/")}); m = m.cuda().train(); m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g4["x"]).cuda(), torch.from_numpy(g4["y"]).cuda()
    lt = m(x); loss = crit(lt, y); loss.backward()
    named = list(m.named_parameters())
    ge = torch.cat([p.grad.detach().float().cpu().flatten() for _, p in named]); gr = torch.cat([torch.from_numpy(g5["grad/"+k]).flatten() for k, _ in named])
    errs = sorted(rel(p.grad, g5["grad/"+k]) for k, p in named)
    print(f"trained: logits {rel(lt, g5['train_logits']):.4f} loss rel {abs(loss.item()-float(g5['loss_main']))/float(g5['loss_main']):.2e} cos {float(torch.dot(ge,gr)/ge.norm()/gr.norm()):.6f} norm rel {abs(float(ge.norm())-float(g5['grad_norm']))/float(g5['grad_norm']):.4f} med {errs[len(errs)//2]:.4f} p90 {errs[int(len(errs)*0.9)]:.4f} worst {errs[-1]:.4f}")
    m = A.AttentionASPPUNet(base_c=8); m.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in g1.items() if k.startswith("init/")}); m = m.cuda().train(); m.bridge.project[3].p = 0.0
    x, y = torch.from_numpy(g1["x"]).cuda(), torch.from_numpy(g1["y"]).cuda()
    lt = m(x); loss = crit(lt, y); loss.backward()
    named = list(m.named_parameters())
    ge = torch.cat([p.grad.detach().float().cpu().flatten() for _, p in named]); gr = torch.cat([torch.from_numpy(g1["grad/"+k]).flatten() for k, _ in named])
    rs = max(rel(m.state_dict()[k[10:]], v) for k, v in g1.items() if k.startswith("after_fwd/") and "num_batches" not in k)
    print(f"  init: logits {rel(lt, g1['train_logits']):.4f} loss rel {abs(loss.item()-float(g1['loss_main']))/float(g1['loss_main']):.2e} cos {float(torch.dot(ge,gr)/ge.norm()/gr.norm()):.5f} norm rel {abs(float(ge.norm())-float(g1['grad_norm']))/float(g1['grad_norm']):.4f} running-stat worst {rs:.4f}")
