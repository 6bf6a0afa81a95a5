import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from att_aspp_unet_amd import ops
N, H, W, C_ = 2, 64, 80, 48
M = N*H*W
g = torch.Generator().manual_seed(21)
w = (torch.randn(C_, 9, 64, generator=g) / (C_*9)**0.5).to(torch.bfloat16).cuda()
dz = torch.randn(N, H, W, C_, generator=g).to(torch.bfloat16).cuda()
z = torch.randn(M, C_, generator=g).to(torch.bfloat16).cuda()
one = torch.ones(C_).cuda(); zero = torch.zeros(C_).cuda()
dd = ops.conv_desc(N, H, W, C_, C_, H, W, C_, C_, 3, 3, 1, 1, 1, 64)
ref = torch.empty(M, C_, dtype=torch.bfloat16, device="cuda")
ops.conv_igemm(dd, dz, w, ref)
out = torch.full((M, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
sums = ops.stats_buffer(C_)
ops.conv_igemm_bnred(dd, dz, w, out, z, C_, one, zero, zero, one, sums)
torch.cuda.synchronize()
bad = (out != ref) | out.isnan()
print("mismatch", int(bad.sum()), "nan", int(out.isnan().sum()))
idx = bad.nonzero()
print(idx[:10].tolist())
pix = idx[:, 0]
n = pix // (H*W); y = (pix // W) % H; x = pix % W
print("rows", sorted(set(y.tolist()))[:40]); print("cols", sorted(set(x.tolist()))[:40]); print("ch", sorted(set(idx[:,1].tolist())))
print(out[idx[0,0], :8], ref[idx[0,0], :8])
i0, c0 = idx[0].tolist()
print("elem", out[i0, c0].item(), ref[i0, c0].item(), out[i0, c0].view(torch.int16).item(), ref[i0, c0].view(torch.int16).item())
d = (out.float() - ref.float()).abs()
print("max abs diff", d.max().item(), "max rel", (d / ref.float().abs().clamp_min(1e-3)).max().item())
out2 = torch.full((M, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
ops.conv_igemm_bnred(dd, dz, w, out2, z, C_, one, zero, zero, one, ops.stats_buffer(C_))
ref2 = torch.empty_like(ref); ops.conv_igemm(dd, dz, w, ref2)
torch.cuda.synchronize()
print("bnred vs bnred", int((out != out2).sum()), "ref vs ref", int((ref != ref2).sum()))
