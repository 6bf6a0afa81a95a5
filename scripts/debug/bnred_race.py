"""Debug: aau_conv_igemm_bnred against aau_conv_igemm, repeated, with mismatch positions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import att_aspp_unet_amd as A
from att_aspp_unet_amd import ops
from test_kernels_gpu import pack_dgrad, dev, R

for (N, H, W) in ((2, 64, 80), (8, 512, 512), (1, 16, 16), (1, 32, 16), (1, 64, 16)):
    C_ = 48
    M = N * H * W
    g = torch.Generator().manual_seed(21)
    w = R.bf16_round(torch.randn(C_, C_, 3, 3, generator=g) / (C_ * 9) ** 0.5)
    dz = R.bf16_round(torch.randn(N, H, W, C_, generator=g))
    z = torch.randn(M, C_, generator=g).to(torch.bfloat16).cuda()
    gamma, beta = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.3
    zf = z.float().cpu()
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    cpd = ops.cpad_of(C_)
    dd = ops.conv_desc(N, H, W, C_, C_, H, W, C_, C_, 3, 3, 1, 1, 1, cpd)
    dzd, wd = dev(dz.to(torch.bfloat16)), dev(pack_dgrad(w, cpd))
    ref = torch.empty(M, C_, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(dd, dzd, wd, ref)
    ref2 = torch.empty(M, C_, dtype=torch.bfloat16, device="cuda")
    ops.conv_igemm(dd, dzd, wd, ref2)
    print(f"shape {N}x{H}x{W}: plain vs plain equal: {bool(torch.equal(ref, ref2))}")
    for it in range(3):
        out = torch.full((M, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
        sums = ops.stats_buffer(C_)
        ops.conv_igemm_bnred(dd, dzd, wd, out, z, C_, dev(scale), dev(shift), dev(mean), dev(invstd), sums)
        torch.cuda.synchronize()
        bad = (out.view(torch.int16) != ref.view(torch.int16))
        nb = int(bad.sum())
        print(f"  it {it}: mismatches {nb} of {M * C_}, nan in out {int(torch.isnan(out.float()).sum())}")
        if nb:
            idx = bad.nonzero()[:12].cpu()
            for p, c in idx.tolist():
                n_, r = divmod(p, H * W); y_, x_ = divmod(r, W)
                print(f"     n {n_} y {y_} x {x_} c {c}: out {float(out[p, c]):.5f} ref {float(ref[p, c]):.5f}")
            pix = bad.any(1).nonzero().flatten().cpu()
            ys = (pix % (H * W)) // W; xs = pix % W
            print("     rows mod 16 hist:", torch.bincount(ys % 16, minlength=16).tolist())
            print("     cols mod 16 hist:", torch.bincount(xs % 16, minlength=16).tolist())
            print("     channel hist:", bad.sum(0).cpu().tolist())
