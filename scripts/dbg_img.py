import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import imgproc_ref as R
from att_aspp_unet_amd import imgproc as I
rng=np.random.default_rng(0)
prob=rng.random((512,512),dtype=np.float32)
for hw in [(562,744),(64,80)]:
    w=R.resize_linear_f32(prob,hw); g=I.resize_bilinear(torch.from_numpy(prob).cuda(),hw).cpu().numpy()
    d=np.abs(w-g); print('resize',hw,d.max(),(d>0).mean(), np.argwhere(d>0)[:5].tolist())
    wg=R.gaussian_blur5(w); gg=I.gaussian_blur5(torch.from_numpy(w).cuda()).cpu().numpy()
    d=np.abs(wg-gg); print('gauss',d.max(),(d>0).mean())
img=rng.integers(10,200,(100,90)).astype(np.uint8)
n=R.normalize_minmax(img); gn=I.normalize_minmax(torch.from_numpy(img).cuda()).cpu().numpy(); print('norm',(n!=gn).mean())
c=R.clahe(n); gc=I.clahe(torch.from_numpy(n).cuda()).cpu().numpy(); print('clahe',(c!=gc).mean(), np.abs(c.astype(int)-gc).max())
m=R.median3(c); gm=I.median3(torch.from_numpy(c).cuda()).cpu().numpy(); print('median',(m!=gm).mean())
r=R.resize_linear_u8(m,(512,512)); gr=I.resize_bilinear(torch.from_numpy(m).cuda(),(512,512)).cpu().numpy(); print('resize u8',(r!=gr).mean(), np.abs(r.astype(int)-gr).max())
