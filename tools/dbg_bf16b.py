import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear.solvers import ista
from lasso_amd import _native as nat
from recipes import recipe_xw
def au(v, a=256): return (v + a - 1) // a * a
n, d, k = 64, 256, 1024
X, W = recipe_xw(n, d, k)
Xb, Wb = X.cuda().bfloat16(), W.cuda().bfloat16()
g = torch.Generator().manual_seed(1)
z0 = (torch.randn(n, k, generator=g) * 0.1).cuda().bfloat16()
z, info = ista(Xb, z0, Wb, 0.5, lr=1.0, maxiter=1, tol=0.0, backtrack=True, return_info=True)
torch.cuda.synchronize()
ws = nat.workspace(Xb.device, 1)
kp, ntiles16 = 1024, (n + 15) // 16
off = 0
def take(b):
    global off
    o = off; off += au(b); return o
o_wp = take(256 * kp * 4); o_wtp = take(kp * 256 * 4); o_part = take(5 * max(ntiles16, 1) * 4)
o_dpart = take(1024 * 4); o_delta = take(256); o_flags = take(256); o_fvals = take(256)
o_G = take(n * k * 4); o_C = take(n * k * 4); o_Y = take(n * k * 4); o_Zf = take(n * k * 4)
def view(o, shape, dt=torch.float32):
    cnt = 1
    for s in shape: cnt *= s
    return ws[o:o + cnt * dt.itemsize if hasattr(dt, 'itemsize') else o + cnt * 4].view(dt).reshape(shape)
G = ws[o_G:o_G + n * k * 4].view(torch.float32).reshape(n, k)
Cc = ws[o_C:o_C + n * k * 4].view(torch.float32).reshape(n, k)
part = ws[o_part:o_part + 5 * ntiles16 * 4].view(torch.float32)
flags = ws[o_flags:o_flags + 16].view(torch.int32)
fvals = ws[o_fvals:o_fvals + 16].view(torch.float32)
Xf, Wf, p = Xb.float(), Wb.float(), z0.float()
r0 = p.bfloat16().float() @ Wf.T - Xf
g_ref = r0.bfloat16().float() @ Wf
print("G err", (G - g_ref).abs().max().item(), "G scale", g_ref.abs().max().item())
print("rss0 partial", part[:1].tolist(), "ref", r0.pow(2).sum().item())
print("flags", flags.tolist(), "fvals", fvals.tolist())
lr_acc = fvals[2].item()
zc = torch.nn.functional.softshrink(p - lr_acc * g_ref, 0.5 * lr_acc)
print("C err", (Cc - zc).abs().max().item())
r1 = zc.bfloat16().float() @ Wf.T - Xf
ntile = 1
print("partials [rss1,l1,dzg,dz2]", [part[ntile * (1 + i)].item() for i in range(4)], "ref", r1.pow(2).sum().item(), zc.abs().sum().item(), ((zc - p) * g_ref).sum().item(), (zc - p).pow(2).sum().item())
print("z err", (z.float() - zc).abs().max().item())
Zf = ws[o_Zf:o_Zf + n * k * 4].view(torch.float32).reshape(n, k)
print("Zf vs C", (Zf - Cc).abs().max().item(), "z vs Zf", (z.float() - Zf).abs().max().item(), "z vs z0", (z.float() - z0.float()).abs().max().item())
print(z[0, :6].float().tolist(), Zf[0, :6].tolist())
