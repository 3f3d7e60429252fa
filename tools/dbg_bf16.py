import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
for (n, d, k) in ((64, 256, 1024), (100, 48, 200), (16384, 256, 1024)):
    X, W = recipe_xw(n, d, k)
    Xb, Wb = X.cuda().bfloat16(), W.cuda().bfloat16()
    z0 = torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
    print("case", n, d, k, flush=True)
    z, info = ista(Xb, z0, Wb, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, return_info=True)
    torch.cuda.synchronize()
    zf = z.float()
    obj = ((0.5 * (zf @ Wb.float().T - Xb.float()).pow(2).sum() + 0.5 * zf.abs().sum()) / n).item()
    z32 = ista(Xb.float(), z0.float(), Wb.float(), 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True)
    obj32 = ((0.5 * (z32 @ Wb.float().T - Xb.float()).pow(2).sum() + 0.5 * z32.abs().sum()) / n).item()
    print(" obj bf16-native %.5f  fp32-kernels %.5f  max|dz| %.4f" % (obj, obj32, (zf - z32).abs().max().item()), info, flush=True)
