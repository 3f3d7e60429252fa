import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw
X,W=recipe_xw(16384,512,4096); Xg,Wg=X.cuda(),W.cuda()
for _ in range(2): sparse_encode(Xg,Wg,alpha=0.5,lr=0.05,maxiter=10,tol=0.0)
torch.cuda.synchronize()
