import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
n=int(sys.argv[1]) if len(sys.argv)>1 else 512
X,W=recipe_xw(n,256,1024); Xg,Wg=X.cuda(),W.cuda(); z0=torch.zeros(n,1024,device='cuda')
for _ in range(300): ista(Xg,z0,Wg,0.5,lr=0.1,maxiter=100,tol=0.0)
torch.cuda.synchronize()
