import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
X,W=recipe_xw(16384,256,1024); Xg,Wg=X.cuda(),W.cuda(); z0=torch.zeros(16384,1024,device='cuda')
for _ in range(4): ista(Xg,z0,Wg,0.5,lr=1.0,maxiter=10,tol=0.0,backtrack=True)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): ista(Xg,z0,Wg,0.5,lr=1.0,maxiter=10,tol=0.0,backtrack=True)
torch.cuda.synchronize(); print('ms_per_solve',(time.perf_counter()-t)/5*1e3)
