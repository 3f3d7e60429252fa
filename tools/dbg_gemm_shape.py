import sys, os, torch, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd import _native as nat
if '--lib' in sys.argv: nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib')+1]))
from lasso_amd.linear import sparse_encode
from recipes import recipe_xw
out=[]
for n,d,k in ((4096,2048,2048),(8192,1024,4096),(2048,300,1500),(1024,512,2048)):
    X,W=recipe_xw(n,d,k); Xg,Wg=X.cuda(),W.cuda()
    for _ in range(2): sparse_encode(Xg,Wg,alpha=0.5,lr=0.01,maxiter=20,tol=0.0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(3): sparse_encode(Xg,Wg,alpha=0.5,lr=0.01,maxiter=20,tol=0.0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
    out.append((n,d,k,round(4.0*n*d*k*20/dt/1e12,1)))
print(out)
