import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')): sys.path.insert(0,p)
import torch
from lasso_amd.linear import sparse_encode
from lasso_amd import _native as nat
from oracle import lasso_oracle as orc
torch.manual_seed(0)
for (n,d,k) in [(16,256,256),(16,256,1024),(37,10,50)]:
    g=torch.Generator().manual_seed(1)
    W=torch.nn.functional.normalize(torch.randn(d,k,generator=g),dim=0); X=torch.randn(n,d,generator=g)
    lr=1.0/orc.lipschitz_constant(W,'exact')
    for fast in (False,True):
        for M in (1,2,3,5):
            ref=orc.sparse_encode(X,W,alpha=0.3,fast=fast,lr=lr,maxiter=M,tol=0.0)
            got=sparse_encode(X.cuda(),W.cuda(),alpha=0.3,fast=fast,lr=lr,maxiter=M,tol=0.0).cpu()
            e=(got-ref).abs()
            print((n,d,k),'fast',fast,'M',M,'maxerr %.3g'%e.max().item(), 'bad rows', (e.max(1).values>1e-4).nonzero().flatten().tolist()[:8], 'bad cols', (e.max(0).values>1e-4).nonzero().flatten().tolist()[:12])
    kp = 256 if k<=256 else (512 if k<=512 else 1024)
    ws=nat._WS[('cuda',0)]
    off=2*256*kp*4
    coef=ws[off:off+4*8].view(torch.float32).cpu()
    print('coef', coef.tolist(), 'expected', orc.momentum_schedule(5))
