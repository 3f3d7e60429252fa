import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd import _native as nat
if '--lib' in sys.argv: nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib')+1]))
from lasso_amd.engine import HipEngine
eng=HipEngine()
for n in (4096, 8192, 16384, 32768, 65536):
    g=torch.Generator().manual_seed(n); Z=(torch.randn(n,1024,generator=g)*(torch.rand(n,1024,generator=g)<0.3)).cuda(); X=torch.randn(n,256,generator=g).cuda()
    buf=torch.zeros(1024*1024+1024*256,device='cuda')
    for _ in range(3): A,B=eng.gram(Z,X,buf)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): A,B=eng.gram(Z,X,buf)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    ref=(Z.double().T@Z.double()); err=(A.double()-ref).abs().max().item()/ref.abs().max().item()
    print(n,'gram_ms %.4f'%(dt*1e3),'rel err %.1e'%err)
