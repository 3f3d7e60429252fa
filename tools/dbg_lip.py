import sys, os, torch, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from lasso_amd import _native as nat
if '--lib' in sys.argv: nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib')+1]))
from lasso_amd.linear.lipschitz import lipschitz_constant
out={}
for (d,k) in [(256,1024),(128,512),(96,300),(200,1000),(224,224),(160,4096),(1024,256),(64,256)]:
    g=torch.Generator().manual_seed(d*7+k); W=torch.randn(d,k,generator=g).cuda()
    out['%dx%d'%(d,k)]=float(lipschitz_constant(W)).hex()
print(json.dumps(out))
