import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,'pytorch-lasso_amd'), os.path.join(ROOT,'tests')]
from recipes import recipe_xw, recipe_c4_init
from lasso_amd.engine import HipEngine
from lasso_amd.parallel import dict_learning_sharded
n=int(sys.argv[1]) if len(sys.argv)>1 else 8192
X,_=recipe_xw(n); Xs=X.cuda(); D0=recipe_c4_init()
for side in ((True,) if '--one' in sys.argv else (False, True, False, True)):
    eng=HipEngine(); eng.side_objective=side
    kw=dict(alpha=0.5, algorithm='ista', progbar=False, init_weight=D0, engine=eng)
    dict_learning_sharded(Xs,1024,steps=5,**kw); torch.cuda.synchronize()
    t=time.perf_counter()
    D,losses=dict_learning_sharded(Xs,1024,steps=40,**kw); torch.cuda.synchronize()
    print('side',side,'step_ms',(time.perf_counter()-t)/40*1e3,'loss',losses[-1].item(), D.double().sum().item())
