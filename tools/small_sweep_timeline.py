"""Per-block time stamps of the one-workgroup sweep (sweep_small_kernel): needs tools/build_variant.sh sweep_t mstep.hip
-DLASSO_SWEEP_TIMING.  Columns (us from the start of block 0): loop top, chain start / end (wave 0), staging done
(waves 1-3), work done (waves 1, 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
nat.use_library(os.path.join(ROOT, 'variants', 'liblasso_sweep_t.so'))
from lasso_amd.engine import HipEngine
eng = HipEngine()
k, d, n = 256, 64, 4096
g = torch.Generator().manual_seed(k)
Z = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.3)).cuda()
X = torch.randn(n, d, generator=g).cuda()
D = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).cuda()
A, B = eng.gram(Z, X, torch.empty(k * k + k * d, device='cuda'))
for _ in range(3):
    eng.sweep(A, B, D, None, 1e-10, False)
torch.cuda.synchronize()
ws = eng._ws(0, "sweep")
al = lambda x: (x + 255) // 256 * 256
off = 2 * al(k * 256 * 4)
nblk = (k + 31) // 32
t = ws.view(torch.uint8)[off: off + (nblk + 1) * 64].view(torch.int64).view(nblk + 1, 8).cpu()
t0 = int(t[0, 0])
names = ["top", "chain0", "chain1", "w1_staged", "w2_staged", "w3_staged", "w1_done", "w2_done"]
for b in range(nblk + 1):
    print(b, " ".join("%s=%.2f" % (names[i], (int(t[b, i]) - t0) / 100.0) for i in range(8) if int(t[b, i]) != 0))
