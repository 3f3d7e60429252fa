"""Per-block time stamps of the single-launch atom sweep (csrc/mstep.hip, sweep_persist_kernel): needs the
debug build  tools/build_variant.sh sweep_t mstep.hip -DLASSO_SWEEP_TIMING  (wall_clock64 stamps in the
workspace).  Columns: loop top, chain start/end (wave 0), publish / A blocks staged (wave 1), rows taken /
worker rows arrived / rows staged (waves 2-3), microseconds from the start of block 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from lasso_amd import _native as nat
nat.use_library(sys.argv[sys.argv.index('--lib') + 1] if '--lib' in sys.argv else os.path.join(ROOT, 'variants', 'liblasso_sweep_t.so'))
from lasso_amd.engine import HipEngine
eng = HipEngine()
k, d, n = 1024, 256, 4096
if '--shape' in sys.argv:
    k, d = [int(v) for v in sys.argv[sys.argv.index('--shape') + 1].split('x')]
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
dp = 256
off_ex = 2 * al(k * dp * 4) + al(32 * dp * 4) + 256
nblk = (k + 31) // 32; rows = nblk * 32 * 256
f = ws.view(torch.uint8)
t = f[off_ex + 3 * rows * 4 + 1024: off_ex + 3 * rows * 4 + 1024 + nblk * 128].view(torch.int64).view(nblk, 16).cpu()
t0 = int(t[0, 0])
names = {0: "top", 1: "chain0", 2: "chain1", 4: "h_start", 5: "h_pub", 6: "h_stageA", 7: "h_taken", 8: "h_worker", 9: "h_rows", 10: "h_prog"}
for b in [x for x in list(range(0, 6)) + [16, 30, 31] if x < nblk]:
    print(b, " ".join("%s=%.2f" % (names[i], (int(t[b, i]) - t0) / 100.0) for i in names if int(t[b, i]) != 0))
print("total us", (int(t[nblk - 1, 2]) - t0) / 100.0)
