#!/usr/bin/env python3
"""A/B of builds on BASELINE config 3 (fp32 line search): each library in its own process, interleaved rounds, median
ms per solve; trace and objective of every build printed.  usage: ab_c3.py [--dtype f32|bf16] lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, time
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch
from lasso_amd import _native as nat
if sys.argv[2] != "-":
    nat.use_library(os.path.abspath(sys.argv[2]))
from lasso_amd.linear.solvers import ista
from recipes import recipe_xw
dt = torch.float32 if sys.argv[3] == "f32" else torch.bfloat16
X, W = recipe_xw(16384, 256, 1024)
Xg, Wg = X.cuda().to(dt), W.cuda().to(dt)
z0 = torch.zeros(16384, 1024, device="cuda", dtype=dt)
f = lambda **kw: ista(Xg, z0, Wg, 0.5, lr=1.0, maxiter=10, tol=0.0, backtrack=True, **kw)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4: f()
torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): z = f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10)
_, info = f(return_info=True)
zf = z.float()
obj = ((0.5 * (zf @ Wg.float().T - Xg.float()).pow(2).sum() + 0.5 * zf.abs().sum()) / 16384).item()
print(json.dumps({"ms": sorted(res)[2], "min": min(res), "trials": info["trials"], "objective": obj}))
'''
def main():
    args = sys.argv[1:]
    dtype = "f32"
    if args and args[0] == "--dtype":
        dtype, args = args[1], args[2:]
    libs = args or ["-"]
    out = {l: [] for l in libs}
    for rnd in range(3):
        for l in libs:
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, l, dtype], capture_output=True, text=True, timeout=300)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(l, "FAILED", r.stderr[-500:]); continue
            out[l].append(json.loads(line[0]))
    for l in libs:
        ms = sorted(x["ms"] for x in out[l])
        if ms:
            print("%-40s median %.4f ms  (rounds %s)  trials %s  objective %.6f" % (
                os.path.basename(l), ms[len(ms) // 2], " ".join("%.4f" % m for m in ms), out[l][0]["trials"], out[l][0]["objective"]))
if __name__ == "__main__":
    main()
