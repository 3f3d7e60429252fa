#!/usr/bin/env python3
"""A/B of two builds of the library on the convolutional solver: per-iteration time of each and the codes compared
bitwise.  usage: ab_conv.py <other.so | env:NAME=VALUE> ... (one process per build: the library is chosen once per
process; env:... runs the product library with that variable set, e.g. env:LASSO_CONV_FUSED=0);
AB_CONV_TOL / AB_CONV_MAXITER in the environment set the stop rule's tolerance (default 0: no rule) and maxiter (20)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(256, 1, 64, 7, 1, 0, 26), (64, 3, 128, 5, 1, 2, 64), (64, 3, 64, 5, 1, 2, 64), (32, 16, 256, 3, 1, 1, 64),
         (128, 1, 32, 5, 2, 1, 15), (96, 2, 48, 3, 1, 1, 20),
         (2048, 1, 16, 3, 1, 1, 8), (2048, 1, 64, 3, 1, 1, 8), (512, 1, 32, 5, 1, 0, 28), (256, 3, 24, 5, 1, 2, 40)]
if os.environ.get("AB_CONV_CASES") == "fused":     # geometries of the one-launch kernel (conv_fused.hip): N >= CUs, small images
    CASES = [(256, 1, 64, 7, 1, 0, 26), (2048, 1, 16, 3, 1, 1, 8), (2048, 1, 64, 3, 1, 1, 8), (512, 1, 32, 5, 1, 0, 28),
             (256, 3, 32, 5, 1, 2, 32), (300, 2, 48, 3, 1, 1, 20), (256, 1, 64, 7, 1, 3, 32), (777, 1, 24, 5, 1, 1, 13),
             (256, 4, 12, 3, 1, 0, 30), (1024, 1, 40, 7, 1, 2, 16),
             # K = 128 (atoms contracted in two halves), and N < CUs: bands of code rows with their halos, one launch per iteration
             (256, 3, 128, 5, 1, 2, 32), (300, 1, 128, 7, 1, 0, 20), (64, 3, 128, 5, 1, 2, 64), (64, 3, 64, 5, 1, 2, 64), (96, 2, 48, 3, 1, 1, 20),
             # more than 4096 residual values per image (16 outputs per thread), N below and above the number of CUs
             (256, 3, 24, 5, 1, 2, 40), (256, 1, 64, 7, 1, 0, 64), (256, 3, 64, 5, 1, 2, 44), (200, 1, 64, 7, 1, 0, 26), (128, 1, 16, 3, 1, 1, 8)]
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd")]
    import hashlib, time, torch
    from lasso_amd import _native as nat
    if sys.argv[2] != "-":
        nat.use_library(os.path.abspath(sys.argv[2]))
    from lasso_amd.conv2d import ista_conv2d
    out = []
    for (N, C, K, ks, st, pd, Hz) in CASES:
        g = torch.Generator().manual_seed(0)
        w = torch.randn(K, C, ks, ks, generator=g) / ks
        H = (Hz - 1) * st - 2 * pd + ks
        x = torch.randn(N, C, H, H, generator=g)
        lr = 0.5 / w.pow(2).sum().item()
        xg, wg, zg = x.cuda(), w.cuda(), torch.zeros(N, K, Hz, Hz, device="cuda")
        tol, mi = float(os.environ.get("AB_CONV_TOL", "0")), int(os.environ.get("AB_CONV_MAXITER", "20"))
        z, info = ista_conv2d(xg, zg, wg, 0.1, stride=st, padding=pd, maxiter=mi, lr=lr, tol=tol, return_info=True)
        torch.cuda.synchronize()
        t = time.perf_counter(); reps = 10
        for _ in range(reps):
            ista_conv2d(xg, zg, wg, 0.1, stride=st, padding=pd, maxiter=mi, lr=lr, tol=tol)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps / max(info["iterations"], 1)
        out.append({"case": [N, C, K, ks, st, pd, Hz], "us_per_iteration": round(dt * 1e6, 2),
                    "tflops": round(4.0 * N * Hz * Hz * C * ks * ks * K / dt / 1e12, 2),
                    "nnz": int((z != 0).sum()), "iterations": info["iterations"], "sha": hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:16]})
    print(json.dumps(out))
    sys.exit(0)
res = {}
libs = [("product", "-")] + [(l if l.startswith("env:") else os.path.basename(l), l) for l in sys.argv[1:]]
for tag, lib in libs:
    env = dict(os.environ)
    if lib.startswith("env:"):
        k, v = lib[4:].split("=", 1)
        env[k] = v
        lib = "-"
    r = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True, env=env)
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
for i, n in enumerate(res["product"]):
    print(json.dumps({"case": n["case"], "us": {t: res[t][i]["us_per_iteration"] for t, _ in libs},
                      "product_tflops": n["tflops"], "bitwise": all(res[t][i]["sha"] == n["sha"] for t, _ in libs),
                      "iterations": [res[t][i]["iterations"] for t, _ in libs], "nnz": n["nnz"]}))
