#!/usr/bin/env python3
"""Compile every kernel source of the library to gfx950 assembly and list the kernels that hold a chain of >= 6 global /
buffer loads each followed by `s_waitcnt vmcnt(0)` (no MFMA, LDS operation, barrier or store between them): loads that
hipcc has put under a condition of their own, i.e. memory round trips in a row.  Spin loops (sc1 loads) are skipped.
usage: scan_serial_loads.py   (CPU only; a minute)"""
import re, subprocess, sys, os, glob
C = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-lasso_amd", "csrc")
for f in sorted(glob.glob(C + "/*.hip")):
    out = "/tmp/scan_" + os.path.basename(f) + ".s"
    extra = ["-fno-slp-vectorize"] if f.endswith("bt16_persist.hip") else []
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-S", "--cuda-device-only", f, "-o", out] + extra, stderr=subprocess.DEVNULL)
    if not os.path.exists(out): continue
    kern, run, best = None, 0, {}
    prev_load = False
    for l in open(out):
        l = l.strip()
        m = re.match(r'^(_Z\w+):', l)
        if m: kern = m.group(1); run = 0; continue
        op = l.split()[0] if l and not l.startswith((';', '.')) else None
        if op is None: continue
        if op.startswith(("global_load", "buffer_load")) and "lds" not in l and "sc1" not in l:
            prev_load = True; continue
        if op == "s_waitcnt" and "vmcnt(0)" in l and prev_load:
            run += 1; best[kern] = max(best.get(kern, 0), run); prev_load = False; continue
        if op.startswith(("v_mfma", "s_barrier", "ds_", "global_store", "buffer_store")):
            run = 0
        prev_load = False
    for k, v in sorted(best.items(), key=lambda kv: -kv[1]):
        if v >= 6: print(os.path.basename(f), v, k[:110])
