#!/usr/bin/env python3
"""Pipelined M-step (lasso_mstep_pipe_*) against the plain one (lasso_gram_accumulate + lasso_dict_sweep) on the same
(Z, X, D): [A | B] against an fp64 product, the dictionaries against each other, repeated to shake the hand-offs.
usage: check_pipe.py [n] [k] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from lasso_amd import _native as nat  # noqa: E402
if '--lib' in sys.argv:
    i = sys.argv.index('--lib')
    nat.use_library(sys.argv[i + 1])
    del sys.argv[i:i + 2]
STAMPS = '--stamps' in sys.argv
if STAMPS:
    sys.argv.remove('--stamps')
from lasso_amd.engine import HipEngine  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    d = 256
    dev = torch.device("cuda", 0)
    eng = HipEngine(dev)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(n, d, generator=g).to(dev)
    Z = (torch.randn(n, k, generator=g) * (torch.rand(n, k, generator=g) < 0.2)).to(dev)
    D0 = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0).to(dev)
    stages = eng.mstep_pipe_stages(d, k)
    print("stages (rows of [A | B]):", stages)
    nrb = len(stages)
    assert nrb > 0
    # plain
    buf = torch.zeros(k * k + k * d, device=dev)
    A, B = eng.gram(Z, X, buf)
    Dp = D0.clone()
    maskp, ndegp = eng.sweep(A, B, Dp, None, 1e-10, False)
    torch.cuda.synchronize()
    A64 = (Z.double().T @ Z.double())
    B64 = (Z.double().T @ X.double())
    print("plain  A err %.3g  B err %.3g" % ((A.double() - A64).abs().max().item(), (B.double() - B64).abs().max().item()))
    ws = eng.mstep_pipe_workspace(n, d, k)
    AB = torch.zeros(k, k + d, device=dev)
    M = torch.cuda.current_stream(dev)
    S = torch.cuda.Stream(dev)
    worst = 0.0
    times = []
    big = torch.zeros(4096, 4096, device=dev)
    for rep in range(reps):
        AB.fill_(float("nan"))
        D = D0.clone()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            big @ big                                    # ~1 ms of work in front: the host gets ahead, as in an EM loop
        t0.record()
        eng.pipe_gram(Z, X, AB, 0, ws)
        eng.pipe_rows(AB, D, n, 0, ws, seq=rep + 1)
        with torch.cuda.stream(S):
            eng.pipe_wait(n, d, k, rep + 1, ws)
            for R in range(1, nrb):
                eng.pipe_gram(Z, X, AB, R, ws)
                eng.pipe_rows(AB, D, n, R, ws)
            evs = torch.cuda.Event(); evs.record(S)
        mask = eng.pipe_sweep(AB, D, n, 1e-10, False, ws)
        M.wait_event(evs)
        res = eng.pipe_finish(D, n, 1e-10, False, mask, ws)
        t1.record()
        _, ndeg = res()
        torch.cuda.synchronize()
        times.append(t0.elapsed_time(t1) * 1e3)
        ea = (AB[:, :k].double() - A64).abs().max().item()
        eb = (AB[:, k:].double() - B64).abs().max().item()
        sym = (AB[:, :k] - AB[:, :k].T).abs().max().item()
        dd = (D - Dp).abs().max().item()
        worst = max(worst, dd)
        if rep < 3 or dd > 1e-4:
            print("rep %d: A err %.3g B err %.3g asym %.3g  max|D - D_plain| %.3g ndeg %d (plain %d)  %.1f us"
                  % (rep, ea, eb, sym, dd, ndeg, ndegp, times[-1]))
    if STAMPS:      # -DLASSO_SWEEP_TIMING build: per-block stamps of the LAST pipelined sweep (tools/sweep_timeline.py's columns)
        al = lambda x: (x + 255) // 256 * 256
        off_ex = 2 * al(k * 256 * 4) + al(32 * 256 * 4) + 256
        nblk = (k + 31) // 32
        rows = nblk * 32 * 256
        t = ws[off_ex + 3 * rows * 4 + 1024: off_ex + 3 * rows * 4 + 1024 + nblk * 128].view(torch.int64).view(nblk, 16).cpu()
        t0s = int(t[0, 0])
        names = {0: "top", 1: "chain0", 2: "chain1", 4: "h_start", 5: "h_pub", 6: "h_stageA", 7: "h_taken", 8: "h_worker", 9: "h_rows", 10: "h_prog"}
        for b in range(nblk):
            print(b, " ".join("%s=%.2f" % (names[i], (int(t[b, i]) - t0s) / 100.0) for i in names if int(t[b, i]) != 0))
    # the sweep alone on the pipelined [A | B] with the plain kernel: must match the pipelined dictionary bit for bit
    Dq = D0.clone()
    eng.sweep(AB[:, :k].contiguous(), AB[:, k:].contiguous(), Dq, None, 1e-10, False)
    torch.cuda.synchronize()
    print("pipelined vs plain sweep on the SAME [A | B]: max diff %.3g (expect 0)" % (D - Dq).abs().max().item())
    times.sort()
    print("worst max|D - D_plain| over %d reps: %.3g;  pipelined M-step median %.1f us, min %.1f us" % (reps, worst, times[len(times) // 2], times[0]))
    # plain timing
    tp = []
    for rep in range(reps):
        Dp2 = D0.clone()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            big @ big
        t0.record()
        A, B = eng.gram(Z, X, buf)
        r = eng.sweep_begin(A, B, Dp2, 1e-10, False)
        t1.record()
        r()
        torch.cuda.synchronize()
        tp.append(t0.elapsed_time(t1) * 1e3)
    tp.sort()
    print("plain M-step median %.1f us, min %.1f us" % (tp[len(tp) // 2], tp[0]))


if __name__ == "__main__":
    main()
