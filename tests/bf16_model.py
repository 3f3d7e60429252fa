"""CPU model of the arithmetic of the single-launch bf16 solve (csrc/bt16_persist.hip) -- test infrastructure.

The reference run on bf16 tensors (ista.py:17-54,57-104) rounds EVERY ATen result to bf16, including the scalars
F and Q its line search compares (8 bits of mantissa: fixture g3_c3_trace's bf16 trial counts differ from the fp32
ones for that reason alone).  The HIP kernel keeps what lives in tensors in bf16 -- the point p, the gradient g,
every candidate, the iterate z -- but accumulates GEMMs and the five sums of a trial in fp32 / double.  This file
restates exactly THAT arithmetic with torch CPU ops (rounding points as in the kernel source, cited below), so the
tests can pin the kernel's trial trace, accepted steps and code against an independent evaluation instead of only
bounding its objective: a wrong accept/reject decision cannot hide inside the objective tolerance.
Differences that remain: summation order inside the fp32 GEMMs and sums (relative 1e-7), which can only move a
decision whose F and Q agree to that precision."""
import math

import numpy as np
import torch


def bf(t):
    """round to bf16 and back (values stay fp32 tensors)"""
    return t.to(torch.bfloat16).to(torch.float32)


def _f32(v):
    return float(np.float32(v))


def _shrink(v, lam):
    # soft_threshold() of tile_device.hpp: v - clamp(v, -lam, lam)
    return v - torch.clamp(v, -lam, lam)


def solve(x, z0, weight, alpha, lr0, maxiter, tol=0.0, fast=True, backtrack=True, eta=1.5, max_trials=1000):
    """x [n,d], z0 [n,k], weight [d,k]: bf16 tensors.  Returns (z bf16, info) with info['trials'],
    info['accepted_lr'], info['iterations'], info['last_delta']."""
    xf, wf = x.float(), weight.float()
    z = z0.float().clone()                    # iterate z: bf16 in memory (bt16_persist.hip: store_z8)
    y = z.clone()                             # the point p: bf16 tile in LDS (y_0 = z_0, ista.py:76-78)
    n, k = z.shape
    budget = _f32(float(n) * float(k) * tol)
    alpha_f = _f32(alpha)
    trials, lrs = [], []
    t_mom, last, it_done = 1.0, float('nan'), 0
    for it in range(maxiter):
        t_next = (1.0 + math.sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0
        coef = _f32((t_mom - 1.0) / t_next) if fast else 0.0
        p = y
        r0 = p @ wf.T - xf                                            # GEMM-1, fp32 accumulation; residual()
        rss0 = _f32((r0 * r0).sum(dtype=torch.float64).item())        # (fp32 terms, double accumulation: no 16M-element fp64 temporaries)
        g = bf(bf(r0) @ wf)                                           # residual tile and g are stored as bf16
        lr_acc, lam_acc, t_acc = _f32(lr0), _f32(alpha * lr0), 0
        if backtrack:
            lr_d, accepted = float(lr0), False
            for s in range(max_trials):
                lr_s, lam_s, hol = _f32(lr_d), _f32(alpha * lr_d), _f32(0.5 / lr_d)
                zc = bf(_shrink(p - lr_s * g, lam_s))                 # candidates(): bf16_round(soft_threshold(v, lam))
                d = zc - p
                l1 = _f32(zc.abs().sum(dtype=torch.float64).item())
                dzg = _f32((d * g).sum(dtype=torch.float64).item())
                dz2 = _f32((d * d).sum(dtype=torch.float64).item())
                r1 = zc @ wf.T - xf
                rss1 = _f32((r1 * r1).sum(dtype=torch.float64).item())
                f0 = _f32(np.float32(0.5) * np.float32(rss0))                                   # decide(): ista.py:23
                al1 = _f32(np.float32(alpha_f) * np.float32(l1))
                F = _f32(np.float32(_f32(np.float32(0.5) * np.float32(rss1))) + np.float32(al1))  # :28
                Q = _f32(np.float32(_f32(np.float32(_f32(np.float32(f0) + np.float32(dzg))) +
                                         np.float32(_f32(np.float32(hol) * np.float32(dz2))))) + np.float32(al1))
                if F <= Q:                                                                       # :45
                    lr_acc, lam_acc, t_acc, accepted = lr_s, lam_s, s, True
                    break
                lr_d = lr_d / eta                                                                # :47
            if not accepted:                      # :48-52: warn, revert to lr0
                t_acc = max_trials - 1
        trials.append(t_acc + 1)
        lrs.append(lr_acc)
        zn = bf(_shrink(p - lr_acc * g, lam_acc))                     # accept: ista.py:40 with the accepted step
        last = _f32((z - zn).abs().sum(dtype=torch.float64).item())   # :93
        y = bf(zn + coef * (zn - z))                                  # :99-100, the next point (bf16 tile)
        z = zn                                                        # :102
        t_mom = t_next
        it_done = it + 1
        if tol > 0 and last <= budget:
            break
    return z.to(torch.bfloat16), dict(trials=trials, accepted_lr=lrs, iterations=it_done, last_delta=last)
