#!/usr/bin/env python3
"""bench.py -- FISTA iterations/sec on BASELINE config 2 (n=4096 d=256 k=1024 fp32,
fixed step 1/L, tol=0) through the C ABI, one process per GPU.

A "step" is one sparse_encode solve of --iters FISTA iterations (default 100, the
count BASELINE.md's config-2 timing uses) over a 4096 x 256 batch against a
1024-atom dictionary, inputs resident in HBM.  Multi-GPU is weak scaling: every
rank owns its own 4096-row shard (rows are independent lasso problems; no
data-path collective), value = iterations/s summed over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

N_ROWS, D, K, ALPHA = 4096, 256, 1024, 0.5
LAMBDA_MAX = 8.877719052098003          # fp64 lambda_max(W^T W) of the recipe dictionary
PEAK_F32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak


def recipe(n_total):
    g = torch.Generator().manual_seed(0)
    W = torch.nn.functional.normalize(torch.randn(D, K, generator=g), dim=0)
    X = torch.randn(n_total, D, generator=g)
    return X, W


def hbm_traffic_from_profiles():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE, profiles/*/hbm_traffic.json);
    PMC counters cannot be collected live inside the timed run."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, name, "hbm_traffic.json")
        if os.path.exists(f):
            with open(f) as fh:
                best = json.load(fh).get("hbm_bytes_per_launch")
    return best


def cpu_baseline(X, W, lr, budget_s=12.0):
    """Time the CPU oracle (restatement of the reference, same ATen ops) on the host
    cores of this box on a bounded sample of the same workload."""
    from oracle import lasso_oracle as orc
    z0 = X.new_zeros(X.shape[0], K)
    # pick the thread count that is fastest on this host (ATen's elementwise passes stop
    # scaling long before all cores are used), then time the bounded sample with it
    default_threads = torch.get_num_threads()
    best = (float("inf"), default_threads)
    for nt in sorted({8, 16, 32, 64, default_threads}):
        if nt > (os.cpu_count() or nt):
            continue
        torch.set_num_threads(nt)
        orc.fista(X, z0, W, ALPHA, lr=lr, maxiter=1, tol=0.0)
        t0 = time.perf_counter()
        orc.fista(X, z0, W, ALPHA, lr=lr, maxiter=3, tol=0.0)
        best = min(best, ((time.perf_counter() - t0) / 3, nt))
    per_it, nthreads = best
    torch.set_num_threads(nthreads)
    iters = int(max(5, min(400, budget_s / max(per_it, 1e-4))))
    t0 = time.perf_counter()
    orc.fista(X, z0, W, ALPHA, lr=lr, maxiter=iters, tol=0.0)
    dt = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": iters / dt, "unit": "iterations/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d FISTA iterations, n=%d d=%d k=%d fp32, oracle/lasso_oracle.py, "
                      "os.cpu_count()=%s, %s" % (iters, X.shape[0], D, K, os.cpu_count(), model)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--iters", type=int, default=100, help="FISTA iterations per step (solve)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-tol", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from lasso_amd.linear.solvers import ista

    X_all, W = recipe(N_ROWS * world)
    X = X_all[rank * N_ROWS:(rank + 1) * N_ROWS]
    Xg, Wg = X.to(dev), W.to(dev)
    z0 = torch.zeros(N_ROWS, K, device=dev)
    lr = 1.0 / LAMBDA_MAX

    def solve():
        return ista(Xg, z0, Wg, ALPHA, fast=True, lr=lr, maxiter=args.iters, tol=0.0)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solve()
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        z = solve()
        e.record()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    kern_ms = sorted(s.elapsed_time(e) for s, e in ev)
    avg_launch_ms = sum(kern_ms) / len(kern_ms)

    out = None
    if rank == 0:
        flop_per_launch = 4.0 * N_ROWS * D * K * args.iters
        achieved = flop_per_launch / (avg_launch_ms * 1e-3) / 1e12
        total_iters = world * args.steps * args.iters
        out = {
            "metric": "fista_iterations_per_sec (n=4096 d=256 k=1024 fp32 per GPU, fixed L, tol=0)",
            "value": total_iters / elapsed,
            "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: FISTA n=4096 d=256 k=1024 fp32, fixed L, "
                                   "no backtrack; step = one solve of %d iterations" % args.iters,
                       "iters_per_step": args.iters, "rows_per_gpu": N_ROWS,
                       "parallelism": "row-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": hbm_traffic_from_profiles(),
                         "kernel": "lasso::sp::fista_tile_sp_kernel<1024, false>",
                         "flop_per_launch": flop_per_launch,
                         "avg_launch_ms": avg_launch_ms, "median_launch_ms": kern_ms[len(kern_ms) // 2]},
        }
        if not args.no_time_to_tol:
            ista(Xg, z0, Wg, ALPHA, lr=lr, maxiter=2000, tol=1e-5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, info = ista(Xg, z0, Wg, ALPHA, lr=lr, maxiter=2000, tol=1e-5, return_info=True)
            torch.cuda.synchronize()
            out["time_to_tol"] = {"ms": 1e3 * (time.perf_counter() - t1),
                                  "iterations": info["iterations"], "tol": 1e-5,
                                  "rule": "sum|z-z_next| <= n*k*tol (ista.py:64,93), exact global"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(X, W, lr)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        # cheap sanity: objective of the result (HIP lasso_loss) next to the reference's known answer
        if args.iters == 100:
            from lasso_amd.linear import lasso_loss
            obj = lasso_loss(Xg, z, Wg, ALPHA).item()
            out["objective_after_100"] = obj
            out["objective_reference"] = 63.609337
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
