#!/usr/bin/env python3
"""bench.py -- BASELINE's metric on MI355X: FISTA iterations/sec (+ time-to-tol) on
n=4096 d=256 k=1024 fp32, fixed step 1/L, through the C ABI, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fista|em|c3|conv|cd] [--scaling strong|weak]
                    [--shape c4|c5] [--dtype bf16|f32] [--rows R] [--backend nccl|gloo] [--share-gpu]

--workload fista (default): a "step" is one sparse_encode solve of --iters FISTA iterations
  (default 100, the count BASELINE.md's config-2 timing uses), inputs resident in HBM, tol=0.
  N = 1: the 4096-row batch of BASELINE config 2 on one GPU.
  N > 1, --scaling strong (default; this IS BASELINE's "n=4096 at 1/2/4/8 GPU"): the SAME
    4096 rows sharded over the ranks (4096/N rows per GPU; rows are independent lasso
    problems, no data-path collective).  value = iterations/s of the whole 4096-row batch
    (every rank runs the same iteration count on its shard; time = max over ranks).
    The weak-scaling figure (4096 rows PER GPU) is measured in the same run and reported
    beside it under "weak_scaling".
  N > 1, --scaling weak: the weak figure is `value`.
--workload em: BASELINE config 4, the dict_learning EM loop (FISTA E-step with the defaults
  lr='auto', maxiter=10, tol=1e-5 + constrained least-squares M-step) on n=65536 rows sharded
  over the ranks, with the RCCL all-reduce of [Z^T Z | Z^T X | objective sums] per step;
  a "step" is one EM step; value = EM steps/s; the all-reduce time is reported separately.

  --shape c5: the same loop at the shape of BASELINE config 5 (8 x 8 patches: d=64, k=256, alpha=0.1, synthetic
  centred patches -- the Omniglot notebook is absent from the reference checkout); --rows sets the batch.
--workload c3: BASELINE config 3, FISTA with the backtracking line search (ista.py:17-54), n=16384 d=256 k=1024,
  lr0=1, 10 outer iterations, --dtype bf16 (default, config 3 as named) or f32 tensors; a "step" is one solve;
  value = outer iterations/s; FLOPs as executed = (4 per outer iteration + 2 per trial) n d k.  N > 1: the rows
  sharded, every F <= Q decision on sums all-reduced over the ranks (lasso_fista_solve_sharded).
--workload conv: convolutional FISTA on one of tools/bench_conv.py's geometries (--conv-case gray | rgb |
  c16), 20 iterations per solve, fixed step, no stop rule; a "step" is one solve; value = iterations/s; the roofline is
  HBM (the code and its momentum copy are streamed every iteration); N > 1: the images sharded, no collective.
--backend gloo --share-gpu: every rank on cuda:0, collectives over gloo (host-staged) -- how the multi-rank
  code paths of this file are executed on a ONE-GPU box (tests/test_bench_gpu.py); never a performance figure.

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this script spawns its N
ranks itself through `python -m torch.distributed.run` on 127.0.0.1; under a launcher whose
WORLD_SIZE differs from --gpus, or with fewer visible GPUs than ranks, it exits non-zero.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

N_ROWS, D, K, ALPHA = 4096, 256, 1024, 0.5
N_EM = 65536
LAMBDA_MAX = 8.877719052098003          # fp64 lambda_max(W^T W) of the recipe dictionary
PEAK_F32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
OBJ_REF_100, OBJ_RTOL = 63.609337, 2e-6  # mean objective after 100 iterations


def recipe(n_total):
    import torch
    g = torch.Generator().manual_seed(0)
    W = torch.nn.functional.normalize(torch.randn(D, K, generator=g), dim=0)
    X = torch.randn(n_total, D, generator=g)
    return X, W


def hbm_traffic_from_profiles(kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE, profiles/*/hbm_traffic.json):
    PMC counters cannot be collected inside the timed run, so this figure is the latest
    profile's, and the JSON says which.  A profile that names the kernel it measured is used only
    when that is the kernel this run launched (a stale figure is dropped, not reported)."""
    best, src = None, None
    pdir = os.path.join(ROOT, "profiles")
    names = sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []
    # the headline kernel's profile of the latest round: profiles/rNN_fista (older rounds: rNNx)
    names = [n for n in names if not n.endswith("_fista")] + [n for n in names if n.endswith("_fista")]
    for name in names:
        f = os.path.join(pdir, name, "hbm_traffic.json")
        if os.path.exists(f) and not name.endswith("_splitk"):
            with open(f) as fh:
                rec = json.load(fh)
            best, src = rec.get("hbm_bytes_per_launch"), "profiles/%s/hbm_traffic.json" % name
            if kernel and rec.get("kernel") and rec["kernel"] != kernel:
                best, src = None, src + " names %s, this run launched %s: stale, not reported" % (rec["kernel"], kernel)
    return best, src


def sources_digest():
    """sha1 over the kernel sources (pytorch-lasso_amd/csrc/*.hip, *.hpp, *.h, the Makefile): a traffic figure measured
    on one build is reported only while the sources are the ones that were measured."""
    import hashlib
    h = hashlib.sha1()
    cdir = os.path.join(ROOT, "pytorch-lasso_amd", "csrc")
    for name in sorted(os.listdir(cdir)):
        if name.endswith((".hip", ".hpp", ".h")) or name == "Makefile":
            with open(os.path.join(cdir, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def hbm_traffic_for(tag):
    """(bytes per step, source note) of a workload from the PMC passes committed for it (tools/prof_hbm.sh:
    profiles/rNN_<tag>/hbm_traffic.json; the scratch copy under gpurun_out/ first, so that a profiling call finds its
    own measurement).  None + the reason when no profile exists or when the kernel sources have changed since."""
    cands = []
    for base in ("gpurun_out", "profiles"):
        bdir = os.path.join(ROOT, base)
        if os.path.isdir(bdir):
            cands += [os.path.join(base, n) for n in sorted(os.listdir(bdir), reverse=True)
                      if n.startswith("r") and n.endswith("_" + tag)]
    why = "no profile of this workload (profiles/rNN_%s/hbm_traffic.json)" % tag
    for c in cands:
        f = os.path.join(ROOT, c, "hbm_traffic.json")
        if not os.path.exists(f):
            continue
        with open(f) as fh:
            rec = json.load(fh)
        if "hbm_bytes_per_step" not in rec:
            continue
        if rec.get("sources_digest") != sources_digest():
            why = "%s/hbm_traffic.json was measured on other kernel sources: stale, not reported" % c
            continue
        top = ", ".join("%s %.1f MB" % (r["kernel"].replace("lasso::", "")[:48], r["hbm_bytes_per_step"] / 1e6)
                        for r in rec.get("kernels", [])[:3])
        return rec["hbm_bytes_per_step"], ("%s/hbm_traffic.json (rocprofv3 PMC passes FETCH_SIZE x2 + WRITE_SIZE of `%s`, all "
                                           "dispatches / %g steps; not collected in this run; largest: %s)%s"
                                           % (c, rec.get("command", "?"), rec.get("steps_divisor", 0), top,
                                              ("; " + rec["conditions"]) if rec.get("conditions") else ""))
    return None, why


def with_traffic(roofline, tag, avg_ms, algorithmic_bytes=None):
    """fill roofline.traffic / hbm_gbps (north_star: achieved HBM GB/s) from the workload's committed PMC profile"""
    traffic, src = hbm_traffic_for(tag) if tag else (None, "not one of the profiled batch sizes")
    roofline["traffic"] = traffic
    roofline["traffic_source"] = src
    roofline["hbm_gbps"] = (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None
    roofline["hbm_peak_gbps"] = 8000.0
    if algorithmic_bytes is not None:
        roofline["algorithmic_bytes"] = algorithmic_bytes
        roofline["traffic_over_algorithmic"] = (traffic / algorithmic_bytes) if traffic else None
    return roofline


def cpu_baseline(X, W, lr, budget_s=12.0):
    """Time the CPU oracle (restatement of the reference, same ATen ops) on the host
    cores of this box on a bounded sample of the same workload."""
    import torch
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


# --------------------------------------------------------------------------------------
# launcher
# --------------------------------------------------------------------------------------
def nat_kernel_name(rows, d, k, dtype=None, backtrack=0):
    from lasso_amd import _native as nat
    return nat.lib().lasso_fista_kernel_name(rows, d, k, nat.LASSO_F32 if dtype is None else dtype, backtrack).decode()


def kernel_tokens(name):
    """the kernel names a `roofline.kernel` string mentions, each cut in front of its closing '>' (so that a name given
    with its leading template arguments matches the full instantiation in a trace): what tests/test_bench_gpu.py looks
    for in the rocprofv3 kernel trace of the same command"""
    import re
    return [t.rstrip(">") for t in re.findall(r"[A-Za-z_0-9]+_kernel(?:<[^<>]*>?)?", name or "")]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """--gpus N without a launcher: run the N ranks through torch.distributed.run on
    127.0.0.1 (the same command line the driver uses) and relay rank 0's JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def world_from_env(args):
    """(rank, local_rank, world); exits non-zero when the launcher and --gpus disagree."""
    if "WORLD_SIZE" not in os.environ:
        return 0, 0, 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    return rank, local_rank, world


class Ranks:
    """barrier + max-over-ranks timing, on RCCL ('nccl') or -- launcher self-test -- gloo."""

    def __init__(self, rank, world, device, backend, force=False):
        self.rank, self.world, self.device, self.dist, self.backend = rank, world, device, None, backend
        # force (--force-dist, N = 1 only): a ONE-rank process group, so that the collectives of the N > 1 paths -- RCCL
        # on device tensors when the backend is nccl -- execute on a one-GPU box (lasso_amd.parallel._sharded)
        if world > 1 or force:
            import torch.distributed as dist
            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                os.environ["LASSO_FORCE_COLLECTIVES"] = "1"
            kw = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            self.dist = dist
        self.sharded = self.dist is not None       # the multi-rank code paths run (N > 1, or --force-dist)

    def sync(self):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        if self.device.type == "cuda":
            torch.cuda.synchronize()

    def max(self, seconds):
        import torch
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.device if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def sum_(self, t):
        """in-place sum over the ranks of a device tensor (gloo: staged through the host)"""
        if self.dist is not None:
            from lasso_amd import parallel
            parallel._all_reduce(t, None)
        return t

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_steps(ranks, step, steps, warmup, events=True):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both
    sides; returns (max-over-ranks seconds, sorted per-step device ms on this rank)."""
    import torch
    for _ in range(warmup):
        step()
    ranks.sync()
    # ONE pair of HIP events brackets the K steps on the stream they are enqueued on (a pair per step costs ~10 us of idle
    # GPU per step -- the kernel trace of profiles/r03_fista shows it --, 0.3 % of a 3.1 ms step)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if events else None
    t0 = time.perf_counter()
    if events:
        ev[0].record()
    for i in range(steps):
        step()
    if events:
        ev[1].record()
    ranks.sync()
    elapsed = ranks.max(time.perf_counter() - t0)
    return elapsed, ([ev[0].elapsed_time(ev[1]) / steps] if events else [])


# --------------------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------------------
def run_fista(args, ranks):
    import torch
    from lasso_amd.linear.solvers import ista
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    lr = 1.0 / LAMBDA_MAX

    def shard_solver(n_total, rows):
        X_all, W = recipe(n_total)
        X = X_all[rank * rows:(rank + 1) * rows]
        Xg, Wg = X.to(dev), W.to(dev)
        z0 = torch.zeros(rows, K, device=dev)
        return X, W, Xg, Wg, z0, (lambda: ista(Xg, z0, Wg, ALPHA, fast=True, lr=lr, maxiter=args.iters, tol=0.0))

    def kernel_name(rows):
        from lasso_amd import _native as nat
        return nat.lib().lasso_fista_kernel_name(rows, D, K, nat.LASSO_F32, 0).decode()

    n_rows = args.rows
    results = {}
    modes = ["strong"] if world == 1 else ["strong", "weak"]
    for mode in modes:
        rows = n_rows // world if mode == "strong" else n_rows
        if mode == "strong" and n_rows % world:
            raise SystemExit("bench.py: %d rows do not split over %d ranks" % (n_rows, world))
        X, W, Xg, Wg, z0, solve = shard_solver(rows * world, rows)
        elapsed, kern_ms = timed_steps(ranks, solve, args.steps, args.warmup)
        results[mode] = dict(rows=rows, elapsed=elapsed, kern_ms=kern_ms, X=X, W=W, Xg=Xg, Wg=Wg, z0=z0,
                             z=solve())
    main_mode = args.scaling if world > 1 else "strong"
    r = results[main_mode]
    ttt = None
    if not args.no_time_to_tol:
        # time-to-tol with the reference's rule (ista.py:64,93: ONE sum over ALL rows of the batch, also when the rows
        # are spread over the ranks): every rank runs the solve on its shard, the per-iteration sums of a chunk of <= 64
        # iterations meet in one small all-reduce, every rank stops at the same iteration; time = max over the ranks
        from lasso_amd import parallel
        from lasso_amd.engine import HipEngine
        eng = HipEngine(dev)
        n_total = r["rows"] * world

        def to_tol():
            if not ranks.sharded:
                return ista(r["Xg"], r["z0"], r["Wg"], ALPHA, lr=lr, maxiter=2000, tol=1e-5, return_info=True)
            return parallel.sharded_encode(eng, r["Xg"], r["Wg"], ALPHA, None, lr=lr, maxiter=2000, tol=1e-5,
                                           n_global=n_total, return_info=True)
        to_tol()
        ranks.sync()
        t1 = time.perf_counter()
        _, info = to_tol()
        torch.cuda.synchronize()
        ttt = {"ms": 1e3 * ranks.max(time.perf_counter() - t1), "iterations": info["iterations"], "tol": 1e-5,
               "rows_total": n_total, "rows_per_gpu": r["rows"],
               "rule": "sum|z-z_next| <= n*k*tol (ista.py:64,93), global over all ranks" if ranks.sharded else
                       "sum|z-z_next| <= n*k*tol (ista.py:64,93), global (one rank holds the whole batch)",
               "reference_iterations": 263 if n_total == N_ROWS else None}
    out = None
    if rank == 0:
        def line(mode):
            q = results[mode]
            batches = world if mode == "weak" else 1
            return batches * args.steps * args.iters / q["elapsed"]
        avg_launch_ms = sum(r["kern_ms"]) / len(r["kern_ms"])
        flop_per_launch = 4.0 * r["rows"] * D * K * args.iters          # this rank's launch
        achieved = flop_per_launch / (avg_launch_ms * 1e-3) / 1e12
        traffic, traffic_src = hbm_traffic_for("fista") if (r["rows"] == N_ROWS and world == 1) else (None, None)
        new_format = traffic is not None
        if traffic is None:
            traffic, traffic_src = hbm_traffic_from_profiles(kernel_name(r["rows"]))
        out = {
            "metric": "fista_iterations_per_sec (n=%d d=256 k=1024 fp32, fixed L, tol=0)" % n_rows,
            "value": line(main_mode),
            "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * r["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": main_mode, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2: FISTA n=%d d=256 k=1024 fp32, fixed L, no backtrack; "
                                    "step = one solve of %d iterations" % (n_rows, args.iters)) +
                                   ("" if n_rows == N_ROWS else " (NOT the 4096-row headline batch)"),
                       "iters_per_step": args.iters, "rows_per_gpu": r["rows"],
                       "rows_total": r["rows"] * world,
                       "parallelism": "row-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic if world == 1 else None,
                         "traffic_source": (traffic_src + (" (rocprofv3 PMC pass of this command; not collected in "
                                                           "this run)" if traffic is not None and not new_format else ""))
                                           if traffic_src and world == 1 else None,
                         "hbm_gbps": (traffic / (avg_launch_ms * 1e-3) / 1e9) if (traffic and world == 1) else None,
                         "hbm_peak_gbps": 8000.0,
                         # resident-state model: x, W read, z0 read, z written -- per launch
                         "algorithmic_bytes": 4.0 * (r["rows"] * D + D * K + 2 * r["rows"] * K),
                         "traffic_over_algorithmic": (traffic / (4.0 * (r["rows"] * D + D * K + 2 * r["rows"] * K)))
                                                     if (traffic and world == 1) else None,
                         "kernel": kernel_name(r["rows"]),
                         "flop_per_launch": flop_per_launch, "per": "GPU (rank 0)",
                         "avg_launch_ms": avg_launch_ms,
                         "avg_launch_note": "HIP events around the K timed steps / K: the solve's launches (prepare + "
                                            "the persistent kernel) back to back"},
        }
        if world > 1:
            other = "weak" if main_mode == "strong" else "strong"
            q = results[other]
            out[other + "_scaling"] = {"value": line(other), "unit": "iterations/s", "rows_per_gpu": q["rows"],
                                       "rows_total": q["rows"] * world,
                                       "ms_per_step": 1e3 * q["elapsed"] / args.steps}
        if world == 1 and not args.no_shards and n_rows == N_ROWS:
            # what ONE rank of an N-GPU strong-scaling run of this batch does (4096/N rows, no
            # data-path collective): timed here on this GPU -- the per-rank ceiling of the N-GPU value
            shards = {}
            for nshard in (2, 4, 8):
                rows_s = N_ROWS // nshard
                Xs, Ws = recipe(N_ROWS)
                Xsg, Wsg = Xs[:rows_s].to(dev), Ws.to(dev)
                z0s = torch.zeros(rows_s, K, device=dev)
                # these legs are 12-40 ms long, so one host hiccup (a page-in on a fresh box: 3.1 ms/step was seen once
                # for the 0.59 ms leg with every kernel of its trace at its usual duration) would decide them: median of 3
                els = sorted(timed_steps(ranks, lambda: ista(Xsg, z0s, Wsg, ALPHA, fast=True, lr=lr, maxiter=args.iters,
                                                             tol=0.0), args.steps, args.warmup)[0] for _ in range(3))
                el = els[1]
                # The floor of the split-k form for this shard: its matrix-pipe time -- 4 rows d k flop
                # spread over all CUs at the fp32-MFMA peak -- plus ONE hand-off between compute units per iteration that no
                # schedule can hide: GEMM-2 of an iteration needs the residual rows that GEMM-1's partial sums of OTHER
                # workgroups complete (the price list of MI355X_MICROARCH.md: 1.0 us for a 4 KB one-to-one hand-off on an
                # idle chip).  What is measured above it is the rest of the exchange (a second hop in the reduce-scatter
                # form at T >= 2, eight waves meeting at a barrier twice per tile: every wave waits out the slowest hop).
                kname = kernel_name(rows_s)
                mfma_us = 4.0 * rows_s * D * K / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6
                floor_us = mfma_us + (1.0 if "splitk" in kname else 0.0)
                shards[str(nshard)] = {"rows": rows_s, "iterations_per_s": args.steps * args.iters / el,
                                       "ms_per_step": 1e3 * el / args.steps, "repeats": "median of 3 timed regions",
                                       "kernel": kname,
                                       "us_per_iteration": 1e6 * el / args.steps / args.iters,
                                       "floor_us_per_iteration": floor_us,
                                       "floor_note": "matrix-pipe time at the fp32 peak (%.2f us) + one 4 KB hand-off between "
                                                     "compute units per iteration that the split-k form cannot hide (1.0 us, "
                                                     "MI355X_MICROARCH.md price list): what an 8-GPU strong-scaling curve of this "
                                                     "batch should be read against" % mfma_us}
            out["strong_scaling_shards_on_one_gpu"] = shards
        if ttt is not None:
            out["time_to_tol"] = ttt
        if not args.no_extras and world == 1:
            # SURVEY 8d's other figures for this configuration, each with its own timed region
            Xg, Wg, z0 = r["Xg"], r["Wg"], r["z0"]
            el, _ = timed_steps(ranks, lambda: ista(Xg, z0, Wg, ALPHA, fast=False, lr=lr, maxiter=args.iters, tol=0.0),
                                args.steps, args.warmup)
            out["ista_iterations_per_sec"] = {"value": args.steps * args.iters / el, "fast": False,
                                              "ms_per_step": 1e3 * el / args.steps,
                                              "note": "plain ISTA (ista.py:84 z_prev = z): same kernel, momentum coefficient 0"}
            from lasso_amd.engine import HipEngine
            eng = HipEngine(dev)
            el, _ = timed_steps(ranks, lambda: eng.lipschitz(Wg), args.steps, args.warmup, events=False)
            out["lipschitz_ms"] = {"value": 1e3 * el / args.steps,
                                   "note": "lasso_lipschitz (ista.py:8-14) incl. its host synchronisation: the "
                                           "reference returns a python float the same way"}
            ista(Xg, z0, Wg, ALPHA, lr="auto", maxiter=2000, tol=1e-5)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _, info = ista(Xg, z0, Wg, ALPHA, lr="auto", maxiter=2000, tol=1e-5, return_info=True)
            torch.cuda.synchronize()
            out["time_to_tol_lr_auto"] = {"ms": 1e3 * (time.perf_counter() - t1), "iterations": info["iterations"],
                                          "tol": 1e-5, "note": "lr='auto': lambda_max on the stream inside the call "
                                          "(LASSO_LR_AUTO), Lipschitz time included"}
            long_iters = 10 * args.iters
            el, kms = timed_steps(ranks, lambda: ista(Xg, z0, Wg, ALPHA, fast=True, lr=lr, maxiter=long_iters, tol=0.0),
                                  10, 1)
            out["sustained"] = {"iterations_per_s": 10 * long_iters / el, "iters_per_launch": long_iters, "launches": 10,
                                "seconds": el, "tflops": 4.0 * r["rows"] * D * K * long_iters * 10 / el / 1e12,
                                "note": "the same kernel held busy ~10x longer than the headline region (clock under "
                                        "sustained load)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(r["X"], r["W"], lr)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
    # objective of the timed result (HIP lasso_loss) against the reference's known answer; the
    # whole 4096-row batch when it is sharded (strong): sums all-reduced
    if args.iters == 100 and n_rows == N_ROWS:
        from lasso_amd.engine import HipEngine
        q = results["strong"]
        _, sums = HipEngine(dev).objective_sums(q["Xg"], q["z"], q["Wg"], ALPHA)
        ranks.sum_(sums)
        obj = ((0.5 * sums[0] + ALPHA * sums[1]) / N_ROWS).item()
        if rank == 0:
            out["objective_after_100"] = obj
            out["objective_reference"] = OBJ_REF_100
            if abs(obj - OBJ_REF_100) > OBJ_RTOL * OBJ_REF_100:
                print(json.dumps(out))
                raise SystemExit("bench.py: objective after 100 iterations %.6f differs from the reference's "
                                 "%.6f by more than rtol %g -- the timed kernel is wrong" % (obj, OBJ_REF_100, OBJ_RTOL))
    return out


EM_SHAPES = {
    # name: (config, d, k, alpha, default rows)
    "c4": ("BASELINE config 4", D, K, ALPHA, N_EM),
    "c5": ("BASELINE config 5 shape (8x8 patches; synthetic centred patches stand in for Omniglot)", 64, 256, 0.1, N_EM),
}


def run_em(args, ranks):
    """BASELINE config 4 (or config 5's shape): one EM step = E-step (FISTA, reference defaults) + objective +
    Gram + all-reduce + atom sweep on the rows sharded over the ranks (dict_learning.py:23-53)."""
    import torch
    from lasso_amd.engine import HipEngine
    from lasso_amd import parallel
    from recipes import recipe_c4_init, recipe_c5
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    cfg, d, k, alpha, n_default = EM_SHAPES[args.shape]
    n_all = args.rows or n_default
    if n_all % world:
        raise SystemExit("bench.py: %d rows do not split over %d ranks" % (n_all, world))
    rows = n_all // world
    X_all = recipe(n_all)[0] if args.shape == "c4" else recipe_c5(n_all)
    Xg = X_all[rank * rows:(rank + 1) * rows].to(dev)
    eng = HipEngine(dev)
    state = {"D": recipe_c4_init(d, k).to(dev)}
    comm_ms = []
    real_all_reduce = parallel._all_reduce

    def timed_all_reduce(t, group):        # the collective of the M-step, timed on the device
        if ranks.dist is None or t.numel() < 1024:
            return real_all_reduce(t, group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_all_reduce(t, group)
        e1.record()
        comm_ms.append((e0, e1, t.numel() * t.element_size()))
        return t
    parallel._all_reduce = timed_all_reduce
    eng.em_stats = {}

    def em(steps):
        state["D"], state["loss"] = parallel.em_loop(eng, Xg, state["D"], alpha, constrained=True, steps=steps,
                                                     solver_kwargs=dict(algorithm="ista"))
    try:
        if args.warmup:
            em(args.warmup)
        ranks.sync()
        del comm_ms[:]
        eng.em_stats.clear()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        ev[0].record()
        em(args.steps)                     # exactly K EM steps between the two barriers
        ev[1].record()
        ranks.sync()
        elapsed = ranks.max(time.perf_counter() - t0)
    finally:
        parallel._all_reduce = real_all_reduce
    out = None
    if rank == 0:
        # messages per EM step: one ([A | B | tail]), or -- pipelined M-step (d = 256, k a multiple of 256) -- one per
        # STAGE of block rows, the first on the step's dependent chain, the others behind the running sweep
        per_step = len(comm_ms) // args.steps if comm_ms else 0
        msgs = comm_ms[-args.steps * per_step:] if per_step else []
        step_ms = sorted(sum(a.elapsed_time(b) for a, b, _ in msgs[i * per_step:(i + 1) * per_step])
                         for i in range(args.steps)) if per_step else []
        head_ms = sorted(msgs[i * per_step][0].elapsed_time(msgs[i * per_step][1]) for i in range(args.steps)) if per_step else []
        ar = step_ms
        # E-step flops dominate: 10 iterations x 4 n d k per rank, + Gram 2nk^2 + 2nkd + objective 2ndk
        flop = (10 * 4.0 + 2.0 + 2.0) * rows * d * k + 2.0 * rows * k * k
        ms = 1e3 * elapsed / args.steps
        dev_ms = ev[0].elapsed_time(ev[1]) / args.steps
        out = {
            "metric": "dict_learning_em_steps_per_sec (n=%d d=%d k=%d fp32, constrained, defaults)" % (n_all, d, k),
            "value": args.steps / elapsed, "unit": "EM steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: dict_learning EM step, n=%d d=%d k=%d alpha=%g, FISTA E-step "
                                   "(lr='auto', maxiter=10, tol=1e-5) + constrained M-step" % (cfg, n_all, d, k, alpha),
                       "rows_per_gpu": rows, "rows_total": n_all,
                       "parallelism": "row-sharded x%d, all-reduce of [A|B|sums] per step (one message per stage of the "
                                      "pipelined M-step)" % world},
            "all_reduce_ms": {"median": ar[len(ar) // 2] if ar else 0.0, "max": ar[-1] if ar else 0.0,
                              "median_first_message": head_ms[len(head_ms) // 2] if head_ms else 0.0,
                              "bytes": 4 * (k * k + k * d + 2 + 10),
                              "bytes_sent": sorted({nb for _, _, nb in comm_ms}),
                              "bytes_per_step": sum(nb for _, _, nb in msgs) // args.steps if msgs else 0,
                              "per_step": len(comm_ms) / float(args.steps),
                              "note": "device time of the RCCL all-reduces of one EM step, summed -- [A | B | objective "
                              "sums of the previous step | the E-step's 10 stop-rule sums]: ONE message, or with the "
                              "pipelined M-step one per stage of block rows (`per_step` of them, `bytes_per_step` in "
                              "all; only the first -- median_first_message -- sits on the step's dependent chain, the "
                              "others travel behind the running sweep) -- (0 at N=1: no collective)"},
            "roofline": with_traffic(
                {"bound": "mfma", "achieved": flop / (dev_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                 "unit": "TFLOP/s", "frac": flop / (dev_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                 "kernel": "whole EM step: %s (E-step; dominates) + the Gram product, the atom sweep and the Lipschitz "
                           "squarings (latency chains)" % nat_kernel_name(rows, d, k),
                 "flop_per_launch": flop, "per": "GPU (rank 0)", "avg_launch_ms": dev_ms,
                 "avg_launch_note": "HIP events around the K timed EM steps / K"},
                ("em_%s%s" % (args.shape, "" if n_all == n_default else "_shard"))
                if (world == 1 and n_all in (n_default, 8192)) else None, dev_ms,
                # per EM step (resident-state model): X read by the E-step, the objective and the Gram product (3x),
                # Z written once and read twice, the dictionary-sized operands (W, A, B, U: a few k^2 + k d words)
                algorithmic_bytes=4.0 * (3 * rows * d + 3 * rows * k + 4 * k * k + 6 * k * d)),
            "em_path": dict(eng.em_stats),
            "objective_last_step": float(state["loss"][-1]),
        }
    return out


C3_ROWS, C3_OUTER = 16384, 10
C3_TRIALS = [5, 3, 5, 4, 4, 4, 4, 3, 5, 5]          # the reference's fp32 trace (tests/golden/g3_c3_trace.npz)
C3_OBJ = {"f32": (64.142166, 1e-5), "bf16": (64.151779, 2e-3)}   # reference objective, rtol
PEAK_BF16_MFMA_TFLOPS = 2500.0                      # MI355X_MICROARCH.md: dense bf16 MFMA


def run_c3(args, ranks):
    """BASELINE config 3: FISTA with the backtracking line search (ista.py:17-54), n=16384 d=256 k=1024, lr0 = 1,
    10 outer iterations, bf16 (or fp32) tensors; a step = one solve."""
    import torch
    from lasso_amd.linear.solvers import ista
    from lasso_amd import parallel
    from lasso_amd.engine import HipEngine
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    n_all = args.rows or C3_ROWS
    if n_all % world:
        raise SystemExit("bench.py: %d rows do not split over %d ranks" % (n_all, world))
    rows = n_all // world
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    X_all, W = recipe(n_all)
    Xg, Wg = X_all[rank * rows:(rank + 1) * rows].to(dev).to(dt), W.to(dev).to(dt)
    z0 = torch.zeros(rows, K, device=dev, dtype=dt)
    eng = HipEngine(dev)
    kw = dict(lr=1.0, maxiter=C3_OUTER, tol=0.0, backtrack=True)

    def solve(info=False):
        if not ranks.sharded:
            return ista(Xg, z0, Wg, ALPHA, return_info=info, **kw)
        return parallel.sharded_encode(eng, Xg, Wg, ALPHA, z0, n_global=n_all, return_info=info, **kw)
    elapsed, kern_ms = timed_steps(ranks, solve, args.steps, args.warmup)
    z, info = solve(True)
    _, sums = eng.objective_sums(Xg.float(), z.float(), Wg.float(), ALPHA)
    ranks.sum_(sums)
    obj = ((0.5 * sums[0] + ALPHA * sums[1]) / n_all).item()
    if rank != 0:
        return None
    trials = list(info["trials"])
    flop = (4.0 * len(trials) + 2.0 * sum(trials)) * rows * D * K          # as executed, this rank
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    achieved = flop / (kern_ms[0] * 1e-3) / 1e12
    from lasso_amd import _native as nat
    name = nat.lib().lasso_fista_kernel_name(rows, D, K, nat.LASSO_BF16 if args.dtype == "bf16" else nat.LASSO_F32, 1)
    out = {
        "metric": "fista_backtracking_outer_iterations_per_sec (n=%d d=256 k=1024 %s, lr0=1, eta=1.5)" % (n_all, args.dtype),
        "value": args.steps * len(trials) / elapsed, "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE config 3: FISTA with backtracking line search, n=%d d=256 k=1024 %s tensors, "
                               "lr0=1.0, %d outer iterations; step = one solve" % (n_all, args.dtype, C3_OUTER) +
                               ("" if n_all == C3_ROWS else " (NOT the 16384-row batch of config 3)"),
                   "rows_per_gpu": rows, "rows_total": n_all,
                   "parallelism": "row-sharded x%d%s" % (world, ", every F<=Q decision on all-reduced sums" if ranks.sharded else "")},
        "roofline": with_traffic(
            {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
             "kernel": name.decode() if name else None,
             "flop_per_launch": flop, "per": "GPU (rank 0)", "avg_launch_ms": kern_ms[0],
             "flop_note": "(4 per outer iteration + 2 per trial) n d k as executed: %d outer iterations, "
                          "%d trials" % (len(trials), sum(trials)),
             "avg_launch_note": "HIP events around the K timed solves / K"},
            ("c3_%s" % args.dtype) if (world == 1 and n_all == C3_ROWS) else None, kern_ms[0],
            # resident-state model: x, W, z0 read, z written, once per solve
            algorithmic_bytes=(2.0 if args.dtype == "bf16" else 4.0) * (rows * D + D * K + 2 * rows * K)),
        "trials": trials, "reference_trials_fp32": C3_TRIALS,
        "objective": obj, "objective_reference": C3_OBJ[args.dtype][0],
    }
    if n_all == C3_ROWS:
        ref, rtol = C3_OBJ[args.dtype]
        bad = abs(obj - ref) > rtol * ref or (args.dtype == "f32" and trials != C3_TRIALS)
        if bad:
            print(json.dumps(out))
            raise SystemExit("bench.py: config 3 (%s) ended at objective %.6f, trials %s -- reference %.6f (rtol %g), "
                             "fp32 trace %s: the timed solve is wrong" % (args.dtype, obj, trials, ref, rtol, C3_TRIALS))
    return out


CONV_CASES = {   # tools/bench_conv.py's geometries: N, C, K, kernel size, padding, code height = width
    "gray": (256, 1, 64, 7, 0, 26), "rgb": (64, 3, 128, 5, 2, 64), "c16": (32, 16, 256, 3, 1, 64)}
CONV_ITERS = 20


def run_conv(args, ranks):
    """convolutional FISTA (lasso/conv2d/ista.py:7-49) on one of tools/bench_conv.py's geometries; a
    step = one solve of 20 iterations with a fixed step size and no stop rule.  The batch of images is split over the
    ranks (images are independent: no collective).  Roofline: HBM -- the code z and its momentum copy y, [N, K, Hz, Wz]
    fp32 each, are streamed every iteration (y read by the synthesis and by the gradient step, z read, both written):
    9.8 flop per byte at 1 x 7 x 7 taps and 64 atoms, below the fp32-MFMA ridge."""
    import torch
    from lasso_amd.conv2d import ista_conv2d
    from lasso_amd import _native as nat
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    N_all, C, Kc, ks, pd, Hz = CONV_CASES[args.conv_case]
    if args.rows:
        N_all = args.rows
    if N_all % world:
        raise SystemExit("bench.py: %d images do not split over %d ranks" % (N_all, world))
    N = N_all // world
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Kc, C, ks, ks, generator=g) / ks
    H = (Hz - 1) - 2 * pd + ks
    x = torch.randn(N_all, C, H, H, generator=g)[rank * N:(rank + 1) * N]
    lr = 0.5 / w.pow(2).sum().item()
    xg, wg, zg = x.to(dev), w.to(dev), torch.zeros(N, Kc, Hz, Hz, device=dev)

    def solve():
        return ista_conv2d(xg, zg, wg, 0.1, stride=1, padding=pd, maxiter=CONV_ITERS, lr=lr, tol=0.0)
    elapsed, kern_ms = timed_steps(ranks, solve, args.steps, args.warmup)
    z = solve()
    nnz = torch.tensor([float((z != 0).sum())], device=dev, dtype=torch.float64)
    ranks.sum_(nnz)
    if rank != 0:
        return None
    code_bytes = 4.0 * N * Kc * Hz * Hz
    # per iteration: y read twice (synthesis, gradient step), z read, z and y written, x read; per solve: z0 -> rows
    # (read, z and y written) and rows -> z (read, written)
    alg = CONV_ITERS * (5.0 * code_bytes + 4.0 * N * C * H * H) + 5.0 * code_bytes
    flop = CONV_ITERS * 4.0 * N * Hz * Hz * C * ks * ks * Kc
    gbps = alg / (kern_ms[0] * 1e-3) / 1e9
    name = nat.lib().lasso_conv_ista_kernel_name(N, C, H, H, Kc, Hz, Hz, ks, ks, 1, 1, pd, pd)
    return {
        "metric": "conv_fista_iterations_per_sec (N=%d %dx%dx%d images, %d %dx%d atoms, fp32, fixed step)" % (N_all, C, H, H, Kc, ks, ks),
        "value": args.steps * CONV_ITERS / elapsed, "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "convolutional FISTA (ista_conv2d), N=%d images %dx%dx%d, %d atoms %dx%d, padding %d, code "
                               "%dx%d, alpha=0.1, %d iterations per solve, tol=0; step = one solve"
                               % (N_all, C, H, H, Kc, ks, ks, pd, Hz, Hz, CONV_ITERS),
                   "images_per_gpu": N, "images_total": N_all, "parallelism": "images sharded x%d (no collective)" % world},
        "roofline": with_traffic(
            {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
             "kernel": name.decode() if name else None,
             "bytes_per_launch": alg, "per": "GPU (rank 0)", "avg_launch_ms": kern_ms[0],
             "bytes_note": "algorithmic: per iteration y read twice, z read, z and y written (4 N K Hz Wz bytes each) + x; "
                           "per solve the two layout changes (5 code-sized passes)",
             "avg_launch_note": "HIP events around the K timed solves / K",
             "tflops": flop / (kern_ms[0] * 1e-3) / 1e12, "mfma_frac": flop / (kern_ms[0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            ("conv_%s" % args.conv_case) if (world == 1 and not args.rows) else None, kern_ms[0], algorithmic_bytes=alg),
        "nonzeros": int(nnz.item()),
    }


CD_STEPS = 1000


def run_cd(args, ranks):
    """greedy coordinate descent (lasso/linear/solvers/coordinate_descent.py:5-54) at BASELINE config
    2's shape; a step = one solve of 1000 coordinate steps per row (tol = 1e-6 k: no row of this problem converges
    earlier).  Rows are independent: sharded over the ranks, no collective.  Roofline: neither MFMA nor HBM -- every
    row-step reads one row of S = I - W^T W (4 K bytes) out of L2 / MALL (S is 4 MiB) behind a dependent chain
    argmax -> address -> load -> update: the block reports achieved L2 bytes against the L2 peak of
    MI355X_MICROARCH.md (34.5 TB/s) and says so in `bound`."""
    import torch
    from lasso_amd.linear.solvers import coord_descent
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    n_all = args.rows or N_ROWS
    if n_all % world:
        raise SystemExit("bench.py: %d rows do not split over %d ranks" % (n_all, world))
    rows = n_all // world
    X, W = recipe(n_all)
    Xg, Wg = X[rank * rows:(rank + 1) * rows].to(dev), W.to(dev)

    def solve():
        return coord_descent(Xg, Wg, None, ALPHA, maxiter=CD_STEPS)
    elapsed, kern_ms = timed_steps(ranks, solve, args.steps, args.warmup)
    z, info = coord_descent(Xg, Wg, None, ALPHA, maxiter=CD_STEPS, return_info=True)
    # the objective of the returned codes, all-reduced (dict_learning.py:10-13 on all rows)
    from lasso_amd.engine import HipEngine
    _, sums = HipEngine(dev).objective_sums(Xg, z, Wg, ALPHA)
    ranks.sum_(sums)
    if rank != 0:
        return None
    row_steps = float(rows) * CD_STEPS                     # per rank and solve
    l2_bytes = row_steps * 4.0 * K
    gbps = l2_bytes / (kern_ms[0] * 1e-3) / 1e9
    out = {
        "metric": "coord_descent_row_steps_per_sec (n=%d d=%d k=%d fp32, %d steps per row)" % (n_all, D, K, CD_STEPS),
        "value": args.steps * float(n_all) * CD_STEPS / elapsed, "unit": "row-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "greedy coordinate descent (algorithm='cd') at BASELINE config 2's shape: n=%d d=%d k=%d "
                               "alpha=%g, maxiter=%d, tol=1e-6 (no row stops earlier); step = one solve"
                               % (n_all, D, K, ALPHA, CD_STEPS),
                   "rows_per_gpu": rows, "rows_total": n_all, "parallelism": "rows sharded x%d (no collective)" % world},
        "roofline": with_traffic(
            {"bound": "l2", "achieved": gbps, "peak": 34500.0, "unit": "GB/s", "frac": gbps / 34500.0,
             "kernel": "lasso::cd_rows_kernel<%d>" % (K // 256),
             "bytes_per_launch": l2_bytes, "per": "GPU (rank 0)", "avg_launch_ms": kern_ms[0],
             "bytes_note": "algorithmic: one row of S = I - W^T W (4 k bytes) per row-step, served by L2 / MALL (S is 4 MiB); "
                           "peak = the aggregate L2 bandwidth of MI355X_MICROARCH.md.  The step is a dependent chain "
                           "(argmax -> address -> load -> k multiply-adds): latency-bound at the occupancy n allows",
             "avg_launch_note": "HIP events around the K timed solves / K (set-up GEMMs b = xW, S = I - W^T W included)"},
            "cd" if (world == 1 and not args.rows) else None, kern_ms[0], algorithmic_bytes=4.0 * (rows * D + D * K + rows * K)),
        "max_steps": info["max_steps"], "n_active": info["n_active"],
        "objective": float((0.5 * sums[0] + ALPHA * sums[1]) / n_all),
    }
    if not args.no_cpu_baseline and world == 1:
        from oracle import lasso_oracle as orc
        sample = 256
        orc.coordinate_descent(X[:8], W, None, ALPHA, maxiter=10)
        t0 = time.perf_counter()
        orc.coordinate_descent(X[:sample], W, None, ALPHA, maxiter=CD_STEPS)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": sample * CD_STEPS / dt, "unit": "row-steps/s", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "%d rows x %d steps of oracle/lasso_oracle.py coordinate_descent, "
                               "%.1f s, os.cpu_count()=%s" % (sample, CD_STEPS, dt, os.cpu_count())}
    return out


def run_launcher_selftest(args, ranks):
    """No compute: the launcher / barrier / max-over-ranks protocol only (CPU, gloo)."""
    def step():
        time.sleep(0.002 * (ranks.rank + 1))
    elapsed, _ = timed_steps(ranks, step, args.steps, args.warmup, events=False)
    if ranks.rank != 0:
        return None
    return {"metric": "launcher_selftest", "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": ranks.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
            "config": {"workload": "launcher self-test (no kernel)"}}


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # 0.32 s timed at the headline shape
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--iters", type=int, default=100, help="FISTA iterations per step (solve)")
    ap.add_argument("--workload", choices=["fista", "em", "c3", "conv", "cd", "launcher-selftest"], default="fista")
    ap.add_argument("--conv-case", choices=sorted(CONV_CASES), default="gray",
                    help="conv workload: N=256 1x32x32 images with 64 7x7 atoms (gray), N=64 3x64x64 with 128 5x5 (rgb), "
                         "N=32 16x64x64 with 256 3x3 (c16)")
    ap.add_argument("--shape", choices=sorted(EM_SHAPES), default="c4", help="em workload: config 4 or config 5's shape")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16", help="c3 workload: tensor dtype")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collectives: nccl (= RCCL, the product) or gloo (host-staged; tests on a one-GPU box)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 only: create a ONE-rank process group (RCCL with --backend nccl) and run the multi-rank "
                         "code paths on it -- every collective of an N > 1 run executes on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank on cuda:0 (needs --backend gloo): executes the multi-rank code paths on ONE GPU")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="which figure is `value` at N > 1 (both are measured and reported)")
    ap.add_argument("--rows", type=int, default=None,
                    help="total rows of the batch (default: fista 4096 = BASELINE config 2 -- e.g. 512 = the shard one "
                         "rank of an 8-GPU strong-scaling run works on --, em 65536, c3 16384)")
    ap.add_argument("--no-shards", action="store_true", help="skip the per-shard timings of the N=1 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ISTA / Lipschitz / lr='auto' / sustained legs")
    ap.add_argument("--no-time-to-tol", action="store_true")
    return ap


def main():
    args = parser().parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL wants one GPU per rank)")
    if args.force_dist and args.gpus != 1:
        raise SystemExit("bench.py: --force-dist is for --gpus 1 (N > 1 runs its collectives anyway)")
    if args.workload == "fista" and args.rows is None:
        args.rows = N_ROWS

    selftest = args.workload == "launcher-selftest"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not selftest:
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus and not (args.share_gpu and have >= 1):
                raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible -- one process per GPU, "
                                 "ranks cannot share a device" % (args.gpus, have))
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank, local_rank, world = world_from_env(args)
    import torch
    if selftest:
        ranks = Ranks(rank, world, torch.device("cpu"), "gloo")
        out = run_launcher_selftest(args, ranks)
    else:
        gpu = 0 if args.share_gpu else local_rank
        if torch.cuda.device_count() <= gpu:
            raise SystemExit("bench.py: rank %d has no GPU (visible: %d)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(gpu)
        ranks = Ranks(rank, world, torch.device("cuda", gpu), args.backend, force=args.force_dist)
        out = {"em": run_em, "c3": run_c3, "fista": run_fista, "conv": run_conv, "cd": run_cd}[args.workload](args, ranks)
        if out is not None and ranks.sharded:
            out["backend"] = args.backend + (" (all ranks share cuda:0: a code-path run, not a performance figure)"
                                             if args.share_gpu else
                                             " (ONE-rank process group, --force-dist: the N > 1 code paths and their "
                                             "collectives on one GPU -- a code-path run, not a scaling figure)"
                                             if world == 1 else " (RCCL over xGMI)" if args.backend == "nccl" else "")
    ranks.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
