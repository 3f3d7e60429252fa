#!/usr/bin/env python3
"""Time the pieces of one EM step at BASELINE config 4's per-GPU shard (n=8192) and at the
full n=65536 on one GPU: E-step (10 FISTA its, lr given), objective, Gram, sweep, Lipschitz."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-lasso_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from recipes import recipe_xw, recipe_c4_init, LAMBDA_MAX_C4
if '--lib' in sys.argv:                       # an A/B build (tools/build_variant.sh)
    from lasso_amd import _native as _nat
    _nat.use_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1]))
from lasso_amd.engine import HipEngine
from lasso_amd.linear import sparse_encode, dict_learning
from lasso_amd.parallel import constrained_mstep

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

def _rank_worker(rank, world, port, rows, steps, tmp):
    """--ranks R: one of R processes sharing THIS GPU, collectives over gloo (RCCL wants one GPU per rank; the
    driver's multi-GPU tier is the only place that can run it).  Each rank holds `rows` rows of the recipe."""
    import torch.distributed as dist
    from lasso_amd.parallel import dict_learning_sharded, _all_reduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, _ = recipe_xw(rows * world)
    Xs = X[rank * rows:(rank + 1) * rows].cuda()
    D0 = recipe_c4_init()
    eng = HipEngine()
    eng.em_stats = {}
    kw = dict(alpha=0.5, algorithm='ista', progbar=False, init_weight=D0, engine=eng)
    dict_learning_sharded(Xs, 1024, steps=2, **kw)
    torch.cuda.synchronize(); dist.barrier()
    eng.em_stats.clear()
    t = time.perf_counter()
    D, losses = dict_learning_sharded(Xs, 1024, steps=steps, **kw)
    torch.cuda.synchronize(); dist.barrier()
    step_ms = (time.perf_counter() - t) / steps * 1e3
    buf = torch.zeros(1024 * 1024 + 1024 * 256 + 12, device='cuda')
    _all_reduce(buf, None); torch.cuda.synchronize(); dist.barrier()
    t = time.perf_counter()
    for _ in range(10):
        _all_reduce(buf, None)
    torch.cuda.synchronize(); dist.barrier()
    ar_ms = (time.perf_counter() - t) / 10 * 1e3
    torch.save({"step_ms": step_ms, "allreduce_ms": ar_ms, "stats": dict(eng.em_stats), "D": D.cpu(),
                "losses": losses.cpu()}, os.path.join(tmp, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__' and '--ranks' in sys.argv:
    import tempfile
    import torch.multiprocessing as mp
    world = int(sys.argv[sys.argv.index('--ranks') + 1])
    rows = int(sys.argv[sys.argv.index('--rows') + 1]) if '--rows' in sys.argv else 4096
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 30
    # the yardstick: ONE process, the same rows, the GPU to itself
    X, _ = recipe_xw(rows * world)
    D0 = recipe_c4_init().cuda()
    Xs = X[:rows].cuda()
    dict_learning(Xs, 1024, alpha=0.5, steps=2, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    dict_learning(Xs, 1024, alpha=0.5, steps=steps, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t) / steps * 1e3
    Xall = X.cuda()
    dict_learning(Xall, 1024, alpha=0.5, steps=2, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    Dref, lref = dict_learning(Xall, 1024, alpha=0.5, steps=steps, algorithm='ista', progbar=False, device='cuda',
                               init_weight=D0)
    torch.cuda.synchronize()
    whole_ms = (time.perf_counter() - t) / steps * 1e3
    Dref, lref = Dref.cpu(), lref.cpu()
    del Xs, Xall
    with tempfile.TemporaryDirectory() as tmp:
        port = 36500 + os.getpid() % 2000
        mp.start_processes(_rank_worker, args=(world, port, rows, steps, tmp), nprocs=world, join=True,
                           start_method="spawn")
        res = [torch.load(os.path.join(tmp, "rank%d.pt" % r)) for r in range(world)]
    step_ms = max(r["step_ms"] for r in res)
    ar_ms = max(r["allreduce_ms"] for r in res)
    out = {"ranks": world, "rows_per_rank": rows, "steps": steps, "backend": "gloo (ranks share one GPU)",
           "em_step_ms": step_ms, "single_process_step_ms(rows_per_rank)": single_ms,
           "single_process_step_ms(all rows)": whole_ms, "staged_allreduce_ms(5 MiB)": ar_ms,
           "yardstick_ms(ranks x single + allreduce)": world * single_ms + ar_ms,
           "ratio_to_yardstick": step_ms / (world * single_ms + ar_ms),
           "overlapped_steps": res[0]["stats"].get("overlapped_steps", 0),
           "replayed_steps": res[0]["stats"].get("replayed_steps", 0),
           "D_bitwise_equal_across_ranks": all(torch.equal(res[0]["D"], r["D"]) for r in res[1:]),
           "max|D - single process D|": float((res[0]["D"] - Dref).abs().max()),
           "max|loss - single process loss|": float((res[0]["losses"] - lref).abs().max())}
    print(json.dumps(out))
    sys.exit(0)

def single_process():
    eng = HipEngine()
    D0 = recipe_c4_init().cuda()
    NS = (int(sys.argv[sys.argv.index('--n') + 1]),) if '--n' in sys.argv else (8192, 65536)
    for n in NS:
        X, _ = recipe_xw(n)
        X = X.cuda()
        lr = 1.0 / LAMBDA_MAX_C4
        Z = sparse_encode(X, D0, 0.5, lr=lr, maxiter=10, tol=0.0)
        buf = torch.empty(1024 * 1024 + 1024 * 256, device='cuda')
        out = {"n": n}
        out["estep_10it_ms"] = timeit(lambda: sparse_encode(X, D0, 0.5, lr=lr, maxiter=10, tol=0.0))
        out["estep_default_ms(lr=auto,tol=1e-5)"] = timeit(lambda: sparse_encode(X, D0, 0.5))
        out["lipschitz_ms"] = timeit(lambda: eng.lipschitz(D0))
        out["objective_ms"] = timeit(lambda: eng.objective_sums(X, Z, D0, 0.5))
        out["gram_ms"] = timeit(lambda: eng.gram(Z, X, buf))
        A, B = eng.gram(Z, X, buf)
        D = D0.clone()
        out["sweep_ms"] = timeit(lambda: constrained_mstep(eng, A, B, D))
        out["ridge_ms"] = timeit(lambda: eng.ridge(A, B, 1e-2 * n))
        for steps in (10, 40):          # the first call also pays the one-off allocations (workspaces, pinned words)
            dict_learning(X, 1024, alpha=0.5, steps=2, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
            torch.cuda.synchronize()
            t = time.perf_counter()
            dict_learning(X, 1024, alpha=0.5, steps=steps, algorithm='ista', progbar=False, device='cuda', init_weight=D0)
            torch.cuda.synchronize()
            out["em_step_ms(constrained, %d steps avg)" % steps] = (time.perf_counter() - t) / steps * 1e3
        print(json.dumps(out))


if __name__ == '__main__':
    single_process()
