"""Two ranks on the HIP engine: the batch-sharded EM driver (lasso_amd.parallel) with
world_size = 2, both ranks sharing the one GPU of the test box, collectives over gloo
(RCCL needs one GPU per rank; the driver runs the 8-GPU job).  Ragged shards; constrained
and ridge M-steps; fixed iteration count and the global stop rule (tol > 0)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from margins import record_margins

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K, SPLIT = 1100, 96, 320, 413
CASES = {"bcd": dict(constrained=True, lr=0.08, maxiter=12, tol=0.0),
         "ridge": dict(constrained=False, lr=0.08, maxiter=12, tol=0.0),
         "tol": dict(constrained=True, lr=0.08, maxiter=200, tol=2e-3),
         "auto": dict(constrained=True, maxiter=10),                      # lr='auto', default tol
         "tol_short": dict(constrained=True, lr=0.08, maxiter=40, tol=5e-3),   # the rule fires at ~19 of 40: replay
         "persist": dict(constrained=True, persist=True, lr=0.08, maxiter=8, tol=0.0)}


def _problem():
    g = torch.Generator().manual_seed(21)
    X = torch.randn(N, D, generator=g)
    D0 = torch.nn.functional.normalize(torch.randn(D, K, generator=g), dim=0)
    return X, D0


def _worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd.parallel import dict_learning_sharded
    from lasso_amd.engine import HipEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, D0 = _problem()
    lo, hi = (0, SPLIT) if rank == 0 else (SPLIT, N)           # ragged shards on purpose
    out = {}
    for tag, kw in CASES.items():
        kw = dict(kw)
        torch.manual_seed(1)
        eng = HipEngine()
        eng.em_stats = {}
        Dl, losses = dict_learning_sharded(X[lo:hi], K, alpha=0.3, steps=4, init_weight=D0,
                                           engine=eng, **kw)
        out[tag + "_D"], out[tag + "_l"] = Dl.cpu().numpy(), losses.cpu().numpy()
        out[tag + "_stats"] = np.array([eng.em_stats.get("overlapped_steps", 0), eng.em_stats.get("replayed_steps", 0)])
    # a rank WITHOUT rows (rank 1) takes the same path and issues the same collectives as its peer: the asynchronous
    # form ("auto"), its replay ("tol_short") and the synchronous chunked form ("tol")
    lo, hi = (0, N) if rank == 0 else (N, N)
    for tag in ("auto", "tol_short", "tol"):
        torch.manual_seed(1)
        eng = HipEngine()
        eng.em_stats = {}
        Dl, losses = dict_learning_sharded(X[lo:hi], K, alpha=0.3, steps=3, init_weight=D0, engine=eng, **CASES[tag])
        out["empty_" + tag + "_D"], out["empty_" + tag + "_l"] = Dl.cpu().numpy(), losses.cpu().numpy()
        out["empty_" + tag + "_stats"] = np.array([eng.em_stats.get("overlapped_steps", 0),
                                                   eng.em_stats.get("replayed_steps", 0)])
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_engine(tmp_path):
    from lasso_amd.linear import dict_learning
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    port = 31500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    X, D0 = _problem()
    # the multi-rank step takes the no-host-wait path (E-step enqueued with lr='auto' on the stream, stop-rule sums
    # in the tail of the ONE all-reduce, verdict on the device) whenever maxiter <= 64; "tol" (maxiter = 200) is the
    # synchronous chunked path, "tol_short" stops early in every step and replays
    for tag in CASES:
        assert np.array_equal(r0[tag + "_stats"], r1[tag + "_stats"]), tag
        assert (int(r0[tag + "_stats"][0]) >= 4) == (tag != "tol"), (tag, r0[tag + "_stats"])
    assert int(r0["tol_short_stats"][1]) >= 1 and int(r0["auto_stats"][1]) == 0
    margins = {}
    for tag, kw in CASES.items():
        # both ranks hold the same replicated dictionary and the same global objective, bit for bit
        assert np.array_equal(r0[tag + "_D"], r1[tag + "_D"]), tag
        assert np.array_equal(r0[tag + "_l"], r1[tag + "_l"]), tag
        torch.manual_seed(1)
        Dref, lref = dict_learning(X.cuda(), K, alpha=0.3, steps=4, init_weight=D0, progbar=False,
                                   device="cuda", **kw)
        assert np.abs(r0[tag + "_l"] - lref.cpu().numpy()).max() <= 1e-4, tag
        assert np.abs(r0[tag + "_D"] - Dref.cpu().numpy()).max() <= 1e-4, tag
        # ... and directly against the oracle (the reference's arithmetic on the whole batch, CPU)
        torch.manual_seed(1)
        Do, lo_ = orc.dict_learning(X, K, alpha=0.3, steps=4, init_weight=D0, **kw)
        margins[tag] = (float(np.abs(r0[tag + "_l"] - lo_.numpy()).max()), float(np.abs(r0[tag + "_D"] - Do.numpy()).max()))
        assert margins[tag][0] <= 1e-5 and margins[tag][1] <= 2e-5, (tag, margins[tag])   # measured: <= 3e-6 / 4e-6 (profiles/r04)
        if "empty_" + tag + "_D" in r0.files:      # rank 1 without rows: the whole batch on rank 0, same collectives
            assert np.array_equal(r0["empty_" + tag + "_D"], r1["empty_" + tag + "_D"]), tag
            assert np.array_equal(r0["empty_" + tag + "_l"], r1["empty_" + tag + "_l"]), tag
            assert np.array_equal(r0["empty_" + tag + "_stats"], r1["empty_" + tag + "_stats"]), tag
            assert (int(r0["empty_" + tag + "_stats"][0]) >= 3) == (tag != "tol"), tag
            assert np.abs(r0["empty_" + tag + "_l"] - lo_.numpy()[:3]).max() <= 1e-4, tag
    record_margins("two_ranks_vs_oracle", {t: {"max_dloss": a, "max_dD": b} for t, (a, b) in margins.items()})


# ---- the PIPELINED M-step on two ranks (d = 256, k a multiple of 256: em_loop's two-stream form) ------------------
PN, PD, PK, PSPLIT = 900, 256, 512, 333
PCASES = {"auto": dict(maxiter=10),                                        # lr='auto', default tol: one message per stage
          "tol_short": dict(lr=0.05, maxiter=40, tol=2e-2),                 # the rule fires early: speculated step discarded, replay
          "persist": dict(persist=True, lr=0.05, maxiter=8, tol=1e-7)}


# ---- ... and the DOUBLE-BUFFERED loop of a small dictionary (d <= 64, k <= 256: the one-workgroup sweep) ------------
SN, SD, SK, SSPLIT = 900, 64, 256, 333


def _pipe_problem(shape=None):
    n, d, k, _ = shape or (PN, PD, PK, PSPLIT)
    g = torch.Generator().manual_seed(77)
    X = torch.randn(n, d, generator=g)
    D0 = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    return X, D0


def _pipe_worker(rank, world, port, tmp, shape=None, stat="pipelined_steps", name="pipe", form=None):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd import parallel
    from lasso_amd.engine import HipEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, D0 = _pipe_problem(shape)
    PN, PD, PK, PSPLIT = shape or (globals()["PN"], globals()["PD"], globals()["PK"], globals()["PSPLIT"])
    if shape is None or form:
        os.environ["LASSO_EM_FORM"] = form or "pipeline"   # (by itself the loop picks its form from the rows per rank)
    out = {}
    sent = []
    real = parallel._all_reduce

    def counting(t, group):
        sent.append(t.numel())
        return real(t, group)
    parallel._all_reduce = counting
    for empty in (False, True):
        lo, hi = ((0, PSPLIT) if rank == 0 else (PSPLIT, PN)) if not empty else ((0, PN) if rank == 0 else (PN, PN))
        for tag, kw in PCASES.items():
            if empty and tag != "auto":
                continue
            torch.manual_seed(1)
            eng = HipEngine()
            eng.em_stats = {}
            del sent[:]
            Dl, losses = parallel.dict_learning_sharded(X[lo:hi], PK, alpha=0.3, steps=4, init_weight=D0, engine=eng, **kw)
            key = ("empty_" if empty else "") + tag
            out[key + "_D"], out[key + "_l"] = Dl.cpu().numpy(), losses.cpu().numpy()
            out[key + "_stats"] = np.array([eng.em_stats.get(stat, 0), eng.em_stats.get("replayed_steps", 0)])
            out[key + "_sent"] = np.array(sent)
    np.savez(os.path.join(tmp, "%s%d.npz" % (name, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_mstep_on_two_ranks(tmp_path):
    """d = 256, k = 512: em_loop takes its two-stream form -- [A | B] all-reduced per STAGE of block rows (head on the
    step's chain, the rest behind the running sweep), the objective's sums riding in the next step's message.  Ragged
    shards, a rank without rows, a stop rule that fires early (replay) and persist=True: both ranks bit for bit, and the
    reference's arithmetic on the whole batch (oracle) within the two-rank margins of the plain form."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    port = 33500 + (os.getpid() % 2000)
    mp.start_processes(_pipe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "pipe0.npz"), np.load(tmp_path / "pipe1.npz")
    X, D0 = _pipe_problem()
    stage_rows = 256 * (PK + PD)
    margins = {}
    for key in [k[:-2] for k in r0.files if k.endswith("_D")]:
        tag = key.replace("empty_", "")
        assert np.array_equal(r0[key + "_D"], r1[key + "_D"]) and np.array_equal(r0[key + "_l"], r1[key + "_l"]), key
        assert np.array_equal(r0[key + "_stats"], r1[key + "_stats"]) and np.array_equal(r0[key + "_sent"], r1[key + "_sent"]), key
        assert int(r0[key + "_stats"][0]) >= 4, (key, r0[key + "_stats"])                  # every step on the pipelined form
        assert (int(r0[key + "_stats"][1]) >= 1) == (tag == "tol_short"), (key, r0[key + "_stats"])
        big = [int(v) for v in r0[key + "_sent"] if v >= 1024]
        # every large message is a stage of [A | B]: 256 rows, the last one with the tail (2 sums + maxiter stop-rule sums)
        assert set(big) <= {stage_rows, stage_rows + 2 + PCASES[tag].get("maxiter", 10)}, (key, sorted(set(big)))
        torch.manual_seed(1)
        Do, lo_ = orc.dict_learning(X, PK, alpha=0.3, steps=4, init_weight=D0, **PCASES[tag])
        margins[key] = (float(np.abs(r0[key + "_l"] - lo_.numpy()).max()), float(np.abs(r0[key + "_D"] - Do.numpy()).max()))
        assert margins[key][0] <= 1e-5 and margins[key][1] <= 2e-5, (key, margins[key])
    record_margins("two_ranks_pipelined_vs_oracle", {t: {"max_dloss": a, "max_dD": b} for t, (a, b) in margins.items()})


def test_double_buffered_em_loop_on_two_ranks(tmp_path):
    """d = 64, k = 256 on two ranks: em_loop's two-stream form with the double-buffered dictionary -- ONE message per step
    ([A | B | the previous step's objective sums | this step's stop-rule sums]), the sweep enqueued before the host wait,
    the objective on the side stream after it.  Ragged shards, a rank without rows, a stop rule that fires early (the
    speculated sweep is dropped, replay) and persist=True: both ranks bit for bit, the oracle within the two-rank margins."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    port = 35500 + (os.getpid() % 2000)
    shape = (SN, SD, SK, SSPLIT)
    mp.start_processes(_pipe_worker, args=(2, port, str(tmp_path), shape, "speculative_sweeps", "small"), nprocs=2, join=True,
                       start_method="spawn")
    r0, r1 = np.load(tmp_path / "small0.npz"), np.load(tmp_path / "small1.npz")
    X, D0 = _pipe_problem(shape)
    margins = {}
    for key in [k[:-2] for k in r0.files if k.endswith("_D")]:
        tag = key.replace("empty_", "")
        assert np.array_equal(r0[key + "_D"], r1[key + "_D"]) and np.array_equal(r0[key + "_l"], r1[key + "_l"]), key
        assert np.array_equal(r0[key + "_stats"], r1[key + "_stats"]) and np.array_equal(r0[key + "_sent"], r1[key + "_sent"]), key
        assert int(r0[key + "_stats"][0]) >= 4, (key, r0[key + "_stats"])                  # every step on the speculative form
        assert (int(r0[key + "_stats"][1]) >= 1) == (tag == "tol_short"), (key, r0[key + "_stats"])
        big = [int(v) for v in r0[key + "_sent"] if v >= 1024]
        assert set(big) == {SK * (SK + SD) + 2 + PCASES[tag].get("maxiter", 10)}, (key, sorted(set(big)))
        torch.manual_seed(1)
        Do, lo_ = orc.dict_learning(X, SK, alpha=0.3, steps=4, init_weight=D0, **PCASES[tag])
        margins[key] = (float(np.abs(r0[key + "_l"] - lo_.numpy()).max()), float(np.abs(r0[key + "_D"] - Do.numpy()).max()))
        assert margins[key][0] <= 1e-5 and margins[key][1] <= 2e-5, (key, margins[key])
    record_margins("two_ranks_double_buffered_vs_oracle", {t: {"max_dloss": a, "max_dD": b} for t, (a, b) in margins.items()})


def test_double_buffered_em_loop_of_a_large_dictionary_on_two_ranks(tmp_path):
    """d = 256, k = 512 on two ranks in the form em_loop takes beyond 8192 rows per rank (asked for here): double-buffered
    dictionary, the co-operating sweep written by its transposing launch into the other buffer, the objective on the side
    stream held back until the sweep starts, ONE message per step.  Both ranks bit for bit, the oracle within the bars."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    port = 37500 + (os.getpid() % 2000)
    shape = (PN, PD, PK, PSPLIT)
    mp.start_processes(_pipe_worker, args=(2, port, str(tmp_path), shape, "speculative_sweeps", "large", "double-buffer"),
                       nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "large0.npz"), np.load(tmp_path / "large1.npz")
    X, D0 = _pipe_problem(shape)
    for key in [k[:-2] for k in r0.files if k.endswith("_D")]:
        tag = key.replace("empty_", "")
        assert np.array_equal(r0[key + "_D"], r1[key + "_D"]) and np.array_equal(r0[key + "_l"], r1[key + "_l"]), key
        assert np.array_equal(r0[key + "_stats"], r1[key + "_stats"]) and np.array_equal(r0[key + "_sent"], r1[key + "_sent"]), key
        assert int(r0[key + "_stats"][0]) >= 4, (key, r0[key + "_stats"])
        assert (int(r0[key + "_stats"][1]) >= 1) == (tag == "tol_short"), (key, r0[key + "_stats"])
        big = [int(v) for v in r0[key + "_sent"] if v >= 1024]
        assert set(big) == {PK * (PK + PD) + 2 + PCASES[tag].get("maxiter", 10)}, (key, sorted(set(big)))
        torch.manual_seed(1)
        Do, lo_ = orc.dict_learning(X, PK, alpha=0.3, steps=4, init_weight=D0, **PCASES[tag])
        dl, dD = float(np.abs(r0[key + "_l"] - lo_.numpy()).max()), float(np.abs(r0[key + "_D"] - Do.numpy()).max())
        assert dl <= 1e-5 and dD <= 2e-5, (key, dl, dD)


# ---- row-sharded line search (ista.py:23-52 on two ranks) -------------------------------------
BT = dict(n=700, d=64, k=200, split=263, alpha=0.25, lr=0.6, maxiter=8, eta=1.5)


BT_LARGE = dict(n=300, d=300, k=1100, split=117, alpha=0.25, lr=0.6, maxiter=6, eta=1.5)   # beyond the fused shapes


def _bt_problem(cfg=None):
    cfg = cfg or BT
    g = torch.Generator().manual_seed(33)
    X = torch.randn(cfg["n"], cfg["d"], generator=g)
    W = torch.nn.functional.normalize(torch.randn(cfg["d"], cfg["k"], generator=g), dim=0)
    return X, W


def _bt_large_worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd.parallel import sharded_encode
    from lasso_amd.engine import HipEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = BT_LARGE
    X, W = _bt_problem(c)
    lo, hi = (0, c["split"]) if rank == 0 else (c["split"], c["n"])
    out = {}
    for tag, tol in (("f32", 0.0), ("f32tol", 2e-3)):
        z, info = sharded_encode(HipEngine(), X[lo:hi].cuda(), W.cuda(), c["alpha"], None, lr=c["lr"],
                                 maxiter=30 if tol else c["maxiter"], tol=tol, backtrack=True,
                                 eta_backtrack=c["eta"], return_info=True)
        out[tag + "_z"] = z.cpu().numpy()
        out[tag + "_trials"] = np.array(info["trials"])
        out[tag + "_it"] = np.array(info["iterations"])
    np.savez(os.path.join(tmp, "btl%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_line_search_on_two_row_shards_beyond_the_fused_shapes(tmp_path):
    """d = 300, k = 1100: the unfused line search (general MFMA GEMMs) with the same all-reduced decisions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    from lasso_amd.linear.solvers import ista
    port = 35500 + (os.getpid() % 2000)
    mp.start_processes(_bt_large_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "btl0.npz"), np.load(tmp_path / "btl1.npz")
    c = BT_LARGE
    X, W = _bt_problem(c)
    for tag, tol, iters in (("f32", 0.0, c["maxiter"]), ("f32tol", 2e-3, 30)):
        tr = orc.FistaTrace()
        zo = orc.fista(X, torch.zeros(c["n"], c["k"]), W, alpha=c["alpha"], lr=c["lr"], maxiter=iters, tol=tol,
                      backtrack=True, eta_backtrack=c["eta"], trace=tr)
        assert list(r0[tag + "_trials"]) == list(r1[tag + "_trials"]) == list(tr.trials), tag
        assert int(r0[tag + "_it"]) == int(r1[tag + "_it"]) == tr.iterations, tag
        z = np.concatenate([r0[tag + "_z"], r1[tag + "_z"]])
        # 20+ momentum iterations on a 3.7x overcomplete dictionary amplify fp32 summation-order differences between
        # the oracle's GEMMs and the MFMA GEMMs (same trials, same iteration count); the single-process HIP solve runs
        # the same kernels and differs only in how the five sums of a trial are added
        assert np.abs(z - zo.numpy()).max() <= (1e-4 if tol == 0.0 else 5e-3), tag
        zs, info = ista(X.cuda(), torch.zeros(c["n"], c["k"], device="cuda"), W.cuda(), alpha=c["alpha"], lr=c["lr"],
                        maxiter=iters, tol=tol, backtrack=True, eta_backtrack=c["eta"], return_info=True)
        assert info["trials"] == list(r0[tag + "_trials"]) and info["iterations"] == int(r0[tag + "_it"]), tag
        assert np.abs(z - zs.cpu().numpy()).max() <= 1e-5, tag


def _bt_worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd.parallel import sharded_encode
    from lasso_amd.engine import HipEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, W = _bt_problem()
    lo, hi = (0, BT["split"]) if rank == 0 else (BT["split"], BT["n"])
    out = {}
    for tag, dt, tol in (("f32", torch.float32, 0.0), ("f32tol", torch.float32, 3e-3), ("bf16", torch.bfloat16, 0.0)):
        z, info = sharded_encode(HipEngine(), X[lo:hi].cuda().to(dt), W.cuda().to(dt), BT["alpha"], None, lr=BT["lr"],
                                 maxiter=40 if tol else BT["maxiter"], tol=tol, backtrack=True,
                                 eta_backtrack=BT["eta"], return_info=True)
        out[tag + "_z"] = z.float().cpu().numpy()
        out[tag + "_trials"] = np.array(info["trials"])
        out[tag + "_lr"] = np.array(info["accepted_lr"])
    np.savez(os.path.join(tmp, "bt%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_line_search_on_two_row_shards_equals_the_whole_batch(tmp_path):
    """sharded_encode(backtrack=True): every F <= Q decision and the stop rule on sums over BOTH
    ranks (lasso_fista_solve_sharded).  The trial trace equals the oracle's on the whole batch and
    the single-process HIP solve; the code matches row for row."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lasso_oracle as orc
    from lasso_amd.linear.solvers import ista
    port = 33500 + (os.getpid() % 2000)
    mp.start_processes(_bt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "bt0.npz"), np.load(tmp_path / "bt1.npz")
    X, W = _bt_problem()
    for tag, tol, iters in (("f32", 0.0, BT["maxiter"]), ("f32tol", 3e-3, 40)):
        tr = orc.FistaTrace()
        zo = orc.fista(X, torch.zeros(BT["n"], BT["k"]), W, alpha=BT["alpha"], lr=BT["lr"], maxiter=iters, tol=tol,
                      backtrack=True, eta_backtrack=BT["eta"], trace=tr)
        assert list(r0[tag + "_trials"]) == list(r1[tag + "_trials"]) == list(tr.trials), tag
        assert np.array_equal(r0[tag + "_lr"], r1[tag + "_lr"]), tag
        z = np.concatenate([r0[tag + "_z"], r1[tag + "_z"]])
        assert np.abs(z - zo.numpy()).max() <= 1e-4, tag
        zs, info = ista(X.cuda(), torch.zeros(BT["n"], BT["k"], device="cuda"), W.cuda(), alpha=BT["alpha"], lr=BT["lr"],
                        maxiter=iters, tol=tol, backtrack=True, eta_backtrack=BT["eta"], return_info=True)
        assert info["trials"] == list(r0[tag + "_trials"]), tag
        assert np.abs(z - zs.cpu().numpy()).max() <= 1e-5, tag
    # bf16 tensors: same decisions on both ranks; the trace of the single-process bf16 multi-launch solve
    assert list(r0["bf16_trials"]) == list(r1["bf16_trials"])
    zs, info = ista(X.cuda().bfloat16(), torch.zeros(BT["n"], BT["k"], device="cuda").bfloat16(), W.cuda().bfloat16(),
                    alpha=BT["alpha"], lr=BT["lr"], maxiter=BT["maxiter"], tol=0.0, backtrack=True,
                    eta_backtrack=BT["eta"], return_info=True, kernel="tile")
    assert info["trials"] == list(r0["bf16_trials"])
    z = np.concatenate([r0["bf16_z"], r1["bf16_z"]])
    assert np.abs(z - zs.float().cpu().numpy()).max() <= 2e-2
