"""Host logic of the EM driver on CPU: single-process against the oracle's own
dict_learning, and world_size=2 over gloo against the single-process result
(the compute engine is the oracle stand-in of tests/oracle_engine.py; the HIP
engine is exercised by the -m gpu tests)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import lasso_oracle as orc
from oracle_engine import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n=96, d=12, k=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=g)
    D0 = torch.nn.functional.normalize(torch.randn(d, k, generator=g), dim=0)
    return X, D0


@pytest.mark.parametrize("constrained,persist", [(True, False), (True, True), (False, False)])
def test_em_loop_matches_oracle_dict_learning(constrained, persist):
    from lasso_amd.parallel import em_loop
    X, D0 = _problem()
    torch.manual_seed(11)
    Dref, lref = orc.dict_learning(X, 40, alpha=0.3, constrained=constrained, persist=persist,
                                   steps=6, init_weight=D0, lr=0.1, maxiter=15)
    torch.manual_seed(11)
    D, losses = em_loop(OracleEngine(), X, D0.clone(), 0.3, constrained=constrained,
                        persist=persist, steps=6, solver_kwargs=dict(lr=0.1, maxiter=15))
    assert (losses - lref).abs().max().item() < 2e-5
    assert (D - Dref).abs().max().item() < 2e-4


def test_degenerate_atoms_follow_reference_rng():
    """An atom nobody uses is re-drawn from torch's generator exactly like
    dict_learning.py:92-98 (same values, same generator advance), and its codes
    are zeroed."""
    from lasso_amd.parallel import constrained_mstep
    X, D0 = _problem(n=64, d=10, k=12)
    Z = orc.sparse_encode(X, D0, 0.2, lr=0.1, maxiter=20)
    Z[:, 3] = 0
    Z[:, 7] = 0
    Dref, Zref = D0.clone(), Z.clone()
    torch.manual_seed(5)
    orc.update_dict(Dref, X, Zref)
    after_ref = torch.rand(1)
    eng = OracleEngine()
    D = D0.clone()
    buf = torch.empty(12 * 12 + 12 * 10)
    A, B = eng.gram(Z, X, buf)
    torch.manual_seed(5)
    mask = constrained_mstep(eng, A, B, D)
    after = torch.rand(1)
    assert mask is not None and mask.tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0]
    assert torch.equal(after, after_ref)
    assert (D - Dref).abs().max().item() < 1e-5
    assert torch.equal(D[:, 3], Dref[:, 3]) and torch.equal(D[:, 7], Dref[:, 7])


def _worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd.parallel import dict_learning_sharded
    from oracle_engine import OracleEngine as Eng
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, D0 = _problem(n=100, d=12, k=40)
    lo, hi = (0, 37) if rank == 0 else (37, 100)       # ragged shards on purpose
    out = {}
    for tag, kw in {"bcd": dict(constrained=True), "ridge": dict(constrained=False),
                    "tol": dict(constrained=True, tol=3e-3, maxiter=80)}.items():
        skw = dict(lr=0.1, maxiter=15, tol=0.0)
        skw.update({k: v for k, v in kw.items() if k in ("tol", "maxiter")})
        torch.manual_seed(1)
        D, losses = dict_learning_sharded(X[lo:hi], 40, alpha=0.3, steps=4, init_weight=D0,
                                          engine=Eng(), constrained=kw["constrained"], **skw)
        out[tag + "_D"], out[tag + "_l"] = D.numpy(), losses.numpy()
        # the same with ALL rows on rank 0 and none on rank 1: the empty rank issues the same collectives
        torch.manual_seed(1)
        elo, ehi = (0, 100) if rank == 0 else (100, 100)
        D, losses = dict_learning_sharded(X[elo:ehi], 40, alpha=0.3, steps=4, init_weight=D0,
                                          engine=Eng(), constrained=kw["constrained"], **skw)
        out["e" + tag + "_D"], out["e" + tag + "_l"] = D.numpy(), losses.numpy()
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True,
                       start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    X, D0 = _problem(n=100, d=12, k=40)
    for tag, kw in {"bcd": dict(constrained=True, tol=0.0, maxiter=15),
                    "ridge": dict(constrained=False, tol=0.0, maxiter=15),
                    "tol": dict(constrained=True, tol=3e-3, maxiter=80)}.items():
        torch.manual_seed(1)
        Dref, lref = orc.dict_learning(X, 40, alpha=0.3, steps=4, init_weight=D0, lr=0.1, **kw)
        # both ranks hold the same replicated dictionary and the global objective
        assert np.array_equal(r0[tag + "_D"], r1[tag + "_D"])
        assert np.array_equal(r0[tag + "_l"], r1[tag + "_l"])
        assert np.abs(r0[tag + "_l"] - lref.numpy()).max() < 2e-5, tag
        assert np.abs(r0[tag + "_D"] - Dref.numpy()).max() < 2e-4, tag
        assert np.array_equal(r0["e" + tag + "_D"], r1["e" + tag + "_D"]), tag      # rank 1 held no rows
        assert np.array_equal(r0["e" + tag + "_l"], r1["e" + tag + "_l"]), tag
        assert np.abs(r0["e" + tag + "_l"] - lref.numpy()).max() < 2e-5, tag
        assert np.abs(r0["e" + tag + "_D"] - Dref.numpy()).max() < 2e-4, tag


def _bt_worker(rank, world, port, tmp):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-lasso_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from lasso_amd.parallel import sharded_encode
    from oracle_engine import OracleEngine as Eng
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, D0 = _problem(n=100, d=12, k=40)
    lo, hi = (0, 37) if rank == 0 else (37, 100)
    z, info = sharded_encode(Eng(), X[lo:hi], D0, 0.3, None, lr=1.5, maxiter=12, tol=0.0, backtrack=True,
                             eta_backtrack=1.5, return_info=True)
    # fixed step, the reference's stop rule on the rows of BOTH ranks (ista.py:64,93): iterations-to-tol
    zt, it = sharded_encode(Eng(), X[lo:hi], D0, 0.3, None, lr=0.1, maxiter=300, tol=1e-4, return_info=True)
    np.savez(os.path.join(tmp, "bt%d.npz" % rank), z=z.numpy(), trials=np.array(info["trials"]),
             zt=zt.numpy(), it=np.array(it["iterations"]), last=np.array(it["last_delta"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_line_search_equals_the_whole_batch(tmp_path):
    """sharded_encode(backtrack=True) over gloo: the F <= Q decisions use the sums of both ranks
    (ista.py:23,28,32-35), so the trial trace equals the oracle's on the whole batch."""
    port = 27500 + (os.getpid() % 2000)
    mp.start_processes(_bt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "bt0.npz"), np.load(tmp_path / "bt1.npz")
    X, D0 = _problem(n=100, d=12, k=40)
    tr = orc.FistaTrace()
    zo = orc.fista(X, torch.zeros(100, 40), D0, alpha=0.3, lr=1.5, maxiter=12, tol=0.0, backtrack=True,
                  eta_backtrack=1.5, trace=tr)
    assert list(r0["trials"]) == list(r1["trials"]) == list(tr.trials)
    assert max(tr.trials) > 1                                   # the search did backtrack
    assert np.abs(np.concatenate([r0["z"], r1["z"]]) - zo.numpy()).max() <= 1e-5
    tr = orc.FistaTrace()
    zt = orc.fista(X, torch.zeros(100, 40), D0, alpha=0.3, lr=0.1, maxiter=300, tol=1e-4, trace=tr)
    assert int(r0["it"]) == int(r1["it"]) == tr.iterations and 64 < tr.iterations < 300     # several chunks + a replay
    assert float(r0["last"]) == float(r1["last"]) and float(r0["last"]) <= 100 * 40 * 1e-4
    assert np.abs(np.concatenate([r0["zt"], r1["zt"]]) - zt.numpy()).max() <= 1e-5
