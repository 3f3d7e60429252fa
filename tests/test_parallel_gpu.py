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

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K, SPLIT = 1100, 96, 320, 413
CASES = {"bcd": dict(constrained=True, lr=0.08, maxiter=12, tol=0.0),
         "ridge": dict(constrained=False, lr=0.08, maxiter=12, tol=0.0),
         "tol": dict(constrained=True, lr=0.08, maxiter=200, tol=2e-3),
         "auto": dict(constrained=True, maxiter=10),                      # lr='auto', default tol
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
        Dl, losses = dict_learning_sharded(X[lo:hi], K, alpha=0.3, steps=4, init_weight=D0,
                                           engine=HipEngine(), **kw)
        out[tag + "_D"], out[tag + "_l"] = Dl.cpu().numpy(), losses.cpu().numpy()
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_engine(tmp_path):
    from lasso_amd.linear import dict_learning
    port = 31500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    X, D0 = _problem()
    for tag, kw in CASES.items():
        # both ranks hold the same replicated dictionary and the same global objective, bit for bit
        assert np.array_equal(r0[tag + "_D"], r1[tag + "_D"]), tag
        assert np.array_equal(r0[tag + "_l"], r1[tag + "_l"]), tag
        torch.manual_seed(1)
        Dref, lref = dict_learning(X.cuda(), K, alpha=0.3, steps=4, init_weight=D0, progbar=False,
                                   device="cuda", **kw)
        assert np.abs(r0[tag + "_l"] - lref.cpu().numpy()).max() <= 1e-4, tag
        assert np.abs(r0[tag + "_D"] - Dref.cpu().numpy()).max() <= 1e-4, tag
