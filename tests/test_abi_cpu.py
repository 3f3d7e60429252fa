"""CPU-side checks of the boundary: the C-ABI library loads and exports every
symbol include/lasso_hip.h declares; the host mirror keeps the reference's
signatures and error behaviour; no compute call is made (no GPU here)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lasso_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lasso_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lasso_amd import _native
    assert os.path.exists(_native.lib_path()), "build the extension first (__graft_entry__.build())"
    lib = ctypes.CDLL(_native.lib_path())
    names = _declared_symbols()
    assert len(names) >= 8
    for name in names:
        assert hasattr(lib, name), name
    L = _native.lib()
    assert L.lasso_hip_abi_version() == 7
    assert L.lasso_hip_status_string(0) == b"ok"
    assert L.lasso_fista_workspace_bytes(4096, 256, 1024, 0, 100, 0.0, 0, 0) > 2 * 1024 * 1024
    assert L.lasso_fista_workspace_bytes(4096, 256, 4096, 0, 100, 0.0, 0, 0) > 0    # unfused path
    assert L.lasso_fista_workspace_bytes(4096, 256, 4096, 0, 100, 0.0, 0, 1) > L.lasso_fista_workspace_bytes(4096, 256, 4096, 0, 100, 0.0, 0, 0)    # unfused line search: + the candidate matrix


def test_signatures_match_reference():
    from lasso_amd.linear import sparse_encode, initialize_code
    from lasso_amd.linear.solvers import ista
    sig = inspect.signature(ista)
    names = list(sig.parameters)
    assert names[:11] == ["x", "z0", "weight", "alpha", "fast", "lr", "maxiter", "tol",
                          "backtrack", "eta_backtrack", "verbose"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["alpha"], d["fast"], d["lr"], d["maxiter"], d["tol"], d["backtrack"],
            d["eta_backtrack"], d["verbose"]) == (1.0, True, "auto", 10, 1e-5, False, 1.5, False)
    sig = inspect.signature(sparse_encode)
    assert list(sig.parameters)[:6] == ["x", "weight", "alpha", "z0", "algorithm", "init"]
    assert sig.parameters["algorithm"].default == "ista"
    assert list(inspect.signature(initialize_code).parameters) == ["x", "weight", "alpha", "mode"]


def test_host_side_errors_without_gpu():
    from lasso_amd.linear import sparse_encode, initialize_code
    from lasso_amd import NativeError
    x, w = torch.randn(4, 3), torch.randn(3, 5)
    with pytest.raises(ValueError):
        sparse_encode(x, w, algorithm="nope")
    with pytest.raises(ValueError):
        initialize_code(x, w, 1.0, "nope")
    with pytest.raises(AssertionError):
        sparse_encode(x, w, z0=torch.zeros(4, 4))
    assert initialize_code(x, w, 1.0, "zero").shape == (4, 5)
    if not torch.cuda.is_available():
        with pytest.raises(NativeError):     # no CPU fallback: fails loudly
            sparse_encode(x, w, lr=0.1)
