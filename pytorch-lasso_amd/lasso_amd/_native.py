"""ctypes binding of liblasso_hip.so (the C ABI of include/lasso_hip.h).

This is the binding a maintainer of the reference would add next to
lasso/linear/solvers/ista.py; see INTEGRATION.md.  The library is built in-tree
by ``__graft_entry__.build()`` / ``make -C pytorch-lasso_amd/csrc``.
"""
import collections
import ctypes as C
import os
import threading
import warnings

import torch

LASSO_OK, LASSO_ERR_BAD_ARG, LASSO_ERR_UNSUPPORTED = 0, 1, 2
LASSO_ERR_WORKSPACE, LASSO_ERR_HIP, LASSO_WARN_LINESEARCH = 3, 4, 5
LASSO_PENDING, LASSO_WARN_ABORTED, LASSO_PENDING_MAPPED, LASSO_PENDING_DEFERRED = 6, 7, 8, 9
LASSO_F32, LASSO_BF16 = 0, 1
STOP_GLOBAL, STOP_NONE, STOP_GLOBAL_CHUNKED = 0, 1, 2
ABI_VERSION = 7
KERNEL_AUTO, KERNEL_TILE, KERNEL_SPLITK = 0, 0x100, 0x200
SOLVE_ASYNC = 0x4000
SOLVE_ONE_CHUNK = 0x10000
SOLVE_SHARDED = 0x8000
SOLVE_STATUS_MAPPED = 0x20000       # iters_out = four words of pinned (device-writable) host memory
SOLVE_DEFER_VERDICT = 0x40000       # the verdict's launch is left to lasso_fista_solve_verdict_deferred (another stream)
LR_AUTO = -1.0

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class NativeError(RuntimeError):
    """The HIP extension is missing, or a native call failed."""


_LIB_PATH = os.path.join(_HERE, "liblasso_hip.so")


def lib_path():
    return _LIB_PATH


# int (*lasso_allreduce_fn)(void* ctx, double* sums, int count)   (include/lasso_hip.h)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


def use_library(path):
    """Bind a different build of the library (A/B builds of the kernels under tools/); must be
    called before the first native call."""
    global _LIB_PATH, _LIB
    _LIB_PATH, _LIB = path, None


def _declare(lib):
    i64, i32, dbl, vp, sz = C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_size_t
    lib.lasso_hip_abi_version.restype = i32
    lib.lasso_hip_status_string.restype = C.c_char_p
    lib.lasso_hip_status_string.argtypes = [i32]
    lib.lasso_hip_last_error.restype = C.c_char_p
    lib.lasso_hip_device_cus.argtypes = [C.POINTER(i32)]
    lib.lasso_debug_force_standby.restype = i32
    lib.lasso_debug_force_standby.argtypes = [i32]
    lib.lasso_fista_workspace_bytes.restype = sz
    lib.lasso_fista_workspace_bytes.argtypes = [i64, i64, i64, i32, i32, dbl, i32, i32]
    lib.lasso_fista_kernel_name.restype = C.c_char_p
    lib.lasso_fista_kernel_name.argtypes = [i64, i64, i64, i32, i32]
    lib.lasso_fista_solve.restype = i32
    lib.lasso_fista_solve.argtypes = [
        vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32,
        dbl, dbl, i32, i32, dbl, i32, i32, dbl, C.POINTER(C.c_int32), C.POINTER(C.c_float),
        C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
        vp, sz, vp]
    lib.lasso_fista_prepare.restype = i32
    lib.lasso_fista_prepare.argtypes = [vp, i64, i64, i64, i32, i32, vp, sz, vp]
    lib.lasso_fista_run.restype = i32
    lib.lasso_fista_run.argtypes = [
        vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32,
        dbl, dbl, i32, i32, i32, i32, i32, vp, vp, sz, vp]
    lib.lasso_lipschitz_workspace_bytes.restype = sz
    lib.lasso_lipschitz_workspace_bytes.argtypes = [i64, i64]
    lib.lasso_lipschitz.restype = i32
    lib.lasso_lipschitz.argtypes = [vp, i64, i64, i64, i32, C.POINTER(dbl), vp, sz, vp]
    lib.lasso_objective_workspace_bytes.restype = sz
    lib.lasso_objective_workspace_bytes.argtypes = [i64, i64, i64]
    lib.lasso_objective.restype = i32
    lib.lasso_objective.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, dbl, vp, vp,
                                    vp, sz, vp]
    lib.lasso_objective_throttled.restype = i32
    lib.lasso_objective_throttled.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, dbl, vp, vp, i32,
                                              vp, sz, vp]
    lib.lasso_gram_accumulate.restype = i32
    lib.lasso_gram_workspace_bytes.restype = sz
    lib.lasso_gram_workspace_bytes.argtypes = [i64, i64, i64]
    lib.lasso_gram_accumulate.argtypes = [vp, i64, vp, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp]
    lib.lasso_gram_accumulate_signal.argtypes = [vp, i64, vp, i64, i64, i64, i64, i32, vp, vp, vp, sz, vp, i32, vp]
    lib.lasso_fista_solve_verdict_deferred.restype = i32
    lib.lasso_fista_solve_verdict_deferred.argtypes = [vp, vp, vp, i32, vp]
    lib.lasso_mstep_pipe_head_word.restype = vp
    lib.lasso_mstep_pipe_head_word.argtypes = [i64, i64, i64, vp, sz]
    lib.lasso_dict_sweep_workspace_bytes.restype = sz
    lib.lasso_dict_sweep_workspace_bytes.argtypes = [i64, i64]
    lib.lasso_dict_sweep.restype = i32
    lib.lasso_dict_sweep.argtypes = [vp, vp, vp, i64, i64, i64, i32, dbl, i32, vp, i64, i64,
                                     C.c_uint64, vp, C.POINTER(C.c_int32), vp, sz, vp]
    lib.lasso_dict_sweep_async.restype = i32
    lib.lasso_dict_sweep_async.argtypes = [vp, vp, vp, i64, i64, i64, i32, dbl, i32, vp, i64, i64,
                                           C.c_uint64, vp, vp, vp, sz, vp]
    lib.lasso_dict_sweep_async_to.restype = i32
    lib.lasso_dict_sweep_async_to.argtypes = [vp, vp, vp, i64, vp, i64, i64, i64, i32, dbl, i32, vp, i64, i64,
                                              C.c_uint64, vp, vp, vp, i32, vp, sz, vp]
    lib.lasso_stream_wait_word.restype = i32
    lib.lasso_stream_wait_word.argtypes = [vp, i32, i32, vp]
    lib.lasso_mstep_pipe_stages.restype = i32
    lib.lasso_mstep_pipe_stages.argtypes = [i64, i64, i64]
    lib.lasso_mstep_pipe_stage_rows.restype = i32
    lib.lasso_mstep_pipe_stage_rows.argtypes = [i64, i64, i64, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.lasso_mstep_pipe_workspace_bytes.restype = sz
    lib.lasso_mstep_pipe_workspace_bytes.argtypes = [i64, i64, i64]
    lib.lasso_mstep_pipe_gram.restype = i32
    lib.lasso_mstep_pipe_gram.argtypes = [vp, i64, vp, i64, i64, i64, i64, i32, vp, i64, i32, vp, sz, vp]
    lib.lasso_mstep_pipe_wait.restype = i32
    lib.lasso_mstep_pipe_wait.argtypes = [i64, i64, i64, i32, vp, sz, vp]
    lib.lasso_mstep_pipe_rows.restype = i32
    lib.lasso_mstep_pipe_rows.argtypes = [vp, i64, vp, i64, i64, i64, i64, i32, i32, i32, vp, sz, vp]
    lib.lasso_mstep_pipe_sweep.restype = i32
    lib.lasso_mstep_pipe_sweep.argtypes = [vp, i64, vp, i64, i64, i64, i64, i32, dbl, i32, vp, vp, sz, vp]
    lib.lasso_mstep_pipe_finish.restype = i32
    lib.lasso_mstep_pipe_finish.argtypes = [vp, i64, i64, i64, i64, i32, dbl, i32, vp, vp, i32, vp, sz, vp]
    lib.lasso_mstep_pipe_signal.restype = i32
    lib.lasso_mstep_pipe_signal.argtypes = [i64, i64, i64, i32, vp, sz, vp]
    lib.lasso_fista_solve_verdict_mapped.restype = i32
    lib.lasso_fista_solve_verdict_mapped.argtypes = [i64, i64, i64, i64, i32, i32, dbl, vp, vp, vp, sz, vp]
    lib.lasso_dict_sweep_count.restype = vp
    lib.lasso_dict_sweep_count.argtypes = [i64, i64, vp, sz]
    lib.lasso_fista_solve_sharded.restype = i32
    lib.lasso_fista_solve_sharded.argtypes = [
        vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i64, i32, dbl, dbl, i32, i32, dbl, dbl,
        ALLREDUCE_FN, vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float),
        C.POINTER(C.c_float), vp, sz, vp]
    lib.lasso_fista_solve_collect.restype = i32
    lib.lasso_fista_solve_collect.argtypes = [i64, i64, i64, i32, i32, dbl, vp, vp, sz, vp]
    lib.lasso_fista_solve_deltas.restype = vp
    lib.lasso_fista_solve_deltas.argtypes = [i64, i64, i64, i32, i32, dbl, vp, sz]
    lib.lasso_fista_solve_verdict.restype = i32
    lib.lasso_fista_solve_verdict.argtypes = [i64, i64, i64, i64, i32, i32, dbl, vp, vp, sz, vp]
    lib.lasso_fista_solve_finish.restype = i32
    lib.lasso_fista_solve_finish.argtypes = [i64, i64, i64, i32, i32, dbl, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                             vp, sz, vp]
    lib.lasso_init_transpose_workspace_bytes.restype = sz
    lib.lasso_init_transpose_workspace_bytes.argtypes = [i64, i64]
    lib.lasso_init_transpose.restype = i32
    lib.lasso_init_transpose.argtypes = [i64, i64, i64, i32, vp, i64, vp, i64, vp, i64, vp, sz, vp]
    lib.lasso_ridge_workspace_bytes.restype = sz
    lib.lasso_ridge_workspace_bytes.argtypes = [i64, i64]
    lib.lasso_ridge_solve.restype = i32
    lib.lasso_ridge_solve.argtypes = [vp, vp, vp, i64, i64, i64, i32, dbl, C.POINTER(C.c_int32), vp, sz, vp]
    lib.lasso_dict_fill_degenerate.restype = i32
    lib.lasso_dict_fill_degenerate.argtypes = [vp, i64, i64, i64, i32, vp, vp, i64, i64, i32, vp]
    lib.lasso_zero_columns.restype = i32
    lib.lasso_zero_columns.argtypes = [vp, i64, i64, i64, i32, vp, vp]
    pi32 = C.POINTER(C.c_int32)
    lib.lasso_cd_workspace_bytes.restype = sz
    lib.lasso_cd_workspace_bytes.argtypes = [i64, i64, i64, i32]
    lib.lasso_cd_prepare.restype = i32
    lib.lasso_cd_prepare.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp, sz, vp]
    lib.lasso_cd_run.restype = i32
    lib.lasso_cd_run.argtypes = [i64, i64, i64, dbl, dbl, i32, pi32, pi32, vp, sz, vp]
    lib.lasso_cd_finish.restype = i32
    lib.lasso_cd_finish.argtypes = [vp, i64, vp, i64, i64, i64, i64, dbl, vp, sz, vp]
    geom = [i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32]
    lib.lasso_conv_ista_workspace_bytes.restype = sz
    lib.lasso_conv_ista_workspace_bytes.argtypes = geom
    lib.lasso_conv_ista_kernel_name.restype = C.c_char_p
    lib.lasso_conv_ista_kernel_name.argtypes = geom
    lib.lasso_conv_ista_solve.restype = i32
    lib.lasso_conv_ista_solve.argtypes = [vp, vp, vp, vp] + geom + [i32, dbl, dbl, i32, i32, dbl, pi32,
                                                                    C.POINTER(C.c_float), vp, sz, vp]
    lib.lasso_conv_objective.restype = i32
    lib.lasso_conv_objective.argtypes = [vp, vp, vp] + geom + [i32, dbl, vp, vp, sz, vp]
    lib.lasso_conv_lip_workspace_bytes.restype = sz
    lib.lasso_conv_lip_workspace_bytes.argtypes = [i64, i64, i32, i32]
    lib.lasso_conv_lip_bound.restype = i32
    lib.lasso_conv_lip_bound.argtypes = [vp, i64, i64, i32, i32, i32, i32, C.POINTER(dbl), vp, sz, vp]
    lib.lasso_fista_backward_workspace_bytes.restype = sz
    lib.lasso_fista_backward_workspace_bytes.argtypes = [i64, i64, i64]
    lib.lasso_fista_backward.restype = i32
    lib.lasso_fista_backward.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i32, dbl, i32, i32,
                                         vp, vp, vp, vp, sz, vp]
    lib.lasso_fista_backward_steps.restype = i32
    lib.lasso_fista_backward_steps.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i32, dbl, C.POINTER(C.c_float),
                                               i32, i32, vp, vp, vp, vp, sz, vp]
    lib.lasso_patches_extract.restype = i32
    lib.lasso_patches_extract.argtypes = [vp, vp, i64, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, vp]
    lib.lasso_patches_reconstruct.restype = i32
    lib.lasso_patches_reconstruct.argtypes = [vp, i64, vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, vp]
    lib.lasso_cd_solve.restype = i32
    lib.lasso_cd_solve.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, dbl, i32,
                                   dbl, pi32, pi32, vp, sz, vp]


def lib():
    """Load the shared library; fail loudly if it is not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise NativeError(
                "lasso_amd: %s is missing -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` (there is no CPU fallback)" % path)
        try:
            handle = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise NativeError("lasso_amd: cannot load %s: %s" % (path, e))
        _declare(handle)
        if handle.lasso_hip_abi_version() != ABI_VERSION:
            raise NativeError("lasso_amd: ABI version mismatch")
        _LIB = handle
    return _LIB


def check(status):
    if status == LASSO_OK:
        return
    L = lib()
    msg = "%s: %s" % (L.lasso_hip_status_string(status).decode(),
                      L.lasso_hip_last_error().decode())
    if status == LASSO_WARN_LINESEARCH:
        warnings.warn("backtracking line search failed. Reverting to initial step size")
        return
    if status == LASSO_ERR_BAD_ARG:
        raise ValueError(msg)
    if status == LASSO_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise NativeError(msg)


def require_gpu():
    if not torch.cuda.is_available():
        raise NativeError("lasso_amd: no HIP device visible (torch.cuda.is_available() is "
                          "False); this package has no CPU fallback")


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class HostWords:
    """A few int32 words of pinned (device-writable) host memory whose LAST word a kernel sets to 1, released, after it
    has written the others (LASSO_SOLVE_STATUS_MAPPED, lasso_dict_sweep_async, lasso_mstep_pipe_finish).  ``arm()``
    zeroes that word before the launch; ``wait()`` polls it -- the result needs neither a copy launch nor an event
    record behind the kernel (an event record between two kernels of a stream costs ~5 us on that stream)."""

    def __init__(self, count):
        self.tensor = torch.zeros(count, dtype=torch.int32).pin_memory()
        self.view = self.tensor.numpy()              # shares the pinned pages

    def arm(self):
        self.view[-1] = 0
        return self.tensor.data_ptr()

    def ready(self):
        return self.view[-1] != 0

    def wait(self, timeout=120.0):
        import time
        view, spins, t0 = self.view, 0, None
        while view[-1] == 0:
            spins += 1
            if (spins & 0x3FFF) == 0:
                now = time.monotonic()
                if t0 is None:
                    t0 = now
                elif now - t0 > timeout:
                    raise NativeError("lasso_amd: no result from the GPU after %.0f s (kernel fault or hang)" % timeout)
        return view


_WS = collections.OrderedDict()       # key -> buffer, least recently used first
_WS_LOCK = threading.Lock()
_WS_MAX_ENTRIES = 64                   # (device, stream, thread, purpose) combinations kept
_WS_MAX_BYTES = 8 << 30                # ... and their total size; beyond either the oldest entries go


def _ws_evict(keep):
    total = sum(b.numel() for b in _WS.values())
    while len(_WS) > 1 and (len(_WS) > _WS_MAX_ENTRIES or total > _WS_MAX_BYTES):
        key = next(iter(_WS))
        if key == keep:                 # never the buffer being handed out
            _WS.move_to_end(key)
            key = next(iter(_WS))
            if key == keep:
                break
        total -= _WS.pop(key).numel()


def workspace(device, nbytes, tag='fista'):
    """A cached device scratch buffer (caller-owned memory of the C ABI).  One buffer per
    (device, HIP stream, host thread, purpose): calls enqueued on one stream are ordered, so
    they may share scratch; different streams or threads never do (the C library itself is
    re-entrant -- all state lives in the workspace the caller passes).  The cache is bounded
    (least recently used entries are dropped beyond `_WS_MAX_ENTRIES` keys or `_WS_MAX_BYTES`
    bytes): thread pools with churn or code that creates many streams do not pin one workspace
    set each for ever.  A dropped buffer goes back to torch's caching allocator, which keeps
    it alive for the kernels already enqueued on its stream."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream,
           threading.get_ident(), tag)
    with _WS_LOCK:
        buf = _WS.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
            _WS[key] = buf
        _WS.move_to_end(key)
        _ws_evict(key)
    return buf


def release_workspaces():
    """Drop every cached scratch buffer (e.g. after a one-off large problem)."""
    with _WS_LOCK:
        _WS.clear()
