"""HIP-backed ISTA/FISTA: same signature, defaults and error behaviour as
``lasso.linear.solvers.ista.ista`` (reference ista.py:57-104)."""
import ctypes as C

import torch

from ... import _native as nat

_DT = {torch.float32: nat.LASSO_F32, torch.bfloat16: nat.LASSO_BF16}
# bf16 tensors (BASELINE config 3).  On the fused shapes bf16 x, W, z0 go to the native
# bf16-MFMA kernels (csrc/bt_bf16.hip: bf16 operands, fp32 accumulation and state), with or
# without the backtracking line search.  Everything else (fp16, larger shapes, verbose,
# autograd) computes in fp32 on exact up-conversions of x, W, z0 with one rounding of the
# result back to the tensor dtype.  Both are at least as accurate as the reference's all-bf16
# arithmetic; parity is judged on the objective (rtol 2e-3).
_UPCAST = (torch.bfloat16, torch.float16)


def _to_device(t, device):
    return t if t.device == device else t.to(device)


def lazy_zeros(like, n, k):
    """An all-zero [n,k] start that costs no memory: a (0,0)-stride view of one element, MARKED as
    the library's own sentinel.  The solvers recognise the marked object and hand the C ABI a NULL z0
    (= zeros, no fill and no read of n*k floats); anything else that touches it sees an ordinary
    read-only zeros tensor.  The mark lives on this python object only -- views, clones and a
    caller's own broadcast tensors (``torch.full((1, 1), c).expand(n, k)``) never carry it and are
    read like any other z0 (sparse_encode.py:44-45)."""
    key = (like.device, like.dtype)
    base = _ZERO_ELEMENT.get(key)
    if base is None:         # one element per device and dtype, made once: an EM loop asks for this every step, and
        base = _ZERO_ELEMENT[key] = like.new_zeros(1, 1)       # the fill launch sat on the step's dependent chain
    z = base.expand(n, k)
    z._lasso_lazy_zeros = True
    return z


_ZERO_ELEMENT = {}


def _is_lazy_zeros(z0):
    return (z0 is not None and getattr(z0, '_lasso_lazy_zeros', False) is True and z0.dim() == 2
            and z0.stride(0) == 0 and z0.stride(1) == 0)


def _ista_verbose(x, z0, weight, alpha, fast, lr, maxiter, tol, dev):
    """verbose=True: the reference prints the mean objective of z before every iteration
    (ista.py:80-81, 'loss: %0.4f').  Same here, one HIP iteration at a time (the state
    round-trips through HBM each step -- a debugging mode, not the fast path)."""
    from ...engine import HipEngine
    eng = HipEngine(dev)
    n, k = z0.shape
    budget = torch.tensor(float(n * k) * tol, dtype=torch.float32).item()
    z, y, done, last = z0, None, 0, float('nan')
    ws = eng.fista_workspace(n, x.shape[1], k, maxiter)
    for it in range(maxiter):
        loss, _ = eng.objective_sums(x, z, weight, alpha)
        print('loss: %0.4f' % loss.item())
        z, y, delta = eng.fista_run(x, weight, z, y, alpha, lr, fast, it, 1, True, ws=ws)
        done, last = it + 1, delta[0].item()
        if last <= budget:
            break
    return z, dict(iterations=done, last_delta=last)


class _UnrolledIsta(torch.autograd.Function):
    """Differentiable solve: the reference's loop is plain torch code, so
    torch.autograd differentiates through its unrolled iterations (ista.py:79-102).  Forward: the
    HIP kernel one iteration per launch, keeping the iterates z_0..z_T; backward:
    lasso_fista_backward_steps (csrc/autograd.hip).  `lr` is the fixed step, or the list of the
    steps a line-search solve accepted (ista.py:17-54: python floats, constants of the graph) --
    then exactly len(lr) iterations are replayed."""

    @staticmethod
    def forward(ctx, x, z0, weight, alpha, fast, lr, maxiter, tol):
        from ...engine import HipEngine
        dev = x.device
        eng = HipEngine(dev)
        n, k = z0.shape
        xg, wg = x.detach().contiguous(), weight.detach().contiguous()
        # The iterates z_0 .. z_T are kept for the reverse pass in ONE buffer of exactly T+1 iterates.
        # With the stop rule active T is not known in advance: an ordinary (untraced, single-launch) solve
        # finds the stopping iteration first -- the solve is bitwise deterministic, so the replay below lands
        # on the same iterate -- instead of growing the trace in blocks and concatenating them (twice the
        # memory at the end) with a host synchronisation per iteration.
        steps = int(maxiter)
        if tol > 0 and steps > 0:
            _, info = _solve_native(xg, z0.detach(), wg, alpha, fast, lr, steps, tol, False, 1.5, False, True,
                                    out_device=dev)
            steps = int(info['iterations'])
        trace = torch.empty((steps + 1, n, k), dtype=torch.float32, device=dev)
        trace[0].copy_(z0.detach())
        ws = eng.fista_workspace(n, xg.shape[1], k, max(steps, 1))
        y = None
        for it in range(steps):
            _, y, _ = eng.fista_run(xg, wg, trace[it], y, alpha, lr[it] if isinstance(lr, list) else lr, fast, it, 1,
                                    False, ws=ws, z_out=trace[it + 1])
        done = steps
        ctx.save_for_backward(xg, wg, trace)
        ctx.lr, ctx.fast = lr, fast
        return trace[done].clone()

    @staticmethod
    def backward(ctx, grad_z):
        from ...engine import HipEngine
        xg, wg, trace = ctx.saved_tensors
        eng = HipEngine(xg.device)
        gx, gw, gz0 = eng.fista_backward(xg, wg, trace.contiguous(), grad_z.contiguous().float(), ctx.lr,
                                         ctx.fast, ctx.needs_input_grad[0], ctx.needs_input_grad[2],
                                         ctx.needs_input_grad[1])
        return gx, gz0, gw, None, None, None, None, None


class PendingSolve:
    """Outcome of a solve whose stop rule was left running on the GPU (``ista(..., begin=True)``):
    calling it waits for the SOLVE only -- work enqueued behind it keeps the GPU busy -- and
    returns True, or False when the solve has to be repeated with ``stop_mode='chunked'``: the
    in-kernel stop rule gave up (a workgroup of the persistent kernel was not resident), or -- batches
    with more tiles than resident workgroups, enqueued as one chunk -- the rule fired before the last
    iteration (z then holds a later iterate).  ``iterations`` / ``last_delta`` are valid after a True call."""

    def __init__(self, status, event, deferred=None):
        # status: a _native.HostWords of 4 words; event: a torch event behind the copy of the words (the in-kernel
        # rule's path), or None when the verdict kernel writes them itself and raises the "valid" word (polled)
        self._status, self._event = status, event
        self.iterations, self.last_delta = None, None
        # deferred: (workspace tensor, device) of a LASSO_SOLVE_DEFER_VERDICT solve whose verdict launch is still owed
        self._deferred = deferred

    @property
    def deferred(self):
        """True while the verdict's launch has not been enqueued (``ista(..., begin='defer')``): call launch_verdict()
        on a stream ordered behind the solve -- calling the object before that would wait for ever"""
        return self._deferred is not None

    def launch_verdict(self, gate=None):
        """(begin='defer') enqueue the stop rule's launch on the CURRENT stream, which the caller has ordered behind the
        solve's kernels (lasso_fista_solve_verdict_deferred); same host thread as the solve, before its next one.
        ``gate``: (address of an int32 in device memory, value) -- the word the stream's wait polled; the launch
        re-checks it and answers "repeat the solve" if the wait gave up instead"""
        ws, dev = self._deferred
        self._deferred = None
        with torch.cuda.device(dev):
            nat.check(nat.lib().lasso_fista_solve_verdict_deferred(
                nat.ptr(ws), self._status.tensor.data_ptr(), C.c_void_p(int(gate[0])) if gate else None,
                int(gate[1]) if gate else 0, nat.stream_ptr(dev)))

    def status_word(self):
        """address of the "valid" word (pinned host memory), for lasso_stream_wait_word; None on the event path"""
        return None if self._event is not None or self._status is None else self._status.tensor.data_ptr() + 12

    def __call__(self):
        if self._deferred is not None:
            raise RuntimeError("PendingSolve: the deferred verdict was never launched (launch_verdict())")
        if self._event is not None:
            self._event.synchronize()
            st = self._status.view
        else:
            st = self._status.wait()
        if int(st[2]) != 0:
            return False
        self.iterations = int(st[0])
        self.last_delta = float(st[1:2].view('float32')[0])
        return True


class PendingShardedSolve(PendingSolve):
    """A LASSO_SOLVE_SHARDED solve (this rank's row shard of a larger batch): the stop rule of ista.py:93
    sums over ALL shards, so the outcome exists only after the ranks' per-iteration sums met.
    ``deltas`` is a device view (inside the solve's workspace) of this shard's `maxiter` sums; the
    driver all-reduces them -- in place, or carried in the tail of another message -- and calls
    ``judge(reduced_or_None, n_global)``, which enqueues the rule on the summed vector and the copy of
    its words.  Calling the object then waits like PendingSolve (False: the rule fired before the last
    iteration -- repeat the E-step on the chunked path)."""

    def __init__(self, deltas, judge):
        super().__init__(None, None)
        self.deltas, self._judge = deltas, judge

    def judge(self, reduced, n_global):
        self._status, self._event = self._judge(reduced, int(n_global))


_PINNED = {}


def _pinned_status(dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    slots = _PINNED.setdefault(key, [])
    # a small ring: a slot is reused only after several later solves on the same stream
    if len(slots) < 4:
        slots.append(nat.HostWords(4))
        return slots[-1]
    slots.append(slots.pop(0))
    return slots[-1]


def ista(x, z0, weight, alpha=1.0, fast=True, lr='auto', maxiter=10,
         tol=1e-5, backtrack=False, eta_backtrack=1.5, verbose=False,
         return_info=False, stop_mode='global', kernel='auto', begin=False, shard=False):
    """Solve min_z 0.5*||z W^T - x||^2 + alpha*||z||_1 on the GPU.

    x [n,d], z0 [n,k], weight [d,k]; returns a NEW tensor z [n,k] with the
    dtype/device of z0 (``maxiter=0`` returns ``z0`` itself, ista.py:76,104).
    Inputs are never modified.  ``return_info`` (extension) additionally
    returns ``dict(iterations=..., last_delta=...)`` (plus ``trials`` / ``accepted_lr`` per
    outer iteration with the line search; ``return_info='objective'`` adds the mean objective
    of the result).  ``stop_mode`` (extension): 'global' = the reference's rule (default),
    'chunked' = the same rule without the in-kernel handshake, 'none' = run maxiter iterations.
    ``kernel`` (extension): 'auto' | 'tile' | 'splitk' -- which fused kernel runs the batch
    (include/lasso_hip.h, LASSO_KERNEL_*); the code is bitwise the same either way.
    ``begin`` (extension): return ``(z, pending)`` without waiting for the stop rule's outcome
    (``begin='defer'``, with stop_mode='one-chunk': the rule's launch itself is left to the caller, who enqueues it on
    another stream -- ``pending.deferred`` / ``pending.launch_verdict()``);
    ``pending`` is a :class:`PendingSolve`, or None when the solve completed inside the call.
    ``shard`` (extension, with ``begin``): x is one rank's row shard of a larger batch -- ``pending`` is a
    :class:`PendingShardedSolve` whose sums the multi-GPU driver all-reduces (lasso_amd/parallel.py).
    Tensors that live on the CPU are staged through the current HIP device
    (the arithmetic still runs in the HIP kernels; there is no CPU fallback).
    """
    nat.require_gpu()
    if backtrack and eta_backtrack <= 1:
        raise ValueError('eta must be > 1.')                       # ista.py:18-19
    if x.dim() != 2 or weight.dim() != 2 or z0.dim() != 2:
        raise RuntimeError("ista expects 2-D x, z0, weight")
    n, d = x.shape
    k = weight.shape[1]
    if weight.shape[0] != d or tuple(z0.shape) != (n, k):
        raise RuntimeError("shape mismatch: x %s, weight %s, z0 %s"
                           % (tuple(x.shape), tuple(weight.shape), tuple(z0.shape)))
    if not (x.dtype == weight.dtype == z0.dtype):
        raise RuntimeError("expected x, weight, z0 of one dtype")
    if x.dtype in _UPCAST:
        if lr == 'auto':
            # the reference raises here as well: `.numpy()` has no bf16 (ista.py:12)
            raise TypeError("lasso_amd: lr='auto' is not supported for %s inputs" % x.dtype)
        if maxiter == 0:
            return (z0, dict(iterations=0, last_delta=float('nan'))) if return_info else z0
        native = (x.dtype == torch.bfloat16 and d <= 256 and k <= 1024 and n > 0 and not (verbose and not backtrack)
                  and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or z0.requires_grad)))
        if native:
            return _solve_native(x, z0, weight, alpha, fast, float(lr), maxiter, tol, backtrack, eta_backtrack,
                                 verbose, return_info, stop_mode=stop_mode, kernel=kernel)
        out = ista(x.float(), z0.float(), weight.float(), alpha, fast, lr, maxiter, tol, backtrack,
                   eta_backtrack, verbose, return_info, stop_mode, kernel)
        if return_info:
            return out[0].to(x.dtype), out[1]
        return out.to(x.dtype)
    if x.dtype not in _DT:
        raise NotImplementedError("lasso_amd: dtype %s is not implemented on the HIP path" % x.dtype)
    if begin and (return_info or x.dtype != torch.float32):
        raise ValueError("begin=True: fp32 tensors, return_info=False")
    if shard and not (begin and tol > 0 and not backtrack and not verbose and 0 < maxiter <= 64 and d <= 256 and k <= 1024):
        raise NotImplementedError("shard=True: asynchronous fixed-step fp32 solves with 0 < maxiter <= 64 and tol > 0 "
                                  "on the fused shapes")
    if maxiter == 0:
        if _is_lazy_zeros(z0):
            z0 = z0.contiguous()
        if begin:
            return z0, None
        return (z0, dict(iterations=0, last_delta=float('nan'))) if return_info else z0
    if n == 0:      # empty batch: nothing to solve (the reference's loop stops at once: 0 <= 0)
        z = z0.clone()
        if begin:
            return z, None
        return (z, dict(iterations=1, last_delta=0.0)) if return_info else z

    out_device = z0.device
    dev = x.device if x.is_cuda else (weight.device if weight.is_cuda else
                                      (z0.device if z0.is_cuda else torch.device('cuda', torch.cuda.current_device())))
    xg = _to_device(x.detach(), dev).contiguous()
    wg = _to_device(weight.detach(), dev).contiguous()
    zg = None if _is_lazy_zeros(z0) else _to_device(z0.detach(), dev).contiguous()   # None: zeros, never materialised

    wants_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or z0.requires_grad)
    if lr == 'auto':
        if not wants_grad and not verbose and not backtrack and d <= 256 and k <= 1024:
            # the fused fp32 kernels read the step from device memory: lambda_max and 1/L are
            # computed on the stream inside lasso_fista_solve, no host round trip (ista.py:72-73)
            if wg.dtype != torch.float32:
                raise TypeError("lasso_amd: lr='auto' needs an fp32 dictionary, got %s" % wg.dtype)
            lr = nat.LR_AUTO
        else:
            from ..lipschitz import lipschitz_constant
            lr = 1.0 / lipschitz_constant(wg)                      # ista.py:59-63
    lr = float(lr)

    if wants_grad:
        # differentiable path (the reference's loop is autograd-traceable, README "autograd")
        if not (x.is_cuda and weight.is_cuda and z0.is_cuda):
            raise NotImplementedError("lasso_amd: the differentiable path needs x, weight, z0 on the HIP device")
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise NotImplementedError("lasso_amd: the differentiable path computes in float32")
        if backtrack:
            # the line search picks the steps (one ordinary solve, no graph); the accepted steps are constants of the
            # reference's graph (python floats, ista.py:40,47), so the derivative is that of the replayed iterations
            _, binfo = _solve_native(xg, zg if zg is not None else z0, wg, alpha, fast, lr, maxiter, tol, True,
                                     eta_backtrack, False, True, out_device=dev, stop_mode=stop_mode, kernel=kernel)
            steps = [float(v) for v in binfo['accepted_lr']]
            z = _UnrolledIsta.apply(x, z0, weight, float(alpha), bool(fast), steps, len(steps), 0.0)
            if begin:
                return z, None
            return (z, dict(iterations=binfo['iterations'], last_delta=binfo['last_delta'], trials=binfo['trials'],
                            accepted_lr=steps)) if return_info else z
        z = _UnrolledIsta.apply(x, z0, weight, float(alpha), bool(fast), lr, int(maxiter), float(tol))
        if begin:
            return z, None
        return (z, dict(iterations=None, last_delta=None)) if return_info else z

    if verbose and not backtrack:
        if zg is None:
            zg = torch.zeros((n, k), dtype=xg.dtype, device=dev)
        z, info = _ista_verbose(xg, zg, wg, alpha, fast, lr, maxiter, tol, dev)
        z = z if z.device == out_device else z.to(out_device)
        if begin:
            return z, None
        return (z, info) if return_info else z

    return _solve_native(xg, zg if zg is not None else z0, wg, alpha, fast, lr, maxiter, tol, backtrack, eta_backtrack,
                         verbose, return_info, out_device=out_device, stop_mode=stop_mode, kernel=kernel, begin=begin,
                         shard=shard)


_STOP = {'global': nat.STOP_GLOBAL, 'chunked': nat.STOP_GLOBAL_CHUNKED, 'none': nat.STOP_NONE,
         # begin=True only: the rule is not expected to fire before maxiter (<= 64) iterations -- one chunk on the plain
         # kernels, judged on the device (LASSO_SOLVE_ONE_CHUNK); "repeat" when it does fire early
         'one-chunk': nat.STOP_GLOBAL | nat.SOLVE_ONE_CHUNK}
_KERNEL = {'auto': nat.KERNEL_AUTO, 'tile': nat.KERNEL_TILE, 'splitk': nat.KERNEL_SPLITK,
           'splitk1': nat.KERNEL_SPLITK | 0x1000, 'splitk2': nat.KERNEL_SPLITK | 0x2000,
           'splitk4': nat.KERNEL_SPLITK | 0x3000,
           # A/B knobs: T = 2 / 4 with the register-gather exchange (the default streams the partials by LDS-DMA)
           'splitk2g': nat.KERNEL_SPLITK | 0x400 | 0x2000, 'splitk4g': nat.KERNEL_SPLITK | 0x400 | 0x3000,
           'splitk1s': nat.KERNEL_SPLITK | 0x800 | 0x1000, 'unfused': 0x300, 'narrow': 0x800}


def _solve_native(x, z0, weight, alpha, fast, lr, maxiter, tol, backtrack, eta_backtrack, verbose,
                  return_info, out_device=None, stop_mode='global', kernel='auto', begin=False, shard=False):
    """One call of lasso_fista_solve on tensors of one dtype (float32, or bfloat16 with the
    line search)."""
    n, d = x.shape
    k = weight.shape[1]
    if out_device is None:
        out_device = z0.device
    dev = x.device if x.is_cuda else (weight.device if weight.is_cuda else
                                      (z0.device if z0.is_cuda else torch.device('cuda', torch.cuda.current_device())))
    xg = _to_device(x.detach(), dev).contiguous()
    wg = _to_device(weight.detach(), dev).contiguous()
    zg = None if _is_lazy_zeros(z0) else _to_device(z0.detach(), dev).contiguous()     # None -> NULL z0 = zeros
    L = nat.lib()
    z = torch.empty((n, k), dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        nbytes = L.lasso_fista_workspace_bytes(n, d, k, _DT[x.dtype], int(maxiter), float(tol),
                                               nat.STOP_GLOBAL, int(bool(backtrack)))
        ws = nat.workspace(dev, nbytes)
        want_async = bool(begin) and not backtrack and x.dtype == torch.float32
        iters = C.c_int32(0)
        last = C.c_float(float('nan'))
        want_host = bool(return_info) or bool(verbose)
        # asynchronous solves: the verdict kernel writes its four words straight into pinned host memory
        # (LASSO_SOLVE_STATUS_MAPPED) -- no copy launch behind it on the EM step's dependent chain
        status = _pinned_status(dev) if want_async and not shard else None
        want_trace = bool(backtrack) and bool(return_info)
        trials = (C.c_int32 * max(int(maxiter), 1))() if want_trace else None
        acc_lr = (C.c_float * max(int(maxiter), 1))() if want_trace else None
        acc_f = (C.c_float * max(int(maxiter), 1))() if (want_trace or (verbose and backtrack)) else None
        obj = C.c_float(float('nan')) if return_info == 'objective' else None
        st = L.lasso_fista_solve(
            nat.ptr(xg), xg.stride(0), nat.ptr(wg), wg.stride(0), nat.ptr(zg), zg.stride(0) if zg is not None else 0,
            nat.ptr(z), z.stride(0), n, d, k, _DT[x.dtype], float(alpha), lr, int(bool(fast)),
            int(maxiter), float(tol), _STOP[stop_mode] | _KERNEL[kernel] | (nat.SOLVE_ASYNC if want_async else 0) |
            (nat.SOLVE_SHARDED if shard else 0) | (nat.SOLVE_STATUS_MAPPED if status is not None else 0) |
            (nat.SOLVE_DEFER_VERDICT if (begin == 'defer' and status is not None and stop_mode == 'one-chunk') else 0),
            int(bool(backtrack)), float(eta_backtrack),
            C.cast(status.arm(), C.POINTER(C.c_int32)) if status is not None else
            (C.byref(iters) if want_host else None), C.byref(last) if want_host else None, trials, acc_lr, acc_f,
            C.byref(obj) if obj is not None else None, nat.ptr(ws), ws.numel(), nat.stream_ptr(dev))
        pending = None
        if st == nat.LASSO_PENDING and shard:
            shape = (n, d, k, _DT[x.dtype], int(maxiter), float(tol))
            dptr = L.lasso_fista_solve_deltas(*shape, nat.ptr(ws), ws.numel())
            if not dptr:
                raise nat.NativeError("lasso_fista_solve_deltas: no pending sharded solve")
            off = int(dptr) - ws.data_ptr()
            deltas = ws[off:off + 4 * int(maxiter)].view(torch.float32)

            def judge(reduced, n_global, ws=ws, shape=shape, dev=dev):
                with torch.cuda.device(dev):
                    status = _pinned_status(dev)
                    nat.check(L.lasso_fista_solve_verdict_mapped(shape[0], n_global, *shape[1:], nat.ptr(reduced),
                                                                 status.arm(), nat.ptr(ws), ws.numel(),
                                                                 nat.stream_ptr(dev)))
                return status, None
            pending = PendingShardedSolve(deltas, judge)
        elif st == nat.LASSO_PENDING_MAPPED:
            pending = PendingSolve(status, None)      # the verdict kernel raises the buffer's "valid" word itself
        elif st == nat.LASSO_PENDING_DEFERRED:
            pending = PendingSolve(status, None, deferred=(ws, dev))     # ... once the caller has launched it
        elif st == nat.LASSO_PENDING:
            nat.check(L.lasso_fista_solve_collect(n, d, k, _DT[x.dtype], int(maxiter), float(tol),
                                                  status.tensor.data_ptr(), nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            pending = PendingSolve(status, ev)
        else:
            nat.check(st)
    if begin:
        return (z if z.device == out_device else z.to(out_device)), pending
    if z.device != out_device:
        z = z.to(out_device)
    if verbose and backtrack:
        # the reference prints the mean objective of z before every iteration (ista.py:80-81):
        # that of z0, then F(z_next)/n of each accepted line-search trial but the last
        from ..dict_learning import lasso_loss
        z_start = zg.float() if zg is not None else torch.zeros((n, k), dtype=torch.float32, device=dev)
        print('loss: %0.4f' % lasso_loss(xg.float(), z_start, wg.float(), alpha).item())
        for v in list(acc_f[:max(iters.value - 1, 0)]):
            print('loss: %0.4f' % (v / n))
    if return_info:
        info = dict(iterations=iters.value, last_delta=last.value)
        if want_trace:      # what ista.py:43-47 prints per trial with verbose=True, condensed
            info['trials'] = list(trials[:iters.value])
            info['accepted_lr'] = list(acc_lr[:iters.value])
            info['accepted_f'] = list(acc_f[:iters.value])
        if obj is not None:
            info['objective'] = obj.value
        return z, info
    return z
