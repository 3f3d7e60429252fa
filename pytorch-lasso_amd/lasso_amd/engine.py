"""Compute engines behind the EM driver (linear/dict_learning.py, parallel.py).

``HipEngine`` is the product: every method is a call through the C ABI of
include/lasso_hip.h into the HIP kernels.  The driver only talks to this small
interface so that its host logic (sharding, collectives, persist/loss ordering)
can be exercised in CPU unit tests with a stand-in engine defined under tests/
-- nothing in this package falls back to CPU arithmetic.
"""
import ctypes as C

import torch

from . import _native as nat


class HipEngine:
    name = "hip"

    def __init__(self, device=None):
        nat.require_gpu()
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise nat.NativeError("HipEngine needs a HIP device, got %s" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = nat.lib()

    # -- plumbing ---------------------------------------------------------------
    def to_device(self, t):
        return t.detach().to(self.device).contiguous()

    def _ws(self, nbytes, tag):
        return nat.workspace(self.device, nbytes, tag)

    def _stream(self):
        return nat.stream_ptr(self.device)

    # -- E-step -------------------------------------------------------------------
    def encode(self, X, W, alpha, z0, **solver_kwargs):
        from .linear.sparse_encode import sparse_encode
        return sparse_encode(X, W, alpha, z0, **solver_kwargs)

    def encode_begin_sharded(self, X, W, alpha, z0, **solver_kwargs):
        """E-step on a ROW SHARD without a host wait: returns (Z, pending) with ``pending`` a
        PendingShardedSolve (its ``deltas`` are to be all-reduced, then ``judge``d), None when the solve
        needs no verdict (stop rule off), or returns None when these arguments have no asynchronous
        sharded form (line search, maxiter > 64, shapes beyond the fused kernels, ...): the caller then
        takes the synchronous sharded path."""
        from .linear.solvers import ista
        if not self.sharded_async_ok(X, W, **solver_kwargs) or not (z0 is None or z0.is_cuda) or X.shape[0] == 0:
            return None
        kw = dict(solver_kwargs)
        for name in ('n_global', 'algorithm', 'init'):
            kw.pop(name, None)
        tol = kw.get('tol', 1e-5)
        if z0 is None:
            from .linear.solvers.ista import lazy_zeros
            z0 = lazy_zeros(X, X.shape[0], W.shape[1])
        if not tol > 0:
            return ista(X, z0, W, alpha, begin=True, **kw)          # no rule: nothing to agree on
        return ista(X, z0, W, alpha, begin=True, shard=True, **kw)

    def sharded_async_ok(self, X, W, **solver_kwargs):
        """Whether encode_begin_sharded has a form for these arguments.  Depends on the solver's keyword
        arguments, d, k, the dtype and the device only -- NOT on this rank's row count -- so that every rank of
        an EM loop reaches the same answer (parallel.em_loop sums the answers over the ranks all the same)."""
        kw = solver_kwargs
        d, k = W.shape
        maxiter = kw.get('maxiter', 10)
        return bool(kw.get('algorithm', 'ista') == 'ista' and kw.get('init', None) is None and not kw.get('verbose')
                    and not kw.get('backtrack') and not kw.get('return_info') and X.dtype == torch.float32
                    and X.is_cuda and W.is_cuda and d <= 256 and k <= 1024
                    and kw.get('stop_mode', 'global') == 'global' and 0 < maxiter <= 64)

    def encode_begin(self, X, W, alpha, z0, defer_verdict=False, **solver_kwargs):
        """sparse_encode that does not wait for the stop rule's outcome: returns (Z, pending);
        ``pending`` is None (complete) or a callable that waits for the solve alone and returns
        False when it has to be repeated with stop_mode='chunked' (solvers/ista.py PendingSolve).
        ``defer_verdict``: where the solve allows it, the stop rule's launch is left to the caller
        (``pending.deferred`` -> ``pending.launch_verdict()`` on a stream ordered behind the solve)."""
        from .linear.solvers import ista
        kw = dict(solver_kwargs)
        plain = (kw.pop('algorithm', 'ista') == 'ista' and kw.pop('init', None) is None and not kw.get('verbose')
                 and not kw.get('backtrack') and not kw.get('return_info') and X.dtype == torch.float32
                 and X.is_cuda and W.is_cuda and (z0 is None or z0.is_cuda))
        if not plain:
            return self.encode(X, W, alpha, z0, **solver_kwargs), None
        if z0 is None:
            from .linear.solvers.ista import lazy_zeros
            z0 = lazy_zeros(X, X.shape[0], W.shape[1])
        return ista(X, z0, W, alpha, begin='defer' if defer_verdict else True, **kw)

    def encode_sharded_backtrack(self, X, W, alpha, z0, lr, fast, maxiter, tol, eta, n_global, all_reduce):
        """ISTA/FISTA with the backtracking line search on this rank's row shard
        (lasso_fista_solve_sharded).  ``all_reduce(t)`` sums a float64 CPU tensor over the ranks in
        place.  Returns (z, info) -- info holds the same trace on every rank."""
        n, d = X.shape
        k = W.shape[1]
        dt = {torch.float32: nat.LASSO_F32, torch.bfloat16: nat.LASSO_BF16}[X.dtype]
        L = self.lib
        failure = []

        def _cb(_ctx, buf, count):
            try:
                t = torch.tensor([buf[i] for i in range(count)], dtype=torch.float64)
                all_reduce(t)
                for i in range(count):
                    buf[i] = float(t[i])
                return 0
            except Exception as e:          # never let an exception cross the C frame
                failure.append(e)
                return 1
        cb = nat.ALLREDUCE_FN(_cb)
        X, W = X.contiguous(), W.contiguous()
        z0 = z0.contiguous() if z0 is not None else None
        z = torch.empty((n, k), dtype=X.dtype, device=self.device)
        iters, last = C.c_int32(0), C.c_float(float('nan'))
        cap = max(int(maxiter), 1)
        trials, acc_lr, acc_f = (C.c_int32 * cap)(), (C.c_float * cap)(), (C.c_float * cap)()
        with torch.cuda.device(self.device):
            nbytes = L.lasso_fista_workspace_bytes(n, d, k, dt, int(maxiter), float(tol), nat.STOP_GLOBAL, 1)
            ws = nat.workspace(self.device, nbytes)
            st = L.lasso_fista_solve_sharded(
                nat.ptr(X), X.stride(0), nat.ptr(W), W.stride(0), nat.ptr(z0), z0.stride(0) if z0 is not None else 0,
                nat.ptr(z), z.stride(0), n, int(n_global), d, k, dt, float(alpha), float(lr), int(bool(fast)),
                int(maxiter), float(tol), float(eta), cb, None, C.byref(iters), C.byref(last), trials, acc_lr, acc_f,
                nat.ptr(ws), ws.numel(), self._stream())
        if failure:
            raise failure[0]
        nat.check(st)
        it = iters.value
        return z, dict(iterations=it, last_delta=last.value, trials=list(trials[:it]),
                       accepted_lr=list(acc_lr[:it]), accepted_f=list(acc_f[:it]))

    sweep_out_of_place = True        # sweep_begin(out=...): lasso_dict_sweep_async_to
    defer_verdict = True             # encode_begin(defer_verdict=True): LASSO_SOLVE_DEFER_VERDICT

    def signal_words(self):
        """(int32[2] in device memory, [count0, count1] on the host): words that launches of one stream raise for waves
        of another to poll (lasso_gram_accumulate_signal, lasso_dict_sweep_async_to's started_word) -- they count up
        over the engine's lifetime, so a loop neither allocates nor clears them"""
        w = getattr(self, "_signal_words", None)
        if w is None:
            with torch.cuda.device(self.device):
                w = self._signal_words = (torch.zeros(2, dtype=torch.int32, device=self.device), [0, 0])
        return w

    def sweep_begin(self, A, B, D, eps, positive, out=None, started=None):
        """The atom sweep without the host round trip for the number of degenerate atoms: returns
        a callable giving (mask, ndeg) that waits only for the sweep itself.  ``out``: another [d, k] tensor that
        receives the new dictionary -- D is then only read (lasso_dict_sweep_async_to: the call may be enqueued before
        the host knows whether the step stands, and another stream may keep reading D).  ``started`` (with ``out``):
        (int32 device tensor, value) -- a launch in front of the sweep writes the value: "the sweep starts now"."""
        d, k = D.shape
        L = self.lib
        with torch.cuda.device(self.device):
            ws = self._ws(L.lasso_dict_sweep_workspace_bytes(d, k), "sweep")
            mask = torch.empty(k, dtype=torch.int32, device=self.device)     # the sweep writes every flag
            # the count goes straight into a pinned host word, written by the sweep's last kernel (no copy launch)
            host = self._ndeg_word()
            if out is None:
                nat.check(L.lasso_dict_sweep_async(
                    nat.ptr(A), nat.ptr(B), nat.ptr(D), D.stride(0), d, k, nat.LASSO_F32, float(eps),
                    int(bool(positive)), None, 0, 0, 0, nat.ptr(mask), host.arm(), nat.ptr(ws), ws.numel(),
                    self._stream()))
            else:
                if tuple(out.shape) != (d, k) or out.dtype != D.dtype or out.stride(1) != 1:
                    raise RuntimeError("sweep_begin: `out` must be a [d, k] tensor like D")
                nat.check(L.lasso_dict_sweep_async_to(
                    nat.ptr(A), nat.ptr(B), nat.ptr(D), D.stride(0), nat.ptr(out), out.stride(0), d, k, nat.LASSO_F32,
                    float(eps), int(bool(positive)), None, 0, 0, 0, nat.ptr(mask), host.arm(),
                    nat.ptr(started[0]) if started else None, int(started[1]) if started else 0, nat.ptr(ws), ws.numel(),
                    self._stream()))

        def result():        # (polls the word the sweep's last kernel raises: no event record behind the sweep)
            return mask, int(host.wait()[0])
        return result

    def _ndeg_word(self):
        """{count, valid}: two int32 words of pinned (device-writable) host memory from a small ring: a slot is
        reused only after several later sweeps have been enqueued."""
        ring = getattr(self, "_ndeg_ring", None)
        if ring is None:
            ring = self._ndeg_ring = [nat.HostWords(2) for _ in range(8)]
        ring.append(ring.pop(0))
        return ring[-1]

    def side_stream(self):
        """The engine's second stream (the EM loop's objective and the later stages of the pipelined M-step)."""
        st = getattr(self, "_side", None)
        if st is None:
            st = self._side = torch.cuda.Stream(self.device)
        return st

    def stream_wait_word(self, address, value, host_memory):
        """(on the current stream) one wave that returns when the int32 at `address` equals `value`"""
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_stream_wait_word(C.c_void_p(int(address)), int(value), int(bool(host_memory)),
                                                      self._stream()))

    # -- pipelined constrained M-step (lasso_mstep_pipe_*) -------------------------------------
    def mstep_pipe_stages(self, d, k):
        """Row ranges [(lo, hi), ...] of [A | B] that the stages of the pipelined M-step produce ([]: this dictionary
        shape has no pipelined form).  Depends on d and k only -- every rank of an EM loop reaches the same answer."""
        L = self.lib
        out = []
        for s_ in range(int(L.lasso_mstep_pipe_stages(1, d, k))):
            lo, hi = C.c_int64(0), C.c_int64(0)
            nat.check(L.lasso_mstep_pipe_stage_rows(1, d, k, s_, C.byref(lo), C.byref(hi)))
            out.append((lo.value, hi.value))
        return out

    def pipe_head_word(self, n, d, k, ws):
        """address of the word pipe_wait polls (device memory inside ws)"""
        p = self.lib.lasso_mstep_pipe_head_word(max(int(n), 1), d, k, nat.ptr(ws), ws.numel())
        if not p:
            raise nat.NativeError("lasso_mstep_pipe_head_word: no pipelined M-step for this shape")
        return int(p)

    def mstep_pipe_workspace(self, n, d, k):
        with torch.cuda.device(self.device):
            nbytes = self.lib.lasso_mstep_pipe_workspace_bytes(max(int(n), 1), d, k)
            return torch.zeros(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)   # (zeroed once: the tickets)

    def pipe_gram(self, Z, X, AB, R, ws):
        n, k = Z.shape
        d = X.shape[1]
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_mstep_pipe_gram(nat.ptr(Z) if n else None, Z.stride(0), nat.ptr(X) if n else None,
                                                     X.stride(0), n, d, k,
                                                     nat.LASSO_F32, nat.ptr(AB), AB.stride(0), int(R),
                                                     nat.ptr(ws), ws.numel(), self._stream()))

    def pipe_wait(self, n, d, k, seq, ws):
        """(on the side stream) one wave that returns once block row 0's Gram launch of step `seq` has finished"""
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_mstep_pipe_wait(max(int(n), 1), d, k, int(seq), nat.ptr(ws), ws.numel(),
                                                     self._stream()))

    def pipe_rows(self, AB, D, n, R, ws, seq=0):
        d, k = D.shape
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_mstep_pipe_rows(nat.ptr(AB), AB.stride(0), nat.ptr(D), D.stride(0), max(int(n), 1),
                                                     d, k, nat.LASSO_F32, int(R), int(seq), nat.ptr(ws), ws.numel(),
                                                     self._stream()))

    def pipe_sweep(self, AB, D, n, eps, positive, ws):
        """The gated sweep; returns the mask tensor its kernels fill.  D is only read."""
        d, k = D.shape
        with torch.cuda.device(self.device):
            mask = torch.empty(k, dtype=torch.int32, device=self.device)
            nat.check(self.lib.lasso_mstep_pipe_sweep(nat.ptr(AB), AB.stride(0), nat.ptr(D), D.stride(0),
                                                      max(int(n), 1), d, k, nat.LASSO_F32, float(eps),
                                                      int(bool(positive)), nat.ptr(mask), nat.ptr(ws), ws.numel(),
                                                      self._stream()))
        return mask

    def pipe_signal(self, n, d, k, seq, ws):
        """(on the side stream, last) raises the word pipe_finish(wait_seq=seq) waits for"""
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_mstep_pipe_signal(max(int(n), 1), d, k, int(seq), nat.ptr(ws), ws.numel(),
                                                       self._stream()))

    def pipe_finish(self, D, n, eps, positive, mask, ws, wait_seq=0):
        """Writes the new dictionary; returns a callable giving (mask, ndeg) that waits for this launch only."""
        d, k = D.shape
        with torch.cuda.device(self.device):
            host = self._ndeg_word()
            nat.check(self.lib.lasso_mstep_pipe_finish(nat.ptr(D), D.stride(0), max(int(n), 1), d, k, nat.LASSO_F32,
                                                       float(eps), int(bool(positive)), nat.ptr(mask),
                                                       host.arm(), int(wait_seq), nat.ptr(ws), ws.numel(),
                                                       self._stream()))

        def result():
            ndeg = int(host.wait()[0])
            if ndeg < 0:
                raise nat.NativeError("lasso_amd: the pipelined M-step's side stream did not finish in time "
                                      "(GPU shared with other work?); set LASSO_EM_PIPELINE=0")
            return mask, ndeg
        return result

    def lipschitz(self, W):
        from .linear.lipschitz import lipschitz_constant
        return lipschitz_constant(W)

    def fista_workspace(self, n, d, k, cap):
        """A caller-held workspace for a SEQUENCE of fista_run calls against one dictionary
        (iterations it0 .. < cap): W is packed into it by the first call that uses it, later
        calls skip the packing.  Held by the caller, never shared through the scratch cache."""
        with torch.cuda.device(self.device):
            nbytes = self.lib.lasso_fista_workspace_bytes(n, d, k, nat.LASSO_F32, int(cap), 0.0,
                                                          nat.STOP_NONE, 0)
            return {"buf": torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device),
                    "packed": False, "shape": (n, d, k), "cap": int(cap)}

    def fista_run(self, X, W, z_in, y_in, alpha, lr, fast, it0, iters, want_delta, ws=None, z_out=None,
                  kernel=nat.KERNEL_AUTO):
        """`iters` iterations from (z_in, y_in); returns (z, y, delta[iters] or None).
        Building block of the distributed exact-stop E-step and of the traced forward pass
        of the autograd path (lasso_fista_run).  `z_out`: optional [n,k] destination.
        `ws`: a workspace from fista_workspace() (W is packed once per workspace); None = a
        scratch buffer of this stream, W packed by this call."""
        n, d = X.shape
        k = W.shape[1]
        L = self.lib
        with torch.cuda.device(self.device):
            if ws is None:
                cap = it0 + iters
                nbytes = L.lasso_fista_workspace_bytes(n, d, k, nat.LASSO_F32, cap, 0.0, nat.STOP_NONE, 0)
                buf, packed = self._ws(nbytes, "fista"), False
            else:
                if ws["shape"] != (n, d, k) or it0 + iters > ws["cap"]:
                    raise ValueError("fista_run: workspace was sized for %s, cap %d" % (ws["shape"], ws["cap"]))
                buf, packed, cap = ws["buf"], ws["packed"], ws["cap"]
            if not packed:
                nat.check(L.lasso_fista_prepare(nat.ptr(W), W.stride(0), d, k, nat.LASSO_F32, int(cap),
                                                nat.ptr(buf), buf.numel(), self._stream()))
                if ws is not None:
                    ws["packed"] = True
            z = z_out if z_out is not None else torch.empty((n, k), dtype=torch.float32, device=self.device)
            y = torch.empty((n, k), dtype=torch.float32, device=self.device)
            delta = torch.empty(iters, dtype=torch.float32, device=self.device) if want_delta else None
            nat.check(L.lasso_fista_run(
                nat.ptr(X), X.stride(0), nat.ptr(z_in), z_in.stride(0) if z_in is not None else 0,
                nat.ptr(y_in), y_in.stride(0) if y_in is not None else 0,
                nat.ptr(z), z.stride(0), nat.ptr(y), y.stride(0), n, d, k, nat.LASSO_F32,
                float(alpha), float(lr), int(bool(fast)), int(it0), int(iters), int(cap), int(kernel),
                nat.ptr(delta),
                nat.ptr(buf), buf.numel(), self._stream()))
        return z, y, delta

    def fista_backward(self, X, W, trace, grad_z, lr, fast, need_x, need_w, need_z0):
        """Reverse pass through the unrolled solve (lasso_fista_backward_steps); trace [T+1,n,k] holds
        z_0..z_T, `lr` is the step or the list of the T steps of a line-search solve.
        Returns (gx, gw, gz0), None where not needed."""
        n, d = X.shape
        k = W.shape[1]
        T = trace.shape[0] - 1
        L = self.lib
        with torch.cuda.device(self.device):
            ws = self._ws(L.lasso_fista_backward_workspace_bytes(n, d, k), "bw")
            gx = torch.empty((n, d), dtype=torch.float32, device=self.device) if need_x else None
            gw = torch.empty((d, k), dtype=torch.float32, device=self.device) if need_w else None
            gz0 = torch.empty((n, k), dtype=torch.float32, device=self.device) if need_z0 else None
            steps = None
            if isinstance(lr, (list, tuple)):
                if len(lr) != T:
                    raise ValueError("fista_backward: %d steps for %d iterations" % (len(lr), T))
                steps, lr = (C.c_float * max(T, 1))(*lr), (lr[0] if T else 1.0)
            nat.check(L.lasso_fista_backward_steps(nat.ptr(X), X.stride(0), nat.ptr(W), W.stride(0), nat.ptr(trace),
                                                   nat.ptr(grad_z), n, d, k, nat.LASSO_F32, float(lr), steps,
                                                   int(bool(fast)), int(T), nat.ptr(gx), nat.ptr(gw), nat.ptr(gz0),
                                                   nat.ptr(ws), ws.numel(), self._stream()))
        return gx, gw, gz0

    # -- objective ---------------------------------------------------------------
    objective_loss_out = True      # objective_sums(..., loss_out=) exists (parallel.em_loop)

    def objective_sums(self, X, Z, W, alpha, loss_out=None, max_workgroups=0):
        """-> (loss_local 0-d float tensor, sums double[2] = {sum r^2, sum |z|}) on device.
        ``loss_out``: an optional 0-d fp32 device tensor (e.g. one slot of a loss history) the loss is
        written to directly."""
        n, d = X.shape
        k = W.shape[1]
        L = self.lib
        with torch.cuda.device(self.device):
            ws = self._ws(L.lasso_objective_workspace_bytes(n, d, k), "obj")
            loss = loss_out if loss_out is not None else torch.empty((), dtype=torch.float32, device=self.device)
            sums = torch.empty(2, dtype=torch.float64, device=self.device)
            nat.check(L.lasso_objective_throttled(nat.ptr(X), X.stride(0), nat.ptr(W), W.stride(0),
                                                  nat.ptr(Z), Z.stride(0), n, d, k, nat.LASSO_F32, float(alpha),
                                                  nat.ptr(loss), nat.ptr(sums), int(max_workgroups), nat.ptr(ws),
                                                  ws.numel(), self._stream()))
        return loss, sums

    # -- M-step --------------------------------------------------------------------
    def gram(self, Z, X, out, started=None):
        """out: flat fp32 buffer of k*k + k*d (+extra) floats; A and B are written at its
        start so one all-reduce covers both.  ``started``: (int32 device tensor, value) -- the product's first launch
        writes the value there as it starts (lasso_gram_accumulate_signal: a start signal for another stream)."""
        n, k = Z.shape
        d = X.shape[1]
        A = out[:k * k].view(k, k)
        B = out[k * k:k * k + k * d].view(k, d)
        with torch.cuda.device(self.device):
            ws = self._ws(self.lib.lasso_gram_workspace_bytes(n, d, k), "gram")
            if started is None:
                nat.check(self.lib.lasso_gram_accumulate(nat.ptr(Z), Z.stride(0), nat.ptr(X), X.stride(0),
                                                         n, d, k, nat.LASSO_F32, nat.ptr(A), nat.ptr(B),
                                                         nat.ptr(ws), ws.numel(), self._stream()))
            else:
                nat.check(self.lib.lasso_gram_accumulate_signal(nat.ptr(Z), Z.stride(0), nat.ptr(X), X.stride(0),
                                                                n, d, k, nat.LASSO_F32, nat.ptr(A), nat.ptr(B),
                                                                nat.ptr(ws), ws.numel(), nat.ptr(started[0]), int(started[1]),
                                                                self._stream()))
        return A, B

    def sweep(self, A, B, D, pool, eps, positive, seed=0):
        """In-place Gauss-Seidel atom sweep on D [d,k].  Returns (mask int32[k] on device,
        ndeg python int)."""
        d, k = D.shape
        L = self.lib
        with torch.cuda.device(self.device):
            ws = self._ws(L.lasso_dict_sweep_workspace_bytes(d, k), "sweep")
            mask = torch.zeros(k, dtype=torch.int32, device=self.device)
            ndeg = C.c_int32(0)
            pool_dev = pool.to(self.device).contiguous() if pool is not None else None
            nat.check(L.lasso_dict_sweep(
                nat.ptr(A), nat.ptr(B), nat.ptr(D), D.stride(0), d, k, nat.LASSO_F32, float(eps),
                int(bool(positive)), nat.ptr(pool_dev),
                pool_dev.shape[0] if pool_dev is not None else 0,
                pool_dev.stride(0) if pool_dev is not None else 0, int(seed) & (2 ** 64 - 1),
                nat.ptr(mask), C.byref(ndeg), nat.ptr(ws), ws.numel(), self._stream()))
        return mask, ndeg.value

    def fill_degenerate(self, D, mask, pool, positive):
        """The i-th flagged atom of D becomes pool row i, normalised (dict_learning.py:93-96)."""
        d, k = D.shape
        pool = pool.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_dict_fill_degenerate(
                nat.ptr(D), D.stride(0), d, k, nat.LASSO_F32, nat.ptr(mask), nat.ptr(pool), pool.shape[0],
                pool.stride(0), int(bool(positive)), self._stream()))

    def zero_columns(self, Z, mask):
        n, k = Z.shape
        with torch.cuda.device(self.device):
            nat.check(self.lib.lasso_zero_columns(nat.ptr(Z), Z.stride(0), n, k, nat.LASSO_F32,
                                                  nat.ptr(mask), self._stream()))

    def init_transpose(self, X, W):
        """z0 = X W (sparse_encode.py:24-25) on the library's NT GEMM."""
        n, d = X.shape
        k = W.shape[1]
        L = self.lib
        with torch.cuda.device(self.device):
            ws = self._ws(L.lasso_init_transpose_workspace_bytes(d, k), "init_t")
            z0 = torch.empty((n, k), dtype=torch.float32, device=self.device)
            nat.check(L.lasso_init_transpose(n, d, k, nat.LASSO_F32, nat.ptr(X), X.stride(0), nat.ptr(W), W.stride(0),
                                             nat.ptr(z0), z0.stride(0), nat.ptr(ws), ws.numel(), self._stream()))
        return z0

    def ridge(self, A, B, lam_n, check=False):
        """V = ((A + lam_n I)^-1 B)^T  [d,k] (dict_learning.py:117-121): blocked Cholesky and
        triangular solves of csrc/ridge.hip (lasso_ridge_solve).  `check` synchronises and raises
        like torch.linalg.cholesky when the matrix is not positive definite.  Beyond k = 4096
        (the kernels' limit) the k x k factorisation goes to torch.linalg on the device."""
        k, d = B.shape
        L = self.lib
        nbytes = L.lasso_ridge_workspace_bytes(d, k)
        if nbytes == 0:
            M = A.clone()
            M.diagonal().add_(lam_n)
            return torch.cholesky_solve(B, torch.linalg.cholesky(M)).T.contiguous()
        with torch.cuda.device(self.device):
            ws = self._ws(nbytes, "ridge")
            V = torch.empty((d, k), dtype=torch.float32, device=self.device)
            info = C.c_int32(0)
            status = L.lasso_ridge_solve(nat.ptr(A), nat.ptr(B), nat.ptr(V), V.stride(0), d, k, nat.LASSO_F32,
                                         float(lam_n), C.byref(info) if check else None, nat.ptr(ws), ws.numel(),
                                         self._stream())
            if info.value != 0:          # what torch.linalg.cholesky raises at dict_learning.py:120
                raise torch.linalg.LinAlgError(
                    "linalg.cholesky: The factorization could not be completed because the input is not "
                    "positive-definite (the leading minor of order %d is not positive-definite)." % info.value)
            nat.check(status)
        return V
