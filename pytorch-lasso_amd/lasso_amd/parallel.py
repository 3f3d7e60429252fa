"""EM driver shared by the single-GPU ``dict_learning`` and the batch-sharded
multi-GPU run (one process per GPU, torch.distributed backend 'nccl' = RCCL).

Sharding: rank r holds a row shard of X (and of Z); the dictionary is
replicated.  Per EM step:
  * E-step: local FISTA on the shard.  With the reference's global stop rule active
    (tol > 0) the per-iteration |z - z_next| sums of a chunk are all-reduced ONCE per
    chunk and every rank replays to the same stopping iteration (exact global rule);
  * objective + M-step: ONE all-reduce of the fp32 buffer
    [A = Z^T Z | B = Z^T X | sum r^2, sum |z|], then every rank runs the identical
    deterministic atom sweep (no broadcast of D; replacement directions for degenerate
    atoms are broadcast only in the steps where an atom actually degenerated).
The reference has no distributed code; this is the only parallelism strategy the build
adds (BASELINE.json north_star).
"""
import math

import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


def _world(group):
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _sharded(group):
    """True when the multi-rank code paths run: more than one rank -- or (LASSO_FORCE_COLLECTIVES=1) ONE rank with an
    initialised process group.  The second form exists so that every line a multi-GPU run executes -- the RCCL arms of
    _all_reduce / _broadcast on device tensors, the sharded E-step with its stop-rule sums in the M-step message, the
    line search's all-reduce callback -- has run on hardware before an 8-GPU node sees it: a 1-rank RCCL group on one
    GPU reduces over itself (tests/test_bench_gpu.py, bench.py --force-dist), and the first N > 1 run then differs
    from tested code in N only."""
    world, _ = _world(group)
    if world > 1:
        return True
    import os
    return (os.environ.get("LASSO_FORCE_COLLECTIVES", "0") == "1" and dist is not None and dist.is_available()
            and dist.is_initialized())


def _host_staged(t, group):
    """gloo moves host memory: device tensors are staged through the host explicitly (the
    production backend is 'nccl' = RCCL, which reduces device buffers over xGMI in place)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(t, group):
    if _sharded(group):
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _broadcast(t, group):
    src = dist.get_global_rank(group, 0) if group else 0
    if _host_staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def draw_directions(d, count):
    """`count` replacement directions for degenerate atoms, drawn the way the reference does
    (``dictionary[:, k].normal_()`` on a CPU tensor, dict_learning.py:93): same generator,
    same non-contiguous-view code path, one draw per degenerate atom in atom order."""
    buf = torch.empty(d, 2)
    col = buf[:, 0]
    out = torch.empty(count, d)
    for i in range(count):
        col.normal_()
        out[i] = col
    return out


def constrained_mstep(engine, A, B, D, eps=1e-10, positive=False, group=None):
    """Atom sweep on the (already all-reduced) Gram matrices; D updated in place.
    Returns the degenerate mask (device int32[k]) or None when no atom degenerated.
    The sweep never reads a replacement direction (a degenerate atom leaves the model,
    dict_learning.py:92-98), so directions are drawn -- on rank 0, then broadcast -- only
    in the steps where an atom actually degenerated, and exactly as many as needed: the
    common step costs no RNG call, no broadcast and no copy of D."""
    world, rank = _world(group)
    mask, ndeg = engine.sweep(A, B, D, None, eps, positive)
    if ndeg == 0:
        return None
    cand = draw_directions(D.shape[0], ndeg).to(D.device)      # every rank advances its generator alike
    if _sharded(group):   # every rank must use rank 0's directions
        _broadcast(cand, group)
    engine.fill_degenerate(D, mask, cand, positive)
    return mask


def sharded_encode(engine, X, W, alpha, z0, group=None, **kw):
    """E-step on this rank's shard.  world == 1: plain sparse_encode.  world > 1 with an
    active stop rule: exact GLOBAL rule through chunked speculation + one all-reduce of
    the chunk's delta vector (see module docstring)."""
    if not _sharded(group) or kw.get('algorithm', 'ista') == 'cd':
        # coordinate descent stops every row on its own (coordinate_descent.py:45-48): a row
        # shard needs no collective and reproduces the full batch exactly
        kw.pop('n_global', None)
        return engine.encode(X, W, alpha, z0, **kw)
    fast = kw.pop('fast', True)
    lr = kw.pop('lr', 'auto')
    maxiter = kw.pop('maxiter', 10)
    tol = kw.pop('tol', 1e-5)
    backtrack = kw.pop('backtrack', False)
    eta = kw.pop('eta_backtrack', 1.5)
    kw.pop('verbose', None)
    return_info = kw.pop('return_info', False)
    n_global = kw.pop('n_global', None)
    if kw.get('algorithm', 'ista') != 'ista':
        raise NotImplementedError("sharded E-step supports algorithm='ista' and 'cd' only")
    kw.pop('algorithm', None)
    if kw:
        raise TypeError("ista() got unexpected keyword arguments %s" % sorted(kw))
    if backtrack and eta <= 1:
        raise ValueError('eta must be > 1.')                                              # ista.py:18-19
    if lr == 'auto':
        lr = 1.0 / engine.lipschitz(W)
    n, k = X.shape[0], W.shape[1]
    if z0 is None:
        z0 = X.new_zeros(n, k)
    if maxiter == 0:
        return (z0, dict(iterations=0)) if return_info else z0
    if backtrack:
        # the line search decides on sums over the WHOLE batch (ista.py:23,28,32-35,93): the HIP
        # library hands this rank's sums to the callback below at every decision, all ranks add
        # theirs and take the same decision (lasso_fista_solve_sharded)
        if n_global is None:
            n_glob = torch.tensor([float(n)], dtype=torch.float64, device=X.device)
            _all_reduce(n_glob, group)
            n_global = n_glob.item()
        # a rank without rows would fail its argument check before the first collective and leave its peers
        # blocked in the line search's all-reduce: agree on that up front and fail on EVERY rank instead
        empty = torch.tensor([1.0 if n == 0 else 0.0], dtype=torch.float64, device=X.device)
        _all_reduce(empty, group)
        if empty.item() > 0:
            raise ValueError("sharded line search: %d rank(s) hold an empty row shard; give every rank at least "
                             "one row (or run those rows elsewhere)" % int(empty.item()))

        def reduce_host(t):          # a few float64 words in host memory; RCCL reduces device buffers
            if dist.get_backend(group) == "gloo":
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                g = t.to(X.device)
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                t.copy_(g)
        z, info = engine.encode_sharded_backtrack(X, W, alpha, z0, lr, fast, maxiter, tol, eta, int(n_global),
                                                  reduce_host)
        return (z, info) if return_info else z
    def result(z, iterations, last):
        return (z, dict(iterations=iterations, last_delta=last)) if return_info else z

    def run(z, y, it0, c, want_delta):
        if n == 0:      # a rank without rows: nothing to launch, zero sums -- but the SAME collectives as its peers
            return z, y, (X.new_zeros(c, dtype=torch.float32) if want_delta else None)
        return engine.fista_run(X, W, z, y, alpha, lr, fast, it0, c, want_delta)

    if not tol > 0:
        z, _, _ = run(z0, None, 0, maxiter, False)
        return result(z, maxiter, float('nan'))
    if n_global is None:          # (the EM driver passes the row count of the whole batch)
        n_glob = torch.tensor([float(n)], dtype=torch.float64, device=X.device)
        _all_reduce(n_glob, group)
        n_global = n_glob.item()
    # ista.py:64,93 on the rows of ALL ranks: sum_i |z - z_next| <= n_global * k * tol, compared in fp32
    budget = torch.tensor(float(n_global) * k * tol, dtype=torch.float32).item()
    chunk, done = 64, 0
    z, y = z0, None
    last = float('nan')
    while done < maxiter:
        c = min(chunk, maxiter - done)
        z2, y2, delta = run(z, y, done, c, True)
        _all_reduce(delta, group)           # ONE small all-reduce per chunk of <= 64 iterations, not per iteration
        hdelta = delta.tolist()             # the chunk's one host read; every rank sees the same summed vector
        hit = next((i for i, v in enumerate(hdelta) if v <= budget), -1)
        if hit >= 0:
            if hit + 1 < c:                 # replay the chunk from its intact input state up to the stopping iteration
                z2, _, _ = run(z, y, done, hit + 1, False)
            return result(z2, done + hit + 1, hdelta[hit])
        z, y, done, last = z2, y2, done + c, hdelta[-1]
        # scheduling only (the stop decision stays exact; every rank holds the same summed vector, so every rank sizes
        # the next chunk alike): iterations left from the geometric decay over this chunk, approached with short chunks
        # so that little is speculated past the stopping iteration -- the heuristic of lasso_fista_solve's chunked path
        chunk = 64
        if c >= 8 and budget > 0:
            h = c // 2
            hi, lo = max(hdelta[:h]), max(hdelta[h:])
            if lo > budget and hi > lo:
                left = math.log(lo / budget) / (math.log(hi / lo) / h)
                if left < 128:
                    chunk = max(8, min(int(left) - 6, 64))
            elif lo <= 4 * budget:
                chunk = 8
    return result(z, done, last)


class _EmptyShardPending:
    """The asynchronous sharded E-step of a rank that holds NO rows: zero stop-rule sums in the message, and
    the verdict every rank reads -- 'the rule fired before the last iteration' (chunk_verdict_kernel's third
    word) -- taken from the summed vector with the same fp32 comparison."""

    def __init__(self, ndelta, k, tol, device):
        self.deltas = torch.zeros(ndelta, dtype=torch.float32, device=device)
        self._k, self._tol, self._reduced, self._n_global = k, tol, None, 0
        self.iterations, self.last_delta = None, None

    def judge(self, reduced, n_global):
        self._reduced, self._n_global = reduced, n_global
        self._ev = None
        if reduced.is_cuda:      # (the two-stream loop reduces on its side stream: wait for THAT stream's work)
            self._ev = torch.cuda.Event()
            self._ev.record(torch.cuda.current_stream(reduced.device))

    def __call__(self):
        if getattr(self, '_ev', None) is not None:
            self._ev.synchronize()
        budget = torch.tensor(float(self._n_global) * self._k * self._tol, dtype=torch.float32).item()
        h = self._reduced.tolist()
        hit = next((i for i, v in enumerate(h) if v <= budget), -1)
        self.iterations, self.last_delta = (hit + 1 if hit >= 0 else len(h)), h[hit]
        return not (hit >= 0 and hit + 1 < len(h))


def em_loop(engine, X, weight, alpha, constrained=True, persist=False, lambd=1e-2, steps=60,
            progbar=False, solver_kwargs=None, group=None):
    """The EM loop of dict_learning.py:35-53 on this rank's row shard ``X`` [n_local, d].
    ``weight`` [d,k] must be identical on every rank.  Returns (weight, losses[steps])."""
    solver_kwargs = dict(solver_kwargs or {})
    world, rank = _world(group)
    multi = _sharded(group)          # (world > 1, or one rank with LASSO_FORCE_COLLECTIVES=1: the same code paths)
    n_local, d = X.shape
    k = weight.shape[1]
    # Which form the E-step takes must not depend on anything rank-local (a rank on another path would issue
    # other collectives than its peers: hang or corruption), so "this rank cannot take the asynchronous sharded
    # form" is summed over the ranks next to the row count, once, before the loop.
    can_async = (multi and hasattr(engine, 'encode_begin_sharded') and hasattr(engine, 'sweep_begin')
                 and getattr(engine, 'sharded_async_ok', lambda *a, **kw: True)(X, weight, **solver_kwargs))
    n_glob = torch.tensor([float(n_local), 0.0 if can_async else 1.0], dtype=torch.float64, device=X.device)
    _all_reduce(n_glob, group)
    n_total = n_glob[0].item()
    every_rank_async = n_glob[1].item() == 0
    losses = torch.zeros(steps, device=X.device)
    # One GPU: the step is enqueued without waiting on it -- the stop rule's outcome and the
    # sweep's count of degenerate atoms are collected at ONE host wait per step, placed where the
    # objective and the Gram product are still queued, so the GPU does not idle behind the host.
    overlap = not multi and hasattr(engine, 'encode_begin') and hasattr(engine, 'sweep_begin')
    # Several ranks: the same, with the stop rule's per-iteration sums (ista.py:93 sums over the rows of ALL
    # ranks) riding in the tail of the M-step message -- E-step (lr='auto' and the solve on the stream),
    # objective, Gram product, ONE all-reduce, the rule judged on the device from the summed vector -- and
    # the one host wait per step behind all of it.  (RCCL reduces on the stream; gloo, used when ranks
    # share a GPU in tests, stages the message through the host -- that copy is then the wait.)
    shard_async = multi and every_rank_async
    ndelta = int(solver_kwargs.get('maxiter', 10)) if shard_async else 0
    if not 0 < ndelta <= 64:
        ndelta = 0
    # ONE message per EM step: [A = Z^T Z | B = Z^T X | sum r^2, sum |z| | stop-rule sums]
    buf = torch.zeros(k * k + k * d + 2 + ndelta, dtype=torch.float32, device=X.device)
    tail = buf[k * k + k * d:k * k + k * d + 2]
    dtail = buf[k * k + k * d + 2:]
    Z0 = None
    bar = None
    if progbar and rank == 0:
        from tqdm import tqdm
        bar = tqdm(total=steps)
    # Any number of ranks: the sweep's count of degenerate atoms is not waited for -- it is looked at
    # after the NEXT E-step has been enqueued (every rank holds the same A, B, D bit for bit, so every
    # rank sees the same count and takes the same branch).
    defer = hasattr(engine, 'sweep_begin')
    deferred = None          # the previous step's sweep: callable -> (mask, ndeg)
    stats = getattr(engine, 'em_stats', None)     # optional counters (tests, tools/bench_em.py)

    def repair(mask, ndeg, Zprev):
        cand = draw_directions(d, ndeg).to(weight.device)       # every rank advances its generator alike
        if multi:
            _broadcast(cand, group)                              # ... and uses rank 0's directions
        engine.fill_degenerate(weight, mask, cand, False)                                 # :93-96
        if Zprev is not None and Zprev.shape[0] > 0:
            engine.zero_columns(Zprev, mask)                                              # :98

    def encode_sync():
        return sharded_encode(engine, X, weight, alpha, Z0, group=group, n_global=n_total, **solver_kwargs)

    tol = float(solver_kwargs.get('tol', 1e-5))

    def local_stats(Z, **kw):
        """(loss, {sum r^2, sum |z|}, A, B) of this rank's rows; a rank without rows contributes zeros"""
        if n_local == 0 and multi:
            buf[:k * k + k * d].zero_()
            return (torch.zeros((), device=X.device), torch.zeros(2, dtype=torch.float64, device=X.device),
                    buf[:k * k].view(k, k), buf[k * k:k * k + k * d].view(k, d))
        loss_local, sums = engine.objective_sums(X, Z, weight, alpha, **kw)               # :39
        A, B = engine.gram(Z, X, buf)
        return loss_local, sums, A, B

    # An E-step practically never stops before its `maxiter` (default 10) iterations: ask for the one-chunk form of the
    # asynchronous solve (plain kernels + the verdict on the device) instead of the in-kernel rule's per-iteration
    # exchange; a step where the rule does fire early is repeated on the chunked path below, like an aborted handshake.
    begin_kwargs = dict(solver_kwargs)
    if ('stop_mode' not in begin_kwargs and begin_kwargs.get('algorithm', 'ista') == 'ista'
            and 0 < int(begin_kwargs.get('maxiter', 10)) <= 64):
        begin_kwargs['stop_mode'] = 'one-chunk'
    # Round 6: the constrained loop on TWO streams -- the objective beside the M-step instead of in front of it, host
    # results polled instead of waited for through events.  Two forms: d == 256 and k a multiple
    # of 256 pipeline the M-step (later stages + objective beside the atom sweep); small dictionaries (d <= 64, k <= 256:
    # the one-workgroup sweep) double-buffer the dictionary, which lets the sweep be enqueued before the step's host wait
    # and the objective after it, on the other stream.  Decided from rank-invariant inputs only (every rank takes the
    # same path).
    import os
    side = os.environ.get("LASSO_EM_SIDE_STREAM", "1")
    if (constrained and defer and X.is_cuda and hasattr(engine, 'side_stream') and tol > 0
            and (overlap or (shard_async and ndelta > 0)) and side != "0"):
        piped = (os.environ.get("LASSO_EM_PIPELINE", "1") != "0" and hasattr(engine, 'mstep_pipe_stages')
                 and len(engine.mstep_pipe_stages(d, k)) > 0)
        # Which form, by rows per rank (the average: the same number on every rank).  Measured on one MI355X, config 4's
        # dictionary (tools/r6_em_forms.sh; ms per step: one-stream / pipelined / double-buffered):
        #    2048: 0.745 / 0.898 / 0.906    (objective and Gram product are ~20 us each there: less than what running beside
        #                                   the sweep costs it, and than the fixed launches of a stage; the host is not the
        #                                   limit -- it spends 0.5 ms per step waiting, cProfile)
        #    4096: 0.866 / 0.831 / 0.841    8192: 1.260 / 1.202 / 1.207   (equal; the pipelined form has the smaller message
        #                                                                  on the chain when there are several ranks)
        #   12288: 1.657 / 1.640 / 1.579   16384: 2.050 / 2.091 / 1.962
        #   32768: 3.63  / 3.92  / 3.44    65536: 6.80  / 7.66  / 6.50    (the stages of a pipelined M-step then take
        #                                  longer than the sweep can wait for; the double-buffered form holds the objective
        #                                  back until the sweep starts instead of letting it fight the Gram product for HBM)
        # Small dictionaries (the one-workgroup sweep; config 5's shape): double-buffered at every size measured
        # (2048 rows 0.2025 -> 0.190, 8192 0.2205 -> 0.201, 65536 0.673 -> 0.641).
        rows = n_total / max(world, 1)
        oop = getattr(engine, 'sweep_out_of_place', False)
        if oop and d <= 64 and k <= 256:
            form = "double-buffer"
        elif piped and 4096 <= rows <= 8192:
            form = "pipeline"
        elif oop and piped and rows > 8192:
            form = "double-buffer"
        elif side == "force" and (piped or oop):
            form = "pipeline" if piped else "double-buffer"
        else:
            form = None
        knob = os.environ.get("LASSO_EM_FORM")               # A/B knob (tools/r6_em_forms.sh): pipeline | double-buffer
        if knob == "pipeline" and piped or knob == "double-buffer" and oop:
            form = knob
        if form:
            return _em_loop_two_streams(engine, X, weight, alpha, persist, steps, bar, solver_kwargs, begin_kwargs, group,
                                        n_total, ndelta, losses, stats, pipeline=(form == "pipeline"))
    i, Zlast = 0, None
    while i < steps:
        pending, sharded = None, False
        if overlap:
            Z, pending = engine.encode_begin(X, weight, alpha, Z0, **begin_kwargs)        # :38
        elif not ndelta:
            Z = encode_sync()
        else:
            if n_local == 0:
                Z = Z0 if Z0 is not None else X.new_zeros(0, k)
                pending = _EmptyShardPending(ndelta, k, tol, X.device) if tol > 0 else None
            else:
                # (Z0 is None or a code this engine returned: besides sharded_async_ok(), which every rank agreed on
                # before the loop, encode_begin_sharded only asks for a start on the device -- make that hold here
                # rather than leave one rank raising while its peers wait in the step's all-reduce)
                if Z0 is not None and Z0.device != X.device:
                    Z0 = Z0.to(X.device)
                began = engine.encode_begin_sharded(X, weight, alpha, Z0, **solver_kwargs)
                if began is None:       # sharded_async_ok() said yes on every rank: this is a bug, not a fallback
                    raise RuntimeError("encode_begin_sharded refused arguments sharded_async_ok accepted")
                Z, pending = began
            sharded = pending is not None
            if stats is not None:
                stats['overlapped_steps'] = stats.get('overlapped_steps', 0) + 1
        direct = not multi and getattr(engine, 'objective_loss_out', False)   # losses[i] written in place: no copy launch
        loss_local, sums, A, B = local_stats(Z, **(dict(loss_out=losses[i]) if direct else {}))
        if deferred is not None:
            mask, ndeg = deferred()
            deferred = None
            if ndeg:     # rare: an atom degenerated in the previous sweep -- repair, redo this E-step
                if pending is not None and not sharded:
                    pending()
                repair(mask, ndeg, Zlast)
                continue
        if pending is not None and not sharded and not pending():
            # the in-kernel stop rule gave up (CUs held by other work): same rule, chunked
            Z = engine.encode(X, weight, alpha, Z0, **dict(solver_kwargs, stop_mode='chunked'))
            loss_local, sums, A, B = local_stats(Z)
            direct = False
        if multi:
            tail.copy_(sums)                 # the two objective sums ride in the Gram message
            if sharded:
                dtail.copy_(pending.deltas)  # ... and so do this shard's stop-rule sums
            _all_reduce(buf, group)
            if sharded:
                pending.judge(dtail, n_total)
                if not pending():            # the step's host wait; every rank reads the same verdict
                    # the rule fired before the last iteration (rare in an EM loop): exact rule, chunked replay
                    Z = encode_sync()
                    loss_local, sums, A, B = local_stats(Z)
                    tail.copy_(sums)
                    _all_reduce(buf, group)
                    if stats is not None:
                        stats['replayed_steps'] = stats.get('replayed_steps', 0) + 1
            losses[i] = (0.5 * tail[0] + alpha * tail[1]) / n_total
        elif not direct:
            losses[i] = loss_local
        if persist:
            Z0 = Z                                                                        # :40-41
        if constrained:
            if defer:
                deferred = engine.sweep_begin(A, B, weight, 1e-10, False)                 # :44-45
                Zlast = Z
            else:
                mask = constrained_mstep(engine, A, B, weight, group=group)
                if mask is not None and Z.shape[0] > 0:
                    engine.zero_columns(Z, mask)                                          # :98
        else:
            weight = engine.ridge(A, B, lambd * n_total, check=True)                      # :46-47
        if bar is not None:
            bar.set_postfix(loss=losses[i].item())                                        # :50
            bar.update(1)
        i += 1
    if deferred is not None:
        mask, ndeg = deferred()
        if ndeg:
            repair(mask, ndeg, Zlast)
    if bar is not None:
        bar.close()
    return weight, losses


def _em_loop_two_streams(engine, X, weight, alpha, persist, steps, bar, solver_kwargs, begin_kwargs, group, n_total,
                         ndelta, losses, stats, pipeline=True):
    """The constrained EM loop (dict_learning.py:35-53) on two streams -- see em_loop for the one-stream form, which
    this one reproduces step for step (same kernels on the same operands; only [A | B] of the pipelined M-step has
    another summation order).  Per step, stream M (the caller's): E-step, the head of the M-step (Gram product of the
    first block rows, its all-reduce, its U rows), the sweep, the new dictionary; stream S (the engine's): the later
    stages of the M-step (Gram, all-reduce, U rows -- each announced to the running sweep by a flag word), the stop
    rule's verdict on the reduced sums, and the objective (dict_learning.py:39) -- so the step's dependent chain is
    E-step -> head -> sweep -> Lipschitz, and the chain carries no event record and no copy: S is started by a wave
    that polls a word in device memory (raised by the head's U rows), M is ordered behind S where it changes the
    dictionary S reads (a word the writing launch waits for itself / an event that completed a step ago), and the host
    POLLS the verdict's and the sweep's words in pinned memory at its one wait per step.  On one GPU the E-step's stop
    rule is judged on S too (LASSO_SOLVE_DEFER_VERDICT): nothing on M needs its outcome.  With several ranks the objective's two sums ride in the NEXT step's message (one small
    all-reduce flushes the last step's): a message per stage, the first on the critical chain.  Shapes without a
    pipelined M-step (e.g. 8 x 8 patches: d = 64) keep lasso_gram_accumulate + the sweep on M, with the dictionary
    DOUBLE-BUFFERED: the sweep writes the new dictionary into the other buffer (lasso_dict_sweep_async_to), so it is
    enqueued BEFORE the step's host wait (a step that has to be repeated keeps the old buffer and drops the other), and
    the objective is enqueued on S AFTER that wait (behind the next E-step's launches) -- the host has then seen the
    E-step's verdict, so S needs no device-side start signal, and nothing on M waits for it: the old dictionary it reads
    is only overwritten by the NEXT step's sweep.  Batches beyond 8192 rows per rank of a pipelinable dictionary take this
    form as well (`pipeline=False`), with the objective held back until the sweep has started."""
    import torch as _t
    world, rank = _world(group)
    multi = _sharded(group)
    n_local, d = X.shape
    k = weight.shape[1]
    dev = X.device
    tol = float(solver_kwargs.get('tol', 1e-5))
    M = _t.cuda.current_stream(dev)
    S = engine.side_stream()
    import os
    stages = engine.mstep_pipe_stages(d, k) if (pipeline and hasattr(engine, 'mstep_pipe_stages')
                                                and os.environ.get("LASSO_EM_PIPELINE", "1") != "0") else []
    pipe = len(stages) > 0
    # ONE buffer per step's messages: [A | B] (pipelined: one matrix [k][k + d]; else A [k][k] then B [k][d]) and the
    # tail [sum r^2, sum |z| of the PREVIOUS step | this step's stop-rule sums]
    buf = _t.zeros(k * (k + d) + 2 + ndelta, dtype=_t.float32, device=dev)
    tail = buf[k * (k + d):k * (k + d) + 2]
    dtail = buf[k * (k + d) + 2:]
    AB = buf[:k * (k + d)].view(k, k + d) if pipe else None
    ws = engine.mstep_pipe_workspace(n_local, d, k) if pipe else None
    ev_S = _t.cuda.Event()
    can_defer = (not multi and os.environ.get("LASSO_EM_DEFER_VERDICT", "1") != "0"
                 and getattr(engine, 'defer_verdict', False))
    defer_kw = dict(defer_verdict=True) if can_defer else {}
    # two words in device memory that count up over the engine's lifetime (one allocation, no clearing per loop):
    # [0] raised by the Gram launch of a step ("the E-step has completed"), [1] by the launch in front of a sweep
    words, count = engine.signal_words() if hasattr(engine, 'signal_words') else (None, None)
    sig = words[0:1] if can_defer and not pipe else None
    # a sweep of co-operating workgroups (not the one-workgroup sweep of small dictionaries) leaves most of the chip idle
    # for hundreds of microseconds: the objective is held back until it starts (a word a launch in front of it raises)
    # instead of running beside the Gram product, with which it competes for HBM
    gate_obj = not pipe and not (d <= 64 and k <= 256) and words is not None
    sig2 = words[1:2] if gate_obj else None
    # "cur": the dictionary in force; "alt" (shapes without a pipelined M-step): the buffer the running step's sweep fills
    state = {"seq": 0, "prev_sums": None, "prev_index": -1, "cur": weight,
             "alt": None if pipe else _t.empty_like(weight)}
    Z0, Zlast, deferred = None, None, None

    def repair(mask, ndeg, Zprev):
        cand = draw_directions(d, ndeg).to(weight.device)       # every rank advances its generator alike
        if multi:
            _broadcast(cand, group)                              # ... and uses rank 0's directions
        engine.fill_degenerate(state["cur"], mask, cand, False)                           # :93-96
        if Zprev is not None and Zprev.shape[0] > 0:
            engine.zero_columns(Zprev, mask)                                              # :98

    def fill_tail(pending):
        """(on the stream of the message that carries the tail) the previous step's objective sums, this step's deltas"""
        if state["prev_sums"] is not None:
            tail.copy_(state["prev_sums"])
        else:
            tail.zero_()
        if ndelta:
            if pending is not None:
                dtail.copy_(pending.deltas)
            else:
                dtail.zero_()

    def after_tail(i):
        """(behind the all-reduce of the tail) the previous step's loss from the summed pair"""
        j = state["prev_index"]
        if j >= 0:
            losses[j] = (0.5 * tail[0] + alpha * tail[1]) / n_total

    # beside the sweep the objective runs on a quarter of the chip: four times as long (it has the time) and 8 us less
    # disturbance of the sweep's memory round trips (measured: 64 of 256 workgroups; 32 make the step wait for it)
    cap = 64 if pipe else 0

    def objective(Z, i, D=None):
        """(on S) dict_learning.py:39 for this step's code with the OLD dictionary"""
        D = state["cur"] if D is None else D
        if n_local == 0:
            sums = _t.zeros(2, dtype=_t.float64, device=dev)
        elif multi:
            _, sums = engine.objective_sums(X, Z, D, alpha, max_workgroups=cap)
        else:
            engine.objective_sums(X, Z, D, alpha, loss_out=losses[i], max_workgroups=cap)
            sums = None
        state["next_sums"] = sums

    def flush_objective():
        """(double-buffered form) the objective of the last accepted step, enqueued on S -- the host has seen that step's
        verdict, so S needs no start signal; it is put off until the NEXT E-step has been enqueued on M (its launches
        are what the host must get out before the running sweep ends) unless somebody reads the loss first"""
        job = state.pop("objective_job", None)
        if job is None:
            return
        with _t.cuda.stream(S):
            if job[3]:
                engine.stream_wait_word(sig2.data_ptr(), job[3], False)      # "that step's sweep has started"
            objective(*job[:3])
            ev_S.record(S)
        if multi:
            state["prev_sums"], state["prev_index"] = state["next_sums"], job[1]

    def produce(Z, pending, i):
        """Enqueue everything of step i between the E-step and the new dictionary.  Returns a callable to be called once
        the host has seen the verdict; it gives the sweep's handle (() -> (mask, ndeg)).  Pipelined: that call enqueues
        the launch that writes the dictionary.  Otherwise the sweep has been enqueued already, into state["alt"]."""
        sharded = multi and pending is not None
        if pipe:
            state["seq"] += 1
            seq = state["seq"]
            last = len(stages) - 1

            def stage(s_):
                lo, hi = stages[s_]
                engine.pipe_gram(Z, X, AB, s_, ws)               # (a rank without rows: zeros, and the flag words cleared)
                if multi:
                    if s_ == last:
                        fill_tail(pending)
                        _all_reduce(buf[lo * (k + d):], group)           # the last stage's rows + the tail: contiguous
                        after_tail(i)
                    else:
                        _all_reduce(buf[lo * (k + d):hi * (k + d)], group)
                engine.pipe_rows(AB, weight, n_local, s_, ws, seq=seq)

            stage(0)                                                     # the head: on the step's dependent chain
            with _t.cuda.stream(S):
                engine.pipe_wait(n_local, d, k, seq, ws)                 # (a wave that polls the head's word: no event on M)
                if getattr(pending, 'deferred', False):                  # the E-step's stop rule (its kernels are long done)
                    pending.launch_verdict(gate=(engine.pipe_head_word(n_local, d, k, ws), seq))
                    if stats is not None:
                        stats['deferred_verdicts'] = stats.get('deferred_verdicts', 0) + 1
                for s_ in range(1, last + 1):
                    stage(s_)
                if sharded:
                    pending.judge(dtail, n_total)                        # the verdict, mirrored into pinned memory
                objective(Z, i)
                engine.pipe_signal(n_local, d, k, seq, ws)               # "S has read the old dictionary"
                ev_S.record(S)
            mask = engine.pipe_sweep(AB, weight, n_local, 1e-10, False, ws)

            def finish():
                # (the kernel that writes the dictionary waits for S's word itself: no cross-stream event on M's chain)
                return engine.pipe_finish(weight, n_local, 1e-10, False, mask, ws, wait_seq=seq)
            return finish
        # no pipelined form: Gram product and message on M, then the sweep into the OTHER dictionary buffer -- enqueued here,
        # before the host has seen the verdict (the objective follows on S after the host wait, see the loop)
        A = buf[:k * k].view(k, k)
        B = buf[k * k:k * k + k * d].view(k, d)
        if n_local > 0 and getattr(pending, 'deferred', False):
            count[0] += 1
            state["sig"] = count[0]
            engine.gram(Z, X, buf, started=(sig, state["sig"]))          # its first launch raises the word as it starts
            with _t.cuda.stream(S):
                engine.stream_wait_word(sig.data_ptr(), state["sig"], False)
                pending.launch_verdict(gate=(sig.data_ptr(), state["sig"]))
            if stats is not None:
                stats['deferred_verdicts'] = stats.get('deferred_verdicts', 0) + 1
        elif n_local > 0:
            engine.gram(Z, X, buf)
        else:
            buf[:k * (k + d)].zero_()
        # S's last objective (enqueued a step ago: long complete) wrote the sums this message carries and read the buffer
        # this sweep overwrites
        M.wait_event(ev_S)
        if multi:
            fill_tail(pending)
            _all_reduce(buf, group)
            after_tail(i)
            if sharded:
                pending.judge(dtail, n_total)
        if gate_obj:
            count[1] += 1
            state["sig2"] = count[1]
        handle = engine.sweep_begin(A, B, state["cur"], 1e-10, False, out=state["alt"],
                                    started=(sig2, state["sig2"]) if gate_obj else None)
        return lambda: handle

    def encode_sync():
        return sharded_encode(engine, X, state["cur"], alpha, Z0, group=group, n_global=n_total, **solver_kwargs)

    i = 0
    while i < steps:
        # ---- E-step (dict_learning.py:38), enqueued without a wait
        D = state["cur"]
        if not multi:
            # (the stop rule's launch goes to S -- behind a wave that polls a word M's next launch raises -- where the
            # engine can leave it out of the solve: nothing of the step's chain on M needs its outcome)
            Z, pending = engine.encode_begin(X, D, alpha, Z0, **dict(begin_kwargs, **defer_kw))
        elif n_local == 0:
            Z = Z0 if Z0 is not None else X.new_zeros(0, k)
            pending = _EmptyShardPending(ndelta, k, tol, dev)
        else:
            if Z0 is not None and Z0.device != dev:
                Z0 = Z0.to(dev)
            began = engine.encode_begin_sharded(X, D, alpha, Z0, **solver_kwargs)
            if began is None:
                raise RuntimeError("encode_begin_sharded refused arguments sharded_async_ok accepted")
            Z, pending = began
        flush_objective()        # (the previous step's; before this step's message, which carries its sums)
        if stats is not None:
            stats['overlapped_steps'] = stats.get('overlapped_steps', 0) + 1
            if pipe:
                stats['pipelined_steps'] = stats.get('pipelined_steps', 0) + 1
            else:
                stats['speculative_sweeps'] = stats.get('speculative_sweeps', 0) + 1
        finish = produce(Z, pending, i)
        # ---- the step's ONE host wait: the previous sweep's count of degenerate atoms, this E-step's verdict
        if deferred is not None:
            mask, ndeg = deferred()
            deferred = None
            if ndeg:     # rare: an atom degenerated in the previous sweep -- repair, redo this step
                if pending is not None:
                    pending()
                M.wait_event(ev_S)
                _t.cuda.current_stream(dev).synchronize()
                repair(mask, ndeg, Zlast)
                continue
        if pending is not None and not pending():
            # the stop rule fired before the last iteration (or the in-kernel rule gave up): the same rule on the
            # chunked path, then the step's products once more for the right code.  Nothing of the speculated step
            # has touched the dictionary in force (pipelined: finish() was not called; else the sweep wrote the other
            # buffer, which the repeated sweep overwrites).
            M.wait_event(ev_S)
            _t.cuda.current_stream(dev).synchronize()
            if multi:
                Z = encode_sync()
            else:
                Z = engine.encode(X, D, alpha, Z0, **dict(solver_kwargs, stop_mode='chunked'))
            finish = produce(Z, None, i)
            if stats is not None:
                stats['replayed_steps'] = stats.get('replayed_steps', 0) + 1
        if not pipe:
            # the host has seen this E-step's verdict: Z is final.  The objective reads the dictionary of THIS step, which
            # stays untouched until the next step's sweep (ordered behind ev_S) refills it: see flush_objective()
            state["objective_job"] = (Z, i, state["cur"], state.get("sig2", 0) if gate_obj else 0)
        elif multi:
            state["prev_sums"], state["prev_index"] = state["next_sums"], i
        if persist:
            Z0 = Z                                                                        # :40-41
        deferred = finish()                                                               # :44-45
        if not pipe:
            state["cur"], state["alt"] = state["alt"], state["cur"]
        Zlast = Z
        if bar is not None:
            flush_objective()
            ev_S.synchronize()
            j = i if not multi else state["prev_index"] - 1            # (several ranks: a loss is known a step later)
            bar.set_postfix(loss=losses[j].item() if j >= 0 else float('nan'))            # :50
            bar.update(1)
        i += 1
    flush_objective()
    M.wait_event(ev_S)
    if multi and state["prev_index"] >= 0:           # flush: the last step's objective sums
        t2 = state["prev_sums"].to(_t.float32).clone()
        _all_reduce(t2, group)
        losses[state["prev_index"]] = (0.5 * t2[0] + alpha * t2[1]) / n_total
    if deferred is not None:
        mask, ndeg = deferred()
        if ndeg:
            repair(mask, ndeg, Zlast)
    if state["cur"] is not weight:                   # (an odd number of accepted sweeps: the caller's tensor is the other buffer)
        weight.copy_(state["cur"])
    if bar is not None:
        bar.close()
    return weight, losses


def dict_learning_sharded(X_shard, n_components, alpha=1.0, constrained=True, persist=False,
                          lambd=1e-2, steps=60, progbar=False, init_weight=None, group=None,
                          engine=None, **solver_kwargs):
    """Multi-GPU dict_learning: call on every rank with its row shard of X (one process
    per GPU).  The initial dictionary is drawn on rank 0 exactly like the reference
    (orthogonal_ + normalisation on the CPU generator) and broadcast."""
    from .engine import HipEngine
    engine = engine or HipEngine()
    world, rank = _world(group)
    d = X_shard.shape[1]
    if init_weight is None:
        weight = torch.empty(d, n_components)
        torch.nn.init.orthogonal_(weight)
        if constrained:
            weight = torch.nn.functional.normalize(weight, dim=0)
    else:
        weight = init_weight.detach().clone()
    Xd = engine.to_device(X_shard)
    weight = engine.to_device(weight).clone()
    if _sharded(group):
        _broadcast(weight, group)
    return em_loop(engine, Xd, weight, alpha, constrained=constrained, persist=persist,
                   lambd=lambd, steps=steps, progbar=progbar, solver_kwargs=solver_kwargs,
                   group=group)
