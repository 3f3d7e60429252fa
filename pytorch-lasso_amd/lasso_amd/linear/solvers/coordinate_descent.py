"""HIP-backed greedy coordinate descent: same signature, defaults and side effects as
``lasso.linear.solvers.coordinate_descent.coord_descent`` (reference
coordinate_descent.py:5-54)."""
import ctypes as C

import torch

from ... import _native as nat


def _pick_device(*tensors):
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    return torch.device('cuda', torch.cuda.current_device())


def coord_descent(x, W, z0=None, alpha=1.0, maxiter=1000, tol=1e-6, verbose=False,
                  return_info=False):
    """x [n,d], W [d,k], z0 [n,k] or None -> z [n,k] = S_alpha(b) (:52).

    Like the reference, a caller-supplied ``z0`` is updated IN PLACE and ends up holding
    the tracked code (:14,47); ``b`` starts at ``x W`` whatever ``z0`` is (:19).  Every
    row stops on its own once its committed change is <= tol*k (:9,45-48).
    ``verbose`` prints the reference's line per step (:49-50) by stepping the HIP solver
    one step at a time.  ``return_info`` (extension) also returns
    ``dict(max_steps=..., n_active=...)``.  No CPU fallback.
    """
    nat.require_gpu()
    if x.dim() != 2 or W.dim() != 2:
        raise RuntimeError("coord_descent expects 2-D x and W")
    d, k = W.shape
    n, d1 = x.shape
    assert d1 == d                                                  # :8
    if z0 is not None:
        assert z0.shape == (n, k)                                   # :13
    if x.dtype != torch.float32 or W.dtype != torch.float32 or (z0 is not None and z0.dtype != torch.float32):
        raise NotImplementedError("lasso_amd: coord_descent is implemented for float32 tensors")
    out_device = x.device
    dev = _pick_device(x, W, z0)
    xg = x.detach().to(dev).contiguous()
    wg = W.detach().to(dev).contiguous()
    # the tracked z must land in the caller's z0 storage: work on it directly when it is a
    # row-major device tensor, else on a staged copy that is copied back at the end
    z0g = None
    if z0 is not None:
        z0g = z0.detach()
        if z0g.device != dev or z0g.stride(1) != 1 or (n > 1 and z0g.stride(0) < k):
            z0g = z0g.to(dev).contiguous()
    L = nat.lib()
    z = torch.empty((n, k), dtype=torch.float32, device=dev)
    n_active, max_steps = C.c_int32(0), C.c_int32(0)
    if n > 0:
        with torch.cuda.device(dev):
            ws = nat.workspace(dev, L.lasso_cd_workspace_bytes(n, d, k, nat.LASSO_F32), "cd")
            wsp, wsn, st = nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)
            ldz0 = z0g.stride(0) if z0g is not None else 0
            if n == 1 and z0g is not None:
                ldz0 = max(ldz0, k)
            if not verbose:
                nat.check(L.lasso_cd_solve(
                    nat.ptr(xg), xg.stride(0), nat.ptr(wg), wg.stride(0), nat.ptr(z0g), ldz0,
                    nat.ptr(z), z.stride(0), n, d, k, nat.LASSO_F32, float(alpha), int(maxiter),
                    float(tol), C.byref(n_active) if return_info else None,
                    C.byref(max_steps) if return_info else None, wsp, wsn, st))
            else:
                from ...engine import HipEngine
                eng = HipEngine(dev)
                nat.check(L.lasso_cd_prepare(nat.ptr(xg), xg.stride(0), nat.ptr(wg), wg.stride(0),
                                             nat.ptr(z0g), ldz0, n, d, k, nat.LASSO_F32, wsp, wsn, st))
                n_active.value = n
                for i in range(int(maxiter)):                       # :42-50
                    if n_active.value == 0:
                        break
                    nat.check(L.lasso_cd_run(n, d, k, float(alpha), float(tol) * k, 1,
                                             C.byref(n_active), C.byref(max_steps), wsp, wsn, st))
                    nat.check(L.lasso_cd_finish(nat.ptr(z), z.stride(0), None, 0, n, d, k,
                                                float(alpha), wsp, wsn, st))
                    _, sums = eng.objective_sums(xg, z, wg, alpha)
                    s = sums.tolist()
                    print('iter %i - loss: %0.4f' % (i, 0.5 * s[0] + alpha * s[1]))
                nat.check(L.lasso_cd_finish(nat.ptr(z), z.stride(0), nat.ptr(z0g), ldz0, n, d, k,
                                            float(alpha), wsp, wsn, st))
        if z0 is not None and z0g.data_ptr() != z0.data_ptr():
            z0.detach().copy_(z0g)                                   # :47 (in-place on the caller's z0)
    if z.device != out_device:
        z = z.to(out_device)
    if return_info:
        return z, dict(max_steps=max_steps.value, n_active=n_active.value)
    return z
