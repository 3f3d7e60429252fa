"""CPU stand-in for lasso_amd.engine.HipEngine built on the oracle -- TEST ONLY.
Lets the host logic of the EM driver (sharding, collectives, persist/loss ordering,
degenerate-atom RNG protocol) run under gloo on CPU."""
import torch

from oracle import lasso_oracle as orc


class OracleEngine:
    name = "oracle"

    def __init__(self):
        self.device = torch.device("cpu")

    def to_device(self, t):
        return t.detach().contiguous()

    def encode(self, X, W, alpha, z0, **kw):
        return orc.sparse_encode(X, W, alpha, z0, **kw)

    def encode_sharded_backtrack(self, X, W, alpha, z0, lr, fast, maxiter, tol, eta, n_global, all_reduce):
        """The line search on a row shard with the five batch sums of ista.py:23,28,32-35 and the
        stop-rule sum (:93) added over the ranks -- the protocol of lasso_fista_solve_sharded."""
        z = z0.clone()
        y, t = z.clone(), 1.0
        info = dict(trials=[], accepted_lr=[], accepted_f=[], iterations=0, last_delta=float("nan"))
        for _ in range(int(maxiter)):
            p = y if fast else z
            r0 = p @ W.T - X
            g = r0 @ W
            lr_t, ntr = float(lr), 0
            while True:
                zn = orc.soft_threshold(p - lr_t * g, alpha * lr_t)
                r1, dz = zn @ W.T - X, zn - p
                s = torch.stack([r0.pow(2).sum(), r1.pow(2).sum(), zn.abs().sum(), (dz * g).sum(),
                                 dz.pow(2).sum()]).double()
                all_reduce(s)
                s = s.float()
                F = 0.5 * s[1] + alpha * s[2]
                Q = 0.5 * s[0] + s[3] + (0.5 / lr_t) * s[4] + alpha * s[2]
                ntr += 1
                if F <= Q:
                    break
                lr_t /= eta
            d = (z - zn).abs().sum().double().reshape(1)
            all_reduce(d)
            t_next = (1.0 + (1.0 + 4.0 * t * t) ** 0.5) / 2.0
            y = zn + ((t - 1.0) / t_next) * (zn - z) if fast else zn
            z, t = zn, t_next
            info["trials"].append(ntr); info["accepted_lr"].append(lr_t); info["accepted_f"].append(float(F))
            info["iterations"] += 1
            info["last_delta"] = float(d)
            if tol > 0 and float(d) <= n_global * W.shape[1] * tol:
                break
        return z, info

    def lipschitz(self, W):
        return orc.lipschitz_constant(W, "exact")

    def fista_run(self, X, W, z_in, y_in, alpha, lr, fast, it0, iters, want_delta, ws=None, z_out=None):
        coefs = orc.momentum_schedule(it0 + iters) if fast else [0.0] * (it0 + iters)
        z = z_in if z_in is not None else X.new_zeros(X.shape[0], W.shape[1])
        y = y_in if y_in is not None else z
        deltas = []
        for i in range(it0, it0 + iters):
            grad = torch.matmul(torch.matmul(y, W.T) - X, W)
            zn = orc.soft_threshold(y - lr * grad, alpha * lr)
            deltas.append((z - zn).abs().sum())
            y = zn + coefs[i] * (zn - z)
            z = zn
        return z, y, (torch.stack(deltas) if want_delta else None)

    def objective_sums(self, X, Z, W, alpha):
        r = X - Z @ W.T
        sums = torch.stack([r.double().pow(2).sum(), Z.double().abs().sum()])
        return orc.lasso_objective(X, Z, W, alpha), sums

    def gram(self, Z, X, out):
        k, d = Z.shape[1], X.shape[1]
        A = out[:k * k].view(k, k)
        B = out[k * k:k * k + k * d].view(k, d)
        A.copy_(Z.T @ Z)
        B.copy_(Z.T @ X)
        return A, B

    def sweep(self, A, B, D, pool, eps, positive, seed=0):
        used = [0]

        def fresh(j):       # pool None: a placeholder direction, overwritten by fill_degenerate
            v = pool[min(used[0], pool.shape[0] - 1)] if pool is not None else torch.ones(D.shape[0])
            used[0] += 1
            return v
        _, deg = orc.update_dict_gram(D, A.clone(), B.clone(), positive=positive, eps=eps,
                                      fresh_atom=fresh)
        return deg.to(torch.int32), int(deg.sum())

    def fill_degenerate(self, D, mask, pool, positive):
        for i, j in enumerate(torch.nonzero(mask).flatten().tolist()):
            v = pool[i].clone()
            if positive:
                v.clamp_(0, None)
            D[:, j] = v / v.norm()

    def zero_columns(self, Z, mask):
        Z[:, mask.bool()] = 0

    def ridge(self, A, B, lam_n, check=False):
        M = A.clone()
        M.diagonal().add_(lam_n)
        return torch.cholesky_solve(B, torch.linalg.cholesky(M)).T.contiguous()
