/* Plain-C fp64 restatement of the hot path -- TEST INFRASTRUCTURE ONLY (never linked
 * into the product).  It is an independent ground truth for small cases: the torch
 * oracle (oracle/lasso_oracle.py) reproduces the reference's fp32 ATen arithmetic, this
 * file evaluates the same algorithm in double precision with naive loops, so the two
 * must agree to fp32 round-off.  Pinned against tests/golden/small_cases.npz (outputs of
 * the real reference) in tests/test_oracle_c.py.
 *
 * Follows: lasso/linear/solvers/ista.py:57-104 (fista), :17-54 (line search),
 *          lasso/linear/dict_learning.py:10-13 (loss), :82-101 (atom sweep).
 * Layout: X [n][d], W [d][k] (atoms = columns), Z [n][k], row-major.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double soft(double v, double lam) { /* ATen softshrink, ista.py:90 */
  return v > lam ? v - lam : (v < -lam ? v + lam : 0.0);
}

/* r = p W^T - x  [n][d] ; returns 0.5*||r||^2   (ista.py:22-23 / :72) */
static double residual(const double* p, const double* X, const double* W, double* r, int n, int d, int k) {
  double f = 0.0;
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < d; ++a) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += p[(size_t)i * k + j] * W[(size_t)a * k + j];
      s -= X[(size_t)i * d + a];
      r[(size_t)i * d + a] = s;
      f += 0.5 * s * s;
    }
  return f;
}

/* g = r W  [n][k]   (ista.py:24 / :73) */
static void gradient(const double* r, const double* W, double* g, int n, int d, int k) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int a = 0; a < d; ++a) s += r[(size_t)i * d + a] * W[(size_t)a * k + j];
      g[(size_t)i * k + j] = s;
    }
}

/* Returns the number of iterations executed; trials_out (nullable) gets the line-search
 * trial count of every iteration. */
int oracle_fista(const double* X, const double* W, const double* z0, double* z_out, int n, int d, int k,
                 double alpha, double lr0, int fast, int maxiter, double tol, int backtrack, double eta,
                 int* trials_out) {
  const size_t nk = (size_t)n * k;
  double* z = malloc(nk * sizeof(double));
  double* y = malloc(nk * sizeof(double));
  double* zn = malloc(nk * sizeof(double));
  double* g = malloc(nk * sizeof(double));
  double* r = malloc((size_t)n * d * sizeof(double));
  memcpy(z, z0, nk * sizeof(double));
  memcpy(y, z0, nk * sizeof(double));
  const double budget = (double)nk * tol;                       /* ista.py:64 */
  double t = 1.0;
  int it = 0;
  for (; it < maxiter; ++it) {
    const double* p = fast ? y : z;                              /* :84 */
    const double f0 = residual(p, X, W, r, n, d, k);
    gradient(r, W, g, n, d, k);
    double lr = lr0;
    int trials = 0;
    for (;;) {
      for (size_t e = 0; e < nk; ++e) zn[e] = soft(p[e] - lr * g[e], alpha * lr);   /* :40 / :90 */
      ++trials;
      if (!backtrack) break;
      double l1 = 0.0, dzg = 0.0, dz2 = 0.0;
      for (size_t e = 0; e < nk; ++e) {
        const double dz = zn[e] - p[e];
        l1 += fabs(zn[e]); dzg += dz * g[e]; dz2 += dz * dz;
      }
      const double F = residual(zn, X, W, r, n, d, k) + alpha * l1;               /* :26-28 */
      const double Q = f0 + dzg + (0.5 / lr) * dz2 + alpha * l1;                   /* :30-35 */
      if (F <= Q) break;                                                           /* :45 */
      if (trials >= 1000) {                                                        /* :48-52 */
        lr = lr0;
        for (size_t e = 0; e < nk; ++e) zn[e] = soft(p[e] - lr * g[e], alpha * lr);
        break;
      }
      lr /= eta;                                                                   /* :47 */
    }
    if (trials_out) trials_out[it] = trials;
    double delta = 0.0;
    for (size_t e = 0; e < nk; ++e) delta += fabs(z[e] - zn[e]);                   /* :93 */
    if (delta <= budget) { memcpy(z, zn, nk * sizeof(double)); ++it; break; }
    if (fast) {                                                                    /* :98-101 */
      const double tn = (1.0 + sqrt(1.0 + 4.0 * t * t)) / 2.0;
      const double c = (t - 1.0) / tn;
      for (size_t e = 0; e < nk; ++e) y[e] = zn[e] + c * (zn[e] - z[e]);
      t = tn;
    }
    memcpy(z, zn, nk * sizeof(double));
  }
  memcpy(z_out, z, nk * sizeof(double));
  free(z); free(y); free(zn); free(g); free(r);
  return it;
}

/* (0.5*||X - Z W^T||^2 + alpha*||Z||_1)/n   (dict_learning.py:10-13) */
double oracle_lasso_loss(const double* X, const double* Z, const double* W, int n, int d, int k, double alpha) {
  double* r = malloc((size_t)n * d * sizeof(double));
  double f = residual(Z, X, W, r, n, d, k), l1 = 0.0;
  for (size_t e = 0; e < (size_t)n * k; ++e) l1 += fabs(Z[e]);
  free(r);
  return (f + alpha * l1) / n;
}

/* Residual-form Gauss-Seidel atom sweep (dict_learning.py:82-101).  D [d][k] and Z are
 * updated in place; `fresh` [k][d] supplies the replacement direction of atom j if it
 * degenerates (the reference draws it from torch's RNG, :93).  Returns #degenerate. */
int oracle_update_dict(double* D, const double* X, double* Z, int n, int d, int k, int positive, double eps,
                       const double* fresh) {
  double* R = malloc((size_t)n * d * sizeof(double));
  int ndeg = 0;
  for (int i = 0; i < n; ++i)                                                      /* :82 */
    for (int a = 0; a < d; ++a) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += Z[(size_t)i * k + j] * D[(size_t)a * k + j];
      R[(size_t)i * d + a] = X[(size_t)i * d + a] - s;
    }
  for (int j = 0; j < k; ++j) {
    for (int i = 0; i < n; ++i)                                                    /* :85 */
      for (int a = 0; a < d; ++a) R[(size_t)i * d + a] += Z[(size_t)i * k + j] * D[(size_t)a * k + j];
    double nrm = 0.0;
    for (int a = 0; a < d; ++a) {                                                  /* :86 */
      double s = 0.0;
      for (int i = 0; i < n; ++i) s += Z[(size_t)i * k + j] * R[(size_t)i * d + a];
      if (positive && s < 0.0) s = 0.0;                                            /* :87-88 */
      D[(size_t)a * k + j] = s;
      nrm += s * s;
    }
    nrm = sqrt(nrm);                                                               /* :91 */
    if (nrm < eps) {                                                               /* :92-98 */
      double fn = 0.0;
      for (int a = 0; a < d; ++a) {
        double s = fresh ? fresh[(size_t)j * d + a] : (a == j % d ? 1.0 : 0.0);
        if (positive && s < 0.0) s = 0.0;
        D[(size_t)a * k + j] = s;
        fn += s * s;
      }
      fn = sqrt(fn);
      for (int a = 0; a < d; ++a) D[(size_t)a * k + j] /= fn;
      for (int i = 0; i < n; ++i) Z[(size_t)i * k + j] = 0.0;
      ++ndeg;
    } else {
      for (int a = 0; a < d; ++a) D[(size_t)a * k + j] /= nrm;                     /* :100 */
      for (int i = 0; i < n; ++i)                                                  /* :101 */
        for (int a = 0; a < d; ++a) R[(size_t)i * d + a] -= Z[(size_t)i * k + j] * D[(size_t)a * k + j];
    }
  }
  free(R);
  return ndeg;
}
