/* A host in plain C -- no Python, no torch: hipMalloc'ed buffers, include/lasso_hip.h, and the
 * C oracle (oracle/lasso_oracle.c, fp64) as the checker.  What a non-Python maintainer's binding
 * of the path looks like; built and run by tests/test_c_host_gpu.py.
 *
 *   lasso_fista_solve   fixed step and backtracking line search  (ista.py:57-104, 17-54)
 *   lasso_objective     lasso_loss                               (dict_learning.py:10-13)
 *   lasso_gram_accumulate + lasso_ridge_solve   update_dict_ridge (dict_learning.py:106-123)
 * Exit code 0 = every check inside its tolerance. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "lasso_hip.h"

int oracle_fista(const double* X, const double* W, const double* z0, double* z_out, int n, int d, int k,
                 double alpha, double lr0, int fast, int maxiter, double tol, int backtrack, double eta,
                 int* trials_out);
double oracle_lasso_loss(const double* X, const double* Z, const double* W, int n, int d, int k, double alpha);

#define HIP_OK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); return 2; } } while (0)
#define LASSO_OKAY(c) do { int s_ = (c); if (s_ != LASSO_OK && s_ != LASSO_WARN_LINESEARCH) { fprintf(stderr, "%s: %s: %s\n", #c, lasso_hip_status_string(s_), lasso_hip_last_error()); return 3; } } while (0)

static unsigned long long rng = 88172645463325252ull;
static double unif(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; }
static double gauss(void) { return sqrt(-2.0 * log(unif() + 1e-300)) * cos(6.283185307179586 * unif()); }

int main(void) {
  const int n = 300, d = 48, k = 160, maxiter = 12;
  const double alpha = 0.3, lr = 0.08;
  float* X = malloc(sizeof(float) * n * d); float* W = malloc(sizeof(float) * d * k);
  double* Xd = malloc(sizeof(double) * n * d); double* Wd = malloc(sizeof(double) * d * k);
  double* z0d = calloc((size_t)n * k, sizeof(double)); double* zref = malloc(sizeof(double) * n * k);
  float* Z = malloc(sizeof(float) * n * k);
  for (int j = 0; j < k; ++j) {                    /* unit-norm atoms */
    double nrm = 0.0;
    for (int r = 0; r < d; ++r) { Wd[r * k + j] = gauss(); nrm += Wd[r * k + j] * Wd[r * k + j]; }
    for (int r = 0; r < d; ++r) { W[r * k + j] = (float)(Wd[r * k + j] / sqrt(nrm)); Wd[r * k + j] = W[r * k + j]; }
  }
  for (int i = 0; i < n * d; ++i) { X[i] = (float)gauss(); Xd[i] = X[i]; }

  if (lasso_hip_abi_version() != LASSO_HIP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  float *dX, *dW, *dZ; void* ws;
  HIP_OK(hipMalloc((void**)&dX, sizeof(float) * n * d));
  HIP_OK(hipMalloc((void**)&dW, sizeof(float) * d * k));
  HIP_OK(hipMalloc((void**)&dZ, sizeof(float) * n * k));
  HIP_OK(hipMemcpy(dX, X, sizeof(float) * n * d, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW, W, sizeof(float) * d * k, hipMemcpyHostToDevice));
  int failures = 0;
  for (int backtrack = 0; backtrack <= 1; ++backtrack) {
    const double lr0 = backtrack ? 1.0 : lr;
    const size_t wsb = lasso_fista_workspace_bytes(n, d, k, LASSO_F32, maxiter, 0.0, LASSO_STOP_GLOBAL, backtrack);
    HIP_OK(hipMalloc(&ws, wsb));
    int32_t iters = 0, trials[12]; float last = 0.f, obj = 0.f;
    LASSO_OKAY(lasso_fista_solve(dX, d, dW, k, NULL, 0, dZ, k, n, d, k, LASSO_F32, alpha, lr0, 1, maxiter, 0.0,
                                 LASSO_STOP_GLOBAL, backtrack, 1.5, &iters, &last, backtrack ? trials : NULL, NULL, NULL,
                                 &obj, ws, wsb, NULL));
    HIP_OK(hipMemcpy(Z, dZ, sizeof(float) * n * k, hipMemcpyDeviceToHost));
    int rtrials[12];
    const int rit = oracle_fista(Xd, Wd, z0d, zref, n, d, k, alpha, lr0, 1, maxiter, 0.0, backtrack, 1.5, rtrials);
    double worst = 0.0;
    for (int i = 0; i < n * k; ++i) { const double e = fabs((double)Z[i] - zref[i]); if (e > worst) worst = e; }
    const double robj = oracle_lasso_loss(Xd, zref, Wd, n, d, k, alpha);
    int trace_ok = 1;
    if (backtrack) for (int i = 0; i < maxiter; ++i) trace_ok = trace_ok && trials[i] == rtrials[i];
    printf("solve backtrack=%d: iterations %d (oracle %d), max|dz| %.2e, objective %.6f (oracle %.6f), trace %s\n",
           backtrack, iters, rit, worst, obj, robj, trace_ok ? "equal" : "DIFFERENT");
    if (iters != rit || worst > 5e-5 || fabs(obj - robj) > 2e-6 * robj || !trace_ok) ++failures;
    HIP_OK(hipFree(ws));
  }
  /* unconstrained M-step: V = ((Z^T Z + lam n I)^-1 Z^T X)^T against a normal-equations residual check in fp64 */
  {
    float *dA, *dB, *dV; void *gws, *rws;
    HIP_OK(hipMalloc((void**)&dA, sizeof(float) * k * k));
    HIP_OK(hipMalloc((void**)&dB, sizeof(float) * k * d));
    HIP_OK(hipMalloc((void**)&dV, sizeof(float) * d * k));
    const size_t gb = lasso_gram_workspace_bytes(n, d, k), rb = lasso_ridge_workspace_bytes(d, k);
    HIP_OK(hipMalloc(&gws, gb)); HIP_OK(hipMalloc(&rws, rb));
    LASSO_OKAY(lasso_gram_accumulate(dZ, k, dX, d, n, d, k, LASSO_F32, dA, dB, gws, gb, NULL));
    int32_t info = -1;
    const double lam_n = 1e-2 * n;
    LASSO_OKAY(lasso_ridge_solve(dA, dB, dV, k, d, k, LASSO_F32, lam_n, &info, rws, rb, NULL));
    float* V = malloc(sizeof(float) * d * k);
    HIP_OK(hipMemcpy(V, dV, sizeof(float) * d * k, hipMemcpyDeviceToHost));
    /* residual of (A + lam I) V^T = B with A, B in fp64 from the returned code */
    double worst = 0.0, scale = 0.0;
    for (int a = 0; a < k; ++a)
      for (int c = 0; c < d; ++c) {
        double lhs = lam_n * V[c * k + a], rhs = 0.0;
        for (int b = 0; b < k; ++b) {
          double Aab = 0.0;
          for (int i = 0; i < n; ++i) Aab += (double)Z[i * k + a] * Z[i * k + b];
          lhs += Aab * V[c * k + b];
        }
        for (int i = 0; i < n; ++i) rhs += (double)Z[i * k + a] * X[i * d + c];
        if (fabs(lhs - rhs) > worst) worst = fabs(lhs - rhs);
        if (fabs(rhs) > scale) scale = fabs(rhs);
      }
    printf("ridge solve: info %d, residual %.2e of %.2e\n", info, worst, scale);
    if (info != 0 || worst > 1e-4 * (scale + 1.0)) ++failures;
  }
  printf(failures ? "FAILED (%d)\n" : "ok\n", failures);
  return failures ? 1 : 0;
}
