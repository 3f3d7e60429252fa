/* lasso_hip.h -- C ABI of the MI355X-native ISTA/FISTA sparse-encode engine.
 *
 * Drop-in boundary for the hot path of rfeinman/pytorch-lasso.
 * The reference has no FFI: its boundary is two Python call sites,
 *   lasso/linear/sparse_encode.py:62-63   ista(x, z0, weight, alpha, **kwargs)
 *   lasso/linear/dict_learning.py:38,39,45,47   E-step, objective, M-step
 * Each entry point below names the reference function it replaces.
 *
 * Conventions
 *   - every pointer named *_dev is DEVICE memory (HBM) owned by the caller; the
 *     library allocates nothing persistent and never frees caller memory;
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS:
 *       X [n][d] (ldx), W [d][k] (ldw, atoms are columns), Z [n][k] (ldz);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is
 *     enqueued on it; a call only blocks the host where a host-visible result
 *     (iteration count, stop decision) is produced -- stated per function;
 *   - return value: lasso_status (0 = ok).  No exceptions cross the boundary;
 *     lasso_hip_last_error() gives the detail string for the calling thread;
 *   - dtype: LASSO_F32 only in this revision (LASSO_BF16 is reserved).
 */
#ifndef LASSO_HIP_H_
#define LASSO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASSO_HIP_ABI_VERSION 1

typedef enum {
  LASSO_OK = 0,
  LASSO_ERR_BAD_ARG = 1,      /* reference raises ValueError / AssertionError          */
  LASSO_ERR_UNSUPPORTED = 2,  /* shape or dtype outside what the HIP path implements   */
  LASSO_ERR_WORKSPACE = 3,    /* workspace missing or too small                        */
  LASSO_ERR_HIP = 4,          /* HIP runtime error (no device, launch failure, ...)    */
  LASSO_WARN_LINESEARCH = 5   /* backtracking failed, reverted to lr0 (ista.py:48-52)  */
} lasso_status;

typedef enum { LASSO_F32 = 0, LASSO_BF16 = 1 } lasso_dtype;

/* How the global stopping rule of ista.py:93 is evaluated. */
typedef enum {
  LASSO_STOP_GLOBAL = 0,  /* exact reference rule: sum over the whole batch <= n*k*tol */
  LASSO_STOP_NONE = 1     /* run exactly maxiter iterations (same as tol = 0)          */
} lasso_stop_mode;

int lasso_hip_abi_version(void);
const char* lasso_hip_status_string(int status);
const char* lasso_hip_last_error(void);

/* Number of compute units of the current HIP device (0 and LASSO_ERR_HIP if none). */
int lasso_hip_device_cus(int* cus_out);

/* ---- FISTA / ISTA solve: replaces lasso/linear/solvers/ista.py:57-104 ------------
 *   min_z 0.5*||z W^T - x||^2 + alpha*||z||_1,  fixed step `lr`.
 *   z0_dev == NULL means zero initialisation (sparse_encode.py:22-23).
 *   fast != 0 -> FISTA (Nesterov momentum, ista.py:98-101), else ISTA.
 *   tol is the reference's RELATIVE tolerance; the absolute budget n*k*tol is
 *   formed inside (ista.py:64).  tol == 0 or stop_mode == LASSO_STOP_NONE runs
 *   exactly `maxiter` iterations with no host synchronisation.
 *   Otherwise the call synchronises `stream` once per chunk of iterations to
 *   evaluate the global stop rule exactly (speculate-and-replay).
 *   iters_out / last_delta_out (HOST pointers, nullable): iterations executed and
 *   the last evaluated sum|z - z_next| (only when the stop rule is active).
 *   x, W, z0 are never written; z_out may alias z0.
 */
size_t lasso_fista_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype,
                                   int maxiter, double tol, int stop_mode);

int lasso_fista_solve(const void* x_dev, int64_t ldx,
                      const void* w_dev, int64_t ldw,
                      const void* z0_dev, int64_t ldz0,
                      void* z_out_dev, int64_t ldz,
                      int64_t n, int64_t d, int64_t k, int dtype,
                      double alpha, double lr, int fast, int maxiter,
                      double tol, int stop_mode,
                      int32_t* iters_out, float* last_delta_out,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- building blocks for multi-GPU / custom drivers ---------------------------------
 * lasso_fista_prepare packs W into the padded layouts the kernels stream
 * (W [256][Kp] and W^T [Kp][256]) at the start of `workspace_dev`.
 * lasso_fista_run executes `iters` iterations numbered it0 .. it0+iters-1 of the
 * momentum schedule from state (z_in, y_in) to (z_out, y_out) and writes
 * delta_dev[i] = sum over THIS shard of |z - z_next| for each executed iteration
 * (device array of `iters` floats, nullable).  No host synchronisation.
 * y_in_dev == NULL means y = z_in (start of a solve); y_out_dev may be NULL.
 */
int lasso_fista_prepare(const void* w_dev, int64_t ldw, int64_t d, int64_t k, int dtype,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

int lasso_fista_run(const void* x_dev, int64_t ldx,
                    const void* z_in_dev, int64_t ldz_in,
                    const void* y_in_dev, int64_t ldy_in,
                    void* z_out_dev, int64_t ldz_out,
                    void* y_out_dev, int64_t ldy_out,
                    int64_t n, int64_t d, int64_t k, int dtype,
                    double alpha, double lr, int fast, int it0, int iters,
                    float* delta_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LASSO_HIP_H_ */
