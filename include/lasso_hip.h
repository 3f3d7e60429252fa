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
 *   - dtype: LASSO_F32; LASSO_BF16 (x, W, z0, z_out all bf16, leading dimensions in bf16
 *     elements) is accepted by lasso_fista_solve on the fused shapes (bf16-MFMA kernels,
 *     fp32 accumulation and state; fixed step or line search) and LASSO_ERR_UNSUPPORTED
 *     elsewhere;
 *   - shapes: d <= 256 and k <= 1024 run the fused kernels; beyond that lasso_fista_solve
 *     (fixed step, and the line search on fp32 tensors), lasso_objective and
 *     lasso_gram_accumulate take any d, k (unfused MFMA GEMM paths), lasso_dict_sweep
 *     d <= 1024 and k <= 4096, lasso_cd_* k <= 4096; lasso_fista_prepare/_run take any
 *     d, k too (unfused: state in HBM, kernel_hint ignored); lasso_fista_solve_sharded takes
 *     any d, k on fp32 tensors (bf16: fused shapes only).
 */
#ifndef LASSO_HIP_H_
#define LASSO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASSO_HIP_ABI_VERSION 7

typedef enum {
  LASSO_OK = 0,
  LASSO_ERR_BAD_ARG = 1,      /* reference raises ValueError / AssertionError          */
  LASSO_ERR_UNSUPPORTED = 2,  /* shape or dtype outside what the HIP path implements   */
  LASSO_ERR_WORKSPACE = 3,    /* workspace missing or too small                        */
  LASSO_ERR_HIP = 4,          /* HIP runtime error (no device, launch failure, ...)    */
  LASSO_WARN_LINESEARCH = 5,  /* backtracking failed, reverted to lr0 (ista.py:48-52)  */
  LASSO_PENDING = 6,          /* LASSO_SOLVE_ASYNC: enqueued; call lasso_fista_solve_finish */
  LASSO_WARN_ABORTED = 7,     /* lasso_fista_solve_finish: the asynchronous solve must be repeated
                                 with LASSO_STOP_GLOBAL_CHUNKED -- the in-kernel stop rule gave up
                                 (a workgroup was not resident; z_out untouched), or the rule fired
                                 before the end of an asynchronously enqueued chunk (z_out holds a
                                 later iterate)                                            */
  LASSO_PENDING_DEFERRED = 9, /* LASSO_SOLVE_DEFER_VERDICT: enqueued WITHOUT the verdict's launch (lasso_fista_solve_verdict_deferred) */
  LASSO_PENDING_MAPPED = 8    /* LASSO_SOLVE_ASYNC | LASSO_SOLVE_STATUS_MAPPED: enqueued, and the verdict's
                                 four words will be written to the caller's mapped host buffer by the
                                 verdict kernel itself -- no lasso_fista_solve_collect needed          */
} lasso_status;

typedef enum { LASSO_F32 = 0, LASSO_BF16 = 1 } lasso_dtype;

/* How the global stopping rule of ista.py:93 is evaluated. */
typedef enum {
  LASSO_STOP_GLOBAL = 0,  /* exact reference rule: sum over the whole batch <= n*k*tol */
  LASSO_STOP_NONE = 1,    /* run exactly maxiter iterations (same as tol = 0)          */
  LASSO_STOP_GLOBAL_CHUNKED = 2  /* the same exact rule, but always evaluated by the
                             chunked speculate-and-replay path (no in-kernel handshake
                             between workgroups): for callers that share the GPU with
                             other streams / processes; LASSO_STOP_GLOBAL falls back to
                             it by itself when its workgroups are not all resident      */
} lasso_stop_mode;

/* Kernel-selection hint for the fused shapes: OR it into `stop_mode` of lasso_fista_solve, pass
 * it as `kernel_hint` to lasso_fista_run.  AUTO picks by batch size: the tile kernel (one
 * workgroup per 16-row tile, no traffic between workgroups) fills the chip from n ~ 16 x #CUs
 * rows; below that the split-k kernel shares each tile among Kpad/128 workgroups.  Both give
 * bitwise the same code for a row.  SPLITK is a request, not a guarantee (in-place calls,
 * d <= 128 tall tiles and lock-step stop-rule launches with more tiles than groups keep the
 * tile kernel). */
#define LASSO_KERNEL_AUTO 0
#define LASSO_KERNEL_TILE 0x100
#define LASSO_KERNEL_SPLITK 0x200
#define LASSO_KERNEL_UNFUSED 0x300   /* fixed-step fp32 solves: the general-GEMM path (state in HBM) also on the fused shapes */
/* tuning knob: split-k with exactly T = 1, 2 or 4 tiles per workgroup group (default: by cost model) */
#define LASSO_KERNEL_SPLITK_TILES(T) (0x200 | ((T) == 1 ? 0x1000 : (T) == 2 ? 0x2000 : 0x3000))
/* tuning knob (A/B measurements): split-k with the register-gather exchange for every T (the default streams the
 * partials of T >= 2 tiles through an LDS-DMA ring, csrc/fista_splitk.hip) */
#define LASSO_KERNEL_SPLITK_GATHER (0x200 | 0x400)
#define LASSO_KERNEL_MASK 0x3F00
/* OR into stop_mode of lasso_fista_solve (fp32, fixed step): do not wait for the stop rule's
 * outcome.  Returns LASSO_PENDING when the solve was enqueued without a wait -- the single
 * persistent launch with the in-kernel rule, or, with more tiles than resident workgroups, one
 * chunk of maxiter <= 64 iterations whose deltas are judged on the device -- (then iters_out /
 * last_delta_out are not written and lasso_fista_solve_finish / _collect fetch them), or
 * LASSO_OK when the solve completed inside the call (stop rule off, longer chunked solves). */
#define LASSO_SOLVE_ASYNC 0x4000
/* OR into stop_mode together with LASSO_SOLVE_ASYNC when the batch is a ROW SHARD of a larger one
 * (one process per GPU): the stop rule of ista.py:93 sums over the rows of ALL shards, so nothing
 * on this device can judge it alone.  The solve is enqueued as one chunk of maxiter <= 64 iterations
 * (LASSO_ERR_UNSUPPORTED beyond) that leaves this shard's per-iteration sums |z - z_next| in the
 * workspace and returns LASSO_PENDING; the caller all-reduces the `maxiter` floats at
 * lasso_fista_solve_deltas() across the ranks ON THE STREAM (RCCL), enqueues
 * lasso_fista_solve_verdict(n_global, ...) -- the one-thread kernel that applies the rule to the
 * summed vector -- and then lasso_fista_solve_collect() as for any other asynchronous solve.  No
 * host wait anywhere: every rank reads the same verdict ("redo" when the rule fired before the last
 * iteration) at its one synchronisation per EM step. */
#define LASSO_SOLVE_SHARDED 0x8000
/* OR into stop_mode together with LASSO_SOLVE_ASYNC when the rule is NOT expected to fire before `maxiter`
 * (<= 64) iterations -- the E-step of an EM loop: the solve is then always enqueued as one chunk on the plain
 * kernels and judged on the device (the in-kernel rule costs a cross-workgroup exchange per iteration that only
 * pays when it stops a long solve early).  Same words from lasso_fista_solve_finish / _collect; when the rule
 * does fire early they say "redo" (LASSO_WARN_ABORTED) exactly as for any other one-chunk solve. */
#define LASSO_SOLVE_ONE_CHUNK 0x10000
/* OR into stop_mode together with LASSO_SOLVE_ASYNC: `iters_out` is not a plain host int but FOUR int32 words of
 * device-writable host memory (pinned and mapped, e.g. hipHostMalloc / torch pin_memory()).  When the solve is
 * enqueued as one chunk judged on the device (always with LASSO_SOLVE_ONE_CHUNK; otherwise when the batch has more
 * tiles than resident workgroups) the verdict kernel writes {iterations, last delta (float bits), redo, 0} there
 * itself -- the fourth word, "valid" = 1, last and released -- and the call returns LASSO_PENDING_MAPPED: a host that
 * zeroed word 3 before the call POLLS it (or waits for an event recorded behind the call).  No copy launch, no
 * collect call and no event record on the step's dependent chain (round 6: copy + event cost ~10 us of every EM step;
 * an event record between two kernels of a stream is ~5 us by itself).  A solve that takes the in-kernel rule returns LASSO_PENDING as before
 * (the buffer is then not written; collect as usual). */
#define LASSO_SOLVE_STATUS_MAPPED 0x20000
#define LASSO_SOLVE_DEFER_VERDICT 0x40000 /* with STATUS_MAPPED | ONE_CHUNK: see lasso_fista_solve_verdict_deferred */
/* lr: the reference's lr='auto' (ista.py:72-73): 1 / lambda_max(W^T W) computed by the library on
 * the stream (csrc/lipschitz.hip).  The fp32 fixed-step kernels read the step from device
 * memory -- no host round trip; other paths synchronise once to fetch it. */
#define LASSO_LR_AUTO (-1.0)

int lasso_hip_abi_version(void);
const char* lasso_hip_status_string(int status);
const char* lasso_hip_last_error(void);

/* Number of compute units of the current HIP device (0 and LASSO_ERR_HIP if none). */
int lasso_hip_device_cus(int* cus_out);

/* Test hook.  The atom sweep (lasso_dict_sweep) and the Lipschitz squarings (lasso_lipschitz, LASSO_LR_AUTO) are
 * single launches of co-operating workgroups, each followed by a stand-by launch that redoes the work in ONE
 * workgroup when the grid could not get all its workgroups resident (and returns at once otherwise).  on != 0 makes
 * the library enqueue the stand-by form alone, so that a test can hold it against the co-operative one bit for bit
 * without having to saturate the GPU.  Returns the previous setting.  Process-wide; not for production use. */
int lasso_debug_force_standby(int on);

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
 *   trials_out / accepted_lr_out (HOST arrays of `maxiter` entries, nullable; written
 *   for the executed iterations when backtrack != 0): the number of line-search trials
 *   evaluated up to and including the accepted one, and the step size the iteration
 *   used -- the quantities ista.py:43-47 prints with verbose=True.  accepted_f_out (same
 *   shape): F(z_next) = 0.5*||z_next W^T - x||^2 + alpha*||z_next||_1 of the accepted trial
 *   (:28) -- divided by n it is the objective ista() prints for the NEXT iteration
 *   with verbose=True (:66-69,80-81), obtained without an extra pass over the data.
 *   objective_out (HOST float, nullable): mean objective (0.5*||x - z W^T||^2 +
 *   alpha*||z||_1)/n of the returned code, evaluated in fp32 (dict_learning.py:10-13,
 *   ista.py:66-69); synchronises `stream`.
 *   x, W, z0 are never written; z_out may alias z0.
 *   backtrack != 0: Beck-Teboulle backtracking line search (ista.py:17-54) from lr
 *   every outer iteration (the accepted step is discarded, ista.py:87), eta_backtrack
 *   > 1 (else LASSO_ERR_BAD_ARG, the reference's ValueError :18-19).  Line search and
 *   stop rule are global over the batch; the call synchronises `stream` once per outer
 *   iteration.  If no step is accepted within 1000 trials the iteration falls back to lr
 *   and the call returns LASSO_WARN_LINESEARCH after completing (ista.py:48-52).
 */
size_t lasso_fista_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype,
                                   int maxiter, double tol, int stop_mode, int backtrack);
/* Name of the device kernel lasso_fista_solve dispatches this problem to on the current
 * device (the name rocprofv3 --kernel-trace reports; for benchmark / profile bookkeeping). */
const char* lasso_fista_kernel_name(int64_t n, int64_t d, int64_t k, int dtype, int backtrack);

int lasso_fista_solve(const void* x_dev, int64_t ldx,
                      const void* w_dev, int64_t ldw,
                      const void* z0_dev, int64_t ldz0,
                      void* z_out_dev, int64_t ldz,
                      int64_t n, int64_t d, int64_t k, int dtype,
                      double alpha, double lr, int fast, int maxiter,
                      double tol, int stop_mode,
                      int backtrack, double eta_backtrack,
                      int32_t* iters_out, float* last_delta_out,
                      int32_t* trials_out, float* accepted_lr_out, float* accepted_f_out,
                      float* objective_out,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- building blocks for multi-GPU / custom drivers ---------------------------------
 * lasso_fista_prepare packs W into the padded layouts the kernels stream
 * (W [256][Kp] and W^T [Kp][256]) at the start of `workspace_dev` and builds the
 * momentum table of iterations 0 .. maxiter-1 (ista.py:98-99) behind them; the
 * workspace is sized by lasso_fista_workspace_bytes(n, d, k, dtype, maxiter, 0,
 * LASSO_STOP_NONE, 0) with n = the largest batch it will be used with.
 * lasso_fista_run executes `iters` iterations numbered it0 .. it0+iters-1 (< maxiter, the
 * value given to prepare) of the
 * momentum schedule from state (z_in, y_in) to (z_out, y_out) and writes
 * delta_dev[i] = sum over THIS shard of |z - z_next| for each executed iteration
 * (device array of `iters` floats, nullable).  No host synchronisation.
 * y_in_dev == NULL means y = z_in (start of a solve); y_out_dev may be NULL.
 */
int lasso_fista_prepare(const void* w_dev, int64_t ldw, int64_t d, int64_t k, int dtype, int maxiter,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

int lasso_fista_run(const void* x_dev, int64_t ldx,
                    const void* z_in_dev, int64_t ldz_in,
                    const void* y_in_dev, int64_t ldy_in,
                    void* z_out_dev, int64_t ldz_out,
                    void* y_out_dev, int64_t ldy_out,
                    int64_t n, int64_t d, int64_t k, int dtype,
                    double alpha, double lr, int fast, int it0, int iters, int maxiter,
                    int kernel_hint, float* delta_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- Lipschitz constant: replaces _lipschitz_constant, ista.py:8-14 -----------------
 * L = lambda_max(W^T W), deterministic, fp64, computed on the device (Gram of the
 * smaller side + repeated squaring; see csrc/lipschitz.hip).  Synchronises `stream`
 * to return the value through the HOST pointer l_out (the reference returns a python
 * float the same way).  ((double*)workspace_dev)[0] also holds the value on the device.
 */
size_t lasso_lipschitz_workspace_bytes(int64_t d, int64_t k);
int lasso_lipschitz(const void* w_dev, int64_t ldw, int64_t d, int64_t k, int dtype,
                    double* l_out, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- objective: replaces lasso_loss, dict_learning.py:10-13 ------------------------
 * loss = (0.5*||X - Z W^T||^2 + alpha*||Z||_1) / n.  No host synchronisation:
 * loss_dev (device float, nullable) receives the scalar for THIS shard's n;
 * sums_dev (device double[2], nullable) receives {sum r^2, sum |z|} of the shard so a
 * multi-GPU driver can all-reduce them and normalise by the global n.
 */
size_t lasso_objective_workspace_bytes(int64_t n, int64_t d, int64_t k);
int lasso_objective(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                    const void* z_dev, int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype,
                    double alpha, float* loss_dev, double* sums_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);
/* lasso_objective on at most `max_workgroups` workgroups of the fused kernel (0 = no cap): for callers that run it
 * beside latency-bound work of another stream (ABI 7; the EM loop's objective beside the atom sweep). */
int lasso_objective_throttled(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                              const void* z_dev, int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype,
                              double alpha, float* loss_dev, double* sums_dev, int max_workgroups,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- constrained M-step in Gram form: replaces update_dict, dict_learning.py:56-103 --
 * lasso_gram_accumulate: A = Z^T Z [k][k] (ld k), B = Z^T X [k][d] (ld d) of this row
 *   shard (fp32 MFMA).  The caller all-reduces A and B across GPUs (they may be two
 *   slices of one buffer).  The workspace (optional, may be NULL) holds the partial
 *   products of the sample splits that keep all CUs busy; they are summed in a fixed
 *   order, so the result is deterministic either way.
 * lasso_dict_sweep: Gauss-Seidel sweep over the atoms on (A, B), updating the
 *   dictionary D [d][k] (ldd) IN PLACE like the reference (:86,:100).  Atoms with
 *   ||u|| < eps (:92) are replaced by the next unused row of `pool_dev`
 *   [pool_rows][d] (ld pool_ld) normalised to unit length -- the reference draws that
 *   direction from torch's RNG (:93) -- or, if pool_dev is NULL, by a counter-based
 *   N(0,1) vector keyed by (seed, atom).  degenerate_dev [k] (int32, device) is set to
 *   1 for those atoms; ndeg_out (HOST, nullable) receives their count and, when
 *   non-NULL, makes the call synchronise `stream`.
 * lasso_dict_sweep_count: device address (inside the sweep's workspace) of that count, for callers that pass
 *   ndeg_out = NULL and read it at their own synchronisation (4 bytes instead of a reduction over the k flags);
 *   NULL for arguments lasso_dict_sweep would reject.
 * lasso_dict_fill_degenerate: deferred form of that replacement for drivers that want to
 *   draw directions only when an atom actually degenerated (the sweep never READS a
 *   replacement -- the degenerate atom leaves the model): call lasso_dict_sweep with
 *   pool_dev == NULL, and if ndeg > 0 draw ndeg directions [ndeg][d] and pass them here;
 *   the i-th flagged atom (atom order) becomes pool row i, clamped at 0 if `positive`,
 *   normalised (:93-96).
 * lasso_zero_columns: Z[:, j] = 0 where degenerate_dev[j] != 0 (:98).
 */
size_t lasso_gram_workspace_bytes(int64_t n, int64_t d, int64_t k);
int lasso_gram_accumulate(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx,
                          int64_t n, int64_t d, int64_t k, int dtype,
                          float* a_dev, float* b_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream);
/* lasso_gram_accumulate whose first launch writes *started_word = started_value (device memory) as it STARTS: everything
 * enqueued on `stream` before the call has completed by then.  A start signal for work on ANOTHER stream
 * (lasso_stream_wait_word polls the word) that costs this stream nothing -- an event record between two kernels is ~5 us. */
int lasso_gram_accumulate_signal(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx,
                                 int64_t n, int64_t d, int64_t k, int dtype, float* a_dev, float* b_dev,
                                 void* workspace_dev, size_t workspace_bytes, int32_t* started_word,
                                 int32_t started_value, void* stream);
size_t lasso_dict_sweep_workspace_bytes(int64_t d, int64_t k);
int lasso_dict_sweep(const float* a_dev, const float* b_dev, void* d_dev, int64_t ldd,
                     int64_t d, int64_t k, int dtype, double eps, int positive,
                     const float* pool_dev, int64_t pool_rows, int64_t pool_ld, uint64_t seed,
                     int32_t* degenerate_dev, int32_t* ndeg_out,
                     void* workspace_dev, size_t workspace_bytes, void* stream);
int32_t* lasso_dict_sweep_count(int64_t d, int64_t k, void* workspace_dev, size_t workspace_bytes);

/* lasso_dict_sweep without a host wait (ABI 7): the count of degenerate atoms is written by the sweep's last kernel to
 * `ndeg_mapped`, TWO int32 words of device-writable host memory (pinned, mapped): {count, 1}, the second written last
 * and released -- a host that zeroed it before the call polls it (or waits for an event recorded behind the call).  Replaces the copy of lasso_dict_sweep_count()'s word on the EM step's dependent chain. */
int lasso_dict_sweep_async(const float* a_dev, const float* b_dev, void* d_dev, int64_t ldd,
                           int64_t d, int64_t k, int dtype, double eps, int positive,
                           const float* pool_dev, int64_t pool_rows, int64_t pool_ld, uint64_t seed,
                           int32_t* degenerate_dev, int32_t* ndeg_mapped, void* workspace_dev,
                           size_t workspace_bytes, void* stream);

/* lasso_dict_sweep_async with the new dictionary written to ANOTHER buffer (d_out_dev, pitch ldo >= k, not overlapping
 * d_dev): d_dev is only read.  The call may then be enqueued BEFORE the host knows whether the EM step it belongs to
 * stands (the E-step's verdict, the previous sweep's count of degenerate atoms) -- a step that is repeated keeps d_dev
 * and ignores d_out_dev, degenerate_dev and ndeg_mapped -- and launches on another stream may keep reading d_dev beside
 * it (the objective of dict_learning.py:39).  started_word (nullable, device memory): set to started_value by a launch
 * in front of the sweep itself -- "the sweep starts now" for a wave of another stream (lasso_stream_wait_word) whose work
 * should run beside it.  Same kernels, same arithmetic: bitwise the dictionary of lasso_dict_sweep.
 * Replaces the in-place update of dict_learning.py:83-91 where the caller double-buffers the dictionary. */
int lasso_dict_sweep_async_to(const float* a_dev, const float* b_dev, const void* d_dev, int64_t ldd,
                              void* d_out_dev, int64_t ldo, int64_t d, int64_t k, int dtype, double eps, int positive,
                              const float* pool_dev, int64_t pool_rows, int64_t pool_ld, uint64_t seed,
                              int32_t* degenerate_dev, int32_t* ndeg_mapped, int32_t* started_word,
                              int32_t started_value, void* workspace_dev, size_t workspace_bytes, void* stream);

/* One wave on `stream` that returns when a word of device memory has reached `value` (*word >= value: such words count
 * up) -- or, host_memory != 0, when a word of pinned host memory (e.g. the "valid" word of a LASSO_SOLVE_STATUS_MAPPED
 * buffer) equals `value` -- or after ~20 s (host memory: ~0.1 s): "after that kernel of ANOTHER stream" for the
 * launches behind it without an event record on the other stream.  A scheduling tool: a launch behind it whose DATA
 * depend on the order must check the word itself (lasso_fista_solve_verdict_deferred's gate).  Polling host memory
 * slows kernels running beside the wave (measured: an E-step by 12-19 %); polling device memory does not. */
int lasso_stream_wait_word(const int32_t* word, int32_t value, int host_memory, void* stream);

/* ---- Pipelined constrained M-step (ABI 7; dict_learning.py:44-45,82-101 in Gram form) -------------
 * The sweep of atom block b needs rows b of A = Z^T Z and of U = B - A D^T only, and walks the blocks far slower
 * than the chip produces them.  For d == 256, k a multiple of 256 (512 ... 4096) the M-step therefore exists in a
 * form that overlaps the two: [A | B] is produced in STAGES of whole block rows (256 atoms each) into ONE matrix
 * ab [k][k + d] (A in columns 0 .. k-1, B behind it, row pitch ldab >= k + d); stage 0 (the head: the first half of
 * the block rows) before the sweep starts, the others on a SECOND stream while the sweep runs -- its workgroups wait,
 * block by block, for the rows they need.  Per EM step, with `stages = lasso_mstep_pipe_stages(n, d, k)` (0: no
 * pipelined form, use lasso_gram_accumulate + lasso_dict_sweep) and stage s covering rows
 * lasso_mstep_pipe_stage_rows(s) of ab:
 *     stream M:  pipe_gram(0) [all-reduce the head's rows of ab] pipe_rows(0, seq)                pipe_sweep ... pipe_finish
 *     stream S:  pipe_wait(seq)   for s = 1 .. stages-1: pipe_gram(s) [all-reduce stage s' rows] pipe_rows(s)
 * with everything of stream S ENQUEUED before pipe_sweep (if the two streams ever share a hardware queue the sweep
 * then simply runs behind its producers) and pipe_finish -- the only call that changes the dictionary -- after
 * whatever still reads the old one.  A stage's rows of ab are contiguous: one collective per stage, the first on the
 * critical chain, the others hidden behind the sweep.  All calls of one step take the SAME workspace
 * (lasso_mstep_pipe_workspace_bytes; caller-owned, shared by both streams, ZEROED once before its first use).
 * Results: the dictionary of lasso_dict_sweep on the same (A, B) bit for bit -- the same products in the same order --;
 * A and B themselves differ from lasso_gram_accumulate's in the last bits (other sample splits, hence another
 * summation order). */
int lasso_mstep_pipe_stages(int64_t n, int64_t d, int64_t k);
int lasso_mstep_pipe_stage_rows(int64_t n, int64_t d, int64_t k, int stage, int64_t* row_lo, int64_t* row_hi);
size_t lasso_mstep_pipe_workspace_bytes(int64_t n, int64_t d, int64_t k);
/* the stage's rows of [A | B] from this process's n samples: per block row R the columns >= 256 R computed and folded,
 * the blocks right of the diagonal also written transposed into the rows below (after the stages 0 .. s the rows of
 * stage s are complete).  Stage 0 also resets the sweep's flag words.  n == 0 (a rank without rows, which still runs the
 * identical sweep on the all-reduced matrix): the stage's rows are zeroed, the flag words reset all the same. */
int lasso_mstep_pipe_gram(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n, int64_t d,
                          int64_t k, int dtype, float* ab_dev, int64_t ldab, int stage, void* workspace_dev,
                          size_t workspace_bytes, void* stream);
/* "after the head" for stream S without an event record on stream M (an event between two kernels of a stream costs
 * ~5 us there): pipe_rows(0, seq) leaves `seq` (any value that differs from the previous step's) in a word of the
 * workspace when its last workgroup finishes; this call enqueues ONE wave that returns when it reads `seq` there (or
 * after ~0.1 s: the ordering is a scheduling matter -- launches of the later stages in front of the head's would only
 * take compute units from it -- no data depends on it). */
int lasso_mstep_pipe_wait(int64_t n, int64_t d, int64_t k, int seq, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
/* the word lasso_mstep_pipe_wait polls (device memory inside the workspace; NULL if the shape has no pipelined M-step):
 * for launches behind that wait which must re-check it (the gate of lasso_fista_solve_verdict_deferred) */
const int32_t* lasso_mstep_pipe_head_word(int64_t n, int64_t d, int64_t k, void* workspace_dev, size_t workspace_bytes);
/* U rows of the stage (B - A D^T with the dictionary as it is BEFORE the sweep); its last workgroup raises the
 * "complete" words of the stage's block rows for the running sweep (stage 0: writes `seq` for lasso_mstep_pipe_wait). */
int lasso_mstep_pipe_rows(const float* ab_dev, int64_t ldab, const void* d_dev, int64_t ldd, int64_t n, int64_t d,
                          int64_t k, int dtype, int stage, int seq, void* workspace_dev, size_t workspace_bytes,
                          void* stream);
/* the sweep (one launch of co-operating workgroups + its stand-by); new atoms stay in the workspace, d_dev is only read */
int lasso_mstep_pipe_sweep(const float* ab_dev, int64_t ldab, const void* d_dev, int64_t ldd, int64_t n, int64_t d,
                           int64_t k, int dtype, double eps, int positive, int32_t* degenerate_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream);
/* (on stream S, behind its last launch that reads the old dictionary) one thread that writes `seq` to the workspace
 * word lasso_mstep_pipe_finish(wait_seq = seq) waits for */
int lasso_mstep_pipe_signal(int64_t n, int64_t d, int64_t k, int seq, void* workspace_dev, size_t workspace_bytes,
                            void* stream);
/* writes the new dictionary (degenerate atoms re-drawn from the counter-based generator, flagged in degenerate_dev as
 * by lasso_dict_sweep); `ndeg_mapped` (nullable): {count, valid} in device-writable host memory, as
 * lasso_dict_sweep_async.  wait_seq != 0: the kernel itself waits (up to ~0.3 s) for lasso_mstep_pipe_signal(wait_seq)
 * before it touches the dictionary -- instead of a cross-stream event wait in front of the launch (6-10 us on the
 * step's chain); if the word never arrives the count comes back as -1 and the caller must treat the step as failed.
 * wait_seq == 0: the caller has ordered the launch behind stream S itself. */
int lasso_mstep_pipe_finish(void* d_dev, int64_t ldd, int64_t n, int64_t d, int64_t k, int dtype, double eps, int positive,
                            int32_t* degenerate_dev, int32_t* ndeg_mapped, int wait_seq, void* workspace_dev,
                            size_t workspace_bytes, void* stream);
int lasso_dict_fill_degenerate(void* d_dev, int64_t ldd, int64_t d, int64_t k, int dtype,
                               const int32_t* degenerate_dev, const float* pool_dev, int64_t pool_rows,
                               int64_t pool_ld, int positive, void* stream);
int lasso_zero_columns(void* z_dev, int64_t ldz, int64_t n, int64_t k, int dtype,
                       const int32_t* degenerate_dev, void* stream);

/* Second half of a LASSO_SOLVE_ASYNC solve that returned LASSO_PENDING (same n, d, k, dtype,
 * maxiter, tol, workspace, stream): synchronises the stream, writes iters_out / last_delta_out
 * (HOST, nullable).  LASSO_OK, or LASSO_WARN_ABORTED (see above).  Work enqueued between the
 * two calls (lasso_objective, lasso_gram_accumulate on z_out ...) overlaps the wait. */
/* ---- line search on a ROW SHARD (one process per GPU): replaces the global reductions of
 * ista.py:23,28,32-35 (F <= Q of each trial) and :93 (stop rule) -------------------------
 * Same arguments as lasso_fista_solve with backtrack = 1 on the n rows this process holds
 * (n_global = rows of the whole batch, for the stop budget).  Wherever the reference reduces over
 * the batch the library hands `reduce` a small array of this rank's sums (doubles, HOST memory)
 * to be replaced IN PLACE by their sum over all ranks (return 0; anything else aborts the solve
 * with LASSO_ERR_HIP) -- e.g. an MPI_Allreduce / torch.distributed.all_reduce of `count`
 * doubles.  Every rank must call with the same lr, eta, maxiter, tol: the callback is then
 * invoked the same number of times with the same counts on every rank, and all ranks take the
 * same decisions (trials_out / accepted_lr_out / accepted_f_out are identical everywhere).
 * Workspace: lasso_fista_workspace_bytes(n, d, k, dtype, maxiter, tol, LASSO_STOP_GLOBAL, 1).
 * Requires ldz == k. */
typedef int (*lasso_allreduce_fn)(void* ctx, double* sums, int count);
int lasso_fista_solve_sharded(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                              const void* z0_dev, int64_t ldz0, void* z_out_dev, int64_t ldz,
                              int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype,
                              double alpha, double lr, int fast, int maxiter, double tol,
                              double eta_backtrack, lasso_allreduce_fn reduce, void* reduce_ctx,
                              int32_t* iters_out, float* last_delta_out, int32_t* trials_out,
                              float* accepted_lr_out, float* accepted_f_out, void* workspace_dev,
                              size_t workspace_bytes, void* stream);

/* The asynchronous form: only ENQUEUES the copy of {iterations, last delta (float bits), aborted
 * != 0, 0} into out4_host (HOST; pinned memory keeps the copy asynchronous).  The caller waits
 * on the stream -- or on an event recorded right behind this call, so that work enqueued after
 * it keeps the GPU busy during the wait -- and decodes the four words itself. */
int lasso_fista_solve_collect(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                              int32_t* out4_host, void* workspace_dev, size_t workspace_bytes,
                              void* stream);
/* LASSO_SOLVE_SHARDED: the device address (inside the workspace) of the pending solve's `maxiter`
 * per-iteration sums, to be all-reduced in place; NULL when the arguments describe no such solve. */
float* lasso_fista_solve_deltas(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                                void* workspace_dev, size_t workspace_bytes);
/* LASSO_SOLVE_SHARDED: enqueue the stop rule over the (all-reduced) sums with the budget
 * n_global * k * tol of the whole batch (ista.py:64); lasso_fista_solve_collect fetches its words.
 * sums_dev: the `maxiter` all-reduced sums, or NULL when they were reduced in place at
 * lasso_fista_solve_deltas() (a driver may carry them in the tail of its M-step message instead: the
 * EM step then has ONE collective). */
int lasso_fista_solve_verdict(int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype, int maxiter,
                              double tol, const float* sums_dev, void* workspace_dev, size_t workspace_bytes,
                              void* stream);
/* lasso_fista_solve_verdict + the four words written to `status_mapped` (device-writable host memory, as
 * LASSO_SOLVE_STATUS_MAPPED) by the verdict kernel: no lasso_fista_solve_collect behind it (ABI 7). */
int lasso_fista_solve_verdict_mapped(int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype, int maxiter,
                                     double tol, const float* sums_dev, int32_t* status_mapped, void* workspace_dev,
                                     size_t workspace_bytes, void* stream);
/* LASSO_SOLVE_DEFER_VERDICT (with LASSO_SOLVE_ASYNC | LASSO_SOLVE_ONE_CHUNK | LASSO_SOLVE_STATUS_MAPPED, not with
 * LASSO_SOLVE_SHARDED): the solve enqueues its kernels but NOT the launch that reduces the per-iteration sums and judges
 * the stop rule (ista.py:93-95), and returns LASSO_PENDING_DEFERRED; this call enqueues that launch on `stream`, which the
 * caller has ordered behind the solve's kernels without touching the solve's stream -- e.g. behind lasso_stream_wait_word
 * on the word lasso_gram_accumulate_signal raises.  An EM step's dependent chain (E-step -> Gram product -> sweep) is
 * then one launch shorter; same kernel, same sums, same four words in `status_mapped`.  The arguments are kept per host
 * thread, one solve at a time: call it from the thread that enqueued the solve, with that solve's workspace, before the
 * thread's next LASSO_SOLVE_ASYNC solve.  A solve that cannot defer (maxiter > 64) returns LASSO_PENDING_MAPPED as
 * without the flag.  gate_word (nullable, device memory): the word the stream's wait polled -- the launch re-checks
 * *gate_word == gate_value and, if the wait ran into its bound instead, reports "repeat the solve" (status word 2 = 1)
 * rather than judging sums of kernels that may still be running. */
int lasso_fista_solve_verdict_deferred(void* workspace_dev, int32_t* status_mapped, const int32_t* gate_word,
                                       int32_t gate_value, void* stream);
int lasso_fista_solve_finish(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                             int32_t* iters_out, float* last_delta_out, void* workspace_dev,
                             size_t workspace_bytes, void* stream);

/* ---- init='transpose': replaces torch.matmul(x, weight), sparse_encode.py:24-25 ---------
 * z0 [n][k] (ldz) = x [n][d] W [d][k] on the library's fp32-MFMA NT GEMM (csrc/gemm.hip).
 */
size_t lasso_init_transpose_workspace_bytes(int64_t d, int64_t k);
int lasso_init_transpose(int64_t n, int64_t d, int64_t k, int dtype, const void* x_dev, int64_t ldx,
                         const void* w_dev, int64_t ldw, void* z0_dev, int64_t ldz,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- unconstrained M-step: replaces update_dict_ridge, dict_learning.py:106-123 ----------
 * V [d][k] (ldv) = ((A + lambda_n I)^-1 B)^T with A = Z^T Z, B = Z^T X from
 * lasso_gram_accumulate (all-reduced by the caller across GPUs) and lambda_n = lambd * n
 * (:119).  Blocked Cholesky + the two triangular solves on the library's own fp32-MFMA
 * kernels (csrc/ridge.hip); k <= 4096 (lasso_ridge_workspace_bytes returns 0 beyond).  A and B are not modified.  info_out (HOST, nullable):
 * 0, or 1 + the index of the first non-positive pivot (then LASSO_ERR_BAD_ARG; torch raises
 * in linalg.cholesky there); a non-NULL info_out makes the call synchronise `stream`.
 */
size_t lasso_ridge_workspace_bytes(int64_t d, int64_t k);
int lasso_ridge_solve(const float* a_dev, const float* b_dev, void* v_dev, int64_t ldv,
                      int64_t d, int64_t k, int dtype, double lambda_n, int32_t* info_out,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- greedy coordinate descent: replaces coord_descent(),
 *      lasso/linear/solvers/coordinate_descent.py:5-54 (sparse_encode.py:54-55) --------
 * Per row: b = x W (:19, independent of z0), tracked z = z0 or 0 (:10-14); per step
 * propose S_alpha(b), commit the coordinate with the largest |proposal - z| (first index
 * on ties), b += (I - W^T W)[:, j] * change (:31-39); the row stops once its committed
 * change is <= tol*k (:9,45-48) or after maxiter steps.  z_out = S_alpha(b) (:52).
 *   lasso_cd_prepare : S, b and the per-row state into the workspace.
 *   lasso_cd_run     : up to `iters` further steps for every still-active row
 *                      (tol_abs is already tol*k).  n_active_out / max_steps_out are HOST
 *                      pointers (nullable); when either is given the call synchronises.
 *   lasso_cd_finish  : z_out = S_alpha(b); z_track_out (nullable) receives the tracked z
 *                      (the reference updates a caller-supplied z0 IN PLACE, :14,47).
 *   lasso_cd_solve   : prepare + run(maxiter) + finish; z0_inout may be NULL (zero init),
 *                      otherwise it is overwritten with the tracked z like the reference.
 * The workspace carries the state between the calls and must not be touched in between.
 * k <= 4096; any d.  Rows are independent: a row shard needs no collective.
 */
size_t lasso_cd_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype);
int lasso_cd_prepare(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                     const void* z0_dev, int64_t ldz0, int64_t n, int64_t d, int64_t k, int dtype,
                     void* workspace_dev, size_t workspace_bytes, void* stream);
int lasso_cd_run(int64_t n, int64_t d, int64_t k, double alpha, double tol_abs, int iters,
                 int32_t* n_active_out, int32_t* max_steps_out,
                 void* workspace_dev, size_t workspace_bytes, void* stream);
int lasso_cd_finish(void* z_out_dev, int64_t ldz, void* z_track_out_dev, int64_t ldzt,
                    int64_t n, int64_t d, int64_t k, double alpha,
                    void* workspace_dev, size_t workspace_bytes, void* stream);
int lasso_cd_solve(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                   void* z0_inout_dev, int64_t ldz0, void* z_out_dev, int64_t ldz,
                   int64_t n, int64_t d, int64_t k, int dtype, double alpha, int maxiter, double tol,
                   int32_t* n_active_out, int32_t* max_steps_out,
                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- convolutional ISTA/FISTA: replaces ista_conv2d(), lasso/conv2d/ista.py:7-49, and
 *      lip_bound_conv2d(), lasso/conv2d/lip_const.py:96-135 ---------------------------------
 * All tensors are contiguous NCHW fp32: x [N][C][H][W], weight [K][C][kh][kw] (the layout
 * F.conv2d takes: K code channels out, C image channels in), code z [N][K][Hz][Wz] with
 * H == (Hz-1)*sh - 2*ph + kh (the size conv_transpose2d gives the code; else
 * LASSO_ERR_BAD_ARG, the reference's shape RuntimeError at ista.py:19).
 *   lasso_conv_ista_solve: min_z 0.5*||conv_transpose2d(z, W) - x||^2 + alpha*||z||_1 by
 *     (F)ISTA with the fixed step lr; z0_dev == NULL means zero init; the stop rule
 *     sum|z - z_next| <= numel(z)*tol is global (ista.py:16,44) and, when tol > 0, is read
 *     once per chunk of speculated iterations (the stopping iteration is replayed bitwise
 *     from the chunk's head: the codes, the count and the last sum are those of a solve
 *     that checked after every iteration); tol == 0 runs exactly maxiter iterations without
 *     a synchronisation.  iters_out / last_delta_out: HOST, nullable.
 *   lasso_conv_objective: (0.5*||x - x_hat||^2 + alpha*||z||_1)/N -> loss_dev (ista.py:23-26).
 *   lasso_conv_lip_bound: the Toeplitz bound on lambda_max of the stride-1 operator on a
 *     sample x sample frequency grid (sqrt != 0: its square root); ksize odd (else
 *     LASSO_ERR_BAD_ARG, the reference's ValueError :101-102).  l_out: HOST (synchronises).
 */
size_t lasso_conv_ista_workspace_bytes(int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                       int64_t Hz, int64_t Wz, int kh, int kw,
                                       int sh, int sw, int ph, int pw);
/* Name of the device kernel(s) lasso_conv_ista_solve dispatches this geometry to on the current device (the names
 * rocprofv3 --kernel-trace reports): "lasso::conv_fused_kernel<..>" when whole iterations run as one kernel with a
 * workgroup per image -- up to 64 iterations per launch -- or per band of an image (stride 1, fewer than 8 channels,
 * K <= 128, small images; at least a third as many images as CUs, or K <= 64 and bands whose halo is cheap), else the
 * synthesis kernel +
 * the gradient/prox kernel of the two-launch form. */
const char* lasso_conv_ista_kernel_name(int64_t N, int64_t C, int64_t H, int64_t W, int64_t K,
                                        int64_t Hz, int64_t Wz, int kh, int kw, int sh, int sw, int ph, int pw);
int lasso_conv_ista_solve(const void* x_dev, const void* w_dev, const void* z0_dev, void* z_out_dev,
                          int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz, int64_t Wz,
                          int kh, int kw, int sh, int sw, int ph, int pw, int dtype,
                          double alpha, double lr, int fast, int maxiter, double tol,
                          int32_t* iters_out, float* last_delta_out,
                          void* workspace_dev, size_t workspace_bytes, void* stream);
int lasso_conv_objective(const void* x_dev, const void* w_dev, const void* z_dev,
                         int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz, int64_t Wz,
                         int kh, int kw, int sh, int sw, int ph, int pw, int dtype,
                         double alpha, float* loss_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);
size_t lasso_conv_lip_workspace_bytes(int64_t K, int64_t C, int ksize, int sample);
int lasso_conv_lip_bound(const void* w_dev, int64_t K, int64_t C, int ksize, int padding, int sample,
                         int take_sqrt, double* l_out,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- reverse-mode derivative of the unrolled fixed-step solve ------------------------
 * The reference's ista() is ordinary autograd-traceable torch code (ista.py:57-104; the
 * README advertises back-propagation through the solver).  Given the iterates
 * z_0 .. z_T of a fixed-step run (trace_dev: [T+1][n][k] contiguous, T = iterations;
 * produce it with lasso_fista_run, one iteration per call) and dL/dz_T (grad_z_dev
 * [n][k]), writes dL/dx [n][d], dL/dW [d][k], dL/dz0 [n][k] (each nullable, contiguous).
 * Same derivative as torch.autograd through the reference loop: softshrink passes the
 * gradient where |u| > alpha*lr, the step size and the momentum schedule are constants.
 * Any d, k.  No host synchronisation.
 */
size_t lasso_fista_backward_workspace_bytes(int64_t n, int64_t d, int64_t k);
int lasso_fista_backward(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                         const void* trace_dev, const void* grad_z_dev,
                         int64_t n, int64_t d, int64_t k, int dtype, double lr, int fast, int iterations,
                         void* grad_x_dev, void* grad_w_dev, void* grad_z0_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same with one step size per iteration: lr_steps_host[i] (HOST array, `iterations` entries) is the
 * step iteration i was taken with -- the derivative of a line-search solve (ista.py:17-54: the accepted
 * step is a python float, i.e. a constant of the autograd graph; lasso_fista_solve reports the steps in
 * accepted_lr_out[], lasso_fista_run replays them one iteration per call to produce the trace).
 * lr_steps_host == NULL: every iteration used `lr`. */
int lasso_fista_backward_steps(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                               const void* trace_dev, const void* grad_z_dev,
                               int64_t n, int64_t d, int64_t k, int dtype, double lr,
                               const float* lr_steps_host, int fast, int iterations,
                               void* grad_x_dev, void* grad_w_dev, void* grad_z0_dev,
                               void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- patch front end: image -> patches -> centre -> dict_learning ->
 * reconstruct.  The reference's notebook that did this (examples/dict_learning_omniglot.ipynb)
 * is absent from its checkout, so there is no reference code to mirror: the layout is that of
 * torch.nn.functional.unfold -- images [N][C][H][W]; patch row m = (n, u, v) with
 * u < (H-ph)/sh+1, v < (W-pw)/sw+1; column (c, a, b).
 *   lasso_patches_extract     : patches [M][C*ph*pw] (ld); center != 0 subtracts each patch's
 *                               mean (stored in means_dev [M] when non-NULL).
 *   lasso_patches_reconstruct : overlap-AVERAGE of (patch + its mean) back into images.
 */
int lasso_patches_extract(const void* img_dev, void* patches_dev, int64_t ld, float* means_dev,
                          int64_t N, int64_t C, int64_t H, int64_t W, int ph, int pw, int sh, int sw,
                          int center, void* stream);
int lasso_patches_reconstruct(const void* patches_dev, int64_t ld, const float* means_dev, void* img_out_dev,
                              int64_t N, int64_t C, int64_t H, int64_t W, int ph, int pw, int sh, int sw,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LASSO_HIP_H_ */
