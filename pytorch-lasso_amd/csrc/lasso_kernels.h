// Internal kernel-launch declarations shared by the .hip translation units and
// the C-ABI layer (lasso_hip.cpp).  Not part of the public boundary
// (include/lasso_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace lasso {

constexpr int kTileM = 16;          // batch rows per workgroup (one MFMA M-block)
constexpr int kFistaWaves = 8;      // waves per workgroup (2 per SIMD)
constexpr int kFistaThreads = kFistaWaves * 64;
constexpr int kFistaD = 256;        // padded feature dim handled by the fused kernel
constexpr int kFistaMaxK = 1024;    // largest padded dictionary size (LDS bound)

struct FistaTileParams {
  const float* X;    int64_t ldx;       // [n][d]
  const float* Wp;                       // [kFistaD][Kpad]   zero padded W
  const float* Wtp;                      // [Kpad][kFistaD]   zero padded W^T
  const float* z_in; int64_t ldz_in;     // nullable -> zeros
  const float* y_in; int64_t ldy_in;     // nullable -> y = z_in
  float* z_out;      int64_t ldz_out;
  float* y_out;      int64_t ldy_out;    // nullable
  const float* coef;                     // [iters] momentum coefficients (device)
  float* partials;                       // [iters][ntiles] sum|z-z_next| per tile, nullable
  int n, d, k;                           // real sizes
  int ntiles, iters;
  float lr, lam;                         // step size, alpha*lr
  const float* lr_dev;                   // nullable: {lr, lam} in device memory (LASSO_LR_AUTO), overrides the two above
  // in-kernel exact global stop rule (fista_tile_sp.hip; needs gridDim.x == ntiles):
  unsigned long long* stop_gran;         // [kStopRing][ntiles] {tag = it+1, |dz| partial} granules, zeroed per solve
  int* stop_out;                         // [0] iterations executed, [1] last delta (float bits), [2] abort flag
  float stop_budget;                     // n*k*tol (ista.py:64)
  int stop_on;
  // split-k kernel (fista_splitk.hip): partial-residual exchange between the members of a group
  float* xch;                            // [2][groups][T][C][16 KiB] payload, parity = epoch & 1
  unsigned* xflags;                      // [groups][C][waves][T] epoch tags (+ XCC-id granules), zeroed per launch
  int groups;                            // groups of C = Kpad/128 workgroups; group g owns tiles g + groups * (T * round + slot)
  // tile kernel as the stand-by of a split-k launch: runs only if *run_if != 0 (the split-k
  // kernel's abort flag)
  const int* run_if;                     // nullable
  int part_stride;                       // partials[it * part_stride + part]
  int variant;                           // split-k kernel: 0 = default, 1 = the register-gather form for every T (A/B knob)
};
constexpr int kStopRing = 64;
constexpr int kSplitkMaxParts = 256;   // split-k kernel: groups x members never exceed the CU count
// split-k hand-off words (zeroed before every launch): [parts][waves][tile slots] epoch flags, then [parts] 8-byte XCC-id granules
constexpr int kSplitkMaxTiles = 4;     // tiles a group works on at once
constexpr size_t kSplitkFlagBytes =
    (size_t)kSplitkMaxParts * kFistaWaves * kSplitkMaxTiles * 4 + (size_t)kSplitkMaxParts * 8 +
    (size_t)kSplitkMaxParts * 2 * kSplitkMaxTiles * 4;     // + [parts][2][tile slots] flags of the reduced blocks
// Bound of every in-kernel handshake spin (one poll = a few L2 round trips + s_sleep, roughly
// a microsecond): ~0.1 s, far beyond any skew between co-resident workgroups.  Hitting it
// means part of the grid is not resident (CUs held by another stream / process); the kernels
// then abort as a whole and the host repeats the solve on a path without handshakes.
constexpr int kStopSpinLimit = 1 << 17;
constexpr int kGateSpinLimit = 1 << 24;   // waits for ANOTHER stream's work (may include the caller's collectives): ~20-30 s
// lasso_debug_force_standby (lasso_hip.h): the co-operative launches of the sweep and the Lipschitz squarings are
// skipped, their one-workgroup stand-by forms run alone
extern int g_force_standby;

// lasso_loss tile kernel (objective.hip)
struct ObjectiveParams {
  const float* X; int64_t ldx;
  const float* Wp;                       // [kFistaD][Kpad]
  const float* Z; int64_t ldz;
  float* partials;                       // [ntiles][2]  (sum r^2, sum |z|)
  int n, d, k, ntiles;
};

// constrained M-step (mstep.hip)
constexpr int kSweepBlock = 32;
constexpr int kSweepMaxD = 1024;   // atom sweep: features per atom (4 waves x 256)
constexpr int kSweepMaxK = 4096;   // atom sweep: atoms
struct SweepParams {
  const float* A; int64_t lda;           // [k][k]   Z^T Z
  float* U; int64_t ldu;                 // [k][dp]  B - A D^T, updated in place
  float* Dt;                             // [k][dp]  atoms as rows (in/out)
  const float* Dsrc; int64_t ldd;        // nullable: the dictionary itself, [d][k] -- the single-launch sweep then reads
                                         //   its old atoms from here (no transposed copy), 16-byte aligned, ldd, k % 4 == 0
  float* Dout; int64_t ldo;              // nullable: [d][k] the new dictionary, written by the last launch (repair of
                                         //   the degenerate atoms + transposition in one)
  float* dD;                             // [kSweepBlock][dp] scratch
  int dp;                                // d rounded up to a multiple of 256 (<= kSweepMaxD)
  const float* pool; int pool_rows;      // [pool_rows][pool_ld] replacement directions (nullable)
  int64_t pool_ld; unsigned long long seed;
  int* degenerate;                       // [k] out: 1 where the atom was re-initialised
  int* ndeg_in_out;                      // [1] running count of degenerate atoms
  int* ndeg_mirror;                      // nullable: TWO device-writable HOST words (pinned, mapped): {the final count, 1} -- the
                                         //   second written last and released, so a host that zeroed it may poll it: the caller's one
                                         //   host read per EM step needs neither a copy nor an event record behind the sweep
  const int* wait_word; int wait_value;  // nullable (fixup_transpose_kernel, pipelined M-step): do not touch the dictionary before
                                         //   *wait_word == wait_value -- a word another STREAM raises when it has read the old one
  int k, d;
  float eps; int positive;
  int flags_cleared;                     // the caller had the single-launch sweep's flag words cleared (by the launch in front)
};

// backtracking line search (backtrack.hip)
struct BtParams {
  const float* X; int64_t ldx;
  const float* Wp; const float* Wtp;
  const float* P; int64_t ldp;           // the point p (y or z)
  float* G;                              // [n][k] gradient at p
  float* C;                              // [n][k] candidate z_next
  float* partials;                       // [5][ntiles]
  int* flags; float* fvals;
  int n, d, k, ntiles;
  const int* skip;                       // nullable: *skip != 0 -> the launch is a no-op (a solve enqueued without host waits
                                         // whose stop rule fired, or whose line search ran out of pre-enqueued trials)
  // bf16 variant (bt_bf16.hip): x in bf16, fragment-major bf16 copies of W, 64-row tiles
  const void* Xh; const void* Wq1; const void* Wq2;
};

// persistent bf16 solve (bt16_persist.hip): one launch per solve, one 64-row tile per workgroup
struct Bt16PersistParams {
  const void* X; int64_t ldx;            // bf16 [n][d]
  const void* Wq1; const void* Wq2;      // fragment-major bf16 packs of W (launch_pack_w_bf16)
  const void* Z0; int64_t ldz0;          // bf16 [n][k] initial code, nullable -> zeros
  void* Z; int64_t ldz;                  // bf16 [n][k]: the iterate z, updated in place; the result
  void* G;                               // bf16 [ntiles * 64][Kpad] workspace: the gradient at the current point
  int n, d, k, ntiles;
  int maxiter, fast, backtrack;
  double alpha, lr0, eta;
  float budget;                          // n*k*tol (ista.py:64), < 0: no stop rule
  const float* coef;                     // [maxiter] momentum coefficients (device)
  void* gran;                            // [ring][ntiles][2] 16-byte trial granules, zeroed per launch
  unsigned long long* dgran;             // [ring][ntiles] |dz| granules, zeroed per launch
  int* out;                              // [0] iterations, [1] last delta (float bits), [2] abort, [3] line search failed
  int* trials; float* lrs; float* fvals; // [maxiter] device trace of the line search (nullable)
};
size_t bt16_persist_granule_bytes(int ntiles);
hipError_t bt16_persist_occupancy(int kpad, int* per_cu);
hipError_t launch_bt16_persist(const Bt16PersistParams& p, int kpad, hipStream_t stream);

// greedy coordinate descent (cd.hip): per-row state padded to kp = 256*NC columns
struct CdParams {
  float* B;                // [n][kp] correlation vectors b
  float* Zt;               // [n][kp] tracked code z
  const float* S;          // [kp][kp] I - W^T W (zero padded)
  int* active;             // [n] 1 = row still in the active set
  int* row_steps;          // [n] steps taken so far
  int* counter;            // row hand-out counter
  int n, iters;
  float alpha, tol;        // tol = absolute per-row threshold (reference: tol*k)
};

// convolutional ISTA (conv.hip): x [N][C][H][W], weight [K][C][kh][kw], code z [N][K][Hz][Wz]
constexpr int kConvDpart = 64 * 1024;      // words of the convolutional solver's partial-sum buffer (lasso_hip.hip carves it;
                                           // conv_fused.hip plans against it: 64 iterations per launch x up to 1024 workgroups)
struct ConvGeom {
  int N, C, H, W, K, Hz, Wz, kh, kw, sh, sw, ph, pw;
};

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per
// (kernel, device); thread-safe, cheap on the repeat path.  Defined in lasso_hip.hip.
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes);

hipError_t launch_fista_tile_sp(const FistaTileParams& p, int kpad, int dpad, int grid, hipStream_t stream,
                                int waves = kFistaWaves);
// the same kernel without the all-padding feature chunks of GEMM-2 (fista_tile_sp_ds.hip); hipErrorInvalidValue = none
hipError_t launch_fista_tile_sp_ds(const FistaTileParams& p, int kpad, int dpad, int dsteps, int grid, hipStream_t stream,
                                   int waves);
// workgroups of the stop-rule instantiation the occupancy query admits per CU
hipError_t fista_tile_sp_occupancy(int kpad, int dpad, int* blocks_per_cu, int waves = kFistaWaves);
// small-batch variant: a 16-row tile shared by Kpad/128 workgroups (fista_splitk.hip); needs
// p.xch / p.xflags / p.groups / p.stop_out, flags and stop_out zeroed on the stream before the launch
hipError_t launch_fista_splitk(const FistaTileParams& p, int kpad, int tiles, hipStream_t stream);
hipError_t fista_splitk_occupancy(int kpad, int* blocks_per_cu);
int fista_splitk_members(int kpad);
size_t fista_splitk_exchange_bytes(int kpad, int groups, int tiles);

hipError_t launch_objective(const ObjectiveParams& p, int kpad, int grid, double alpha,
                            double n_total, double* sums, float* loss_out, hipStream_t stream);

hipError_t launch_objective_generic(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* Z,
                                    int64_t ldz, int n, int d, int k, float* R, float* partials, int grid,
                                    double alpha, double n_total, double* sums, float* loss_out,
                                    hipStream_t stream);

hipError_t launch_bt_grad(const BtParams& p, int kpad, int grid, hipStream_t stream);
// several trials of one outer iteration in one launch (backtrack.hip bt_trials_kernel): their steps, by value
constexpr int kBtMultiMax = 8;
struct BtSteps { float lr[kBtMultiMax]; float lam[kBtMultiMax]; float hol[kBtMultiMax]; };   // step, alpha * step, 0.5 / step
hipError_t launch_bt_trials(const BtParams& p, int kpad, int grid, double alpha, const BtSteps& s, int ntrials,
                            int first_index, float* partsM /* [kBtMultiMax][4][ntiles] */, hipStream_t stream);
// one launch per outer iteration (bt_iter.hip): accept step of the previous iteration + gradient + `ntrials` trials per
// tile, then the decision launch that also closes the previous iteration (record, stop rule)
struct BtIterParams {
  const float* X; int64_t ldx;
  const float* Wp; const float* Wtp;
  float* Z;                              // [n][k] flat (ld = k, 16-byte aligned): the iterate z
  float* Y;                              // [n][k] flat: the momentum point y (read and written only when fast)
  float* G;                              // [n][k] flat: the gradient at the point
  float* partials;                       // [ntiles] sum r0^2 per tile
  float* partsM;                         // [kBtMultiMax][4][ntiles] tile sums of the trials
  float* dpart;                          // [ntiles] sum |z - z_next| per tile of the accept step
  const int* acc_flags; const float* acc_fvals;   // decision of the previous iteration; nullptr: no accept step (first
                                         //   iteration of a window: the point is loaded as it is)
  const int* skip;                       // nullable: *skip != 0 -> no-op
  int n, d, k, ntiles;
  int fast, tail, ntrials;               // tail: accept step only (behind the last iteration of a window)
  int zero_start;                        // no accept step and z = y = 0: the launch writes the zeros to Z / Y itself
  float coef;                            // momentum coefficient of the iteration being accepted
};
hipError_t launch_bt_iter(const BtIterParams& p, const BtSteps& s, int kpad, int grid, hipStream_t stream);
hipError_t launch_bt_iter_decide(const float* partials, const float* partsM, int ntiles, double alpha, const BtSteps& s,
                                 int ntrials, int first_index, int last_batch, int* cur_flags, float* cur_fvals,
                                 const int* prev_flags, const float* prev_fvals, const float* dpart, int it_prev,
                                 float budget, int* ctl, float* rec /* [iterations][4]: trials, step, F */,
                                 hipStream_t stream);
hipError_t launch_bt_trials_only(const BtParams& p, int kpad, int grid, const BtSteps& s, int ntrials, float* partsM,
                                 hipStream_t stream);
hipError_t launch_bt_trial(const BtParams& p, int kpad, int grid, double alpha, double lr,
                           int trial_index, int force, hipStream_t stream, double* sums_out = nullptr);
hipError_t launch_bt_decide(const BtParams& p, double alpha, double lr, int trial_index, int force,
                            hipStream_t stream, double* sums_out = nullptr);
hipError_t launch_bt16_grad(const BtParams& p, int kpad, int grid, hipStream_t stream);
hipError_t launch_bt16_trial(const BtParams& p, int kpad, int grid, float lr, float lam, int force,
                             hipStream_t stream);
hipError_t launch_pack_w_bf16(const void* W, int64_t ldw, int d, int k, int kp, int w_is_bf16, void* q1, void* q2,
                              hipStream_t stream);
hipError_t launch_cvt_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int n, int k, int to_f32,
                           hipStream_t stream);
hipError_t launch_bt_finish_recompute(float* Z, float* Y, const float* P, const float* G, int64_t total, float coef,
                                      const int* flags, const float* fvals, float* dpart, int grid,
                                      hipStream_t stream, const int* skip = nullptr);
// end of an outer iteration of a line-search solve enqueued without host waits: sum |z - z_next|, the iteration's
// record (trials, accepted step, F), the stop rule -- all on the device (backtrack.hip)
hipError_t launch_bt_iter_end(const float* dpart, int nparts, int* flags, const float* fvals, int* ctl, int it,
                              float budget, int* trials, float* lrs, float* fs, hipStream_t stream);
// unfused line search (backtrack.hip): element-wise halves around launch_gemm_nt_sub
hipError_t launch_sumsq_partials(const float* v, int64_t total, float* part, int grid, hipStream_t stream);
hipError_t launch_generic_trial(const float* P, const float* G, float* Cand, int64_t total, float lr, float lam,
                                float* partials, int grid, hipStream_t stream);
hipError_t launch_bt_finish(float* Z, int64_t ldz, float* Y, const float* Cand, int n, int k,
                            float coef, const int* flags, float* dpart, int grid, hipStream_t stream);

// Everything a solve needs before its persistent launch, in ONE launch (the EM step is made of
// ~10 such 5-us launches otherwise): pack W (pack_w_kernel), the momentum table
// (momentum_table_kernel, by the last block), zero the stop rule's granule ring and result words,
// and -- lr = LASSO_LR_AUTO -- turn lambda_max into {lr, alpha*lr} (step_from_lipschitz_kernel).
struct PrepareExtras {
  unsigned long long* zero_a; int words_a;      // nullable
  unsigned long long* zero_b; int words_b;      // nullable
  const double* lip; double alpha; float* lr_slot;   // nullable
};
// The momentum coefficients (t_i - 1) / t_{i+1} are a serial recurrence (a double sqrt and two divisions per step):
// computed by one GPU thread, 100 of them took 17 us -- the whole prepare launch.  The first kCoefHead are the same
// numbers for every solve: the host computes them once (same IEEE operations, same bits) and hands them over as a
// launch argument; only a longer schedule continues on the device from t_kCoefHead.
constexpr int kCoefHead = 256;
struct CoefHead { float v[kCoefHead]; double t_next; };
// the arguments of the prepare launch: a grid of gx x gy blocks of 256 threads (32 x 8)
struct PrepareJob {
  const float* W; int64_t ldw; int d, k, kp;
  float* wp; float* wtp; int dpad; float* coef; float* zeros; int count;
  PrepareExtras x; CoefHead head; int gx, gy;
};
#ifdef __HIPCC__
// block (bx, by) of the prepare launch; tid = 32 ty + tx
__device__ __forceinline__ void prepare_block(const PrepareJob& j, int bx, int by, int tid, float (*tile)[33]) {
  const int c0 = bx * 32, r0 = by * 32;
  const int tx = tid & 31, ty = tid >> 5;         // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < j.d && c < j.k) ? j.W[(int64_t)r * j.ldw + c] : 0.0f;
    tile[i][tx] = v;
    j.wp[(size_t)r * j.kp + c] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    j.wtp[(size_t)c * j.dpad + r] = tile[tx][i];
  }
  const int bid = by * j.gx + bx, nb = j.gx * j.gy;
  for (int i = bid * 256 + tid; i < j.x.words_a; i += nb * 256) j.x.zero_a[i] = 0ull;
  for (int i = bid * 256 + tid; i < j.x.words_b; i += nb * 256) j.x.zero_b[i] = 0ull;
  if (bid == nb - 1)
    for (int i = tid; i < j.count && i < kCoefHead; i += 256) { j.coef[i] = j.head.v[i]; j.zeros[i] = 0.0f; }
  if (tid != 0) return;
  if (bid == 0 && j.x.lip) {
    const double lr = 1.0 / j.x.lip[0];
    j.x.lr_slot[0] = (float)lr;
    j.x.lr_slot[1] = (float)(j.x.alpha * lr);
  }
  if (bid == nb - 1) {
    double t = j.head.t_next;
    for (int i = kCoefHead; i < j.count; ++i) {
      const double tt = __dmul_rn(t, t);
      const double s = __dsqrt_rn(__dadd_rn(1.0, __dmul_rn(4.0, tt)));
      const double tn = __ddiv_rn(__dadd_rn(1.0, s), 2.0);
      j.coef[i] = (float)__ddiv_rn(__dsub_rn(t, 1.0), tn);
      j.zeros[i] = 0.0f;
      t = tn;
    }
  }
}
#endif

size_t lipschitz_workspace_bytes(int64_t d, int64_t k);
// {lr, alpha lr} = {1 / lambda_max, alpha / lambda_max} as fp32, written by the launch that finishes lambda_max (slot nullable)
struct LipLr { float* slot; double alpha; };
// job (nullable): the blocks of a solve's prepare launch, carried by the Gram launch of this computation where that is the
// span kernel (*fused = true); otherwise the caller launches them itself
hipError_t launch_lipschitz(const float* W, int64_t ldw, int64_t d, int64_t k, void* workspace,
                            int squarings, hipStream_t stream, const PrepareJob* job = nullptr, bool* fused = nullptr,
                            LipLr lr = LipLr{nullptr, 0.0});

// (max_splits: what the caller's scratch holds -- 16 unless it was sized for more)
int gram_splits(int pc, int qc, int n, int sym, int cus, int max_splits = 16);
// [A | B] = Z^T [Z | X] in one launch on 256 x 256 blocks (k, d multiples of 256, large n); false = not applicable
constexpr int kGramAbMaxSplits = 128;
size_t gram_ab_scratch_bytes(int64_t d, int64_t k);
bool launch_gram_ab(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* A, float* B,
                    float* scratch, size_t scratch_bytes, int cus, hipStream_t stream, hipError_t* err, int* raise = nullptr,
                    int raise_value = 0);
// small dictionaries (k >= 128, d <= 128): A = Z^T Z and B = Z^T X in one product launch + one fold launch
bool launch_gram_ab128(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* A, float* B,
                       float* scratch, size_t scratch_bytes, int cus, int max_splits, hipStream_t stream, hipError_t* err,
                       int* raise = nullptr, int raise_value = 0);
hipError_t launch_gram_tn(const float* P, int64_t ldp, int pc, const float* Q, int64_t ldq, int qc,
                          int n, float* C, int64_t ldc, int sym, float* scratch, int splits,
                          hipStream_t stream);
hipError_t launch_gemm_nt_sub(const float* A, int64_t lda, const float* B, int64_t ldb,
                              const float* C0, int64_t ldc0, float* C, int64_t ldc, int m, int nn,
                              int kk, hipStream_t stream, int add = 0, int* zero_words = nullptr, int nzero = 0);
// the same GEMM with the proximal step of the unfused FISTA path in its epilogue (gemm.hip)
int gemm_nt_prox_parts(int m, int nn);
hipError_t launch_gemm_nt_prox(const float* A, int64_t lda, const float* B, int64_t ldb, float* Z, int64_t ldz,
                               float* Y, int64_t ldy, int m, int nn, int kk, float lr, float lam, float coef,
                               float* dpart, hipStream_t stream);
hipError_t launch_cd_init(const float* z0, int64_t ldz0, float* Zt, int kp, int n, int k, int* active,
                          int* row_steps, float* S, hipStream_t stream);
hipError_t launch_cd_rows(const CdParams& p, int kp, int cus, int* info, hipStream_t stream);
hipError_t launch_cd_finish(const float* B, const float* Zt, int kp, float* z_out, int64_t ldz,
                            float* zt_out, int64_t ldzt, int n, int k, float alpha, hipStream_t stream);
// persist_extra (sweep_persist_extra_bytes(k) bytes) + dt_out: d <= 256 runs the single-launch sweep,
// whose new atoms land in *dt_out (rows of 256 floats) instead of p.Dt
size_t sweep_persist_extra_bytes(int k);
int* sweep_persist_flags(void* persist_extra, int k);      // the 1024 flag words inside `persist_extra`
hipError_t launch_dict_sweep(const SweepParams& p, hipStream_t stream, void* persist_extra = nullptr,
                             float** dt_out = nullptr);
// ---- pipelined constrained M-step (round 6; mstep.hip): [A | B] = Z^T [Z | X] by block rows of 256 atoms, the sweep
// launched once the first block row's U rows exist and gated on the others
constexpr int kPipeMaxBlocks = 16;     // k <= 4096
struct MstepPipePlan {
  int nstages;                          // 0: this shape has no pipelined form
  int lo[kPipeMaxBlocks], hi[kPipeMaxBlocks];   // stage s produces block rows lo .. hi - 1 (256 atoms each)
  int blocks[kPipeMaxBlocks];           // 256 x 256 output blocks of the stage's Gram launch
  int splits[kPipeMaxBlocks];           // its sample splits
  int rps[kPipeMaxBlocks];              // samples per split
  size_t scratch_off[kPipeMaxBlocks];   // byte offset of the stage's partial sums inside the Gram scratch
  size_t scratch_bytes;
};
MstepPipePlan mstep_pipe_plan(int64_t n, int64_t d, int64_t k, int cus);
hipError_t launch_gram_rows(const float* Z, int64_t ldz, int k, const float* X, int64_t ldx, int d, int n, float* AB,
                            int64_t ldab, int stage, const MstepPipePlan& plan, float* scratch, int* clear_words, int nclear,
                            hipStream_t stream);
hipError_t launch_set_flag(int* flag, int value, hipStream_t stream);
hipError_t launch_wait_word(const int* word, int seq, int host_memory, hipStream_t stream);
hipError_t launch_uprod_rows(const float* A, int64_t lda, const float* Bm, int64_t ldb, const float* C0, int64_t ldc0,
                             float* U, int64_t ldu, int rows, int kk, int* ticket, int* flag, int nflags, int flag_value,
                             hipStream_t stream);
// the pipeline's own words behind the sweep's flags and debug stamps (never cleared by a launch: the caller zeroes the
// workspace once): [0] ticket of block row 0's Gram launch, [1] its "finished" sequence number, [8 + R] ticket of
// block row R's U product
int* sweep_pipe_words(void* persist_extra, int k);
hipError_t launch_sweep_gated(const SweepParams& p, void* persist_extra, int gate, int gate0, float** dt_out, hipStream_t stream);
hipError_t launch_sweep_fixup(const SweepParams& p, hipStream_t stream);
constexpr int kSweepRowFlag = 192;     // word R of the sweep's flags from here: block row R of [A | U] is complete
// unconstrained M-step (ridge.hip): V [d][k] = ((A + lam I)^-1 B)^T by blocked Cholesky, k <= 4096
size_t ridge_workspace_bytes(int64_t d, int64_t k);
hipError_t launch_ridge_solve(const float* A, const float* B, float* V, int64_t ldv, int d, int k, float lam,
                              void* workspace, int* info_dev, hipStream_t st);
hipError_t launch_fill_degenerate(float* D, int64_t ldd, int d, int k, const int* degenerate, const float* pool,
                                  int pool_rows, int64_t pool_ld, int positive, hipStream_t stream);
hipError_t launch_bw_prox(float* zb_next, float* zb_cur, const float* yb, const float* z_next, float* ub, float* gb,
                          int64_t total, float c, float lr, hipStream_t stream);
hipError_t launch_bw_point(const float* z, const float* z_prev, float* y, int64_t total, float c, hipStream_t stream);
hipError_t launch_bw_axpy(float* a, const float* b, float s1, const float* c, float s2, int64_t total,
                          hipStream_t stream);
// dst2 (may be null): a second copy of the destination
hipError_t launch_conv_relayout(const float* src, float* dst, float* dst2, int N, int K, int P, int to_rows,
                                hipStream_t stream);
hipError_t launch_conv_pack_w(const float* w, float* wt, float* wp, int K, int ckk, int ldr, hipStream_t stream);
hipError_t launch_conv_residual(const float* Ym, const float* Wt, const float* w, const float* x, float* colst, float* r,
                                const ConvGeom& g, int cus, hipStream_t stream);
// conv_synth.hip: the same residual as one implicit-GEMM kernel (stride 1, C <= 16, square kernels 3/5/7 with
// an instantiated atom count); *done = false -> not covered
hipError_t launch_conv_synth(const float* Ym, const float* w, const float* x, float* r, const ConvGeom& g, int cus,
                             bool* done, hipStream_t stream, int dry = 0);
// conv_synth_few.hip: the same for C < 8 channels (columns = the taps, overlap-add in LDS); stride 1, K <= 128 a multiple
// of 4, C kh kw <= 128
hipError_t launch_conv_synth_few(const float* Ym, const float* w, const float* x, float* r, const ConvGeom& g, int cus,
                                 bool* done, hipStream_t stream, int dry = 0);
// conv_fused.hip: synthesis + gradient + prox of up to 64 iterations in ONE launch, a workgroup per image or band of an
// image (stride 1, C < 8, K <= 128, small images or bands); `tables` = conv_fused_table_bytes() of workspace filled once per solve by
// launch_conv_fused_pack (*covered = false -> not covered, use the two-kernel form)
size_t conv_fused_table_bytes();
hipError_t launch_conv_fused_pack(const float* w, void* tables, const ConvGeom& g, int cus, bool* covered,
                                  hipStream_t stream);
// iterations one launch may take for this geometry (64 for whole images, 1 for images cut into bands; 0: not covered)
int conv_fused_max_iters(const ConvGeom& g, int cus);
// != 0: the launch reads y from one buffer and writes the next y to ANOTHER (bands: neighbours read the old rows)
int conv_fused_two_y_buffers(const ConvGeom& g, int cus);
const char* conv_fused_kernel_name(const ConvGeom& g, int cus);   // null: not covered
// `iters` iterations in one launch, momentum factor coefs[i] in iteration i; delta_out[i] (device, may be null) = the
// iteration's sum |z - z+|; dpart: iters x min(N bands, cus) words
hipError_t launch_conv_fused(const void* tables, const float* x, float* Zm, const float* Yin, float* Yout, float lr,
                             float lam, const float* coefs, int iters, float* dpart, int dpart_cap, float* delta_out,
                             const ConvGeom& g, int cus, hipStream_t stream);
hipError_t launch_conv_grad_prox(const float* r, const float* Wp, int ldr, float* Zm, float* Ym, float lr, float lam,
                                 float coef, float* dpart, int dpart_cap, const ConvGeom& g, int cus, int* count,
                                 hipStream_t stream, int dry = 0);
hipError_t launch_conv_gradient(const float* r, const float* Wp, float* rc, int ldr, float* G, const ConvGeom& g,
                                hipStream_t stream);
hipError_t launch_patches_extract(const float* img, float* out, int64_t ld, float* means, const ConvGeom& g,
                                  int center, hipStream_t stream);
hipError_t launch_patches_reconstruct(const float* pat, int64_t ld, const float* means, float* img,
                                      const ConvGeom& g, hipStream_t stream);
hipError_t launch_conv_lip(const float* taps, int O, int I, int64_t so, int64_t si, int ks, int padding,
                           const float* freq, int sample, int take_sqrt, float* maxes, double* out,
                           hipStream_t stream);
hipError_t launch_objective_reduce(const float* R, int64_t nd, const float* Z, int64_t ldz, int n, int k,
                                   float* partials, int grid, double alpha, double n_total, double* sums,
                                   float* loss_out, hipStream_t stream);
hipError_t launch_transpose_pad(const float* src, int64_t ld_src, int rows, int cols, float* dst,
                                int64_t ld_dst, int drows, int dcols, hipStream_t stream);
hipError_t launch_generic_prox(float* Z, int64_t ldz, float* Y, const float* G, int n, int k, float lr,
                               float lam, float coef, float* dpart, int grid, hipStream_t stream);
hipError_t launch_zero_columns(float* Z, int64_t ldz, int n, int k, const int* degenerate,
                               hipStream_t stream);

}  // namespace lasso
