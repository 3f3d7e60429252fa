// C-ABI layer of the MI355X FISTA engine (include/lasso_hip.h) plus the small
// support kernels (W packing, momentum table, per-iteration delta reduction).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <map>
#include <set>
#include <utility>
#include <vector>

#include "../../include/lasso_hip.h"
#include "lasso_kernels.h"

namespace lasso {
namespace {

thread_local char g_err[512] = "";

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

#define LASSO_HIP_TRY(expr)                                                         \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      return fail(LASSO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
  } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

int pad_k(int64_t k) {
  if (k <= 256) return 256;
  if (k <= 512) return 512;
  if (k <= 1024) return 1024;
  return -1;
}

// Padded feature count of the fused FISTA kernel: tiles are M x D with M*D = 4096, and the
// y tile (M x Kp fp32) must fit 64 KiB of LDS, so small dictionaries get taller tiles.
int pad_d(int64_t d, int kp);

// ---------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------
constexpr int kChunkMax = 64;     // iterations per launch when deltas are recorded

constexpr int kSplitMaxParts = kSplitkMaxParts;

struct Workspace {
  float* wp;        // [256][Kp]
  float* wtp;       // [Kp][256]
  float* coef;      // [coef_cap]  FISTA momentum coefficients
  float* zeros;     // [coef_cap]  ISTA "coefficients"
  float* partials;  // [kChunkMax][ntiles * members]
  float* delta;     // [kChunkMax]
  unsigned long long* gran;  // [kStopRing][max(ntiles, 256)] in-kernel stop-rule granules
  int* stop_out;    // [4]
  float* xch;       // split-k kernel: [2][256][16 KiB] partial-residual exchange
  unsigned* xflags; // split-k kernel: [256][waves] epoch tags
  float* state[4];  // zA, yA, zB, yB  [n][k]   (stop rule only)
  size_t bytes;
};

Workspace carve(void* base, int64_t n, int64_t k, int kp, int coef_cap, bool with_state) {
  Workspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes);
    return reinterpret_cast<float*>(r);
  };
  const int64_t ntiles = (n + kTileM - 1) / kTileM;
  w.wp = take((size_t)kFistaD * kp * 4);
  w.wtp = take((size_t)kp * kFistaD * 4);
  w.coef = take((size_t)std::max(coef_cap, 1) * 4);
  w.zeros = take((size_t)std::max(coef_cap, 1) * 4);
  const int members = kp / 128;                       // split-k: members per group
  // [kChunkMax][ntiles] of the tile kernel, then the split-k kernel's rows: one partial per
  // workgroup and round, rounds * groups * members <= (ntiles + 256) * members
  w.partials = take((size_t)kChunkMax * ntiles * 4 + (size_t)kChunkMax * (ntiles + kSplitMaxParts) * members * 4);
  w.delta = take((size_t)kChunkMax * 4);
  w.gran = reinterpret_cast<unsigned long long*>(
      take((size_t)kStopRing * std::max<int64_t>(ntiles, kSplitMaxParts) * 8));
  w.stop_out = reinterpret_cast<int*>(take(256));
  // exchange payload of the split-k kernel: up to 256 workgroups x T tiles at once, T by batch size
  {
    const int64_t slots = (ntiles * members + kSplitMaxParts - 1) / kSplitMaxParts;   // tiles per group if all 256 CUs take part
    const int tmax = slots <= 1 ? 1 : slots <= 2 ? 2 : kSplitkMaxTiles;
    w.xch = take(fista_splitk_exchange_bytes(kp, kSplitMaxParts / members, tmax));
  }
  w.xflags = reinterpret_cast<unsigned*>(take(kSplitkFlagBytes));
  for (int i = 0; i < 4; ++i) w.state[i] = with_state ? take((size_t)n * k * 4) : nullptr;
  w.bytes = off;
  return w;
}

// ---------------------------------------------------------------------------
// support kernels
// ---------------------------------------------------------------------------
// Wp[r][c] = W[r][c] (zero padded to [256][Kp]); Wtp[c][r] = W[r][c].
__global__ void pack_w_kernel(const float* __restrict__ W, int64_t ldw, int d, int k, int kp,
                              float* __restrict__ wp, float* __restrict__ wtp, int dpad = kFistaD) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < d && c < k) ? W[(int64_t)r * ldw + c] : 0.0f;
    tile[i][tx] = v;
    wp[(size_t)r * kp + c] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    wtp[(size_t)c * dpad + r] = tile[tx][i];
  }
}

// Momentum table (ista.py:78,98-100): t_0 = 1, t_{i+1} = (1+sqrt(1+4 t_i^2))/2,
// coef_i = (float)((t_i - 1)/t_{i+1}).  Evaluated in IEEE double without
// contraction so it reproduces the reference's python-float arithmetic.
__global__ void momentum_table_kernel(float* __restrict__ coef, float* __restrict__ zeros, int count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 1.0;
  for (int i = 0; i < count; ++i) {
    const double tt = __dmul_rn(t, t);
    const double s = __dsqrt_rn(__dadd_rn(1.0, __dmul_rn(4.0, tt)));
    const double tn = __ddiv_rn(__dadd_rn(1.0, s), 2.0);
    coef[i] = (float)__ddiv_rn(__dsub_rn(t, 1.0), tn);
    zeros[i] = 0.0f;
    t = tn;
  }
}

// Everything a solve needs before its persistent launch, in ONE launch (the EM step is made of
// ~10 such 5-us launches otherwise): pack W (pack_w_kernel), the momentum table
// (momentum_table_kernel, by the last block), zero the stop rule's granule ring and result words,
// and -- lr = LASSO_LR_AUTO -- turn lambda_max into {lr, alpha*lr} (step_from_lipschitz_kernel).
// (PrepareExtras, CoefHead, PrepareJob and the launch's device code: lasso_kernels.h -- the Lipschitz launch of an
// lr = LASSO_LR_AUTO solve carries the same blocks, csrc/lipschitz.hip)
static const CoefHead& coef_head() {
  static const CoefHead head = [] {
    CoefHead h;
    double t = 1.0;
    for (int i = 0; i < kCoefHead; ++i) {
      const double tn = (1.0 + sqrt(1.0 + 4.0 * (t * t))) / 2.0;
      h.v[i] = (float)((t - 1.0) / tn);
      t = tn;
    }
    h.t_next = t;
    return h;
  }();
  return head;
}
__global__ void prepare_solve_kernel(const PrepareJob j) {
  __shared__ float tile[32][33];
  prepare_block(j, blockIdx.x, blockIdx.y, threadIdx.y * 32 + threadIdx.x, tile);
}

// delta[i] = sum_t partials[i][t], fixed summation order (deterministic).  `alt` (nullable):
// the rows of the stand-by launch, taken instead when *alt_if != 0 (the split-k kernel gave up).
// `base` (nullable): rows that are always added in front -- a ragged batch whose full rounds ran on the tile kernel
// and whose tail ran on the split-k kernel (PartialSets below).
struct PartialSets {
  const float* src; int n, stride;            // the launch's own partial sums, row i at src + i * stride
  const float* alt; int alt_n, alt_stride;    // stand-by rows, nullable
  const int* alt_if;
  const float* base; int base_n, base_stride; // nullable
};
struct DeferredVerdict { void* workspace; PartialSets ps; float* delta; int iters; float budget; int* out; bool armed; };
// the arguments of the verdict launch an asynchronous solve of THIS thread left out (one slot: the EM loop's E-step)
static thread_local DeferredVerdict t_deferred_verdict = {nullptr, {}, nullptr, 0, 0.0f, nullptr, false};
__global__ void reduce_partial_sets_kernel(const PartialSets ps, float* __restrict__ delta) {
  __shared__ float sh[256];
  const float* partials = ps.src;
  int ntiles = ps.n, stride = ps.stride;
  if (ps.alt && *ps.alt_if != 0) { partials = ps.alt; ntiles = ps.alt_n; stride = ps.alt_stride; }
  float acc = 0.0f;
  if (ps.base) {
    const float* brow = ps.base + (size_t)blockIdx.x * ps.base_stride;
    for (int t = threadIdx.x; t < ps.base_n; t += 256) acc += brow[t];
  }
  const float* row = partials + (size_t)blockIdx.x * stride;
  for (int t = threadIdx.x; t < ntiles; t += 256) acc += row[t];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) delta[blockIdx.x] = sh[0];
}

// the plain form: delta[i] = sum of row i of a dense [rows][ntiles] array (line search, unfused path)
// dst[0 .. n) = src[0 .. n) (both 16-byte aligned): the checkpoints of lasso_conv_ista_solve.  A kernel of our own
// rather than hipMemcpyAsync: with the runtime's device-to-device copies, 2 of 7 timed runs of a 10 / 20-iteration solve
// on 134 MB codes came out 2-5 ms per solve slow (profiles/r04/ab_conv_stop_rule.txt: none in 4 runs with this kernel;
// the cause inside the runtime was not established -- a copy engine instead of a blit kernel would fit the figure).
__global__ __launch_bounds__(256) void copy_words_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    ((v4*)dst)[i] = __builtin_nontemporal_load((const v4*)src + i);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void reduce_partials_kernel(const float* __restrict__ partials, int ntiles, float* __restrict__ delta) {
  __shared__ float sh[256];
  const float* row = partials + (size_t)blockIdx.x * ntiles;
  float acc = 0.0f;
  for (int t = threadIdx.x; t < ntiles; t += 256) acc += row[t];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) delta[blockIdx.x] = sh[0];
}

// reduce_partial_sets_kernel for `iters` rows and chunk_verdict_kernel behind it in ONE workgroup (the asynchronous
// E-step of an EM loop: one launch less on the step's dependent chain).  Every delta[i] is the same sum in the
// same order as reduce_partials_kernel's.
// defer (nullable): do not launch reduce_verdict_kernel -- leave its arguments there (LASSO_SOLVE_DEFER_VERDICT: the caller
// launches it on another stream, lasso_fista_solve_verdict_deferred)
struct DeferredVerdict;
struct ChunkVerdict { float budget; int* out; int* mirror; DeferredVerdict* defer = nullptr; };
// the verdict's four words a second time, into a device-writable HOST buffer (pinned, mapped; LASSO_SOLVE_STATUS_MAPPED):
// the caller's one host read per EM step then needs no copy launch behind the verdict
__device__ __forceinline__ void mirror_status(int* mirror, int w0, int w1, int w2) {
  if (!mirror) return;
  __hip_atomic_store(mirror + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(mirror + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(mirror + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // word 3 = "valid", written last and released: a host that has zeroed it before the call may POLL it instead of
  // recording an event behind the launch (an event record costs the stream ~5 us between two kernels)
  __hip_atomic_store(mirror + 3, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// gate (nullable; the deferred launch on another stream): the word that stream's wait polled -- if it does not hold
// gate_value the wait ran into its bound and the solve's kernels may still be running: the verdict is then "repeat the
// solve" (out[2] = 1), never a judgement of unfinished sums.
__global__ __launch_bounds__(1024) void reduce_verdict_kernel(const PartialSets ps, float* __restrict__ delta, int iters,
                                                              float budget, int* __restrict__ out, int* mirror,
                                                              const int* gate = nullptr, int gate_value = 0) {
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gate_value) {     // (words count up)
    if (threadIdx.x == 0) {
      out[0] = 0; out[1] = __float_as_int(NAN); out[2] = 1; out[3] = 0;
      mirror_status(mirror, out[0], out[1], out[2]);
    }
    return;
  }
  // One WAVE per row, rows w, w + 16, ... (16 waves: the 10 rows of an EM step's chunk in one pass): lane l plays reduce_partials_kernel's threads l, l + 64, l + 128, l + 192
  // (their strided sums), the first two levels of its tree are then this lane's (a0 + a2) + (a1 + a3), the last six
  // the shuffles below -- no barrier per row.
  __shared__ float sd[64];
  const float* partials = ps.src;
  int ntiles = ps.n, stride = ps.stride;
  if (ps.alt && *ps.alt_if != 0) { partials = ps.alt; ntiles = ps.alt_n; stride = ps.alt_stride; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = w; i < iters; i += 16) {
    const float* row = partials + (size_t)i * stride;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (ps.base) {
      const float* brow = ps.base + (size_t)i * ps.base_stride;
#pragma unroll
      for (int h = 0; h < 4; ++h)
        for (int t = lane + 64 * h; t < ps.base_n; t += 256) a[h] += brow[t];
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
      for (int t = lane + 64 * h; t < ntiles; t += 256) a[h] += row[t];
    float v = (a[0] + a[2]) + (a[1] + a[3]);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s, 64);
    if (lane == 0) { delta[i] = v; sd[i] = v; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int hit = -1;
    float last = sd[iters - 1];
    for (int i = 0; i < iters; ++i)
      if (sd[i] <= budget) { hit = i; last = sd[i]; break; }
    out[0] = hit >= 0 ? hit + 1 : iters;
    out[1] = __float_as_int(last);
    out[2] = (hit >= 0 && hit + 1 < iters) ? 1 : 0;
    out[3] = 0;
    mirror_status(mirror, out[0], out[1], out[2]);
  }
}

// the stop rule over one chunk's per-iteration deltas (ista.py:93-95): out = {iterations, last delta, redo, 0};
// redo = the rule fired before the chunk's last iteration
__global__ void chunk_verdict_kernel(const float* __restrict__ delta, int c, float budget, int* __restrict__ out,
                                     int* mirror) {
  int hit = -1;
  float last = delta[c - 1];
  for (int i = 0; i < c; ++i)
    if (delta[i] <= budget) { hit = i; last = delta[i]; break; }
  out[0] = hit >= 0 ? hit + 1 : c;
  out[1] = __float_as_int(last);
  out[2] = (hit >= 0 && hit + 1 < c) ? 1 : 0;
  out[3] = 0;
  mirror_status(mirror, out[0], out[1], out[2]);
}

int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}

bool fused_shape(int64_t d, int64_t k) { return d <= kFistaD && k <= kFistaMaxK; }

// workgroups of the fused kernel that can be resident at once (occupancy query x CUs)
int fista_resident_workgroups(int kp, int dpad, int waves = kFistaWaves) {
  int per_cu = 0;
  if (fista_tile_sp_occupancy(kp, dpad, &per_cu, waves) != hipSuccess) return 0;
  return per_cu * device_cus();
}

// Tile geometry of the one-workgroup-per-tile kernel for a batch of n rows: 8-wave workgroups on
// tiles of 4096 / dpad rows, or -- short rows (dpad <= 128) whose tiles would not give every CU
// its share -- 4-wave workgroups on half-height tiles, two per CU (the same waves per CU on twice
// as many tiles; per CU the longest queue of rows is what counts).
struct TilePlan { int waves, rows, ntiles; };
TilePlan plan_tiles(int64_t n, int dpad, int kp = 0) {
  TilePlan t = {kFistaWaves, 4096 / dpad, 0};
  if (kp > 512 && dpad == 128) {       // 128 columns with more than 512 atoms: the 32-row y tile does not fit -- 4 waves on 16 rows
    t.waves = 4; t.rows = 16;
    t.ntiles = (int)((n + 15) / 16);
    return t;
  }
  t.ntiles = (int)((n + t.rows - 1) / t.rows);
  if (dpad >= kFistaD || n <= 0) return t;
  const int cus = std::max(device_cus(), 1);
  const int64_t full_rows = (int64_t)((t.ntiles + cus - 1) / cus) * t.rows;       // rows queued on the busiest CU
  const int half = t.rows / 2, nh = (int)((n + half - 1) / half);
  const int64_t half_rows = (int64_t)((nh + cus - 1) / cus) * half;
  if (half_rows * 10 <= full_rows * 9) { t.waves = 4; t.rows = half; t.ntiles = nh; }
  return t;
}

// (lasso_fista_solve may narrow the tile for the extent of one call: d <= 128 with more than 512 atoms on a batch that
// fills the chip runs 4-wave workgroups on 16 x 128 tiles -- `NarrowTiles` below; thread-local like the error text)
static thread_local int g_dpad_override = 0;
int pad_d(int64_t d, int kp) {
  if (g_dpad_override) return g_dpad_override;
  if (d <= 64 && kp <= 256) return 64;
  if (d <= 128 && kp <= 512) return 128;
  return kFistaD;
}
struct NarrowTiles {
  explicit NarrowTiles(bool on) : on_(on) { if (on_) g_dpad_override = 128; }
  ~NarrowTiles() { if (on_) g_dpad_override = 0; }
  bool on_;
};

int check_common(int64_t n, int64_t d, int64_t k, int dtype, bool allow_large = false) {
  if (dtype != LASSO_F32)
    return fail(LASSO_ERR_UNSUPPORTED, "dtype %d: only LASSO_F32 is implemented", dtype);
  if (n < 0 || d <= 0 || k <= 0) return fail(LASSO_ERR_BAD_ARG, "bad shape n=%lld d=%lld k=%lld",
                                              (long long)n, (long long)d, (long long)k);
  if (!allow_large && !fused_shape(d, k))
    return fail(LASSO_ERR_UNSUPPORTED,
                "shape d=%lld k=%lld exceeds the fused kernel (d<=%d, k<=%d)", (long long)d,
                (long long)k, kFistaD, kFistaMaxK);
  if (n > (int64_t)INT32_MAX - kTileM) return fail(LASSO_ERR_UNSUPPORTED, "n too large");
  return LASSO_OK;
}

// Which fused kernel runs a batch: the split-k kernel (a 16-row tile shared by Kp/128
// workgroups, fista_splitk.hip) when the one-workgroup-per-tile kernel would leave most of the
// chip idle.  `lockstep`: the in-kernel stop rule needs every tile to own a resident group.
// Cost model of the two kernels in microseconds per iteration and round on MI355X
// (tools/bench_small.py): the tile kernel runs ceil(ntiles / #CUs) rounds of tile_us; the
// split-k kernel with T tiles per group runs ceil(ntiles / (groups * T)) rounds of split_us[T].
struct KernelCost { double tile_us, split_us[3]; };   // split_us: T = 1, 2, 4
KernelCost kernel_cost(int kp) {
  if (kp >= 1024) return {31.0, {5.9, 10.4, 18.9}};      // T >= 2: the reduce-scatter form (fista_splitk_rs_kernel)
  if (kp >= 768) return {23.6, {1e9, 1e9, 1e9}};        // (768 atoms: tile kernel only -- 6 members do not split)
  if (kp >= 512) return {15.9, {5.45, 9.9, 19.4}};
  return {8.3, {5.2, 9.6, 18.9}};
}
struct KernelPlan { bool split; int groups, tiles; double us; };   // us: the model's microseconds per iteration

int splitk_max_groups(int kp) {
  int per_cu = 0;
  if (fista_splitk_occupancy(kp, &per_cu) != hipSuccess) return 0;
  const int resident = std::min(per_cu * device_cus(), kSplitMaxParts);
  const int members = fista_splitk_members(kp);
  return resident / (8 * members) * 8;                // whole rows of 8 groups (one per XCD)
}

KernelPlan plan_kernel(int kp, int dpad, int ntiles, bool lockstep, int hint_bits) {
  KernelPlan plan = {false, 0, 1, 0.0};
  const int hint = hint_bits & 0x300;
  const int cus = std::max(device_cus(), 1);
  const KernelCost cost = kernel_cost(kp);
  plan.us = cost.tile_us * ((ntiles + cus - 1) / cus);
  if (dpad != kFistaD || hint == LASSO_KERNEL_TILE || ntiles <= 0) return plan;
  const int gmax = splitk_max_groups(kp);
  if (gmax <= 0) return plan;
  // (5 % in favour of the tile kernel: it has no cross-workgroup dependencies)
  double best = hint == LASSO_KERNEL_SPLITK ? 1e30 : 0.95 * cost.tile_us * ((ntiles + cus - 1) / cus);
  const int forced = (hint_bits >> 12) & 3;           // LASSO_KERNEL_SPLITK_TILES(T): 1, 2, 3 -> T = 1, 2, 4
  for (int ti = 0; ti < 3; ++ti) {
    if (forced && ti != forced - 1) continue;
    const int T = 1 << ti;
    const int groups = std::min(gmax, (ntiles + T - 1) / T);
    const int rounds = (ntiles + groups * T - 1) / (groups * T);
    if (lockstep && rounds > 1) continue;             // the in-kernel stop rule needs every tile in a resident slot
    // the kernel's cost is linear in the tile slots a group works on, and the last round may fill fewer of them
    const int last_tiles = ntiles - (rounds - 1) * groups * T, last_slots = (last_tiles + groups - 1) / groups;
    const double slot_us[5] = {0.0, cost.split_us[0], cost.split_us[1], 0.5 * (cost.split_us[1] + cost.split_us[2]),
                               cost.split_us[2]};
    const double us = cost.split_us[ti] * (rounds - 1) + slot_us[std::min(last_slots, 4)];
    if (us < best) { best = us; plan.split = true; plan.groups = groups; plan.tiles = T; plan.us = us; }
  }
  return plan;
}

// Tiles of a ragged batch's last, partly filled round that go to the split-k kernel (0: none); see run_impl.
// `whole`: what plan_kernel() chose for the whole batch (tile kernel alone, or split-k alone); the hybrid must beat it.
int ragged_tail(int kp, int dpad, int ntiles, int waves, int hint, const KernelPlan& whole, KernelPlan* tail_plan) {
  const int cus = std::max(device_cus(), 1);
  const int rounds = (ntiles + cus - 1) / cus;
  if (rounds < 2 || dpad != kFistaD || waves != kFistaWaves || (hint & 0x300) != LASSO_KERNEL_AUTO) return 0;
  const int t = ntiles - (rounds - 1) * cus;
  if (t >= cus) return 0;
  const KernelPlan sp = plan_kernel(kp, dpad, t, false, LASSO_KERNEL_SPLITK);
  const double tile_us = kernel_cost(kp).tile_us;
  if (!sp.split) return 0;
  const double hybrid_us = (rounds - 1) * tile_us + sp.us;
  if (hybrid_us >= 0.95 * rounds * tile_us || (whole.split && hybrid_us >= whole.us)) return 0;
  if (tail_plan) *tail_plan = sp;
  return t;
}

// Geometry of a fixed-step fp32 solve on the fused shapes: padded dictionary size and tile width.
//   * 256 < k <= 384 with d <= 128: the 384-atom instantiations of the d <= 128 tile kernels;
//   * 512 < k <= 768: the 768-atom instantiation of the tile kernel (a quarter less work than padding to 1024) where it
//     beats what the cost model picks at 1024 (small batches run the split-k kernel there, which needs a power of two
//     of 128-atom slices);
//   * d <= 128 with more than 512 atoms: NARROW tiles -- 4-wave workgroups on 16 x 128 tiles instead of 8 waves on
//     16 x 256 -- half the padded work per tile at nearly the same rate per wave (d=64, k=1024: 34 -> 60 TFLOP/s useful;
//     d=128: 68 -> 119), again only where that beats the split-k kernel of a small batch.
// Costs in microseconds per iteration and round of 256 tiles, measured on MI355X (round 3).  Every choice gives bitwise
// the same code for a row (canonical 128-atom slices, exact-zero padding).  A function of the shape alone: the second
// halves of an asynchronous solve carve the same workspace layout.
struct SolveGeom { int kp; bool narrow; };
SolveGeom solve_geometry(int64_t n, int64_t d, int64_t k) {
  const int kp0 = pad_k(k);
  if (kp0 == 512 && k <= 384 && d <= 128 && n > 0) return {384, false};
  if (kp0 != 1024 || n <= 0) return {kp0, false};
  const int ntiles = (int)((n + kTileM - 1) / kTileM);
  const int cus = std::max(device_cus(), 1);
  const int rounds = (ntiles + cus - 1) / cus;
  SolveGeom best = {1024, false};
  const KernelPlan whole = plan_kernel(1024, kFistaD, ntiles, false, LASSO_KERNEL_AUTO);
  double best_us = whole.us;
  {   // a ragged batch at 1024 atoms: full rounds on the tile kernel + the tail on the split-k kernel (run_impl)
    KernelPlan tp = {false, 0, 1, 0.0};
    if (ragged_tail(1024, kFistaD, ntiles, kFistaWaves, LASSO_KERNEL_AUTO, whole, &tp))
      best_us = (rounds - 1) * kernel_cost(1024).tile_us + tp.us;
  }
  auto consider = [&](int kp, bool narrow, double us_per_round) {
    if (us_per_round * rounds < best_us) { best_us = us_per_round * rounds; best = {kp, narrow}; }
  };
  if (k <= 768) consider(768, false, kernel_cost(768).tile_us);
  if (d <= 128) {
    consider(1024, true, 18.0);
    if (k <= 768) consider(768, true, 13.7);
  }
  return best;
}
int pad_k_solve(int64_t n, int64_t d, int64_t k) { return solve_geometry(n, d, k).kp; }
bool narrow_tiles(int64_t n, int64_t d, int64_t k, int kp, int hint_bits) {
  if (d > 128 || kp <= 512 || n <= 0) return false;
  if ((hint_bits & 0x800) && (hint_bits & 0x300) == LASSO_KERNEL_AUTO) return true;   // A/B knob "narrow": 0x800 ALONE
  if ((hint_bits & 0x300) == LASSO_KERNEL_SPLITK) return false;   // (with LASSO_KERNEL_SPLITK 0x800 is the exchange variant)
  return solve_geometry(n, d, k).narrow;
}

int run_impl(const Workspace& ws, int kp, const float* x, int64_t ldx, const float* z_in,
             int64_t ldz_in, const float* y_in, int64_t ldy_in, float* z_out, int64_t ldz_out,
             float* y_out, int64_t ldy_out, int64_t n, int64_t d, int64_t k, double alpha, double lr,
             int fast, int it0, int iters, float* delta, hipStream_t stream, float stop_budget = -1.0f,
             int hint = LASSO_KERNEL_AUTO, bool* used_split = nullptr, const float* lr_dev = nullptr,
             const ChunkVerdict* verdict = nullptr) {
  if (used_split) *used_split = false;
  if (n == 0) return LASSO_OK;
  const int dpad = pad_d(d, kp);
  const TilePlan tp = plan_tiles(n, dpad, kp);
  const int ntiles = tp.ntiles;
  FistaTileParams p;
  p.X = x; p.ldx = ldx;
  p.Wp = ws.wp; p.Wtp = ws.wtp;
  p.z_in = z_in; p.ldz_in = ldz_in;
  p.y_in = y_in; p.ldy_in = ldy_in;
  p.z_out = z_out; p.ldz_out = ldz_out;
  p.y_out = y_out; p.ldy_out = ldy_out;
  p.coef = (fast ? ws.coef : ws.zeros) + it0;
  p.partials = delta ? ws.partials : nullptr;
  p.n = (int)n; p.d = (int)d; p.k = (int)k;
  p.ntiles = ntiles; p.iters = iters;
  p.lr = (float)lr;                 // ATen casts the python scalar to the tensor dtype
  p.lam = (float)(alpha * lr);      // softshrink(lambd = alpha*lr), product in double (ista.py:90)
  p.lr_dev = lr_dev;                // LASSO_LR_AUTO: {lr, lam} read from device memory instead
  p.stop_on = stop_budget >= 0.0f;
  p.stop_budget = stop_budget;
  p.stop_gran = ws.gran;
  p.stop_out = ws.stop_out;
  p.xch = nullptr; p.xflags = nullptr; p.groups = 0;
  p.run_if = nullptr; p.part_stride = ntiles;
  // A/B knobs of the split-k kernel (only together with LASSO_KERNEL_SPLITK; 0x800 alone is the "narrow tiles" knob)
  p.variant = (hint & 0x300) != LASSO_KERNEL_SPLITK ? 0 : (hint & 0x400) ? 1 : (hint & 0x800) ? 2 : 0;
  const int cus = device_cus();
  if (cus <= 0) return fail(LASSO_ERR_HIP, "no HIP device");
  // in-place launches keep the tile kernel: a split-k launch that gives up is redone from its
  // inputs, which must still be there
  const bool in_place = (z_in && z_in == z_out) || (y_in && y_in == y_out) || (y_in && y_in == z_out) ||
                        (z_in && y_out && z_in == y_out);
  // (the split-k kernel reads a tile's 16 rows of x / z / y through 32-bit buffer offsets: a row pitch of 2^31 / 64 floats
  // or more keeps the tile kernel, whose loads are 64-bit addressed -- ADVICE r05: the descriptor clamped silently)
  const bool wide_pitch = std::max(std::max(ldx, ldz_in), std::max(ldy_in, std::max(ldz_out, ldy_out))) * 64 >= ((int64_t)1 << 31);
  KernelPlan plan = plan_kernel(kp, dpad, ntiles, p.stop_on != 0, (in_place || wide_pitch) ? LASSO_KERNEL_TILE : hint);
  float* const split_rows = ws.partials + (size_t)kChunkMax * ntiles;    // behind the tile kernel's rows
  // ---- ragged batches (round 4): the tile kernel runs ceil(ntiles / #CUs) rounds, so 257 tiles cost what 512 do.
  // The LAST, partly filled round goes to the split-k kernel instead when the cost model says so: the full rounds on
  // the tile kernel (rows [0, n_main)), then the tail's tiles on the split-k kernel (its own launch on the same
  // stream; rows are independent and a row's code is bitwise the same from either kernel).  n = 5000 at k = 1024:
  // 63 -> 42 us per iteration.
  KernelPlan tplan = {false, 0, 1, 0.0};
  const int tail = (in_place || wide_pitch || p.stop_on) ? 0 : ragged_tail(kp, dpad, ntiles, tp.waves, hint, plan, &tplan);
  if (tail) plan.split = false;                  // (the hybrid beats the split-k kernel on the whole batch as well)
  // the split-k kernel on the tiles of `q` (+ its stand-by): returns the partial sums per row of that launch
  auto launch_split = [&](FistaTileParams q, const KernelPlan& pl, float* standby_partials, int* nparts_out) -> int {
    // cross-workgroup hand-offs: epoch tags and the abort flag start from zero in every launch
    q.xch = ws.xch; q.xflags = ws.xflags; q.groups = pl.groups;
    const int members = fista_splitk_members(kp);
    const int rounds = (q.ntiles + pl.groups * pl.tiles - 1) / (pl.groups * pl.tiles);
    const int np = rounds * pl.groups * members;      // one partial per workgroup and round
    *nparts_out = np;
    q.part_stride = np;
    if (q.partials) q.partials = split_rows;
    LASSO_HIP_TRY(hipMemsetAsync(ws.xflags, 0, kSplitkFlagBytes, stream));
    if (!q.stop_on) LASSO_HIP_TRY(hipMemsetAsync(ws.stop_out, 0, 16, stream));   // (the stop-rule caller zeroed it)
    LASSO_HIP_TRY(launch_fista_splitk(q, kp, pl.tiles, stream));
    if (used_split) *used_split = true;
    if (!q.stop_on) {
      // No host synchronisation on this path, so the stand-by is enqueued right behind: the tile
      // kernel, which returns at once unless the split-k kernel raised its abort flag (a peer
      // workgroup was not resident) -- then it redoes the launch from the untouched inputs.
      // (With the stop rule on, the caller reads the flag at its synchronisation instead.)
      FistaTileParams f = q;
      f.run_if = ws.stop_out + 2;
      f.part_stride = ntiles;
      if (f.partials) f.partials = standby_partials;
      LASSO_HIP_TRY(launch_fista_tile_sp(f, kp, dpad, std::min(q.ntiles, cus), stream, tp.waves));
    }
    return LASSO_OK;
  };
  int nparts = ntiles;
  const int main_tiles = ntiles - tail;
  if (plan.split) {
    if (int st = launch_split(p, plan, ws.partials, &nparts)) return st;
  } else {
    // persistent over tiles: one 8-wave workgroup per CU (LDS bound), or two 4-wave ones
    FistaTileParams pm = p;
    if (tail) { pm.n = main_tiles * kTileM; pm.ntiles = main_tiles; }
    const int grid = std::min(main_tiles, cus * ((tp.waves == 4 && kp <= 512) ? 2 : 1));   // (4 waves, > 512 atoms: 104 KiB of LDS, one per CU)
    LASSO_HIP_TRY(launch_fista_tile_sp(pm, kp, dpad, grid, stream, tp.waves));
    if (tail) {
      const int64_t r0 = (int64_t)main_tiles * kTileM;
      FistaTileParams q = p;
      q.X = p.X + r0 * p.ldx;
      if (p.z_in) q.z_in = p.z_in + r0 * p.ldz_in;
      if (p.y_in) q.y_in = p.y_in + r0 * p.ldy_in;
      q.z_out = p.z_out + r0 * p.ldz_out;
      if (p.y_out) q.y_out = p.y_out + r0 * p.ldy_out;
      q.n = (int)(n - r0); q.ntiles = tail;
      if (int st = launch_split(q, tplan, ws.partials + main_tiles, &nparts)) return st;
    }
  }
  if (delta && iters > 0) {
    PartialSets ps = {ws.partials, ntiles, ntiles, nullptr, 0, 0, nullptr, nullptr, 0, 0};
    if (plan.split) ps = {split_rows, nparts, nparts, ws.partials, ntiles, ntiles, ws.stop_out + 2, nullptr, 0, 0};
    else if (tail) ps = {split_rows, nparts, nparts, ws.partials + main_tiles, tail, ntiles, ws.stop_out + 2,
                         ws.partials, main_tiles, ntiles};
    if (verdict && iters <= 64 && verdict->defer) {     // the caller enqueues that launch itself, on another stream
      DeferredVerdict& dv = *verdict->defer;
      dv.ps = ps; dv.delta = delta; dv.iters = iters; dv.budget = verdict->budget; dv.out = verdict->out; dv.armed = true;
    } else if (verdict && iters <= 64)
      hipLaunchKernelGGL(reduce_verdict_kernel, dim3(1), dim3(1024), 0, stream, ps, delta, iters, verdict->budget, verdict->out,
                         verdict->mirror);
    else
      hipLaunchKernelGGL(reduce_partial_sets_kernel, dim3(iters), dim3(256), 0, stream, ps, delta);
    LASSO_HIP_TRY(hipGetLastError());
  }
  return LASSO_OK;
}

// true if the split-k kernel of the launch(es) enqueued so far gave up (a peer workgroup was
// not resident); synchronises the stream.  The caller then repeats the work with LASSO_KERNEL_TILE.
int splitk_aborted(const Workspace& ws, hipStream_t stream, bool* aborted) {
  int hout[4] = {0, 0, 0, 0};
  LASSO_HIP_TRY(hipMemcpyAsync(hout, ws.stop_out, 16, hipMemcpyDeviceToHost, stream));
  LASSO_HIP_TRY(hipStreamSynchronize(stream));
  *aborted = hout[2] != 0;
  return LASSO_OK;
}

static PrepareJob prepare_job(const Workspace& ws, int kp, const float* w, int64_t ldw, int64_t d, int64_t k, int coef_cap,
                              const PrepareExtras* extras) {
  PrepareJob j;
  j.W = w; j.ldw = ldw; j.d = (int)d; j.k = (int)k; j.kp = kp;
  j.wp = ws.wp; j.wtp = ws.wtp; j.dpad = pad_d(d, kp); j.coef = ws.coef; j.zeros = ws.zeros;
  j.count = std::max(coef_cap, 1);
  j.x = PrepareExtras{nullptr, 0, nullptr, 0, nullptr, 0.0, nullptr};
  if (extras) j.x = *extras;
  j.head = coef_head();
  j.gx = kp / 32; j.gy = j.dpad / 32;
  return j;
}

int prepare_impl(const Workspace& ws, int kp, const float* w, int64_t ldw, int64_t d, int64_t k,
                 int coef_cap, hipStream_t stream, const PrepareExtras* extras = nullptr) {
  const PrepareJob j = prepare_job(ws, kp, w, ldw, d, k, coef_cap, extras);
  hipLaunchKernelGGL(prepare_solve_kernel, dim3(j.gx, j.gy), dim3(32, 8), 0, stream, j);
  LASSO_HIP_TRY(hipGetLastError());
  return LASSO_OK;
}

// lr = LASSO_LR_AUTO on the fused shapes: the Lipschitz launches and the prepare launch of the solve as ONE sequence --
// the prepare blocks ride in the Gram launch of the Lipschitz computation (both read only W), and the launch that
// finishes lambda_max leaves {lr, alpha lr} itself: one launch less on an EM step's dependent chain (round 6)
int prepare_with_lipschitz(const Workspace& ws, int kp, const float* w, int64_t ldw, int64_t d, int64_t k, int coef_cap,
                           hipStream_t stream, const PrepareExtras& extras, void* lip_ws) {
  PrepareExtras x = extras;
  const LipLr lr{x.lr_slot, x.alpha};
  x.lip = nullptr; x.lr_slot = nullptr;               // (the step size is the Lipschitz launches' business now)
  const PrepareJob j = prepare_job(ws, kp, w, ldw, d, k, coef_cap, &x);
  bool fused = false;
  LASSO_HIP_TRY(launch_lipschitz(w, ldw, d, k, lip_ws, 20, stream, &j, &fused, lr));
  if (!fused) {
    hipLaunchKernelGGL(prepare_solve_kernel, dim3(j.gx, j.gy), dim3(32, 8), 0, stream, j);
    LASSO_HIP_TRY(hipGetLastError());
  }
  return LASSO_OK;
}


// ---------------------------------------------------------------------------
// backtracking driver (ista.py:17-54 + the outer loop :79-102)
// ---------------------------------------------------------------------------
constexpr int kBtFinishGrid = 1024;
constexpr int kBtBatch = 8;          // trials enqueued per host round trip
constexpr int kBtMaxTrials = 1000;   // ista.py:17 (maxiter=1000)

struct BtWorkspace {
  float* wp; float* wtp; float* partials; float* dpart; float* delta; int* flags; float* fvals;
  float* partsM;   // [kBtMultiMax][4][ntiles] tile sums of the trials of a multi-trial launch (fp32 tensors)
  float* dtile;    // [ntiles] sum |z - z_next| per tile of bt_iter_kernel's accept step
  double* sums;    // [kBtMaxTrials][5] per-trial sums of a row-sharded solve
  float* G; float* C; float* Y;
  float* Zf;       // bf16 tensors: fp32 working copy of z
  // persistent bf16 solve (bt16_persist.hip)
  void* pG;        // [ntiles64 * 64][kp] bf16 gradient
  void* pZ0;       // [n][k] bf16 copy of z0 when it aliases z_out (a fall-back needs it intact)
  float* pcoef;    // [maxiter] momentum coefficients (+ [maxiter] zeros)
  void* pgran;     // trial granules + |dz| granules
  int* pout;       // [4]
  int* ptrials; float* plrs; float* pfvals;   // [maxiter] each
  int* rtrials; float* rlrs; float* rfs;      // [maxiter] each: record of a multi-launch solve enqueued without host waits
  int* ctl;                                   // its control words (bt_iter_end_kernel)
  float* rrec;                                // [maxiter][4] the same record, one 16-byte entry per iteration (bt_iter_decide_kernel)
  size_t bytes;
};

BtWorkspace carve_bt(void* base, int64_t n, int64_t k, int kp, bool half = false, int maxiter = 0) {
  BtWorkspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes);
    return reinterpret_cast<float*>(r);
  };
  const int64_t ntiles = (n + kTileM - 1) / kTileM;
  w.wp = take((size_t)kFistaD * kp * 4);
  w.wtp = take((size_t)kp * kFistaD * 4);
  w.partials = take((size_t)5 * std::max<int64_t>(ntiles, 1) * 4);
  w.partsM = take((size_t)kBtMultiMax * 4 * std::max<int64_t>(ntiles, 1) * 4);
  w.dpart = take((size_t)kBtFinishGrid * 4);
  w.dtile = take((size_t)std::max<int64_t>(ntiles, 1) * 4);
  w.delta = take(256);
  w.flags = reinterpret_cast<int*>(take(256));
  w.fvals = take(256);
  w.sums = reinterpret_cast<double*>(take((size_t)kBtMaxTrials * 5 * sizeof(double)));
  w.G = take((size_t)n * k * 4);
  w.C = take((size_t)n * k * 4);
  w.Y = take((size_t)n * k * 4);
  w.Zf = half ? take((size_t)n * k * 4) : nullptr;      // (wp / wtp hold the two bf16 packs then)
  w.pG = w.pZ0 = w.pgran = nullptr; w.pcoef = nullptr; w.pout = nullptr; w.ptrials = nullptr; w.plrs = w.pfvals = nullptr;
  if (half) {
    const int64_t nt64 = (n + 63) / 64;
    const int cap = std::max(maxiter, 1);
    w.pG = take((size_t)nt64 * 64 * kp * 2);
    w.pZ0 = take((size_t)n * k * 2);
    w.pcoef = take((size_t)cap * 4 * 2);
    w.pgran = take(bt16_persist_granule_bytes((int)std::max<int64_t>(nt64, 1)));
    w.pout = reinterpret_cast<int*>(take(256));
    w.ptrials = reinterpret_cast<int*>(take((size_t)cap * 4));
    w.plrs = take((size_t)cap * 4);
    w.pfvals = take((size_t)cap * 4);
  }
  {   // line search enqueued without host waits: the per-iteration record and the control words
    const int cap = std::max(maxiter, 1);
    w.rtrials = reinterpret_cast<int*>(take((size_t)cap * 4));
    w.rlrs = take((size_t)cap * 4);
    w.rfs = take((size_t)cap * 4);
    w.ctl = reinterpret_cast<int*>(take(256));
    w.rrec = take((size_t)cap * 16);
  }
  w.bytes = off;
  return w;
}

// Persistent single-launch solve on bf16 tensors (bt16_persist.hip), fixed step or line search.
// Returns LASSO_OK / LASSO_WARN_LINESEARCH when it ran; `*ran` false when it does not apply (more
// tiles than resident workgroups) or the kernel gave up (a workgroup was not resident): the caller
// then takes the multi-launch path.
int solve_bf16_persistent(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z0_dev,
                          int64_t ldz0, void* z_out_dev, int64_t ldz, int64_t n, int64_t d, int64_t k, int kp,
                          double alpha, double lr, int fast, int maxiter, double tol, int backtrack, double eta,
                          int32_t* iters_out, float* last_delta_out, int32_t* trials_out, float* accepted_lr_out,
                          float* accepted_f_out, void* workspace, size_t ws_bytes, hipStream_t st, bool* ran,
                          const void** z0_for_fallback) {
  *ran = false;
  *z0_for_fallback = z0_dev;
  BtWorkspace ws = carve_bt(workspace, n, k, kp, true, maxiter);
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  const int ntiles = (int)((n + 63) / 64);
  int per_cu = 0;
  if (bt16_persist_occupancy(kp, &per_cu) != hipSuccess) return LASSO_OK;
  if (ntiles > per_cu * device_cus()) return LASSO_OK;            // not all tiles resident: multi-launch path
  LASSO_HIP_TRY(launch_pack_w_bf16(w_dev, ldw, (int)d, (int)k, kp, 1, ws.wp, ws.wtp, st));
  hipLaunchKernelGGL(momentum_table_kernel, dim3(1), dim3(64), 0, st, ws.pcoef, ws.pcoef + std::max(maxiter, 1),
                     std::max(maxiter, 1));
  LASSO_HIP_TRY(hipGetLastError());
  const void* z0 = z0_dev;
  if (z0_dev && z0_dev == z_out_dev) {      // in place: keep the start for a fall-back run
    LASSO_HIP_TRY(hipMemcpy2DAsync(ws.pZ0, k * 2, z0_dev, ldz0 * 2, k * 2, n, hipMemcpyDeviceToDevice, st));
    *z0_for_fallback = ws.pZ0;
  }
  LASSO_HIP_TRY(hipMemsetAsync(ws.pgran, 0, bt16_persist_granule_bytes(ntiles), st));
  LASSO_HIP_TRY(hipMemsetAsync(ws.pout, 0, 16, st));
  Bt16PersistParams p;
  p.X = x_dev; p.ldx = ldx; p.Wq1 = ws.wp; p.Wq2 = ws.wtp;
  p.Z0 = z0; p.ldz0 = ldz0; p.Z = z_out_dev; p.ldz = ldz; p.G = ws.pG;
  p.n = (int)n; p.d = (int)d; p.k = (int)k; p.ntiles = ntiles;
  p.maxiter = maxiter; p.fast = fast; p.backtrack = backtrack;
  p.alpha = alpha; p.lr0 = lr; p.eta = eta;
  p.budget = tol > 0.0 ? (float)((double)n * (double)k * tol) : -1.0f;
  p.coef = ws.pcoef;
  p.gran = ws.pgran;
  p.dgran = reinterpret_cast<unsigned long long*>((char*)ws.pgran + (size_t)8 * ntiles * 32);
  p.out = ws.pout;
  p.trials = ws.ptrials; p.lrs = ws.plrs; p.fvals = ws.pfvals;
  LASSO_HIP_TRY(launch_bt16_persist(p, kp, st));
  int hout[4] = {0, 0, 0, 0};
  LASSO_HIP_TRY(hipMemcpyAsync(hout, ws.pout, 16, hipMemcpyDeviceToHost, st));
  LASSO_HIP_TRY(hipStreamSynchronize(st));
  if (hout[2]) {                              // some workgroup was not resident: nothing usable was produced
    if (z0_dev && z0_dev == z_out_dev) {
      // z_out was updated in place by the iterations that did complete: restore the start
      LASSO_HIP_TRY(hipMemcpy2DAsync(z_out_dev, ldz * 2, ws.pZ0, k * 2, k * 2, n, hipMemcpyDeviceToDevice, st));
    }
    return LASSO_OK;
  }
  *ran = true;
  const int its = hout[0];
  if (iters_out) *iters_out = its;
  if (last_delta_out) memcpy(last_delta_out, &hout[1], sizeof(float));
  if (backtrack && its > 0 && (trials_out || accepted_lr_out || accepted_f_out)) {
    if (accepted_f_out) LASSO_HIP_TRY(hipMemcpy(accepted_f_out, ws.pfvals, (size_t)its * 4, hipMemcpyDeviceToHost));
    if (trials_out) LASSO_HIP_TRY(hipMemcpy(trials_out, ws.ptrials, (size_t)its * 4, hipMemcpyDeviceToHost));
    if (accepted_lr_out) LASSO_HIP_TRY(hipMemcpy(accepted_lr_out, ws.plrs, (size_t)its * 4, hipMemcpyDeviceToHost));
  }
  return hout[3] ? fail(LASSO_WARN_LINESEARCH, "backtracking line search failed; reverted to lr0") : LASSO_OK;
}

int solve_backtracking(const void* x_any, int64_t ldx, const void* w_any, int64_t ldw, const void* z0_any,
                       int64_t ldz0, void* zout_any, int64_t ldz_any, int64_t n, int64_t d, int64_t k, int kp,
                       int dtype, double alpha, double lr0, int fast, int maxiter, double tol, double eta,
                       int32_t* iters_out, float* last_delta_out, int32_t* trials_out, float* accepted_lr_out,
                       float* accepted_f_out, void* workspace, size_t ws_bytes, hipStream_t st,
                       lasso_allreduce_fn reduce = nullptr, void* reduce_ctx = nullptr, int64_t n_global = 0,
                       int bt_hint = 0 /* 1: single-trial launches also for fp32 tensors (A/B) */) {
  // reduce != nullptr: this process holds a row shard.  The sums behind the two global decisions
  // of an iteration -- F <= Q of every trial (ista.py:23,28,32-35) and sum |z_next - z| <= n k tol
  // (:93) -- are added over the ranks by the caller's callback; every rank then takes the same
  // decision from the same numbers.
  const bool half = dtype == LASSO_BF16;      // bf16 tensors: bt_bf16.hip kernels, 64-row tiles
  BtWorkspace ws = carve_bt(workspace, n, k, kp, half, maxiter);
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  const int cus = device_cus();
  if (cus <= 0) return fail(LASSO_ERR_HIP, "no HIP device");
  const int tile_rows = half ? 64 : kTileM;
  const int ntiles = (int)((n + tile_rows - 1) / tile_rows);
  const int grid = std::min(ntiles, cus);
  const float* x = (const float*)x_any;
  float* zout = half ? ws.Zf : (float*)zout_any;
  const int64_t ldz = half ? k : ldz_any;
  // fp32 tensors, one process, flat 16-byte-aligned state: the one-launch-per-iteration kernels (bt_iter.hip) run the
  // solve; with an all-zero start (z0 == NULL) their first launch writes the zeros of z and y itself
  const bool fused_form = !half && !reduce && ldz == k && (k & 3) == 0 && k >= 4 && (((uintptr_t)zout) & 15) == 0 &&
                          (((uintptr_t)ws.Y) & 15) == 0 && (((uintptr_t)ws.G) & 15) == 0 && !(bt_hint & 3);
  const bool zero_in_kernel = fused_form && !z0_any && maxiter > 0;
  if (half) {
    LASSO_HIP_TRY(launch_pack_w_bf16(w_any, ldw, (int)d, (int)k, kp, 1, ws.wp, ws.wtp, st));
    if (z0_any) LASSO_HIP_TRY(launch_cvt_bf16(z0_any, ldz0, zout, k, (int)n, (int)k, 1, st));
    else LASSO_HIP_TRY(hipMemsetAsync(zout, 0, (size_t)n * k * 4, st));
  } else {
    const float* z0 = (const float*)z0_any;
    hipLaunchKernelGGL(pack_w_kernel, dim3(kp / 32, kFistaD / 32), dim3(32, 8), 0, st, (const float*)w_any, ldw,
                       (int)d, (int)k, kp, ws.wp, ws.wtp);
    LASSO_HIP_TRY(hipGetLastError());
    // working state: z lives in zout, y in the workspace (y0 = z0, ista.py:76-78)
    if (z0) {
      if (z0 != zout)
        LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, z0, ldz0 * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
    } else if (!zero_in_kernel) {
      LASSO_HIP_TRY(hipMemset2DAsync(zout, ldz * 4, 0, k * 4, n, st));
    }
  }
  if (!zero_in_kernel)
    LASSO_HIP_TRY(hipMemcpy2DAsync(ws.Y, k * 4, zout, ldz * 4, k * 4, n, hipMemcpyDeviceToDevice, st));

  BtParams p;
  p.X = x; p.ldx = ldx; p.Wp = ws.wp; p.Wtp = ws.wtp;
  p.Xh = x_any; p.Wq1 = ws.wp; p.Wq2 = ws.wtp;
  // candidates stay on chip and the finish kernel recomputes the accepted one whenever the
  // state is a flat [n][k] array (always for bf16 tensors; fp32: unless z_out is strided)
  const bool recompute = half || ldz == k;
  if (reduce && !recompute) return fail(LASSO_ERR_UNSUPPORTED, "row-sharded line search needs ldz == k");
  p.G = ws.G; p.C = recompute ? nullptr : ws.C; p.partials = ws.partials; p.flags = ws.flags; p.fvals = ws.fvals;
  p.n = (int)n; p.d = (int)d; p.k = (int)k; p.ntiles = ntiles;
  p.skip = nullptr;
  const float budget = (float)((double)(reduce ? n_global : n) * (double)k * tol);
  bool warned = false;
  double t_mom = 1.0;   // ista.py:78 (python int 1; same arithmetic in double)
  double hsums[kBtMaxTrials * 5];
  struct { int flags[4]; float fvals[4]; float delta; } host;
  int it = 0, prev_trials = kBtBatch - 1;
  float last = NAN;
  // ---- the whole solve enqueued WITHOUT a host wait (one process, flat state): every outer iteration is the
  // gradient, kBtBatch trial launches with the step sequence lr0 / eta^t (a trial behind the accepted one returns at
  // once), the accept step and a one-block kernel that keeps the iteration's record and evaluates the stop rule on
  // the device; once the rule fired -- or a search needed more than kBtBatch trials -- the remaining launches are
  // no-ops (BtParams::skip).  ONE synchronisation at the end; only a search that ran out of trials is continued on
  // the iteration-by-iteration path below, from the untouched state of the iteration that needed them.
  // The solve is enqueued in WINDOWS of kBtWindow outer iterations (all `maxiter` iterations at once meant
  // ~19 launches per remaining iteration of empty work after the rule fired -- maxiter = 1000 stopping at 20: ~19k
  // no-op launches); the control words are read between windows, and after a search that ran out of pre-enqueued
  // trials -- ONE iteration on the synchronous path below -- the next window is enqueued asynchronously again.
  const bool can_async = !reduce && recompute;
  constexpr int kBtWindow = 16;
  constexpr int kBtFirst = 5;          // trials of the first multi-trial launch of an iteration
  // multi-trial launches (bt_trials_kernel): fp32 tensors, flat 16-byte-aligned point and gradient
  // (can_async implies ldz == k here, so a row pitch that is not a multiple of 4 floats never reaches them)
  const bool multi = !half && can_async && (k & 3) == 0 && (ldz & 3) == 0 && k >= 4 && (((uintptr_t)zout) & 15) == 0 &&
                     (((uintptr_t)ws.Y) & 15) == 0 && (((uintptr_t)ws.G) & 15) == 0 && !(bt_hint & 1);
  // Round 5, fp32 tensors: ONE launch per outer iteration (bt_iter.hip: the accept step of the previous iteration, the
  // gradient and the first `n_first` trials per tile) + ONE decision launch that also closes the previous iteration;
  // decisions live in per-iteration slots of ws.flags / ws.fvals (kBtWindow slots of 4 words).  The trials beyond
  // n_first (old kernel, p and g re-read) are enqueued only after a search of THIS solve has run out of trials once
  // (`safety`); until then such a search costs one synchronous iteration.  bt_hint & 2: the multi-launch form (A/B).
  const bool fusedit = multi && !(bt_hint & 2);      // (== fused_form above)
  int n_first = kBtFirst;
  bool safety = false;
  while (it < maxiter) {
  if (can_async && fusedit) {
    const int win0 = it, wlen = std::min(kBtWindow, maxiter - it);
    const int n_first_used = n_first;
    LASSO_HIP_TRY(hipMemsetAsync(ws.flags, 0, (size_t)kBtWindow * 4 * sizeof(int), st));
    LASSO_HIP_TRY(hipMemsetAsync(ws.ctl, 0, 4 * sizeof(int), st));
    BtIterParams q;
    q.X = x; q.ldx = ldx; q.Wp = ws.wp; q.Wtp = ws.wtp; q.Z = zout; q.Y = ws.Y; q.G = ws.G;
    q.partials = ws.partials; q.partsM = ws.partsM; q.dpart = ws.dtile; q.skip = ws.ctl;
    q.n = (int)n; q.d = (int)d; q.k = (int)k; q.ntiles = ntiles; q.fast = fast ? 1 : 0;
    q.zero_start = (zero_in_kernel && it == 0) ? 1 : 0;
    const float stop_budget = tol > 0.0 ? budget : -1.0f;
    double tm = t_mom;
    float coef_prev = 0.0f;
    BtSteps steps;
    for (int i = win0; i < win0 + wlen; ++i) {
      const double t_next = (1.0 + sqrt(1.0 + 4.0 * tm * tm)) / 2.0;                 // :98
      const float coef = fast ? (float)((tm - 1.0) / t_next) : 0.0f;                 // :99
      const int slot = i - win0;
      int* const cur_flags = ws.flags + 4 * slot;
      float* const cur_fvals = ws.fvals + 4 * slot;
      const int* const prev_flags = slot > 0 ? ws.flags + 4 * (slot - 1) : nullptr;
      const float* const prev_fvals = slot > 0 ? ws.fvals + 4 * (slot - 1) : nullptr;
      double lr = lr0;
      for (int b = 0; b < n_first; ++b) {
        steps.lr[b] = (float)lr; steps.lam[b] = (float)(alpha * lr); steps.hol[b] = (float)(0.5 / lr);
        lr = lr / eta;                                                               // :47
      }
      q.acc_flags = prev_flags; q.acc_fvals = prev_fvals; q.coef = coef_prev; q.tail = 0; q.ntrials = n_first;
      LASSO_HIP_TRY(launch_bt_iter(q, steps, kp, grid, st));
      const bool more = safety && n_first < kBtBatch;
      LASSO_HIP_TRY(launch_bt_iter_decide(ws.partials, ws.partsM, ntiles, alpha, steps, n_first, 0, more ? 0 : 1, cur_flags,
                                          cur_fvals, prev_flags, prev_fvals, ws.dtile, i - 1, stop_budget, ws.ctl,
                                          ws.rrec, st));
      if (more) {
        BtSteps rest;
        const int nb = kBtBatch - n_first;
        for (int b = 0; b < nb; ++b) {
          rest.lr[b] = (float)lr; rest.lam[b] = (float)(alpha * lr); rest.hol[b] = (float)(0.5 / lr);
          lr = lr / eta;
        }
        BtParams pb = p;
        pb.P = fast ? ws.Y : zout; pb.ldp = k; pb.flags = cur_flags; pb.fvals = cur_fvals; pb.skip = ws.ctl;
        LASSO_HIP_TRY(launch_bt_trials_only(pb, kp, grid, rest, nb, ws.partsM, st));
        LASSO_HIP_TRY(launch_bt_iter_decide(ws.partials, ws.partsM, ntiles, alpha, rest, nb, n_first, 1, cur_flags,
                                            cur_fvals, nullptr, nullptr, nullptr, 0, -1.0f, ws.ctl, ws.rrec, st));
      }
      coef_prev = coef;
      tm = t_next;
    }
    {   // the accept step of the window's last iteration, and its record
      const int slot = wlen - 1;
      q.acc_flags = ws.flags + 4 * slot; q.acc_fvals = ws.fvals + 4 * slot; q.coef = coef_prev; q.tail = 1; q.ntrials = 0;
      LASSO_HIP_TRY(launch_bt_iter(q, steps, kp, grid, st));
      LASSO_HIP_TRY(launch_bt_iter_decide(ws.partials, ws.partsM, ntiles, alpha, steps, 0, 0, 0, nullptr, nullptr,
                                          ws.flags + 4 * slot, ws.fvals + 4 * slot, ws.dtile, win0 + wlen - 1,
                                          stop_budget, ws.ctl, ws.rrec, st));
    }
    // the window's control words and its per-iteration records: two small copies, ONE wait
    int hctl[4] = {0, 0, 0, 0};
    float hrec[kBtWindow * 4];
    LASSO_HIP_TRY(hipMemcpyAsync(hctl, ws.ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipMemcpyAsync(hrec, ws.rrec + 4 * (size_t)win0, (size_t)wlen * 16, hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    const int done = hctl[1] > 0 ? hctl[1] : win0;
    if (done > win0) {
      const int cnt = done - win0;
      int most = 1, last_trials = 1;
      for (int i = 0; i < cnt; ++i) {
        int tr;
        memcpy(&tr, &hrec[4 * i], sizeof(int));
        if (trials_out) trials_out[win0 + i] = tr;
        if (accepted_lr_out) accepted_lr_out[win0 + i] = hrec[4 * i + 1];
        if (accepted_f_out) accepted_f_out[win0 + i] = hrec[4 * i + 2];
        most = std::max(most, tr);
        last_trials = tr;
      }
      prev_trials = last_trials;
      memcpy(&last, &hctl[2], sizeof(float));
      for (int i = win0; i < done; ++i) t_mom = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
      // the next window computes as many trials per tile as this one's longest search took (every computed trial costs
      // its GEMM whether it is needed or not; a search that outgrows the guess costs one synchronous iteration)
      n_first = std::min(kBtBatch, std::max(2, most));
      // a whole window without a search beyond its first batch: the second trial batch and its decision leave the
      // windows again (two launches per iteration that did nothing; ADVICE r05: `safety` was never cleared)
      if (hctl[0] != 2 && most <= n_first_used) safety = false;
    }
    it = done;
    if (hctl[0] == 1) break;     // the stop rule fired at iteration `done` (:93-95)
    if (hctl[0] != 2) continue;  // the window ran through: next window
    safety = true;               // a search ran out of trials: that iteration below, from its untouched state
  } else if (can_async) {
    const int win0 = it, wlen = std::min(kBtWindow, maxiter - it);
    LASSO_HIP_TRY(hipMemsetAsync(ws.flags, 0, 4 * sizeof(int), st));
    LASSO_HIP_TRY(hipMemsetAsync(ws.ctl, 0, 4 * sizeof(int), st));
    p.skip = ws.ctl;
    double tm = t_mom;
    for (int i = win0; i < win0 + wlen; ++i) {
      const double t_next = (1.0 + sqrt(1.0 + 4.0 * tm * tm)) / 2.0;                 // :98
      const float coef = fast ? (float)((tm - 1.0) / t_next) : 0.0f;                 // :99
      p.P = fast ? ws.Y : zout;
      p.ldp = fast ? k : ldz;
      if (half) LASSO_HIP_TRY(launch_bt16_grad(p, kp, grid, st));
      else LASSO_HIP_TRY(launch_bt_grad(p, kp, grid, st));
      double lr = lr0;
      if (multi) {
        // fp32 tensors: the kBtBatch pre-enqueued trials as TWO launches -- trials 0 .. kBtFirst-1 (p and g of a tile
        // read once, in registers for all of them) and the rest (a no-op once a trial of the first launch passed)
        for (int b0 = 0; b0 < kBtBatch; b0 += kBtFirst) {
          BtSteps steps;
          const int nb = std::min(kBtFirst, kBtBatch - b0);
          for (int b = 0; b < nb; ++b) {
            steps.lr[b] = (float)lr; steps.lam[b] = (float)(alpha * lr); steps.hol[b] = (float)(0.5 / lr);
            lr = lr / eta;                                                             // :47
          }
          LASSO_HIP_TRY(launch_bt_trials(p, kp, grid, alpha, steps, nb, b0, ws.partsM, st));
        }
      } else
      for (int b = 0; b < kBtBatch; ++b) {
        if (half) {
          LASSO_HIP_TRY(launch_bt16_trial(p, kp, grid, (float)lr, (float)(alpha * lr), 0, st));
          LASSO_HIP_TRY(launch_bt_decide(p, alpha, lr, b, 0, st, nullptr));
        } else {
          LASSO_HIP_TRY(launch_bt_trial(p, kp, grid, alpha, lr, b, 0, st, nullptr));
        }
        lr = lr / eta;                                                                 // :47
      }
      LASSO_HIP_TRY(launch_bt_finish_recompute(zout, ws.Y, p.P, ws.G, n * k, coef, ws.flags, ws.fvals, ws.dpart,
                                               kBtFinishGrid, st, ws.ctl));
      LASSO_HIP_TRY(launch_bt_iter_end(ws.dpart, kBtFinishGrid, ws.flags, ws.fvals, ws.ctl, i,
                                       tol > 0.0 ? budget : -1.0f, ws.rtrials, ws.rlrs, ws.rfs, st));
      tm = t_next;
    }
    p.skip = nullptr;
    int hctl[4] = {0, 0, 0, 0};
    LASSO_HIP_TRY(hipMemcpyAsync(hctl, ws.ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    const int done = hctl[1] > 0 ? hctl[1] : win0;      // (ctl[1] = iterations completed, absolute; 0: none in this window)
    if (done > win0) {
      const int cnt = done - win0;
      std::vector<int> ht(cnt);
      std::vector<float> hl(cnt), hf(cnt);
      LASSO_HIP_TRY(hipMemcpy(ht.data(), ws.rtrials + win0, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      LASSO_HIP_TRY(hipMemcpy(hl.data(), ws.rlrs + win0, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      LASSO_HIP_TRY(hipMemcpy(hf.data(), ws.rfs + win0, (size_t)cnt * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; ++i) {
        if (trials_out) trials_out[win0 + i] = ht[i];
        if (accepted_lr_out) accepted_lr_out[win0 + i] = hl[i];
        if (accepted_f_out) accepted_f_out[win0 + i] = hf[i];
      }
      prev_trials = ht[cnt - 1];
      memcpy(&last, &hctl[2], sizeof(float));
      for (int i = win0; i < done; ++i) t_mom = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
    }
    it = done;
    if (hctl[0] == 1) break;     // the stop rule fired at iteration `done` (:93-95)
    if (hctl[0] != 2) continue;  // the window ran through: next window
    // a search ran out of pre-enqueued trials at iteration `it`: that iteration below, from its untouched state
  }
  {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;             // :98
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;                 // :99
    p.P = fast ? ws.Y : zout;
    p.ldp = fast ? k : ldz;
    LASSO_HIP_TRY(hipMemsetAsync(ws.flags, 0, 4 * sizeof(int), st));
    if (half) LASSO_HIP_TRY(launch_bt16_grad(p, kp, grid, st));
    else LASSO_HIP_TRY(launch_bt_grad(p, kp, grid, st));
    double lr = lr0;
    int t = 0;
    bool accepted = false;
    while (!accepted) {
      const bool give_up = t >= kBtMaxTrials;
      // the first batch of an iteration is sized from the previous iteration's trial count
      // (trials enqueued after the accepted one only cost their launch, but that adds up)
      const int want = t == 0 ? std::min(kBtBatch, std::max(2, prev_trials + 1)) : kBtBatch / 2;
      const int batch = give_up ? 1 : std::min(want, kBtMaxTrials - t);
      double lr_batch[kBtMaxTrials];
      for (int b = 0; b < batch; ++b) {
        const double lr_t = give_up ? lr0 : lr;      // ista.py:48-52: warn and revert to the initial step size
        const int idx = give_up ? t : t + b, force = give_up ? 1 : 0;
        double* const sums_dev = reduce ? ws.sums + 5 * b : nullptr;
        lr_batch[b] = lr_t;
        if (half) {
          LASSO_HIP_TRY(launch_bt16_trial(p, kp, grid, (float)lr_t, (float)(alpha * lr_t), force, st));
          LASSO_HIP_TRY(launch_bt_decide(p, alpha, lr_t, idx, force, st, sums_dev));
        } else {
          LASSO_HIP_TRY(launch_bt_trial(p, kp, grid, alpha, lr_t, idx, force, st, sums_dev));
        }
        if (give_up) warned = true;
        else lr = lr / eta;                                                            // :47
      }
      if (reduce) {
        // the batch's sums: all ranks add theirs, then the decision of bt_decide_kernel on the host,
        // trial by trial in order -- same float arithmetic, same outcome on every rank
        LASSO_HIP_TRY(hipMemcpyAsync(hsums, ws.sums, (size_t)batch * 5 * sizeof(double), hipMemcpyDeviceToHost, st));
        LASSO_HIP_TRY(hipStreamSynchronize(st));
        if (reduce(reduce_ctx, hsums, batch * 5) != 0) return fail(LASSO_ERR_HIP, "all-reduce callback failed");
        struct { int flags[4]; float fvals[4]; } dec = {{0, 0, 0, 0}, {0.f, 0.f, 0.f, 0.f}};
        for (int b = 0; b < batch && !dec.flags[0]; ++b) {
          const double* sm = hsums + 5 * b;
          const float rss0 = (float)sm[0], rss1 = (float)sm[1], l1 = (float)sm[2], dzg = (float)sm[3], dz2 = (float)sm[4];
          const float alpha_f = (float)alpha, half_over_lr = (float)(0.5 / lr_batch[b]);
          const float f0 = 0.5f * rss0;                                              // ista.py:23
          const float al1 = alpha_f * l1;
          const float F = 0.5f * rss1 + al1;                                         // :28
          const float Q = ((f0 + dzg) + half_over_lr * dz2) + al1;                   // :32-35
          dec.fvals[0] = F; dec.fvals[1] = Q;
          dec.flags[1] = (give_up ? t : t + b) + 1;
          if (give_up || F <= Q) {                                                   // :45
            dec.flags[0] = 1; dec.flags[2] = give_up ? t : t + b;
            dec.fvals[2] = (float)lr_batch[b]; dec.fvals[3] = (float)(alpha * lr_batch[b]);
          }
        }
        LASSO_HIP_TRY(hipMemcpyAsync(ws.flags, dec.flags, sizeof(dec.flags), hipMemcpyHostToDevice, st));
        LASSO_HIP_TRY(hipMemcpyAsync(ws.fvals, dec.fvals, sizeof(dec.fvals), hipMemcpyHostToDevice, st));
        LASSO_HIP_TRY(hipStreamSynchronize(st));     // (dec lives on this stack frame)
      }
      t += batch;
      if (recompute)   // P is Y (fast) or Z itself: element-wise in place is safe either way
        LASSO_HIP_TRY(launch_bt_finish_recompute(zout, ws.Y, p.P, ws.G, n * k, coef, ws.flags, ws.fvals, ws.dpart,
                                                 kBtFinishGrid, st));
      else
        LASSO_HIP_TRY(launch_bt_finish(zout, ldz, ws.Y, ws.C, (int)n, (int)k, coef, ws.flags, ws.dpart,
                                       kBtFinishGrid, st));
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, kBtFinishGrid, ws.delta);
      LASSO_HIP_TRY(hipGetLastError());
      LASSO_HIP_TRY(hipMemcpyAsync(host.flags, ws.flags, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipMemcpyAsync(&host.delta, ws.delta, sizeof(float), hipMemcpyDeviceToHost, st));
      if (accepted_lr_out || accepted_f_out)
        LASSO_HIP_TRY(hipMemcpyAsync(host.fvals, ws.fvals, 4 * sizeof(float), hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipStreamSynchronize(st));
      accepted = host.flags[0] != 0;
    }
    if (reduce) {                                    // sum |z_next - z| over all ranks (:93)
      double dsum = (double)host.delta;
      if (reduce(reduce_ctx, &dsum, 1) != 0) return fail(LASSO_ERR_HIP, "all-reduce callback failed");
      host.delta = (float)dsum;
    }
    prev_trials = host.flags[2] + 1;
    if (trials_out) trials_out[it] = host.flags[2] + 1;            // trials evaluated up to the accepted one
    if (accepted_lr_out) accepted_lr_out[it] = host.fvals[2];      // the step the iteration used (ista.py:40,52)
    if (accepted_f_out) accepted_f_out[it] = host.fvals[0];        // F(z_next) of the accepted trial (:28)
    last = host.delta;
    t_mom = t_next;
    ++it;
    if (tol > 0.0 && host.delta <= budget) break;                                      // :93-95
  }
  }
  if (half) LASSO_HIP_TRY(launch_cvt_bf16(zout, k, zout_any, ldz_any, (int)n, (int)k, 0, st));
  if (iters_out) *iters_out = it;
  if (last_delta_out) *last_delta_out = last;
  return warned ? fail(LASSO_WARN_LINESEARCH, "backtracking line search failed; reverted to lr0") : LASSO_OK;
}


// ---------------------------------------------------------------------------
// Fixed-step solve on bf16 tensors: the bf16-MFMA gradient kernel of the line search
// (bt_bf16.hip: g = (y W^T - x) W, bf16 operands, fp32 accumulation) + the fp32
// prox/momentum pass, state in fp32 -- no up-converted copies of x and W, and at large n
// faster than the fp32 fused kernel fed such copies.  The stop rule is evaluated on the host
// once per iteration like the reference does (ista.py:93).
// ---------------------------------------------------------------------------
int solve_fixed_bf16(const void* x_any, int64_t ldx, const void* w_any, int64_t ldw, const void* z0_any,
                     int64_t ldz0, void* zout_any, int64_t ldz_any, int64_t n, int64_t d, int64_t k, int kp,
                     double alpha, double lr, int fast, int maxiter, double tol, int32_t* iters_out,
                     float* last_delta_out, void* workspace, size_t ws_bytes, hipStream_t st) {
  BtWorkspace ws = carve_bt(workspace, n, k, kp, true);
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  const int cus = device_cus();
  if (cus <= 0) return fail(LASSO_ERR_HIP, "no HIP device");
  const int ntiles = (int)((n + 63) / 64);
  float* Z = ws.Zf;
  LASSO_HIP_TRY(launch_pack_w_bf16(w_any, ldw, (int)d, (int)k, kp, 1, ws.wp, ws.wtp, st));
  if (z0_any) LASSO_HIP_TRY(launch_cvt_bf16(z0_any, ldz0, Z, k, (int)n, (int)k, 1, st));
  else LASSO_HIP_TRY(hipMemsetAsync(Z, 0, (size_t)n * k * 4, st));
  LASSO_HIP_TRY(hipMemcpyAsync(ws.Y, Z, (size_t)n * k * 4, hipMemcpyDeviceToDevice, st));
  BtParams p;
  p.X = nullptr; p.ldx = ldx; p.Wp = ws.wp; p.Wtp = ws.wtp;
  p.skip = nullptr;
  p.Xh = x_any; p.Wq1 = ws.wp; p.Wq2 = ws.wtp;
  p.G = ws.G; p.C = nullptr; p.partials = ws.partials; p.flags = ws.flags; p.fvals = ws.fvals;
  p.n = (int)n; p.d = (int)d; p.k = (int)k; p.ntiles = ntiles;
  const float budget = (float)((double)n * (double)k * tol);
  const float lr_f = (float)lr, lam = (float)(alpha * lr);
  double t_mom = 1.0;
  float last = NAN;
  int it = 0;
  for (; it < maxiter; ++it) {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;
    p.P = ws.Y; p.ldp = k;                    // ISTA: coef = 0 keeps y == z
    LASSO_HIP_TRY(launch_bt16_grad(p, kp, std::min(ntiles, cus), st));
    LASSO_HIP_TRY(launch_generic_prox(Z, k, ws.Y, ws.G, (int)n, (int)k, lr_f, lam, coef, ws.dpart, kBtFinishGrid, st));
    t_mom = t_next;
    if (tol > 0.0) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, kBtFinishGrid, ws.delta);
      LASSO_HIP_TRY(hipGetLastError());
      LASSO_HIP_TRY(hipMemcpyAsync(&last, ws.delta, sizeof(float), hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipStreamSynchronize(st));
      if (last <= budget) { ++it; break; }
    }
  }
  LASSO_HIP_TRY(launch_cvt_bf16(Z, k, zout_any, ldz_any, (int)n, (int)k, 0, st));
  if (iters_out) *iters_out = it;
  if (last_delta_out) *last_delta_out = last;
  return LASSO_OK;
}

// ---------------------------------------------------------------------------
// The stop rule of the multi-launch solvers (ista.py:93 / conv2d/ista.py:44-46: the first iteration whose sum
// |z - z_next| over ALL elements is <= budget ends the solve with that iteration's z) without a host round trip per
// iteration -- the speculate-and-replay scheme of DESIGN 3.2 at launch granularity: a chunk of <= 64 iterations is
// enqueued with every iteration's sum kept on the device (`iterate(slot)` leaves it in *slot), the host reads the
// chunk's sums ONCE; if iteration j of the chunk met the rule and was not the chunk's last, the state goes back to the
// chunk's head (`save` / `restore`, the momentum scalar *t_mom with it) and exactly j + 1 iterations are replayed --
// every kernel of these paths sums in a fixed order, so the replay is bitwise the state the reference stops in.
// ---------------------------------------------------------------------------
// `flush` puts iterations that `iterate` only queued on the stream (the convolutional solver's many-iterations-per-launch
// kernel); solvers that launch in `iterate` pass a no-op.
template <class Iterate, class Save, class Restore, class Flush>
int speculate_stop_rule(int maxiter, float budget, float* delta_dev, hipStream_t st, double* t_mom, Iterate iterate,
                        Save save, Restore restore, Flush flush, int* it_out, float* last_out, const char* who) {
  constexpr int kChunkMax = 64;                    // delta_dev holds 64 sums
  static const bool trace_chunks = getenv("LASSO_STOP_TRACE") != nullptr;        // the chunks and their verdicts on stderr
  float deltas[kChunkMax];
  float last = NAN;
  int it = 0, chunk = 1;
  while (it < maxiter) {
    const int c = std::min(chunk, maxiter - it);
    const double t_head = *t_mom;
    if (c > 1)
      if (int s = save()) return s;
    for (int j = 0; j < c; ++j)
      if (int s = iterate(delta_dev + j)) return s;
    if (int s = flush()) return s;
    LASSO_HIP_TRY(hipMemcpyAsync(deltas, delta_dev, sizeof(float) * c, hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    int hit = -1;
    for (int j = 0; j < c && hit < 0; ++j)
      if (deltas[j] <= budget) hit = j;                                            // (fp32 compare like the reference)
    if (trace_chunks)
      fprintf(stderr, "%s: iterations %d..%d sums %g .. %g budget %g -> %s %d\n", who, it, it + c - 1, (double)deltas[0],
              (double)deltas[c - 1], (double)budget, hit < 0 ? "no stop" : "stop at", hit < 0 ? 0 : it + hit + 1);
    if (hit < 0) {
      it += c;
      last = deltas[c - 1];
      // Size of the next chunk -- only speed depends on it; a stop inside a chunk costs one chunk (the speculated rest
      // plus the replay).  Within a factor 2 of the budget: one iteration at a time (the reference's own cadence: the
      // sums of a momentum run are not monotone, and an iteration speculated past the stop costs more than the wait it
      // saves).  Further away: the iterations the rule is still away at the chunk's average decay -- that many when it
      // is near (so that it fires at the chunk's END: nothing to replay), half as many when it is far; without a
      // decaying chunk behind us, as many iterations as the solve has done; at most half the iterations done, and
      // never fewer than the sums would need if they halved every iteration.
      int next = 1;
      if (last > 2.0f * budget) {
        next = std::min(kChunkMax, std::max(2, it));
        if (c > 1 && deltas[0] > 0.0f && last < deltas[0] && budget > 0.0f) {
          const double rate = log((double)deltas[0] / (double)last) / (double)(c - 1);
          const double away = log((double)last / (double)budget) / rate;
          next = away <= 8.0 ? std::max(1, (int)ceil(away)) : (int)std::min((double)kChunkMax, away / 2.0);
        }
        const int lg = budget > 0.0f ? (int)std::min((double)kChunkMax, log2((double)last / (double)budget)) : kChunkMax;
        next = std::max(lg, std::min(next, std::max(2, it / 2)));
      }
      chunk = next;
      continue;
    }
    last = deltas[hit];
    if (hit < c - 1) {
      if (int s = restore()) return s;
      *t_mom = t_head;
      for (int j = 0; j <= hit; ++j)
        if (int s = iterate(nullptr)) return s;
      if (int s = flush()) return s;
    }
    it += hit + 1;
    break;
  }
  *it_out = it;
  *last_out = last;
  return LASSO_OK;
}

// ---------------------------------------------------------------------------
// Unfused path for shapes beyond the fused kernel (d > 256 or k > 1024): two MFMA GEMM
// launches + one elementwise launch per iteration, state in HBM.  Same arithmetic
// (ista.py:72-73,90,93,98-102); the stop rule is evaluated on the host every iteration
// like the reference does.  Correctness path, not tuned.
// ---------------------------------------------------------------------------
constexpr int kGenGrid = 1024;
struct GenWorkspace { float* Wt; float* Y; float* NR; float* G; float* dpart; float* delta;
                      float* C; float* part; int* flags; float* fvals;      // line search only
                      double* sums;                                         // (its five sums of a trial: row shards)
                      float* Wc;                                            // [d][k] copy of W (lasso_fista_prepare / _run)
                      float* Yc;                                            // y at the head of a speculated chunk (stop rule; z's copy lives in G)
                      size_t bytes; };

// with_state: room for y's checkpoint at the head of a speculated chunk -- only solves with a live stop rule take it
// (fixed-iteration solves paid n k words for a buffer they never touch)
GenWorkspace carve_generic(void* base, int64_t n, int64_t d, int64_t k, bool backtrack = false, bool with_state = true) {
  GenWorkspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes);
    return reinterpret_cast<float*>(r);
  };
  w.Wt = take((size_t)k * d * 4);
  w.Wc = take((size_t)d * k * 4);             // (both dictionary copies first: their place does not depend on n)
  w.Y = take((size_t)n * k * 4);
  w.NR = take((size_t)n * d * 4);
  w.G = take((size_t)n * k * 4);
  // per-block sums of |z - z_next|: the fused GEMM-2 + prox launch writes one per 64 x 64 (or larger) block
  w.dpart = take(std::max<size_t>((size_t)kGenGrid, (size_t)((n + 63) / 64) * (size_t)((k + 63) / 64)) * 4);
  w.delta = take(256);
  w.C = w.part = w.fvals = nullptr; w.flags = nullptr; w.sums = nullptr;
  if (backtrack) {
    w.C = take((size_t)n * k * 4);
    w.part = take((size_t)5 * kGenGrid * 4);
    w.flags = reinterpret_cast<int*>(take(256));
    w.fvals = take(256);
    w.sums = reinterpret_cast<double*>(take(256));
  }
  w.Yc = with_state ? take((size_t)n * k * 4) : nullptr;
  w.bytes = off;
  return w;
}

int solve_generic(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* z0, int64_t ldz0,
                  float* zout, int64_t ldz, int64_t n, int64_t d, int64_t k, double alpha, double lr,
                  int fast, int maxiter, double tol, int32_t* iters_out, float* last_delta_out,
                  void* workspace, size_t ws_bytes, hipStream_t st) {
  GenWorkspace ws = carve_generic(workspace, n, d, k, false, tol > 0.0 && maxiter > 0);
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  if (n > INT32_MAX || d > INT32_MAX || k > INT32_MAX) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
  LASSO_HIP_TRY(launch_transpose_pad(w, ldw, (int)d, (int)k, ws.Wt, d, (int)k, (int)d, st));
  if (z0) {
    if (z0 != zout)
      LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, z0, ldz0 * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  } else {
    LASSO_HIP_TRY(hipMemset2DAsync(zout, ldz * 4, 0, k * 4, n, st));
  }
  LASSO_HIP_TRY(hipMemcpy2DAsync(ws.Y, k * 4, zout, ldz * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  const float budget = (float)((double)n * (double)k * tol);
  const float lr_f = (float)lr, lam = (float)(alpha * lr);
  double t_mom = 1.0;
  float last = NAN;
  int it = 0;
  const int parts = gemm_nt_prox_parts((int)n, (int)k);
  auto iterate = [&](float* delta_slot) -> int {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;
    // NR = x - y W^T  (= -r);   G = 0 - NR Wt^T = r W
    // the gradient block g = r W never goes to memory: the proximal step runs in GEMM-2's epilogue
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.Y, k, w, ldw, x, ldx, ws.NR, d, (int)n, (int)d, (int)k, st));
    LASSO_HIP_TRY(launch_gemm_nt_prox(ws.NR, d, ws.Wt, d, zout, ldz, ws.Y, k, (int)n, (int)k, (int)d, lr_f, lam, coef,
                                      ws.dpart, st));
    t_mom = t_next;
    if (delta_slot) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, parts, delta_slot);
      LASSO_HIP_TRY(hipGetLastError());
    }
    return LASSO_OK;
  };
  if (!(tol > 0.0)) {
    for (; it < maxiter; ++it)
      if (int s = iterate(nullptr)) return s;
  } else {
    // the stop rule once per chunk of speculated iterations (speculate_stop_rule above; round 3: a host round trip
    // per iteration): z's checkpoint lives in G (unused since the proximal step moved into GEMM-2's epilogue), y's in Yc
    const int64_t words = n * k;
    const bool compact = ldz == k && ((uintptr_t)zout & 15) == 0;      // (else the runtime's strided copy)
    const int copy_grid = (int)std::min<int64_t>((words / 4 + 255) / 256 + 1, (int64_t)std::max(device_cus(), 1) * 16);
    auto save = [&]() -> int {
      if (compact) hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, zout, ws.G, words);
      else LASSO_HIP_TRY(hipMemcpy2DAsync(ws.G, k * 4, zout, ldz * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Y, ws.Yc, words);
      LASSO_HIP_TRY(hipGetLastError());
      return LASSO_OK;
    };
    auto restore = [&]() -> int {
      if (compact) hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.G, zout, words);
      else LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, ws.G, k * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Yc, ws.Y, words);
      LASSO_HIP_TRY(hipGetLastError());
      return LASSO_OK;
    };
    if (int s = speculate_stop_rule(maxiter, budget, ws.delta, st, &t_mom, iterate, save, restore,
                                    [] { return (int)LASSO_OK; }, &it, &last, "lasso_fista_solve (unfused)"))
      return s;
  }
  if (iters_out) *iters_out = it;
  if (last_delta_out) *last_delta_out = last;
  return LASSO_OK;
}

// `iters` iterations it0 .. of the unfused solve from a given (z, y) state: the resumable form behind
// lasso_fista_run for shapes beyond the fused kernels (traced forward pass of the autograd path,
// verbose mode).  Wt must have been prepared in the workspace (lasso_fista_prepare).
int run_generic(const float* x, int64_t ldx, const float* z_in, int64_t ldz_in,
                const float* y_in, int64_t ldy_in, float* z_out, int64_t ldz_out, float* y_out, int64_t ldy_out,
                int64_t n, int64_t d, int64_t k, double alpha, double lr, int fast, int it0, int iters, float* delta_dev,
                void* workspace, size_t ws_bytes, hipStream_t st) {
  GenWorkspace ws = carve_generic(workspace, n, d, k, false, false);   // (no stop-rule checkpoint here)
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  if (n == 0) return LASSO_OK;
  if (z_in) {
    if (z_in != z_out)
      LASSO_HIP_TRY(hipMemcpy2DAsync(z_out, ldz_out * 4, z_in, ldz_in * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  } else {
    LASSO_HIP_TRY(hipMemset2DAsync(z_out, ldz_out * 4, 0, k * 4, n, st));
  }
  const float* ysrc = y_in ? y_in : z_out;
  LASSO_HIP_TRY(hipMemcpy2DAsync(ws.Y, k * 4, ysrc, (y_in ? ldy_in : ldz_out) * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  const float lr_f = (float)lr, lam = (float)(alpha * lr);
  double t_mom = 1.0;
  for (int i = 0; i < it0; ++i) t_mom = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
  for (int i = 0; i < iters; ++i) {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.Y, k, ws.Wc, k, x, ldx, ws.NR, d, (int)n, (int)d, (int)k, st));
    LASSO_HIP_TRY(launch_gemm_nt_prox(ws.NR, d, ws.Wt, d, z_out, ldz_out, ws.Y, k, (int)n, (int)k, (int)d, lr_f, lam,
                                      coef, ws.dpart, st));
    if (delta_dev) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, gemm_nt_prox_parts((int)n, (int)k),
                         delta_dev + i);
      LASSO_HIP_TRY(hipGetLastError());
    }
    t_mom = t_next;
  }
  if (y_out)
    LASSO_HIP_TRY(hipMemcpy2DAsync(y_out, ldy_out * 4, ws.Y, k * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  return LASSO_OK;
}

// The line search of ista.py:17-54 for shapes beyond the fused kernels: per outer iteration two
// launches of the general GEMM for the gradient at p, then per trial one element-wise launch
// (candidate + three of the sums), one GEMM (its residual), the sum of squares and the decision
// kernel of backtrack.hip, one host synchronisation per trial -- the reference's own structure on
// HIP kernels.  Correctness path, not tuned.
int solve_generic_backtracking(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* z0,
                               int64_t ldz0, float* zout, int64_t ldz, int64_t n, int64_t d, int64_t k, double alpha,
                               double lr0, int fast, int maxiter, double tol, double eta, int32_t* iters_out,
                               float* last_delta_out, int32_t* trials_out, float* accepted_lr_out,
                               float* accepted_f_out, void* workspace, size_t ws_bytes, hipStream_t st,
                               lasso_allreduce_fn reduce = nullptr, void* reduce_ctx = nullptr, int64_t n_global = 0) {
  // reduce != nullptr: a row shard -- the five sums of every trial and sum|z - z+| are added over the ranks and the
  // decision of bt_decide_kernel is taken on the host, as in solve_backtracking
  GenWorkspace ws = carve_generic(workspace, n, d, k, true, false);
  if (ws_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, ws.bytes);
  if (n > INT32_MAX || d > INT32_MAX || k > INT32_MAX) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
  LASSO_HIP_TRY(launch_transpose_pad(w, ldw, (int)d, (int)k, ws.Wt, d, (int)k, (int)d, st));
  if (z0) {
    if (z0 != zout)
      LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, z0, ldz0 * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  } else {
    LASSO_HIP_TRY(hipMemset2DAsync(zout, ldz * 4, 0, k * 4, n, st));
  }
  LASSO_HIP_TRY(hipMemcpy2DAsync(ws.Y, k * 4, zout, ldz * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
  BtParams bp;                                             // only what bt_decide_kernel reads
  bp.skip = nullptr;
  bp.partials = ws.part; bp.ntiles = kGenGrid; bp.flags = ws.flags; bp.fvals = ws.fvals;
  const float budget = (float)((double)(reduce ? n_global : n) * (double)k * tol);
  bool warned = false;
  double t_mom = 1.0;
  float last = NAN;
  int it = 0;
  for (; it < maxiter; ++it) {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;             // :98
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;                 // :99 (ISTA: y == z)
    LASSO_HIP_TRY(hipMemsetAsync(ws.flags, 0, 4 * sizeof(int), st));
    // NR = x - p W^T (= -r0, :22);  G = r0 W (:24);  partials[0] = sum r0^2 (:23)
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.Y, k, w, ldw, x, ldx, ws.NR, d, (int)n, (int)d, (int)k, st));
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.NR, d, ws.Wt, d, nullptr, 0, ws.G, k, (int)n, (int)k, (int)d, st));
    LASSO_HIP_TRY(launch_sumsq_partials(ws.NR, n * d, ws.part, kGenGrid, st));
    double lr = lr0;
    int t = 0, accepted_at = -1;
    struct { int flags[4]; float fvals[4]; } host;
    for (;;) {
      const bool give_up = t >= kBtMaxTrials;
      const double lr_t = give_up ? lr0 : lr;                                           // :48-52
      LASSO_HIP_TRY(launch_generic_trial(ws.Y, ws.G, ws.C, n * k, (float)lr_t, (float)(alpha * lr_t), ws.part,
                                         kGenGrid, st));                                // :40, :31-35
      LASSO_HIP_TRY(launch_gemm_nt_sub(ws.C, k, w, ldw, x, ldx, ws.NR, d, (int)n, (int)d, (int)k, st));   // :27
      LASSO_HIP_TRY(launch_sumsq_partials(ws.NR, n * d, ws.part + kGenGrid, kGenGrid, st));
      LASSO_HIP_TRY(launch_bt_decide(bp, alpha, lr_t, t, give_up ? 1 : 0, st, reduce ? ws.sums : nullptr));   // :28, :32-35, :45
      if (reduce) {
        double hs[5];
        LASSO_HIP_TRY(hipMemcpyAsync(hs, ws.sums, sizeof(hs), hipMemcpyDeviceToHost, st));
        LASSO_HIP_TRY(hipStreamSynchronize(st));
        if (reduce(reduce_ctx, hs, 5) != 0) return fail(LASSO_ERR_HIP, "all-reduce callback failed");
        const float rss0 = (float)hs[0], rss1 = (float)hs[1], l1 = (float)hs[2], dzg = (float)hs[3], dz2 = (float)hs[4];
        const float f0 = 0.5f * rss0, al1 = (float)alpha * l1;                          // ista.py:23
        const float F = 0.5f * rss1 + al1;                                              // :28
        const float Q = ((f0 + dzg) + (float)(0.5 / lr_t) * dz2) + al1;                 // :32-35
        memset(&host, 0, sizeof(host));
        host.fvals[0] = F; host.fvals[1] = Q; host.flags[1] = t + 1;
        if (give_up || F <= Q) {                                                        // :45
          host.flags[0] = 1; host.flags[2] = t;
          host.fvals[2] = (float)lr_t; host.fvals[3] = (float)(alpha * lr_t);
        }
        LASSO_HIP_TRY(hipMemcpyAsync(ws.flags, host.flags, sizeof(host.flags), hipMemcpyHostToDevice, st));
        LASSO_HIP_TRY(hipMemcpyAsync(ws.fvals, host.fvals, sizeof(host.fvals), hipMemcpyHostToDevice, st));
        LASSO_HIP_TRY(hipStreamSynchronize(st));
      } else {
        LASSO_HIP_TRY(hipMemcpyAsync(host.flags, ws.flags, sizeof(host.flags), hipMemcpyDeviceToHost, st));
        LASSO_HIP_TRY(hipMemcpyAsync(host.fvals, ws.fvals, sizeof(host.fvals), hipMemcpyDeviceToHost, st));
        LASSO_HIP_TRY(hipStreamSynchronize(st));
      }
      if (give_up) warned = true;
      if (host.flags[0]) { accepted_at = t; break; }
      lr = lr / eta;                                                                    // :47
      ++t;
    }
    LASSO_HIP_TRY(launch_bt_finish(zout, ldz, ws.Y, ws.C, (int)n, (int)k, coef, ws.flags, ws.dpart, kGenGrid, st));
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, kGenGrid, ws.delta);
    LASSO_HIP_TRY(hipGetLastError());
    LASSO_HIP_TRY(hipMemcpyAsync(&last, ws.delta, sizeof(float), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    if (reduce) {                                    // sum |z_next - z| over all ranks (:93)
      double dsum = (double)last;
      if (reduce(reduce_ctx, &dsum, 1) != 0) return fail(LASSO_ERR_HIP, "all-reduce callback failed");
      last = (float)dsum;
    }
    if (trials_out) trials_out[it] = accepted_at + 1;
    if (accepted_lr_out) accepted_lr_out[it] = host.fvals[2];
    if (accepted_f_out) accepted_f_out[it] = host.fvals[0];
    t_mom = t_next;
    if (tol > 0.0 && last <= budget) { ++it; break; }                                   // :93-95
  }
  if (iters_out) *iters_out = it;
  if (last_delta_out) *last_delta_out = last;
  return warned ? fail(LASSO_WARN_LINESEARCH, "backtracking line search failed; reverted to lr0") : LASSO_OK;
}

// ---------------------------------------------------------------------------
// greedy coordinate descent (cd.hip): workspace = persistent per-row state
// ---------------------------------------------------------------------------
int pad_k_cd(int64_t k) {
  for (int kp = 256; kp <= 4096; kp *= 2)
    if (k <= kp) return kp;
  return -1;
}
struct CdWorkspace { float* S; float* Wt; float* B; float* Zt; int* active; int* row_steps; int* counter; int* info; size_t bytes; };

CdWorkspace carve_cd(void* base, int64_t n, int64_t d, int kp) {
  CdWorkspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(bytes);
    return r;
  };
  w.S = (float*)take((size_t)kp * kp * 4);
  w.Wt = (float*)take((size_t)kp * d * 4);
  w.B = (float*)take((size_t)n * kp * 4);
  w.Zt = (float*)take((size_t)n * kp * 4);
  w.active = (int*)take((size_t)n * 4);
  w.row_steps = (int*)take((size_t)n * 4);
  w.counter = (int*)take(256);
  w.info = (int*)take(256);
  w.bytes = off;
  return w;
}

int check_cd(int64_t n, int64_t d, int64_t k, int dtype) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d: only LASSO_F32 is implemented", dtype);
  if (n < 0 || d <= 0 || k <= 0) return fail(LASSO_ERR_BAD_ARG, "bad shape n=%lld d=%lld k=%lld",
                                              (long long)n, (long long)d, (long long)k);
  if (pad_k_cd(k) < 0) return fail(LASSO_ERR_UNSUPPORTED, "coordinate descent: k=%lld > 4096", (long long)k);
  if (n > INT32_MAX / 2 || d > INT32_MAX / 2) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
  return LASSO_OK;
}

// ---------------------------------------------------------------------------
// convolutional ISTA (conv.hip)
// ---------------------------------------------------------------------------
// (kConvDpart, lasso_kernels.h: partial sums |z - z+|: 64 iterations per launch x up to 1024 workgroups, conv_fused.hip)
struct ConvWorkspace { float* Wt; float* Wp; void* Wf; float* Zm; float* Ym; float* G; float* PT; float* R; float* dpart; float* delta; double* sums;
                       float* Zc; float* Yc;      // (z, y) at the head of a speculated chunk of iterations (stop rule, below)
                       size_t bytes; };

ConvWorkspace carve_conv(void* base, const ConvGeom& g) {
  ConvWorkspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(std::max<size_t>(bytes, 4));
    return r;
  };
  const size_t ckk = (size_t)g.C * g.kh * g.kw, M = (size_t)g.N * g.Hz * g.Wz;
  const size_t ldr = (ckk + 3) / 4 * 4;       // row stride of the pixel-major patch matrix
  w.Wt = (float*)take(ckk * g.K * 4);
  w.Wp = (float*)take(ldr * g.K * 4);
  w.Wf = take(conv_fused_table_bytes());
  w.Zm = (float*)take(M * g.K * 4);
  w.Ym = (float*)take(M * g.K * 4);
  w.G = (float*)take(M * g.K * 4);
  w.PT = (float*)take(ldr * M * 4);
  w.R = (float*)take((size_t)g.N * g.C * g.H * g.W * 4);
  w.dpart = (float*)take((size_t)kConvDpart * 4);
  w.delta = (float*)take(256);
  w.sums = (double*)take(256);
  w.Zc = (float*)take(M * g.K * 4);
  w.Yc = (float*)take(M * g.K * 4);
  w.bytes = off;
  return w;
}

int check_conv(const ConvGeom& g, int dtype) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d: only LASSO_F32 is implemented", dtype);
  if (g.N < 0 || g.C <= 0 || g.K <= 0 || g.H <= 0 || g.W <= 0 || g.Hz <= 0 || g.Wz <= 0 || g.kh <= 0 ||
      g.kw <= 0 || g.sh <= 0 || g.sw <= 0 || g.ph < 0 || g.pw < 0)
    return fail(LASSO_ERR_BAD_ARG, "bad convolution geometry");
  // x must have exactly the size conv_transpose2d gives the code (ista.py:19: x_hat - x)
  if ((int64_t)(g.Hz - 1) * g.sh - 2 * g.ph + g.kh != g.H || (int64_t)(g.Wz - 1) * g.sw - 2 * g.pw + g.kw != g.W)
    return fail(LASSO_ERR_BAD_ARG, "image %dx%d does not match code %dx%d under kernel %dx%d stride %d,%d padding %d,%d",
                g.H, g.W, g.Hz, g.Wz, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw);
  const int64_t M = (int64_t)g.N * g.Hz * g.Wz, ckk = (int64_t)g.C * g.kh * g.kw;
  if (M > INT32_MAX / 2 || ckk > INT32_MAX / 2 || (int64_t)g.N * g.C * g.H * g.W > ((int64_t)1 << 40))
    return fail(LASSO_ERR_UNSUPPORTED, "convolution problem too large");
  return LASSO_OK;
}

ConvGeom make_geom(int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz, int64_t Wz, int kh, int kw,
                   int sh, int sw, int ph, int pw) {
  ConvGeom g;
  g.N = (int)N; g.C = (int)C; g.H = (int)H; g.W = (int)W; g.K = (int)K; g.Hz = (int)Hz; g.Wz = (int)Wz;
  g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw;
  return g;
}

}  // namespace
}  // namespace lasso

namespace lasso {
hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
  // (kernel, device) -> the largest size set so far: kernels whose LDS size depends on the geometry (the convolutional
  // kernels) raise the attribute again when a later call needs more (the cache ignored the byte count)
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({kernel, dev});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[{kernel, dev}] = bytes;
  return e;
}
}  // namespace lasso

using namespace lasso;

extern "C" {

int lasso_hip_abi_version(void) { return LASSO_HIP_ABI_VERSION; }

const char* lasso_hip_status_string(int status) {
  switch (status) {
    case LASSO_OK: return "ok";
    case LASSO_ERR_BAD_ARG: return "bad argument";
    case LASSO_ERR_UNSUPPORTED: return "unsupported shape or dtype";
    case LASSO_ERR_WORKSPACE: return "workspace too small";
    case LASSO_ERR_HIP: return "HIP runtime error";
    case LASSO_WARN_LINESEARCH: return "backtracking line search failed; reverted to initial step";
    default: return "unknown status";
  }
}

const char* lasso_hip_last_error(void) { return g_err; }

int lasso_debug_force_standby(int on) {
  const int prev = lasso::g_force_standby;
  lasso::g_force_standby = on ? 1 : 0;
  return prev;
}

int lasso_hip_device_cus(int* cus_out) {
  const int c = device_cus();
  if (cus_out) *cus_out = c;
  return c > 0 ? LASSO_OK : fail(LASSO_ERR_HIP, "no HIP device visible");
}

// bytes of the solver's own workspace (the objective_out region follows it)
static size_t solver_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                                     int stop_mode, int backtrack) {
  stop_mode &= 0xFF;                                   // kernel hint / LASSO_SOLVE_ASYNC bits ride above
  const bool with_state = tol > 0.0 && stop_mode != LASSO_STOP_NONE && maxiter > 0;
  if (!fused_shape(d, k)) return carve_generic(nullptr, n, d, k, backtrack != 0, with_state).bytes;
  const int kp = pad_k(k);
  if (kp < 0) return 0;
  if (backtrack || dtype == LASSO_BF16) return carve_bt(nullptr, n, k, kp, dtype == LASSO_BF16, maxiter).bytes;
  // Every padded dictionary size solve_geometry() / the kernel hints can pick for this shape: a SMALLER kp means more
  // split-k groups, i.e. a LARGER exchange region (carve(384) needs ~2.2 MiB more than carve(512)), so the size is the
  // maximum over the candidates, not the size at pad_k(k) (an exactly-sized workspace failed at
  // d=64, k=300 and for most n at d <= 32).
  size_t bytes = carve(nullptr, n, k, kp, maxiter, with_state).bytes;
  if (kp == 512 && k <= 384 && d <= 128) bytes = std::max(bytes, carve(nullptr, n, k, 384, maxiter, with_state).bytes);
  if (kp == 1024 && k <= 768) bytes = std::max(bytes, carve(nullptr, n, k, 768, maxiter, with_state).bytes);
  // (LASSO_KERNEL_UNFUSED / the cost model may send a fused shape down the general-GEMM path: room for either)
  return std::max(bytes, carve_generic(nullptr, n, d, k, false, with_state).bytes);
}

// region behind the solver's workspace that objective_out needs: the lasso_objective workspace,
// a device float, and -- bf16 tensors -- fp32 copies of x, W and z
static size_t objective_region_bytes(int64_t n, int64_t d, int64_t k, int dtype) {
  size_t b = align_up(lasso_objective_workspace_bytes(n, d, k)) + 256;
  if (dtype == LASSO_BF16)
    b += align_up((size_t)n * d * 4) + align_up((size_t)d * k * 4) + align_up((size_t)n * k * 4);
  return b;
}

// region behind that for lr = LASSO_LR_AUTO: the Lipschitz workspace, then {lr, alpha*lr} as floats
static size_t lipschitz_region_bytes(int64_t d, int64_t k, int dtype) {
  if (dtype != LASSO_F32 || std::min(d, k) > 2048) return 0;
  return align_up(lipschitz_workspace_bytes(d, k)) + 256;
}

// {lr, alpha*lr} in fp32 from L (double) with the roundings of the host path: lr = 1/L in
// double (ista.py:72-73), then the casts of run_impl()
__global__ void step_from_lipschitz_kernel(const double* __restrict__ L, double alpha, float* __restrict__ out) {
  const double lr = 1.0 / L[0];
  out[0] = (float)lr;
  out[1] = (float)(alpha * lr);
}

const char* lasso_fista_kernel_name(int64_t n, int64_t d, int64_t k, int dtype, int backtrack) {
  if (n <= 0 || d <= 0 || k <= 0) return "";
  if (!fused_shape(d, k)) return "lasso::gemm_nt_kernel + lasso::gemm_nt_kernel<.., prox epilogue> (unfused)";
  if (dtype == LASSO_BF16) {
    int per_cu = 0;
    const int kpb = pad_k(k);
    if (bt16_persist_occupancy(kpb, &per_cu) == hipSuccess && (n + 63) / 64 <= (int64_t)per_cu * device_cus())
      return kpb == 1024 ? "lasso::bt16_persist_kernel<1024>" : kpb == 512 ? "lasso::bt16_persist_kernel<512>"
                                                                            : "lasso::bt16_persist_kernel<256>";
    return backtrack ? "lasso::bt16_grad_kernel / bt16_trial_kernel" : "lasso::bt16_grad_kernel + lasso::generic_prox_kernel";
  }
  if (backtrack) {
    // what run_bt() dispatches for fp32 tensors in one process (row pitch == k, 16-byte-aligned state as torch and
    // hipMalloc hand it out): ONE bt_iter_kernel launch per outer iteration (accept + gradient + the first trials per
    // tile; the first launch of a solve is the <K, false> instantiation: nothing to accept) + ONE bt_iter_decide_kernel;
    // k not a multiple of 4 takes the multi-launch kernels of backtrack.hip
    const int kpb = pad_k(k);
    if ((k & 3) == 0 && k >= 4) {
      static thread_local char bname[160];
      snprintf(bname, sizeof(bname), "lasso::bt_iter_kernel<%d, true> / bt_iter_kernel<%d, false> + lasso::bt_iter_decide_kernel",
               kpb, kpb);
      return bname;
    }
    return "lasso::bt_grad_kernel / bt_trial_kernel";
  }
  const SolveGeom geom = solve_geometry(n, d, k);
  const NarrowTiles narrow(geom.narrow);
  const int kp = geom.kp, dpad = pad_d(d, kp);
  const KernelPlan plan = plan_kernel(kp, dpad, (int)((n + kTileM - 1) / kTileM), false, LASSO_KERNEL_AUTO);
  const TilePlan tp = plan_tiles(n, dpad, kp);
  KernelPlan tplan = {false, 0, 1, 0.0};
  const int tail = ragged_tail(kp, dpad, tp.ntiles, tp.waves, LASSO_KERNEL_AUTO, plan, &tplan);
  if (plan.split && !tail) {
    static thread_local char name[96];
    if (kp == 1024 && plan.tiles >= 2) snprintf(name, sizeof(name), "lasso::splitk::fista_splitk_rs_kernel<%d>", plan.tiles);
    else snprintf(name, sizeof(name), "lasso::splitk::fista_splitk_kernel<%d, %d, false>", kp, plan.tiles);
    return name;
  }
  static thread_local char tname[160];
  if (tail)
    snprintf(tname, sizeof(tname), "lasso::sp::fista_tile_sp_kernel<%d, %d, false, %d> + the last %d tiles on the split-k kernel (T = %d)",
             kp, tp.rows, tp.waves, tail, tplan.tiles);
  else
  snprintf(tname, sizeof(tname), "lasso::sp::fista_tile_sp_kernel<%d, %d, false, %d>", kp, tp.rows, tp.waves);
  return tname;
}

size_t lasso_fista_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype, int maxiter,
                                   double tol, int stop_mode, int backtrack) {
  if (n < 0 || d <= 0 || k <= 0) return 0;
  const size_t solver = solver_workspace_bytes(n, d, k, dtype, maxiter, tol, stop_mode, backtrack);
  if (solver == 0) return 0;
  return align_up(solver) + objective_region_bytes(n, d, k, dtype) + lipschitz_region_bytes(d, k, dtype);
}

int lasso_fista_prepare(const void* w_dev, int64_t ldw, int64_t d, int64_t k, int dtype, int maxiter,
                        void* workspace_dev, size_t workspace_bytes, void* stream) {
  // packs W and builds the momentum table for iterations 0 .. maxiter-1 (the workspace must
  // come from lasso_fista_workspace_bytes with the same maxiter)
  if (int s = check_common(0, d, k, dtype, /*allow_large=*/true)) return s;
  if (!w_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldw < k || maxiter < 0) return fail(LASSO_ERR_BAD_ARG, "ldw < k or maxiter < 0");
  if (!fused_shape(d, k)) {          // unfused path: W^T for the second GEMM (the workspace of solve_generic)
    if (d > INT32_MAX || k > INT32_MAX) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
    GenWorkspace gw = carve_generic(workspace_dev, 0, d, k, false, false);
    if (workspace_bytes < gw.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", gw.bytes);
    LASSO_HIP_TRY(launch_transpose_pad((const float*)w_dev, ldw, (int)d, (int)k, gw.Wt, d, (int)k, (int)d,
                                       (hipStream_t)stream));
    LASSO_HIP_TRY(hipMemcpy2DAsync(gw.Wc, k * 4, w_dev, ldw * 4, k * 4, d, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return LASSO_OK;
  }
  const int kp = pad_k(k);
  Workspace ws = carve(workspace_dev, 0, k, kp, maxiter, false);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", ws.bytes);
  return prepare_impl(ws, kp, (const float*)w_dev, ldw, d, k, maxiter, (hipStream_t)stream);
}

int lasso_fista_run(const void* x_dev, int64_t ldx, const void* z_in_dev, int64_t ldz_in,
                    const void* y_in_dev, int64_t ldy_in, void* z_out_dev, int64_t ldz_out,
                    void* y_out_dev, int64_t ldy_out, int64_t n, int64_t d, int64_t k, int dtype,
                    double alpha, double lr, int fast, int it0, int iters, int maxiter, int kernel_hint,
                    float* delta_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (int s = check_common(n, d, k, dtype, /*allow_large=*/true)) return s;
  if (!x_dev || !z_out_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (it0 < 0 || iters < 0) return fail(LASSO_ERR_BAD_ARG, "negative iteration range");
  if (!fused_shape(d, k)) {
    if (it0 + iters > maxiter) return fail(LASSO_ERR_BAD_ARG, "it0 + iters = %d > maxiter = %d", it0 + iters, maxiter);
    if (ldx < d || ldz_out < k || (z_in_dev && ldz_in < k) || (y_in_dev && ldy_in < k) || (y_out_dev && ldy_out < k))
      return fail(LASSO_ERR_BAD_ARG, "leading dimension smaller than the row length");
    return run_generic((const float*)x_dev, ldx, (const float*)z_in_dev, ldz_in,
                       (const float*)y_in_dev, ldy_in, (float*)z_out_dev, ldz_out, (float*)y_out_dev, ldy_out, n, d, k,
                       alpha, lr, fast, it0, iters, delta_dev, workspace_dev, workspace_bytes, (hipStream_t)stream);
  }
  if (delta_dev && iters > kChunkMax)
    return fail(LASSO_ERR_BAD_ARG, "iters=%d > %d with delta recording", iters, kChunkMax);
  if (ldx < d || ldz_out < k || (z_in_dev && ldz_in < k) || (y_in_dev && ldy_in < k) ||
      (y_out_dev && ldy_out < k))
    return fail(LASSO_ERR_BAD_ARG, "leading dimension smaller than the row length");
  const int kp = pad_k(k);
  if (it0 + iters > maxiter) return fail(LASSO_ERR_BAD_ARG, "it0 + iters = %d > maxiter = %d", it0 + iters, maxiter);
  if ((kernel_hint & ~LASSO_KERNEL_MASK) || ((kernel_hint & 0x300) != LASSO_KERNEL_AUTO &&
      (kernel_hint & 0x300) != LASSO_KERNEL_TILE && (kernel_hint & 0x300) != LASSO_KERNEL_SPLITK))
    return fail(LASSO_ERR_BAD_ARG, "kernel hint 0x%x", kernel_hint);
  // same carve as lasso_fista_prepare(maxiter): the momentum table it built is read here
  Workspace ws = carve(workspace_dev, n, k, kp, maxiter, false);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", ws.bytes);
  hipStream_t st = (hipStream_t)stream;
  return run_impl(ws, kp, (const float*)x_dev, ldx, (const float*)z_in_dev, ldz_in,
                  (const float*)y_in_dev, ldy_in, (float*)z_out_dev, ldz_out, (float*)y_out_dev,
                  ldy_out, n, d, k, alpha, lr, fast, it0, iters, delta_dev, st, -1.0f, kernel_hint);
}

static int solve_impl(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                      const void* z0_dev, int64_t ldz0, void* z_out_dev, int64_t ldz, int64_t n,
                      int64_t d, int64_t k, int dtype, double alpha, double lr, int fast,
                      int maxiter, double tol, int stop_mode, int backtrack, double eta_backtrack,
                      int32_t* iters_out, float* last_delta_out, int32_t* trials_out,
                      float* accepted_lr_out, float* accepted_f_out, void* workspace_dev,
                      size_t workspace_bytes, void* stream, const float* lr_dev = nullptr, bool async = false,
                      const double* lip_dev = nullptr, bool sharded = false, bool one_chunk = false,
                      int32_t* status_mapped = nullptr, void* lip_deferred_ws = nullptr, bool defer_verdict = false) {
  // LASSO_BF16 (x, W, z0, z_out all bf16) is native on the fused shapes
  const bool half_any = dtype == LASSO_BF16 && fused_shape(d, k) && maxiter > 0 && n > 0;
  const bool half_bt = half_any && backtrack;
  if (int s = check_common(n, d, k, half_any ? LASSO_F32 : dtype, /*allow_large=*/true)) return s;
  if (!x_dev || !w_dev || !z_out_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (maxiter < 0) return fail(LASSO_ERR_BAD_ARG, "maxiter < 0");
  if (!fused_shape(d, k) && backtrack && dtype != LASSO_F32)
    return fail(LASSO_ERR_UNSUPPORTED, "backtrack=1 beyond d<=%d, k<=%d needs fp32 tensors", kFistaD, kFistaMaxK);
  if (ldx < d || ldw < k || ldz < k || (z0_dev && ldz0 < k))
    return fail(LASSO_ERR_BAD_ARG, "leading dimension smaller than the row length");
  if (!(lr > 0.0) || !(alpha >= 0.0)) return fail(LASSO_ERR_BAD_ARG, "need lr > 0 and alpha >= 0");
  if (backtrack && !(eta_backtrack > 1.0)) return fail(LASSO_ERR_BAD_ARG, "eta must be > 1.");
  hipStream_t st = (hipStream_t)stream;
  if (iters_out) *iters_out = 0;
  if (last_delta_out) *last_delta_out = NAN;
  const float* x = (const float*)x_dev;
  const float* z0 = (const float*)z0_dev;
  float* zout = (float*)z_out_dev;

  if (maxiter == 0 || n == 0) {   // ista.py:76,104: returns z0 itself
    if (n > 0) {
      if (z0) {
        if (z0 != zout)
          LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, z0, ldz0 * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
      } else {
        LASSO_HIP_TRY(hipMemset2DAsync(zout, ldz * 4, 0, k * 4, n, st));
      }
    }
    return LASSO_OK;
  }

  int hint = stop_mode & LASSO_KERNEL_MASK;            // kernel-selection hint rides in stop_mode
  stop_mode &= ~LASSO_KERNEL_MASK;
  if (stop_mode != LASSO_STOP_GLOBAL && stop_mode != LASSO_STOP_NONE && stop_mode != LASSO_STOP_GLOBAL_CHUNKED)
    return fail(LASSO_ERR_BAD_ARG, "stop_mode %d", stop_mode);
  // defined hints: AUTO (optionally | 0x800 = narrow tiles), TILE, UNFUSED, SPLITK with its T / exchange knobs
  if (((hint & 0x300) != LASSO_KERNEL_SPLITK && (hint & 0x3400)) ||
      ((hint & 0x300) != LASSO_KERNEL_SPLITK && (hint & 0x300) != LASSO_KERNEL_AUTO && (hint & 0x800)) ||
      ((hint & 0x400) && (hint & 0x800)))
    return fail(LASSO_ERR_BAD_ARG, "kernel hint 0x%x", hint);
  const bool want_unfused = (hint & 0x300) == LASSO_KERNEL_UNFUSED;
  const bool stop_rule = tol > 0.0 && stop_mode != LASSO_STOP_NONE;
  if (!workspace_dev) return fail(LASSO_ERR_WORKSPACE, "workspace is null");
  if (!fused_shape(d, k) && backtrack)
    return solve_generic_backtracking(x, ldx, (const float*)w_dev, ldw, z0, ldz0, zout, ldz, n, d, k, alpha, lr, fast,
                                      maxiter, stop_rule ? tol : 0.0, eta_backtrack, iters_out, last_delta_out,
                                      trials_out, accepted_lr_out, accepted_f_out, workspace_dev, workspace_bytes, st);
  if (!fused_shape(d, k) || (want_unfused && !backtrack && dtype == LASSO_F32 && !lr_dev && !async))
    return solve_generic(x, ldx, (const float*)w_dev, ldw, z0, ldz0, zout, ldz, n, d, k, alpha, lr, fast,
                         maxiter, stop_rule ? tol : 0.0, iters_out, last_delta_out, workspace_dev,
                         workspace_bytes, st);
  const int kp = pad_k(k);
  if (half_any && (hint & 0x300) != LASSO_KERNEL_TILE) {
    // bf16 tensors: the persistent single-launch kernel when every 64-row tile has its own
    // resident workgroup (LASSO_KERNEL_TILE asks for the multi-launch kernels instead)
    bool ran = false;
    const void* z0_fb = z0_dev;
    int per_cu = 0;
    const int64_t cap = bt16_persist_occupancy(kp, &per_cu) == hipSuccess ? (int64_t)per_cu * device_cus() * 64 : 0;
    if (!backtrack && !stop_rule && cap > 0 && n > cap) {
      // Fixed step without the stop rule: rows are independent, so a batch beyond the persistent kernel's capacity
      // (one resident workgroup per 64-row tile: 16384 rows) is a sequence of launches over row blocks of that size
      // instead of the multi-launch kernels (2.7x slower per row).  With the stop rule or the line search the
      // decisions need sums over ALL rows at once -- those keep the multi-launch path.
      const char* xb = (const char*)x_dev; const char* zb = (const char*)z0_dev; char* ob = (char*)z_out_dev;
      bool all = true;
      int64_t r0 = 0;
      for (; r0 < n; r0 += cap) {
        const int64_t nb = std::min(cap, n - r0);
        bool ran_b = false;
        const void* fb = nullptr;
        const int sb = solve_bf16_persistent(xb + r0 * ldx * 2, ldx, w_dev, ldw, zb ? zb + r0 * ldz0 * 2 : nullptr, ldz0,
                                             ob + r0 * ldz * 2, ldz, nb, d, k, kp, alpha, lr, fast, maxiter, 0.0, 0,
                                             eta_backtrack, nullptr, nullptr, nullptr, nullptr, nullptr, workspace_dev,
                                             workspace_bytes, st, &ran_b, &fb);
        if (sb != LASSO_OK) return sb;
        if (!ran_b) { all = false; break; }       // a block gave up (busy GPU): the multi-launch path redoes everything
      }
      if (all) {
        if (iters_out) *iters_out = maxiter;
        return LASSO_OK;
      }
      if (z0_dev && z0_dev == z_out_dev && r0 > 0)
        return fail(LASSO_ERR_HIP, "in-place bf16 solve interrupted after %lld rows", (long long)r0);
    } else {
    const int s = solve_bf16_persistent(x_dev, ldx, w_dev, ldw, z0_dev, ldz0, z_out_dev, ldz, n, d, k, kp, alpha, lr,
                                        fast, maxiter, stop_rule ? tol : 0.0, backtrack, eta_backtrack, iters_out,
                                        last_delta_out, trials_out, accepted_lr_out, accepted_f_out, workspace_dev,
                                        workspace_bytes, st, &ran, &z0_fb);
    if (ran || (s != LASSO_OK && s != LASSO_WARN_LINESEARCH)) return s;
    z0_dev = z0_fb;
    if (z0_fb != z0) ldz0 = k;
    }
  }
  if (half_any && !backtrack)
    return solve_fixed_bf16(x_dev, ldx, w_dev, ldw, z0_dev, ldz0, z_out_dev, ldz, n, d, k, kp, alpha, lr, fast,
                            maxiter, stop_rule ? tol : 0.0, iters_out, last_delta_out, workspace_dev,
                            workspace_bytes, st);
  if (backtrack)
    return solve_backtracking(x_dev, ldx, w_dev, ldw, z0_dev, ldz0, z_out_dev, ldz, n, d, k, kp,
                              half_bt ? LASSO_BF16 : LASSO_F32, alpha, lr, fast, maxiter,
                              stop_rule ? tol : 0.0, eta_backtrack,
                              iters_out, last_delta_out, trials_out, accepted_lr_out, accepted_f_out, workspace_dev,
                              workspace_bytes, st, nullptr, nullptr, 0,
                              // A/B knobs of the fp32 line search: TILE = one trial per launch, SPLITK = round 4's
                              // multi-launch form (gradient / trials / accept as separate launches)
                              (hint & 0x300) == LASSO_KERNEL_TILE ? 1 : (hint & 0x300) == LASSO_KERNEL_SPLITK ? 2 : 0);
  const int kps = pad_k_solve(n, d, k);                  // (fp32 fixed step from here on: 768 atoms have their own tile kernel)
  const NarrowTiles narrow(narrow_tiles(n, d, k, kps, hint));
  Workspace ws = carve(workspace_dev, n, k, kps, maxiter, stop_rule);
  if (workspace_bytes < ws.bytes)
    return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  // one launch: pack W, momentum table, {lr, alpha*lr} from lambda_max (lr = LASSO_LR_AUTO), and the
  // zeroing of the in-kernel stop rule's granule ring and result words
  PrepareExtras px = {nullptr, 0, nullptr, 0, lip_dev, alpha, const_cast<float*>(lr_dev)};
  const TilePlan tp0 = plan_tiles(n, pad_d(d, kps), kps);
  if (stop_rule && stop_mode == LASSO_STOP_GLOBAL) {
    px.zero_a = ws.gran; px.words_a = kStopRing * std::max(tp0.ntiles, kSplitMaxParts);
    px.zero_b = reinterpret_cast<unsigned long long*>(ws.stop_out); px.words_b = 2;
  }
  // (lr = LASSO_LR_AUTO: the Lipschitz launches were left to this point -- their Gram launch carries the prepare blocks)
  if (lip_deferred_ws) {
    if (int s = prepare_with_lipschitz(ws, kps, (const float*)w_dev, ldw, d, k, maxiter, st, px, lip_deferred_ws)) return s;
  } else if (int s = prepare_impl(ws, kps, (const float*)w_dev, ldw, d, k, maxiter, st, &px)) return s;

  if (!stop_rule) {
    if (int s = run_impl(ws, kps, x, ldx, z0, ldz0, nullptr, 0, zout, ldz, nullptr, 0, n, d, k,
                         alpha, lr, fast, 0, maxiter, nullptr, st, -1.0f, hint, nullptr, lr_dev))
      return s;
    if (iters_out) *iters_out = maxiter;
    return LASSO_OK;
  }

  const float budget = (float)((double)n * (double)k * tol);   // ista.py:64, compared in fp32
  // ---- exact global stop rule, in-kernel: when every tile has its own resident workgroup
  // the persistent kernel evaluates the rule itself (one-iteration lag) and a
  // single launch runs to the stopping iteration -- one host sync, at the end. ------------
  const float* cur_z = z0;  int64_t cur_ldz = ldz0;
  const float* cur_y = nullptr; int64_t cur_ldy = 0;
  if (z0 && z0 == zout) {     // aliasing: keep the initial state intact for a replay / a second attempt
    LASSO_HIP_TRY(hipMemcpy2DAsync(ws.state[2], k * 4, z0, ldz0 * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
    cur_z = ws.state[2]; cur_ldz = k;
  }
  if (sharded) {
    // LASSO_SOLVE_SHARDED: this batch is a row shard -- the rule needs the other ranks' sums.  One chunk, its
    // per-iteration sums left in ws.delta for the caller's all-reduce; lasso_fista_solve_verdict judges them.
    if (!async || stop_mode != LASSO_STOP_GLOBAL)
      return fail(LASSO_ERR_BAD_ARG, "LASSO_SOLVE_SHARDED needs LASSO_SOLVE_ASYNC | LASSO_STOP_GLOBAL");
    if (maxiter > kChunkMax)
      return fail(LASSO_ERR_UNSUPPORTED, "LASSO_SOLVE_SHARDED: maxiter=%d > %d", maxiter, kChunkMax);
    // (y_out = NULL: the one chunk is the whole solve -- a verdict of "redo" repeats it from its start, nothing
    // continues from its momentum point; round 6: the kernel no longer writes the n x k tile of y nobody reads)
    if (int s = run_impl(ws, kps, x, ldx, cur_z, cur_ldz, nullptr, 0, zout, ldz, nullptr, 0, n, d, k,
                         alpha, lr, fast, 0, maxiter, ws.delta, st, -1.0f, hint, nullptr, lr_dev))
      return s;
    return LASSO_PENDING;
  }
  if (stop_mode == LASSO_STOP_GLOBAL) {
    const TilePlan tp = plan_tiles(n, pad_d(d, kps), kps);
    const int ntiles = tp.ntiles;
    // The handshake needs every workgroup of the grid resident at once: one workgroup per CU
    // (LDS-bound), so the grid must not exceed what the occupancy query admits.  CUs held by
    // OTHER work (a second stream, another process) are invisible to that query: then the
    // handshake times out, the kernel aborts as a whole without touching z_out, and the solve
    // is repeated on the chunked path below.
    // (the rule's granule fetch covers 256 tiles: four per lane of one wave)
    // (LASSO_SOLVE_ONE_CHUNK: the caller expects all `maxiter` iterations to run -- the plain kernels + the device
    // verdict below are cheaper per iteration than the in-kernel rule's exchange)
    const bool as_chunk = one_chunk && async && maxiter <= kChunkMax && !(z0 && z0 == zout);
    if (!as_chunk && ntiles <= std::min(fista_resident_workgroups(kps, pad_d(d, kps), tp.waves), 256)) {
      // (granule ring and stop_out were zeroed by the prepare launch)
      if (int s = run_impl(ws, kps, x, ldx, cur_z, cur_ldz, nullptr, 0, zout, ldz, nullptr, 0, n, d, k,
                           alpha, lr, fast, 0, maxiter, nullptr, st, budget, hint, nullptr, lr_dev))
        return s;
      if (async) return LASSO_PENDING;     // the caller collects {iterations, last delta, abort flag} later
      int hout[4] = {0, 0, 0, 0};
      LASSO_HIP_TRY(hipMemcpyAsync(hout, ws.stop_out, 16, hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipStreamSynchronize(st));
      if (!hout[2]) {
        float lastf;
        memcpy(&lastf, &hout[1], sizeof(float));
        if (iters_out) *iters_out = hout[0];
        if (last_delta_out) *last_delta_out = lastf;
        return LASSO_OK;
      }
      hint = LASSO_KERNEL_TILE;      // some workgroup was not resident: no cross-workgroup traffic from here on
    }
  }
  // ---- LASSO_SOLVE_ASYNC beyond the in-kernel rule (more tiles than resident workgroups) when maxiter fits
  // one chunk: the chunk is enqueued, a one-thread kernel turns its per-iteration deltas into the words the
  // collect call copies -- {iterations, last delta, redo} -- and the call returns without waiting.  "redo" is
  // set when the rule fired BEFORE the last iteration (z_out is then one of the later iterates): the caller
  // repeats the solve synchronously, exactly like after an aborted handshake.  The E-step of an EM loop
  // (maxiter = 10) practically never stops early, and no longer makes the GPU wait for the host. -------------
  if (async && stop_mode == LASSO_STOP_GLOBAL && maxiter <= kChunkMax && !(z0 && z0 == zout)) {
    // LASSO_SOLVE_DEFER_VERDICT: the verdict launch is left to lasso_fista_solve_verdict_deferred (another stream)
    const bool defer = defer_verdict && status_mapped && maxiter <= 64 && n > 0;
    t_deferred_verdict.armed = false;
    t_deferred_verdict.workspace = workspace_dev;
    const ChunkVerdict cv{budget, ws.stop_out, status_mapped, defer ? &t_deferred_verdict : nullptr};
    if (int s = run_impl(ws, kps, x, ldx, cur_z, cur_ldz, nullptr, 0, zout, ldz, nullptr, 0, n, d, k,
                         alpha, lr, fast, 0, maxiter, ws.delta, st, -1.0f, hint, nullptr, lr_dev, &cv))
      return s;
    if (maxiter > 64 || n == 0) {
      hipLaunchKernelGGL(chunk_verdict_kernel, dim3(1), dim3(1), 0, st, ws.delta, maxiter, budget, ws.stop_out,
                         status_mapped);
      LASSO_HIP_TRY(hipGetLastError());
    }
    if (defer && t_deferred_verdict.armed) return LASSO_PENDING_DEFERRED;
    return status_mapped ? LASSO_PENDING_MAPPED : LASSO_PENDING;
  }
  // ---- exact global stop rule, chunked: speculate a chunk, read its per-iteration deltas,
  // replay the chunk up to the stopping iteration if one fired ----------
  int done = 0, flip = 0;
  std::vector<float> hdelta(kChunkMax);
  float last = NAN;
  int next_chunk = kChunkMax;
  while (done < maxiter) {
    const int c = std::min(next_chunk, maxiter - done);
    const bool final_chunk = (done + c == maxiter);
    float* nz = final_chunk ? zout : ws.state[2 * flip];
    const int64_t nldz = final_chunk ? ldz : k;
    float* ny = ws.state[2 * flip + 1];
    // the aliasing copy above used state[2]; first chunk writes state[0]/[1] (flip = 0)
    if (int s = run_impl(ws, kps, x, ldx, cur_z, cur_ldz, cur_y, cur_ldy, nz, nldz, ny, k, n, d, k,
                         alpha, lr, fast, done, c, ws.delta, st, -1.0f, hint, nullptr, lr_dev))
      return s;
    LASSO_HIP_TRY(hipMemcpyAsync(hdelta.data(), ws.delta, c * sizeof(float), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    int hit = -1;
    for (int i = 0; i < c; ++i) {
      last = hdelta[i];
      if (hdelta[i] <= budget) { hit = i; break; }
    }
    if (hit >= 0) {
      if (hit + 1 < c) {   // replay the chunk from its (intact) input state
        if (int s = run_impl(ws, kps, x, ldx, cur_z, cur_ldz, cur_y, cur_ldy, zout, ldz, nullptr, 0,
                             n, d, k, alpha, lr, fast, done, hit + 1, nullptr, st, -1.0f, hint, nullptr, lr_dev))
          return s;
      } else if (!final_chunk) {
        LASSO_HIP_TRY(hipMemcpy2DAsync(zout, ldz * 4, nz, nldz * 4, k * 4, n, hipMemcpyDeviceToDevice, st));
      }
      done += hit + 1;
      if (iters_out) *iters_out = done;
      if (last_delta_out) *last_delta_out = last;
      return LASSO_OK;
    }
    done += c;
    cur_z = nz; cur_ldz = nldz; cur_y = ny; cur_ldy = k;
    flip ^= 1;
    // Size the next speculative chunk from the decay of the deltas seen so far (purely a
    // scheduling heuristic -- the stop decision itself stays exact): estimate the
    // iterations left until delta <= budget from the geometric decay over the chunk and
    // approach the predicted stop with short chunks so that little work is wasted or replayed.
    next_chunk = kChunkMax;
    if (c >= 8 && budget > 0.0f) {
      const int h = c / 2;
      float hi = 0.0f, lo = 0.0f;
      for (int i = 0; i < h; ++i) hi = std::max(hi, hdelta[i]);
      for (int i = h; i < c; ++i) lo = std::max(lo, hdelta[i]);
      if (lo > budget && hi > lo) {
        const double rate = log((double)hi / lo) / h;              // per-iteration log decay
        const double left = log((double)lo / budget) / rate;       // iterations still needed
        if (left < 2.0 * kChunkMax) {
          const int guess = (int)left - 6;
          next_chunk = guess >= kChunkMax ? kChunkMax : std::max(8, std::min(guess, kChunkMax));
        }
      } else if (lo <= budget * 4.0f) {
        next_chunk = 8;
      }
    }
  }
  if (iters_out) *iters_out = done;
  if (last_delta_out) *last_delta_out = last;
  return LASSO_OK;
}

int lasso_fista_solve(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw,
                      const void* z0_dev, int64_t ldz0, void* z_out_dev, int64_t ldz, int64_t n,
                      int64_t d, int64_t k, int dtype, double alpha, double lr, int fast,
                      int maxiter, double tol, int stop_mode, int backtrack, double eta_backtrack,
                      int32_t* iters_out, float* last_delta_out, int32_t* trials_out,
                      float* accepted_lr_out, float* accepted_f_out, float* objective_out, void* workspace_dev,
                      size_t workspace_bytes, void* stream) {
  if (objective_out) *objective_out = NAN;
  const bool async = (stop_mode & LASSO_SOLVE_ASYNC) != 0;
  const bool sharded = (stop_mode & LASSO_SOLVE_SHARDED) != 0;
  const bool one_chunk = (stop_mode & LASSO_SOLVE_ONE_CHUNK) != 0;
  int32_t* status_mapped = nullptr;
  if (stop_mode & LASSO_SOLVE_STATUS_MAPPED) {      // iters_out is a device-writable host buffer of four words
    if (!async || !iters_out) return fail(LASSO_ERR_BAD_ARG, "LASSO_SOLVE_STATUS_MAPPED needs LASSO_SOLVE_ASYNC and iters_out");
    status_mapped = iters_out;
    iters_out = nullptr; last_delta_out = nullptr;
  }
  const bool defer_verdict = (stop_mode & LASSO_SOLVE_DEFER_VERDICT) != 0;
  if (defer_verdict && (!(status_mapped && one_chunk) || sharded))
    return fail(LASSO_ERR_BAD_ARG, "LASSO_SOLVE_DEFER_VERDICT needs LASSO_SOLVE_STATUS_MAPPED and LASSO_SOLVE_ONE_CHUNK, without LASSO_SOLVE_SHARDED");
  stop_mode &= ~(LASSO_SOLVE_ASYNC | LASSO_SOLVE_SHARDED | LASSO_SOLVE_ONE_CHUNK | LASSO_SOLVE_STATUS_MAPPED |
                 LASSO_SOLVE_DEFER_VERDICT);
  if (sharded && !(async && tol > 0.0 && maxiter > 0 && n > 0 && fused_shape(d, k)))
    return fail(LASSO_ERR_UNSUPPORTED, "LASSO_SOLVE_SHARDED: asynchronous fp32 solves with tol > 0 on the fused shapes only");
  if (async && (objective_out || backtrack || dtype != LASSO_F32))
    return fail(LASSO_ERR_BAD_ARG, "LASSO_SOLVE_ASYNC: fp32 fixed-step solves without objective_out only");
  const float* lr_dev = nullptr;
  const double* lip_dev = nullptr;
  void* lip_deferred = nullptr;
  if (lr == LASSO_LR_AUTO) {
    // lr = 1/L, L = lambda_max(W^T W) (ista.py:72-73, :8-14) computed here on the stream.  The fp32
    // fixed-step kernels read {lr, alpha*lr} from device memory -- no host round trip; the other
    // paths need the step size on the host (line-search bookkeeping, the unfused driver).
    if (dtype != LASSO_F32) return fail(LASSO_ERR_BAD_ARG, "lr = LASSO_LR_AUTO needs fp32 tensors (ista.py:12)");
    if (!x_dev || !w_dev || d <= 0 || k <= 0 || n < 0 || ldw < k) return fail(LASSO_ERR_BAD_ARG, "bad argument");
    const size_t lip = lipschitz_region_bytes(d, k, dtype);
    if (lip == 0) return fail(LASSO_ERR_UNSUPPORTED, "lr = LASSO_LR_AUTO: min(d,k) > 2048");
    const size_t before = align_up(solver_workspace_bytes(n, d, k, dtype, maxiter, tol, stop_mode, backtrack)) +
                          objective_region_bytes(n, d, k, dtype);
    if (!workspace_dev || workspace_bytes < before + lip)
      return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, before + lip);
    hipStream_t st = (hipStream_t)stream;
    char* const lip_ws = (char*)workspace_dev + before;
    if (fused_shape(d, k) && !backtrack && maxiter > 0 && n > 0) {
      float* const slot = (float*)(lip_ws + align_up(lipschitz_workspace_bytes(d, k)));
      lip_dev = (const double*)lip_ws;                 // solve_impl enqueues the Lipschitz launches with its prepare blocks;
      lip_deferred = lip_ws;                           //   the launch that finishes lambda_max fills the slot
      lr_dev = slot;
      lr = 1.0;                                        // placeholder, never used by the kernels
    } else {
      LASSO_HIP_TRY(launch_lipschitz((const float*)w_dev, ldw, d, k, lip_ws, 20, st));
      double L = 0.0;
      LASSO_HIP_TRY(hipMemcpyAsync(&L, lip_ws, sizeof(double), hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipStreamSynchronize(st));
      lr = 1.0 / L;
    }
  }
  const int status = solve_impl(x_dev, ldx, w_dev, ldw, z0_dev, ldz0, z_out_dev, ldz, n, d, k, dtype, alpha, lr,
                                fast, maxiter, tol, stop_mode, backtrack, eta_backtrack, iters_out,
                                last_delta_out, trials_out, accepted_lr_out, accepted_f_out, workspace_dev,
                                workspace_bytes, stream, lr_dev, async, lip_dev, sharded, one_chunk, status_mapped, lip_deferred,
                                defer_verdict);
  if ((status != LASSO_OK && status != LASSO_WARN_LINESEARCH) || !objective_out || n <= 0) return status;
  // objective_out: (0.5*||x - z W^T||^2 + alpha*||z||_1)/n of the RETURNED code, evaluated in fp32
  // (the verbose print of ista.py:66-69,80-81 for the final iterate; dict_learning.py:10-13)
  char saved[sizeof(g_err)];
  memcpy(saved, g_err, sizeof(saved));          // keep the line-search warning text
  const size_t solver = align_up(solver_workspace_bytes(n, d, k, dtype, maxiter, tol, stop_mode, backtrack));
  if (!workspace_dev || workspace_bytes < solver + objective_region_bytes(n, d, k, dtype))
    return fail(LASSO_ERR_WORKSPACE, "objective_out: workspace %zu < %zu bytes", workspace_bytes,
                solver + objective_region_bytes(n, d, k, dtype));
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace_dev + solver;
  float* loss_dev = (float*)base;
  char* obj_ws = base + 256;
  const size_t obj_bytes = align_up(lasso_objective_workspace_bytes(n, d, k));
  const void* xo = x_dev; const void* wo = w_dev; const void* zo = z_out_dev;
  int64_t ldxo = ldx, ldwo = ldw, ldzo = ldz;
  if (dtype == LASSO_BF16) {
    float* xf = (float*)(obj_ws + obj_bytes);
    float* wf = (float*)((char*)xf + align_up((size_t)n * d * 4));
    float* zf = (float*)((char*)wf + align_up((size_t)d * k * 4));
    LASSO_HIP_TRY(launch_cvt_bf16(x_dev, ldx, xf, d, (int)n, (int)d, 1, st));
    LASSO_HIP_TRY(launch_cvt_bf16(w_dev, ldw, wf, k, (int)d, (int)k, 1, st));
    LASSO_HIP_TRY(launch_cvt_bf16(z_out_dev, ldz, zf, k, (int)n, (int)k, 1, st));
    xo = xf; wo = wf; zo = zf; ldxo = d; ldwo = k; ldzo = k;
  }
  if (int s2 = lasso_objective(xo, ldxo, wo, ldwo, zo, ldzo, n, d, k, LASSO_F32, alpha, loss_dev, nullptr, obj_ws,
                               obj_bytes, stream))
    return s2;
  LASSO_HIP_TRY(hipMemcpyAsync(objective_out, loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
  LASSO_HIP_TRY(hipStreamSynchronize(st));
  memcpy(g_err, saved, sizeof(saved));
  return status;
}

// Line-search solve on a ROW SHARD of the batch: see include/lasso_hip.h
int lasso_fista_solve_sharded(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z0_dev,
                              int64_t ldz0, void* z_out_dev, int64_t ldz, int64_t n, int64_t n_global, int64_t d,
                              int64_t k, int dtype, double alpha, double lr, int fast, int maxiter, double tol,
                              double eta_backtrack, lasso_allreduce_fn reduce, void* reduce_ctx, int32_t* iters_out,
                              float* last_delta_out, int32_t* trials_out, float* accepted_lr_out,
                              float* accepted_f_out, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32 && dtype != LASSO_BF16) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!reduce) return fail(LASSO_ERR_BAD_ARG, "reduce callback is null");
  if (n <= 0 || n_global < n || d <= 0 || k <= 0 || maxiter <= 0) return fail(LASSO_ERR_BAD_ARG, "bad shape");
  if (!x_dev || !w_dev || !z_out_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldx < d || ldw < k || ldz < k || (z0_dev && ldz0 < k)) return fail(LASSO_ERR_BAD_ARG, "leading dimension too small");
  if (!(lr > 0.0) || !(alpha >= 0.0)) return fail(LASSO_ERR_BAD_ARG, "need lr > 0 and alpha >= 0");
  if (!(eta_backtrack > 1.0)) return fail(LASSO_ERR_BAD_ARG, "eta must be > 1.");
  if (!fused_shape(d, k)) {                     // beyond the fused shapes: the unfused line search (fp32 tensors)
    if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "bf16 line search needs d<=%d, k<=%d", kFistaD, kFistaMaxK);
    if (iters_out) *iters_out = 0;
    if (last_delta_out) *last_delta_out = NAN;
    return solve_generic_backtracking((const float*)x_dev, ldx, (const float*)w_dev, ldw, (const float*)z0_dev, ldz0,
                                      (float*)z_out_dev, ldz, n, d, k, alpha, lr, fast, maxiter, tol, eta_backtrack,
                                      iters_out, last_delta_out, trials_out, accepted_lr_out, accepted_f_out,
                                      workspace_dev, workspace_bytes, (hipStream_t)stream, reduce, reduce_ctx, n_global);
  }
  if (n > (int64_t)INT32_MAX - kTileM) return fail(LASSO_ERR_UNSUPPORTED, "n too large");
  if (iters_out) *iters_out = 0;
  if (last_delta_out) *last_delta_out = NAN;
  return solve_backtracking(x_dev, ldx, w_dev, ldw, z0_dev, ldz0, z_out_dev, ldz, n, d, k, pad_k(k), dtype, alpha, lr,
                            fast, maxiter, tol, eta_backtrack, iters_out, last_delta_out, trials_out, accepted_lr_out,
                            accepted_f_out, workspace_dev, workspace_bytes, (hipStream_t)stream, reduce, reduce_ctx,
                            n_global);
}

// LASSO_SOLVE_ASYNC solve that returned LASSO_PENDING: enqueue the copy of what the persistent
// kernel leaves behind -- no synchronisation
int lasso_fista_solve_collect(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                              int32_t* out4_host, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32 || !fused_shape(d, k) || n <= 0 || maxiter <= 0 || !(tol > 0.0) || !out4_host)
    return fail(LASSO_ERR_BAD_ARG, "no pending solve of this shape");
  const int kp = pad_k_solve(n, d, k);
  Workspace ws = carve(workspace_dev, n, k, kp, maxiter, true);
  if (!workspace_dev || workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", ws.bytes);
  LASSO_HIP_TRY(hipMemcpyAsync(out4_host, ws.stop_out, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return LASSO_OK;
}

float* lasso_fista_solve_deltas(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                                void* workspace_dev, size_t workspace_bytes) {
  if (dtype != LASSO_F32 || !fused_shape(d, k) || n <= 0 || maxiter <= 0 || maxiter > kChunkMax || !(tol > 0.0) ||
      !workspace_dev)
    return nullptr;
  Workspace ws = carve(workspace_dev, n, k, pad_k_solve(n, d, k), maxiter, true);
  return workspace_bytes < ws.bytes ? nullptr : ws.delta;
}

static int solve_verdict_impl(int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                              const float* sums_dev, int32_t* status_mapped, void* workspace_dev, size_t workspace_bytes,
                              void* stream) {
  if (dtype != LASSO_F32 || !fused_shape(d, k) || n <= 0 || n_global < n || maxiter <= 0 || maxiter > kChunkMax ||
      !(tol > 0.0))
    return fail(LASSO_ERR_BAD_ARG, "no pending sharded solve of this shape");
  Workspace ws = carve(workspace_dev, n, k, pad_k_solve(n, d, k), maxiter, true);
  if (!workspace_dev || workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", ws.bytes);
  const float budget = (float)((double)n_global * (double)k * tol);   // ista.py:64 on the whole batch
  hipLaunchKernelGGL(chunk_verdict_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums_dev ? sums_dev : ws.delta,
                     maxiter, budget, ws.stop_out, status_mapped);
  LASSO_HIP_TRY(hipGetLastError());
  return LASSO_OK;
}

int lasso_fista_solve_verdict(int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                              const float* sums_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  return solve_verdict_impl(n, n_global, d, k, dtype, maxiter, tol, sums_dev, nullptr, workspace_dev, workspace_bytes, stream);
}

// The verdict launch an asynchronous LASSO_SOLVE_DEFER_VERDICT solve of this thread left out (it returned
// LASSO_PENDING_DEFERRED), on `stream` -- which the caller has ordered behind the solve's kernels (the EM loop: its side
// stream, behind a wave that polls a word the next launch of the solve's stream raises): the stop rule's launch is then
// off the step's dependent chain.  Same kernel, same sums, same words in `status_mapped` as the solve would have left.
int lasso_fista_solve_verdict_deferred(void* workspace_dev, int32_t* status_mapped, const int32_t* gate_word,
                                       int32_t gate_value, void* stream) {
  DeferredVerdict& dv = t_deferred_verdict;
  if (!status_mapped) return fail(LASSO_ERR_BAD_ARG, "status_mapped is NULL");
  if (!dv.armed || dv.workspace != workspace_dev)
    return fail(LASSO_ERR_BAD_ARG, "no deferred verdict for this workspace (LASSO_SOLVE_DEFER_VERDICT, same thread, one at a time)");
  dv.armed = false;
  hipLaunchKernelGGL(reduce_verdict_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dv.ps, dv.delta, dv.iters, dv.budget,
                     dv.out, status_mapped, gate_word, gate_value);
  LASSO_HIP_TRY(hipGetLastError());
  return LASSO_OK;
}

// the same verdict, its four words ALSO written to `status_mapped` (device-writable host memory): no collect call
int lasso_fista_solve_verdict_mapped(int64_t n, int64_t n_global, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                                     const float* sums_dev, int32_t* status_mapped, void* workspace_dev,
                                     size_t workspace_bytes, void* stream) {
  if (!status_mapped) return fail(LASSO_ERR_BAD_ARG, "status_mapped is NULL");
  return solve_verdict_impl(n, n_global, d, k, dtype, maxiter, tol, sums_dev, status_mapped, workspace_dev, workspace_bytes,
                            stream);
}

// the synchronous form: collect, wait, decode
int lasso_fista_solve_finish(int64_t n, int64_t d, int64_t k, int dtype, int maxiter, double tol,
                             int32_t* iters_out, float* last_delta_out, void* workspace_dev,
                             size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32 || !fused_shape(d, k) || n <= 0 || maxiter <= 0 || !(tol > 0.0))
    return fail(LASSO_ERR_BAD_ARG, "no pending solve of this shape");
  const int kp = pad_k_solve(n, d, k);
  Workspace ws = carve(workspace_dev, n, k, kp, maxiter, true);
  if (!workspace_dev || workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", ws.bytes);
  hipStream_t st = (hipStream_t)stream;
  int hout[4] = {0, 0, 0, 0};
  LASSO_HIP_TRY(hipMemcpyAsync(hout, ws.stop_out, 16, hipMemcpyDeviceToHost, st));
  LASSO_HIP_TRY(hipStreamSynchronize(st));
  if (hout[2]) {
    if (iters_out) *iters_out = 0;
    if (last_delta_out) *last_delta_out = NAN;
    snprintf(g_err, sizeof(g_err), "asynchronous solve must be repeated (handshake gave up, or the rule fired before the end of the chunk)");
    return LASSO_WARN_ABORTED;
  }
  float lastf;
  memcpy(&lastf, &hout[1], sizeof(float));
  if (iters_out) *iters_out = hout[0];
  if (last_delta_out) *last_delta_out = lastf;
  return LASSO_OK;
}

// ---------------------------------------------------------------------------
size_t lasso_lipschitz_workspace_bytes(int64_t d, int64_t k) {
  if (d <= 0 || k <= 0) return 0;
  return lipschitz_workspace_bytes(d, k);
}

int lasso_lipschitz(const void* w_dev, int64_t ldw, int64_t d, int64_t k, int dtype, double* l_out,
                    void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!w_dev || !workspace_dev || d <= 0 || k <= 0 || ldw < k)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (std::min(d, k) > 2048) return fail(LASSO_ERR_UNSUPPORTED, "min(d,k) > 2048");
  if (workspace_bytes < lipschitz_workspace_bytes(d, k))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lipschitz_workspace_bytes(d, k));
  hipStream_t st = (hipStream_t)stream;
  LASSO_HIP_TRY(launch_lipschitz((const float*)w_dev, ldw, d, k, workspace_dev, 20, st));
  if (l_out) {
    LASSO_HIP_TRY(hipMemcpyAsync(l_out, workspace_dev, sizeof(double), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
  }
  return LASSO_OK;
}

// ---------------------------------------------------------------------------
constexpr int kObjGenGrid = 1024;
// The tile kernel pads d to 256 columns of residual: at d <= 64 (8 x 8 patches) three quarters of its MFMAs multiply
// zeros, and the general GEMM + one reduction pass is the faster route (d=64, k=256, n=65536: 0.10 -> 0.07 ms).
static bool objective_unfused(int64_t d, int64_t k) { return !fused_shape(d, k) || d <= 64; }

size_t lasso_objective_workspace_bytes(int64_t n, int64_t d, int64_t k) {
  if (n < 0 || d <= 0 || k <= 0) return 0;
  if (objective_unfused(d, k))     // unfused: residual [n][d] + partial pairs
    return align_up((size_t)std::max<int64_t>(n, 1) * d * 4) + align_up((size_t)kObjGenGrid * 2 * 4) + 256;
  const int kp = pad_k(k);
  const int64_t ntiles = (n + kTileM - 1) / kTileM;
  return align_up((size_t)kFistaD * kp * 4) + align_up((size_t)kp * kFistaD * 4) +
         align_up((size_t)std::max<int64_t>(ntiles, 1) * 2 * 4) + 256;
}

static int objective_impl(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z_dev,
                          int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype, double alpha,
                          float* loss_dev, double* sums_dev, int max_workgroups, void* workspace_dev,
                          size_t workspace_bytes, void* stream) {
  if (int s = check_common(n, d, k, dtype, /*allow_large=*/true)) return s;
  if (!x_dev || !w_dev || !z_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldx < d || ldw < k || ldz < k) return fail(LASSO_ERR_BAD_ARG, "leading dimension too small");
  if (workspace_bytes < lasso_objective_workspace_bytes(n, d, k))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lasso_objective_workspace_bytes(n, d, k));
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace_dev;
  if (objective_unfused(d, k)) {
    if (d > INT32_MAX / 2 || k > INT32_MAX / 2) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
    float* R = (float*)base;
    float* partials = (float*)(base + align_up((size_t)std::max<int64_t>(n, 1) * d * 4));
    double* sums = (double*)((char*)partials + align_up((size_t)kObjGenGrid * 2 * 4));
    if (n > 0)
      LASSO_HIP_TRY(launch_objective_generic((const float*)x_dev, ldx, (const float*)w_dev, ldw,
                                             (const float*)z_dev, ldz, (int)n, (int)d, (int)k, R, partials,
                                             kObjGenGrid, alpha, (double)n, sums_dev ? sums_dev : sums,
                                             loss_dev, st));
    return LASSO_OK;
  }
  const int kp = pad_k(k);
  const int ntiles = (int)((n + kTileM - 1) / kTileM);
  float* wp = (float*)base;
  float* wtp = (float*)(base + align_up((size_t)kFistaD * kp * 4));
  float* partials = (float*)((char*)wtp + align_up((size_t)kp * kFistaD * 4));
  double* sums = (double*)((char*)partials + align_up((size_t)std::max(ntiles, 1) * 2 * 4));
  hipLaunchKernelGGL(pack_w_kernel, dim3(kp / 32, kFistaD / 32), dim3(32, 8), 0, st,
                     (const float*)w_dev, ldw, (int)d, (int)k, kp, wp, wtp);
  LASSO_HIP_TRY(hipGetLastError());
  ObjectiveParams p;
  p.X = (const float*)x_dev; p.ldx = ldx; p.Wp = wp;
  p.Z = (const float*)z_dev; p.ldz = ldz; p.partials = partials;
  p.n = (int)n; p.d = (int)d; p.k = (int)k; p.ntiles = ntiles;
  const int cus = device_cus();
  if (cus <= 0) return fail(LASSO_ERR_HIP, "no HIP device");
  if (n > 0) {
    int grid = std::min(ntiles, cus);
    if (max_workgroups > 0) grid = std::max(1, std::min(grid, max_workgroups));
    LASSO_HIP_TRY(launch_objective(p, kp, grid, alpha, (double)n,
                                   sums_dev ? sums_dev : sums, loss_dev, st));
  }
  return LASSO_OK;
}

int lasso_objective(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z_dev,
                    int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype, double alpha,
                    float* loss_dev, double* sums_dev, void* workspace_dev, size_t workspace_bytes,
                    void* stream) {
  return objective_impl(x_dev, ldx, w_dev, ldw, z_dev, ldz, n, d, k, dtype, alpha, loss_dev, sums_dev, 0, workspace_dev,
                        workspace_bytes, stream);
}

// the same on at most `max_workgroups` workgroups of the fused kernel (0: as many as there are CUs): for a caller that
// runs the objective BESIDE latency-bound work on another stream (the EM loop's atom sweep) -- a quarter of the chip
// takes four times as long and disturbs its neighbour's memory round trips less (round 6: 8 us less on the sweep)
int lasso_objective_throttled(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z_dev,
                              int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype, double alpha,
                              float* loss_dev, double* sums_dev, int max_workgroups, void* workspace_dev,
                              size_t workspace_bytes, void* stream) {
  if (max_workgroups < 0) return fail(LASSO_ERR_BAD_ARG, "max_workgroups < 0");
  return objective_impl(x_dev, ldx, w_dev, ldw, z_dev, ldz, n, d, k, dtype, alpha, loss_dev, sums_dev, max_workgroups,
                        workspace_dev, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------
static int gram_max_splits(int64_t d, int64_t k) {
  const int64_t one = k * std::max(k, d) * 4;
  return (int)std::min<int64_t>(128, std::max<int64_t>(16, ((int64_t)64 << 20) / std::max<int64_t>(one, 1)));
}

size_t lasso_gram_workspace_bytes(int64_t n, int64_t d, int64_t k) {
  if (n < 0 || d <= 0 || k <= 0) return 0;
  // sample splits of the larger product: 16, more for small dictionaries (few output blocks: the splits are what
  // fills the chip -- k = 256, d = 64 ran on 48 workgroups with 16) up to 64 MB; the fused [A | B] kernel sizes its own
  return std::max((size_t)gram_max_splits(d, k) * (size_t)k * (size_t)std::max(k, d) * 4, gram_ab_scratch_bytes(d, k)) + 256;
}

static int gram_accumulate_impl(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n,
                                int64_t d, int64_t k, int dtype, float* a_dev, float* b_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream, int32_t* started_word,
                                int32_t started_value);
int lasso_gram_accumulate(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n,
                          int64_t d, int64_t k, int dtype, float* a_dev, float* b_dev,
                          void* workspace_dev, size_t workspace_bytes, void* stream) {
  return gram_accumulate_impl(z_dev, ldz, x_dev, ldx, n, d, k, dtype, a_dev, b_dev, workspace_dev, workspace_bytes, stream,
                              nullptr, 0);
}
// the same products; the first launch raises *started_word = started_value when it STARTS (device memory): "everything
// enqueued on this stream before the call has completed", for a wave of another stream that polls the word
// (lasso_stream_wait_word) -- a start signal without an event record on this stream (~5 us between two kernels)
int lasso_gram_accumulate_signal(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n,
                                 int64_t d, int64_t k, int dtype, float* a_dev, float* b_dev,
                                 void* workspace_dev, size_t workspace_bytes, int32_t* started_word, int32_t started_value,
                                 void* stream) {
  if (!started_word) return fail(LASSO_ERR_BAD_ARG, "started_word is NULL");
  return gram_accumulate_impl(z_dev, ldz, x_dev, ldx, n, d, k, dtype, a_dev, b_dev, workspace_dev, workspace_bytes, stream,
                              started_word, started_value);
}
static int gram_accumulate_impl(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n,
                                int64_t d, int64_t k, int dtype, float* a_dev, float* b_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream, int32_t* started_word,
                                int32_t started_value) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!z_dev || !x_dev || !a_dev || !b_dev || n < 0 || d <= 0 || k <= 0 || ldz < k || ldx < d)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (n > INT32_MAX) return fail(LASSO_ERR_UNSUPPORTED, "n too large");
  hipStream_t st = (hipStream_t)stream;
  const float* Z = (const float*)z_dev;
  // optional scratch for the split-n partial products (NULL = single pass)
  float* scratch = (workspace_dev && workspace_bytes >= lasso_gram_workspace_bytes(n, d, k))
                       ? (float*)workspace_dev : nullptr;
  const int cus = device_cus();
  if (scratch) {
    hipError_t e = hipSuccess;
    if (launch_gram_ab(Z, ldz, (int)k, (const float*)x_dev, ldx, (int)d, (int)n, a_dev, b_dev, scratch,
                       workspace_bytes - 256, cus, st, &e, started_word, started_value)) {
      LASSO_HIP_TRY(e);
      return LASSO_OK;
    }
  }
  if (scratch && !getenv("LASSO_GRAM_TWO_LAUNCHES")) {      // small dictionaries: one product launch, one fold (A/B: the env switch)
    hipError_t e = hipSuccess;
    if (launch_gram_ab128(Z, ldz, (int)k, (const float*)x_dev, ldx, (int)d, (int)n, a_dev, b_dev, scratch,
                          workspace_bytes - 256, cus, gram_max_splits(d, k), st, &e, started_word, started_value)) {
      LASSO_HIP_TRY(e);
      return LASSO_OK;
    }
  }
  // (the other product kernels do not carry the signal: a launch of its own in front of them)
  if (started_word) LASSO_HIP_TRY(launch_set_flag(started_word, started_value, st));
  const int smax = scratch ? gram_max_splits(d, k) : 1;
  const int sa = gram_splits((int)k, (int)k, (int)n, 1, cus, smax), sb = gram_splits((int)k, (int)d, (int)n, 0, cus, smax);
  LASSO_HIP_TRY(launch_gram_tn(Z, ldz, (int)k, Z, ldz, (int)k, (int)n, a_dev, k, 1, scratch, sa, st));
  LASSO_HIP_TRY(launch_gram_tn(Z, ldz, (int)k, (const float*)x_dev, ldx, (int)d, (int)n, b_dev, d, 0,
                               scratch, sb, st));
  return LASSO_OK;
}

size_t lasso_dict_sweep_workspace_bytes(int64_t d, int64_t k) {
  if (d <= 0 || k <= 0 || d > kSweepMaxD || k > kSweepMaxK) return 0;
  const size_t dp = (size_t)(d + 255) / 256 * 256;
  return align_up((size_t)k * dp * 4) * 2 + align_up((size_t)kSweepBlock * dp * 4) + 256 +
         (dp == 256 ? align_up(sweep_persist_extra_bytes((int)k)) : 0);
}

// d_out_dev (pitch ldo): where the new dictionary goes -- d_dev itself (in place), or another buffer: d_dev is then only
// read (lasso_dict_sweep_async_to)
static int dict_sweep_impl(const float* a_dev, const float* b_dev, void* d_dev, int64_t ldd, int64_t d,
                           int64_t k, int dtype, double eps, int positive, const float* pool_dev,
                           int64_t pool_rows, int64_t pool_ld, uint64_t seed, int32_t* degenerate_dev,
                           int32_t* ndeg_out, int32_t* ndeg_mapped, void* workspace_dev, size_t workspace_bytes,
                           void* stream, void* d_out_dev = nullptr, int64_t ldo = 0, int32_t* started_word = nullptr,
                           int32_t started_value = 0) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!a_dev || !b_dev || !d_dev || !degenerate_dev || !workspace_dev || d <= 0 || k <= 0 || ldd < k)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!d_out_dev) { d_out_dev = d_dev; ldo = ldd; }
  if (ldo < k) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (d_out_dev != d_dev) {         // two buffers: they must not overlap (the old atoms are read while the new ones land)
    const char* a0 = (const char*)d_dev, * a1 = a0 + ((size_t)(d - 1) * ldd + k) * 4;
    const char* b0 = (const char*)d_out_dev, * b1 = b0 + ((size_t)(d - 1) * ldo + k) * 4;
    if (a0 < b1 && b0 < a1) return fail(LASSO_ERR_BAD_ARG, "d_out_dev overlaps d_dev");
  }
  if (d > kSweepMaxD || k > kSweepMaxK)
    return fail(LASSO_ERR_UNSUPPORTED, "atom sweep: d=%lld k=%lld (d <= %d, k <= %d)", (long long)d,
                (long long)k, kSweepMaxD, kSweepMaxK);
  if (pool_dev && (pool_rows <= 0 || pool_ld < d)) return fail(LASSO_ERR_BAD_ARG, "bad pool");
  if (workspace_bytes < lasso_dict_sweep_workspace_bytes(d, k))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lasso_dict_sweep_workspace_bytes(d, k));
  hipStream_t st = (hipStream_t)stream;
  const int dp = (int)((d + 255) / 256 * 256);     // row stride of U / Dt / dD: whole 256-feature panels
  char* base = (char*)workspace_dev;
  float* U = (float*)base;
  float* Dt = (float*)(base + align_up((size_t)k * dp * 4));
  float* dD = (float*)((char*)Dt + align_up((size_t)k * dp * 4));
  int* ndeg = (int*)((char*)dD + align_up((size_t)kSweepBlock * dp * 4));
  float* D = (float*)d_dev;
  float* Dnew = (float*)d_out_dev;
  // (ndeg is written by degenerate_fixup_kernel on every path: no clearing launch in front)
  // U[j][dd] = B[j][dd] - sum_i A[j][i] D[dd][i]          (k x d, zero padded to dp columns)
  // The single-launch sweep (dp == 256) reads its old atoms from D itself, never reads the padding of U, and its
  // last launch writes the new dictionary: no transposed copies, no clearing of U (four launches off the chain).
  const bool direct = dp == 256 && k % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)D & 15) == 0 && ldo % 4 == 0 &&
                      ((uintptr_t)Dnew & 15) == 0;
  if (dp != d && !direct) LASSO_HIP_TRY(hipMemsetAsync(U, 0, (size_t)k * dp * 4, st));   // (d == dp: the product writes every column)
  void* const extra = dp == 256 ? (void*)((char*)ndeg + 256) : nullptr;
  // (the product's first workgroup also clears the single-launch sweep's flag words: no fill launch between them)
  int* const flag_words = extra ? sweep_persist_flags(extra, (int)k) : nullptr;
  LASSO_HIP_TRY(launch_gemm_nt_sub(a_dev, k, D, ldd, b_dev, d, U, dp, (int)k, (int)d, (int)k, st, 0, flag_words,
                                   flag_words ? 1024 : 0));
  // Dt[j][dd] = D[dd][j]  (zero padded to dp features)
  if (!direct) LASSO_HIP_TRY(launch_transpose_pad(D, ldd, (int)d, (int)k, Dt, dp, (int)k, dp, st));
  // "the sweep starts now": a word another stream's wave polls (work that should run BESIDE the sweep, which leaves most
  // of the chip idle -- the EM loop's objective on large batches); one 4-us launch on this stream
  if (started_word) LASSO_HIP_TRY(launch_set_flag(started_word, started_value, st));
  SweepParams p;
  p.flags_cleared = flag_words != nullptr;
  p.Dsrc = direct ? D : nullptr; p.ldd = ldd;
  p.Dout = direct ? Dnew : nullptr; p.ldo = ldo;
  p.A = a_dev; p.lda = k; p.U = U; p.ldu = dp; p.Dt = Dt; p.dD = dD; p.dp = dp;
  p.pool = pool_dev; p.pool_rows = (int)pool_rows; p.pool_ld = pool_ld; p.seed = seed;
  p.degenerate = degenerate_dev; p.ndeg_in_out = ndeg; p.ndeg_mirror = ndeg_mapped;
  p.wait_word = nullptr; p.wait_value = 0;
  p.k = (int)k; p.d = (int)d; p.eps = (float)eps; p.positive = positive;
  float* dt_new = Dt;
  LASSO_HIP_TRY(launch_dict_sweep(p, st, extra, &dt_new));
  // D[dd][j] = Dt[j][dd]
  if (!direct) LASSO_HIP_TRY(launch_transpose_pad(dt_new, dp, (int)k, (int)d, Dnew, ldo, (int)d, (int)k, st));
  if (ndeg_out) {
    LASSO_HIP_TRY(hipMemcpyAsync(ndeg_out, ndeg, sizeof(int), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
  }
  return LASSO_OK;
}

int lasso_dict_sweep(const float* a_dev, const float* b_dev, void* d_dev, int64_t ldd, int64_t d,
                     int64_t k, int dtype, double eps, int positive, const float* pool_dev,
                     int64_t pool_rows, int64_t pool_ld, uint64_t seed, int32_t* degenerate_dev,
                     int32_t* ndeg_out, void* workspace_dev, size_t workspace_bytes, void* stream) {
  return dict_sweep_impl(a_dev, b_dev, d_dev, ldd, d, k, dtype, eps, positive, pool_dev, pool_rows, pool_ld, seed,
                         degenerate_dev, ndeg_out, nullptr, workspace_dev, workspace_bytes, stream);
}

// the same sweep without a host wait: the count of degenerate atoms goes to `ndeg_mapped`, a device-writable HOST word
// (pinned, mapped), written by the sweep's last kernel -- valid once an event recorded behind this call has completed
int lasso_dict_sweep_async(const float* a_dev, const float* b_dev, void* d_dev, int64_t ldd, int64_t d,
                           int64_t k, int dtype, double eps, int positive, const float* pool_dev,
                           int64_t pool_rows, int64_t pool_ld, uint64_t seed, int32_t* degenerate_dev,
                           int32_t* ndeg_mapped, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!ndeg_mapped) return fail(LASSO_ERR_BAD_ARG, "ndeg_mapped is NULL");
  return dict_sweep_impl(a_dev, b_dev, d_dev, ldd, d, k, dtype, eps, positive, pool_dev, pool_rows, pool_ld, seed,
                         degenerate_dev, nullptr, ndeg_mapped, workspace_dev, workspace_bytes, stream);
}

// lasso_dict_sweep_async with the new dictionary written to ANOTHER buffer: d_dev is only read, so the call may be
// enqueued before the host knows whether the step it belongs to stands (an EM loop's E-step verdict, the previous sweep's
// count of degenerate atoms: a step that has to be repeated simply keeps d_dev), and work on another stream may go on
// reading the old dictionary beside it (the objective of dict_learning.py:39) -- DESIGN.md 3.3h
int lasso_dict_sweep_async_to(const float* a_dev, const float* b_dev, const void* d_dev, int64_t ldd, void* d_out_dev,
                              int64_t ldo, int64_t d, int64_t k, int dtype, double eps, int positive,
                              const float* pool_dev, int64_t pool_rows, int64_t pool_ld, uint64_t seed,
                              int32_t* degenerate_dev, int32_t* ndeg_mapped, int32_t* started_word,
                              int32_t started_value, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!ndeg_mapped) return fail(LASSO_ERR_BAD_ARG, "ndeg_mapped is NULL");
  if (!d_out_dev || d_out_dev == d_dev) return fail(LASSO_ERR_BAD_ARG, "d_out_dev must be another buffer");
  return dict_sweep_impl(a_dev, b_dev, const_cast<void*>(d_dev), ldd, d, k, dtype, eps, positive, pool_dev, pool_rows, pool_ld,
                         seed, degenerate_dev, nullptr, ndeg_mapped, workspace_dev, workspace_bytes, stream, d_out_dev, ldo,
                         started_word, started_value);
}

// ---- pipelined constrained M-step (ABI 7) -------------------------------------------------------
// workspace = [the sweep's workspace (lasso_dict_sweep_workspace_bytes) | the Gram partial sums of every block row]
namespace {
struct PipeWs { SweepParams p; void* extra; float* gram; MstepPipePlan plan; size_t bytes; };
// the SweepParams of every call of one pipelined M-step (the same carve of the same workspace each time)
bool pipe_carve(int64_t n, int64_t d, int64_t k, void* workspace_dev, PipeWs* w) {
  const int cus = device_cus();
  w->plan = mstep_pipe_plan(n, d, k, cus);
  if (w->plan.nstages == 0) return false;
  const size_t sweep = align_up(lasso_dict_sweep_workspace_bytes(d, k));
  w->bytes = sweep + w->plan.scratch_bytes + 256;
  const int dp = 256;
  char* base = (char*)workspace_dev;
  float* U = (float*)base;
  float* Dt = (float*)(base + align_up((size_t)k * dp * 4));
  float* dD = (float*)((char*)Dt + align_up((size_t)k * dp * 4));
  int* ndeg = (int*)((char*)dD + align_up((size_t)kSweepBlock * dp * 4));
  w->extra = (void*)((char*)ndeg + 256);
  w->gram = (float*)(base + sweep);
  SweepParams& p = w->p;
  p.flags_cleared = 1;
  p.U = U; p.ldu = dp; p.Dt = Dt; p.dD = dD; p.dp = dp;
  p.pool = nullptr; p.pool_rows = 0; p.pool_ld = 0; p.seed = 0;
  p.ndeg_in_out = ndeg; p.ndeg_mirror = nullptr;
  p.wait_word = nullptr; p.wait_value = 0;
  p.k = (int)k; p.d = (int)d;
  return true;
}
}  // namespace

int lasso_mstep_pipe_stages(int64_t n, int64_t d, int64_t k) {
  if (n <= 0 || d <= 0 || k <= 0) return 0;
  return mstep_pipe_plan(n, d, k, device_cus()).nstages;
}

int lasso_mstep_pipe_stage_rows(int64_t n, int64_t d, int64_t k, int stage, int64_t* row_lo, int64_t* row_hi) {
  if (n <= 0 || d <= 0 || k <= 0 || !row_lo || !row_hi) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  const MstepPipePlan pl = mstep_pipe_plan(n, d, k, device_cus());
  if (stage < 0 || stage >= pl.nstages) return fail(LASSO_ERR_BAD_ARG, "stage %d of %d", stage, pl.nstages);
  *row_lo = 256 * (int64_t)pl.lo[stage];
  *row_hi = 256 * (int64_t)pl.hi[stage];
  return LASSO_OK;
}

size_t lasso_mstep_pipe_workspace_bytes(int64_t n, int64_t d, int64_t k) {
  PipeWs w;
  if (n <= 0 || d <= 0 || k <= 0 || !pipe_carve(n, d, k, nullptr, &w)) return 0;
  return w.bytes;
}

int lasso_mstep_pipe_gram(const void* z_dev, int64_t ldz, const void* x_dev, int64_t ldx, int64_t n, int64_t d,
                          int64_t k, int dtype, float* ab_dev, int64_t ldab, int stage, void* workspace_dev,
                          size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  PipeWs w;
  if (!ab_dev || !workspace_dev || n < 0 || (n > 0 && (!z_dev || !x_dev || ldz < k || ldx < d)) || ldab < k + d)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(std::max<int64_t>(n, 1), d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for n=%lld d=%lld k=%lld",
                                                           (long long)n, (long long)d, (long long)k);
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  if (stage < 0 || stage >= w.plan.nstages) return fail(LASSO_ERR_BAD_ARG, "stage %d of %d", stage, w.plan.nstages);
  if ((ldab & 3) || ((uintptr_t)ab_dev & 15)) return fail(LASSO_ERR_BAD_ARG, "[A | B] must be 16-byte aligned, pitch a multiple of 4");
  // the head's launch also clears the sweep's flag words: it sits in front of every launch that sets or reads them
  int* const flags = stage == 0 ? sweep_persist_flags(w.extra, (int)k) : nullptr;
  if (n == 0) {
    // a rank without rows (it still runs the identical sweep on the all-reduced [A | B]): zero contributions, and the
    // flag words cleared all the same -- stale "rows complete" words would let the gated sweep run ahead of its rows
    const int lo = w.plan.lo[stage], hi = w.plan.hi[stage];
    LASSO_HIP_TRY(hipMemset2DAsync(ab_dev + (int64_t)256 * lo * ldab, ldab * 4, 0, (size_t)(k + d) * 4, (size_t)256 * (hi - lo),
                                   (hipStream_t)stream));
    if (flags) LASSO_HIP_TRY(hipMemsetAsync(flags, 0, 4096, (hipStream_t)stream));
    return LASSO_OK;
  }
  LASSO_HIP_TRY(launch_gram_rows((const float*)z_dev, ldz, (int)k, (const float*)x_dev, ldx, (int)d, (int)n, ab_dev, ldab,
                                 stage, w.plan, w.gram, flags, flags ? 1024 : 0, (hipStream_t)stream));
  return LASSO_OK;
}

int lasso_stream_wait_word(const int32_t* word, int32_t value, int host_memory, void* stream) {
  if (!word) return fail(LASSO_ERR_BAD_ARG, "word is NULL");
  LASSO_HIP_TRY(launch_wait_word(word, value, host_memory, (hipStream_t)stream));
  return LASSO_OK;
}

int lasso_mstep_pipe_wait(int64_t n, int64_t d, int64_t k, int seq, void* workspace_dev, size_t workspace_bytes,
                          void* stream) {
  PipeWs w;
  if (!workspace_dev) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(n, d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for this shape");
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  LASSO_HIP_TRY(launch_wait_word(sweep_pipe_words(w.extra, (int)k) + 1, seq, 0, (hipStream_t)stream));
  return LASSO_OK;
}

// the word lasso_mstep_pipe_wait polls (device memory inside the workspace; NULL: no pipelined M-step for the shape) --
// for launches behind the wait that must re-check it themselves (lasso_fista_solve_verdict_deferred's gate)
const int32_t* lasso_mstep_pipe_head_word(int64_t n, int64_t d, int64_t k, void* workspace_dev, size_t workspace_bytes) {
  PipeWs w;
  if (!workspace_dev || !pipe_carve(n, d, k, workspace_dev, &w) || workspace_bytes < w.bytes) return nullptr;
  return sweep_pipe_words(w.extra, (int)k) + 1;
}

int lasso_mstep_pipe_rows(const float* ab_dev, int64_t ldab, const void* d_dev, int64_t ldd, int64_t n, int64_t d,
                          int64_t k, int dtype, int stage, int seq, void* workspace_dev, size_t workspace_bytes,
                          void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  PipeWs w;
  if (!ab_dev || !d_dev || !workspace_dev || ldd < k || ldab < k + d) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(n, d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for this shape");
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  if (stage < 0 || stage >= w.plan.nstages) return fail(LASSO_ERR_BAD_ARG, "stage %d of %d", stage, w.plan.nstages);
  hipStream_t st = (hipStream_t)stream;
  const int lo = w.plan.lo[stage], hi = w.plan.hi[stage];
  const int64_t r0 = 256 * (int64_t)lo;
  // U[j] = B[j] - sum_i A[j][i] D[:, i]   for the rows j of the block (dict_learning.py:82 in Gram form)
  // (uprod_rows_kernel: bitwise launch_gemm_nt_sub's product; its last workgroup raises the block row's flag)
  if ((ldab & 3) || (ldd & 3) || ((uintptr_t)ab_dev & 15) || ((uintptr_t)d_dev & 15))
    return fail(LASSO_ERR_BAD_ARG, "[A | B] and the dictionary must be 16-byte aligned, pitches multiples of 4");
  // (the head: `seq` into the word lasso_mstep_pipe_wait watches -- the head of the chain is through; the other
  // stages: the flags of their block rows for the running sweep)
  int* const words = sweep_pipe_words(w.extra, (int)k);
  int* const flag = stage > 0 ? sweep_persist_flags(w.extra, (int)k) + kSweepRowFlag + lo : words + 1;
  LASSO_HIP_TRY(launch_uprod_rows(ab_dev + r0 * ldab, ldab, (const float*)d_dev, ldd, ab_dev + r0 * ldab + k, ldab,
                                  w.p.U + r0 * 256, 256, 256 * (hi - lo), (int)k, words + 8 + stage, flag,
                                  stage > 0 ? hi - lo : 1, stage > 0 ? 1 : seq, st));
  return LASSO_OK;
}

int lasso_mstep_pipe_sweep(const float* ab_dev, int64_t ldab, const void* d_dev, int64_t ldd, int64_t n, int64_t d,
                           int64_t k, int dtype, double eps, int positive, int32_t* degenerate_dev, void* workspace_dev,
                           size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  PipeWs w;
  if (!ab_dev || !d_dev || !degenerate_dev || !workspace_dev || ldd < k || ldab < k + d)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(n, d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for this shape");
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  if ((k & 3) || (ldd & 3) || ((uintptr_t)d_dev & 15)) return fail(LASSO_ERR_BAD_ARG, "the dictionary must be 16-byte aligned, pitch a multiple of 4");
  SweepParams& p = w.p;
  p.A = ab_dev; p.lda = ldab;
  p.Dsrc = (const float*)d_dev; p.ldd = ldd;
  p.Dout = (float*)const_cast<void*>(d_dev); p.ldo = ldd;      // (only launch_sweep_fixup writes it: lasso_mstep_pipe_finish)
  p.degenerate = degenerate_dev; p.eps = (float)eps; p.positive = positive;
  float* dt_new = nullptr;
  LASSO_HIP_TRY(launch_sweep_gated(p, w.extra, 256 / kSweepBlock, w.plan.hi[0], &dt_new, (hipStream_t)stream));
  return LASSO_OK;
}

int lasso_mstep_pipe_signal(int64_t n, int64_t d, int64_t k, int seq, void* workspace_dev, size_t workspace_bytes,
                            void* stream) {
  PipeWs w;
  if (!workspace_dev) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(n, d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for this shape");
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  LASSO_HIP_TRY(launch_set_flag(sweep_pipe_words(w.extra, (int)k) + 2, seq, (hipStream_t)stream));
  return LASSO_OK;
}

int lasso_mstep_pipe_finish(void* d_dev, int64_t ldd, int64_t n, int64_t d, int64_t k, int dtype, double eps, int positive,
                            int32_t* degenerate_dev, int32_t* ndeg_mapped, int wait_seq, void* workspace_dev,
                            size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  PipeWs w;
  if (!d_dev || !degenerate_dev || !workspace_dev || ldd < k) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (!pipe_carve(n, d, k, workspace_dev, &w)) return fail(LASSO_ERR_UNSUPPORTED, "no pipelined M-step for this shape");
  if (workspace_bytes < w.bytes) return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", w.bytes);
  SweepParams& p = w.p;
  p.A = nullptr; p.lda = 0;
  p.Dsrc = (const float*)d_dev; p.ldd = ldd;
  p.Dout = (float*)d_dev; p.ldo = ldd;
  p.degenerate = degenerate_dev; p.eps = (float)eps; p.positive = positive;
  p.ndeg_mirror = ndeg_mapped;
  if (wait_seq != 0) { p.wait_word = sweep_pipe_words(w.extra, (int)k) + 2; p.wait_value = wait_seq; }
  p.Dt = (float*)w.extra;                      // the single-launch sweep's new atoms (SweepPersist::DtN)
  LASSO_HIP_TRY(launch_sweep_fixup(p, (hipStream_t)stream));
  return LASSO_OK;
}

// Device address of the sweep's count of degenerate atoms inside `workspace_dev` (valid once the sweep enqueued
// with this workspace has run): a caller that does not want lasso_dict_sweep's host wait (ndeg_out = NULL) copies
// these 4 bytes at its own synchronisation instead of reducing the k flags.
int32_t* lasso_dict_sweep_count(int64_t d, int64_t k, void* workspace_dev, size_t workspace_bytes) {
  if (!workspace_dev || d <= 0 || k <= 0 || d > kSweepMaxD || k > kSweepMaxK ||
      workspace_bytes < lasso_dict_sweep_workspace_bytes(d, k))
    return nullptr;
  const int dp = (int)((d + 255) / 256 * 256);
  char* base = (char*)workspace_dev;
  return (int32_t*)(base + 2 * align_up((size_t)k * dp * 4) + align_up((size_t)kSweepBlock * dp * 4));
}

// ---- init='transpose': z0 = x W, sparse_encode.py:24-25 --------------------------------------
size_t lasso_init_transpose_workspace_bytes(int64_t d, int64_t k) {
  if (d <= 0 || k <= 0) return 0;
  return (size_t)k * (size_t)d * sizeof(float) + 256;
}

int lasso_init_transpose(int64_t n, int64_t d, int64_t k, int dtype, const void* x_dev, int64_t ldx,
                         const void* w_dev, int64_t ldw, void* z0_dev, int64_t ldz, void* workspace_dev,
                         size_t workspace_bytes, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (n < 0 || d <= 0 || k <= 0 || n > INT32_MAX || d > INT32_MAX || k > INT32_MAX)
    return fail(LASSO_ERR_BAD_ARG, "bad shape");
  if (!w_dev || !workspace_dev || (n > 0 && (!x_dev || !z0_dev))) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldx < d || ldw < k || ldz < k) return fail(LASSO_ERR_BAD_ARG, "leading dimension too small");
  if (workspace_bytes < lasso_init_transpose_workspace_bytes(d, k))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lasso_init_transpose_workspace_bytes(d, k));
  if (n == 0) return LASSO_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* const Wt = (float*)workspace_dev;                  // [k][d]: z0 = x Wt^T on the NT GEMM
  LASSO_HIP_TRY(launch_transpose_pad((const float*)w_dev, ldw, (int)d, (int)k, Wt, d, (int)k, (int)d, st));
  LASSO_HIP_TRY(launch_gemm_nt_sub((const float*)x_dev, ldx, Wt, d, nullptr, 0, (float*)z0_dev, ldz, (int)n, (int)k,
                                   (int)d, st, /*add=*/1));
  return LASSO_OK;
}

// ---- unconstrained M-step: update_dict_ridge, dict_learning.py:106-123 ---------------------
size_t lasso_ridge_workspace_bytes(int64_t d, int64_t k) {
  if (d <= 0 || k <= 0 || k > 4096) return 0;
  return ridge_workspace_bytes(d, k) + 256;
}

int lasso_ridge_solve(const float* a_dev, const float* b_dev, void* v_dev, int64_t ldv, int64_t d, int64_t k,
                      int dtype, double lambda_n, int32_t* info_out, void* workspace_dev, size_t workspace_bytes,
                      void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!a_dev || !b_dev || !v_dev || !workspace_dev || d <= 0 || k <= 0 || ldv < k)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (k > 4096) return fail(LASSO_ERR_UNSUPPORTED, "ridge solve: k=%lld > 4096", (long long)k);
  if (workspace_bytes < lasso_ridge_workspace_bytes(d, k))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lasso_ridge_workspace_bytes(d, k));
  hipStream_t st = (hipStream_t)stream;
  int* const info_dev = (int*)((char*)workspace_dev + ridge_workspace_bytes(d, k));
  LASSO_HIP_TRY(launch_ridge_solve(a_dev, b_dev, (float*)v_dev, ldv, (int)d, (int)k, (float)lambda_n, workspace_dev,
                                   info_dev, st));
  if (info_out) {
    LASSO_HIP_TRY(hipMemcpyAsync(info_out, info_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
    if (*info_out != 0)
      return fail(LASSO_ERR_BAD_ARG, "Z^T Z + lambd*n*I is not positive definite (pivot %d)", *info_out);
  }
  return LASSO_OK;
}

int lasso_dict_fill_degenerate(void* d_dev, int64_t ldd, int64_t d, int64_t k, int dtype,
                               const int32_t* degenerate_dev, const float* pool_dev, int64_t pool_rows,
                               int64_t pool_ld, int positive, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!d_dev || !degenerate_dev || !pool_dev || d <= 0 || k <= 0 || ldd < k || pool_rows <= 0 || pool_ld < d)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  LASSO_HIP_TRY(launch_fill_degenerate((float*)d_dev, ldd, (int)d, (int)k, degenerate_dev, pool_dev, (int)pool_rows,
                                       pool_ld, positive, (hipStream_t)stream));
  return LASSO_OK;
}

int lasso_zero_columns(void* z_dev, int64_t ldz, int64_t n, int64_t k, int dtype,
                       const int32_t* degenerate_dev, void* stream) {
  if (dtype != LASSO_F32) return fail(LASSO_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (!z_dev || !degenerate_dev || n < 0 || k <= 0 || ldz < k) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  LASSO_HIP_TRY(launch_zero_columns((float*)z_dev, ldz, (int)n, (int)k, degenerate_dev, (hipStream_t)stream));
  return LASSO_OK;
}

// ---- greedy coordinate descent: coordinate_descent.py:5-54 -------------------------
size_t lasso_cd_workspace_bytes(int64_t n, int64_t d, int64_t k, int dtype) {
  (void)dtype;
  const int kp = pad_k_cd(k);
  if (kp < 0 || n < 0 || d <= 0) return 0;
  return carve_cd(nullptr, n, d, kp).bytes;
}

int lasso_cd_prepare(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* z0_dev,
                     int64_t ldz0, int64_t n, int64_t d, int64_t k, int dtype, void* workspace_dev,
                     size_t workspace_bytes, void* stream) {
  if (int s = check_cd(n, d, k, dtype)) return s;
  if (!w_dev || (n > 0 && !x_dev) || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldx < d || ldw < k || (z0_dev && ldz0 < k)) return fail(LASSO_ERR_BAD_ARG, "leading dimension too small");
  const int kp = pad_k_cd(k);
  CdWorkspace ws = carve_cd(workspace_dev, n, d, kp);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Wt [kp][d] (zero rows past k);  S = -Wt Wt^T (:22), +1 on the first k diagonal entries
  // (:23);  b = x Wt^T (:19).  Padded columns of b and S are exact zeros.
  LASSO_HIP_TRY(launch_transpose_pad((const float*)w_dev, ldw, (int)d, (int)k, ws.Wt, d, kp, (int)d, st));
  LASSO_HIP_TRY(launch_gemm_nt_sub(ws.Wt, d, ws.Wt, d, nullptr, 0, ws.S, kp, kp, kp, (int)d, st));
  LASSO_HIP_TRY(launch_cd_init((const float*)z0_dev, ldz0, ws.Zt, kp, (int)n, (int)k, ws.active,
                               ws.row_steps, ws.S, st));
  if (n > 0)
    LASSO_HIP_TRY(launch_gemm_nt_sub((const float*)x_dev, ldx, ws.Wt, d, nullptr, 0, ws.B, kp, (int)n, kp,
                                     (int)d, st, /*add=*/1));
  return LASSO_OK;
}

int lasso_cd_run(int64_t n, int64_t d, int64_t k, double alpha, double tol_abs, int iters,
                 int32_t* n_active_out, int32_t* max_steps_out, void* workspace_dev,
                 size_t workspace_bytes, void* stream) {
  if (int s = check_cd(n, d, k, LASSO_F32)) return s;
  if (!workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null workspace");
  if (iters < 0 || !(alpha >= 0.0)) return fail(LASSO_ERR_BAD_ARG, "iters=%d alpha=%g", iters, alpha);
  const int kp = pad_k_cd(k);
  CdWorkspace ws = carve_cd(workspace_dev, n, d, kp);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool want_info = n_active_out || max_steps_out;
  int info[2] = {0, 0};
  if (n > 0) {
    CdParams p;
    p.B = ws.B; p.Zt = ws.Zt; p.S = ws.S; p.active = ws.active; p.row_steps = ws.row_steps;
    p.counter = ws.counter; p.n = (int)n; p.iters = iters;
    p.alpha = (float)alpha; p.tol = (float)tol_abs;
    LASSO_HIP_TRY(launch_cd_rows(p, kp, device_cus(), want_info ? ws.info : nullptr, st));
    if (want_info) {
      LASSO_HIP_TRY(hipMemcpyAsync(info, ws.info, sizeof(info), hipMemcpyDeviceToHost, st));
      LASSO_HIP_TRY(hipStreamSynchronize(st));
    }
  }
  if (n_active_out) *n_active_out = info[0];
  if (max_steps_out) *max_steps_out = info[1];
  return LASSO_OK;
}

int lasso_cd_finish(void* z_out_dev, int64_t ldz, void* z_track_out_dev, int64_t ldzt, int64_t n,
                    int64_t d, int64_t k, double alpha, void* workspace_dev, size_t workspace_bytes,
                    void* stream) {
  if (int s = check_cd(n, d, k, LASSO_F32)) return s;
  if (!workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null workspace");
  if ((z_out_dev && ldz < k) || (z_track_out_dev && ldzt < k)) return fail(LASSO_ERR_BAD_ARG, "leading dimension too small");
  const int kp = pad_k_cd(k);
  CdWorkspace ws = carve_cd(workspace_dev, n, d, kp);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  LASSO_HIP_TRY(launch_cd_finish(ws.B, ws.Zt, kp, (float*)z_out_dev, ldz, (float*)z_track_out_dev, ldzt,
                                 (int)n, (int)k, (float)alpha, static_cast<hipStream_t>(stream)));
  return LASSO_OK;
}

int lasso_cd_solve(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, void* z0_inout_dev,
                   int64_t ldz0, void* z_out_dev, int64_t ldz, int64_t n, int64_t d, int64_t k, int dtype,
                   double alpha, int maxiter, double tol, int32_t* n_active_out, int32_t* max_steps_out,
                   void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!z_out_dev && n > 0) return fail(LASSO_ERR_BAD_ARG, "null z_out");
  if (int s = lasso_cd_prepare(x_dev, ldx, w_dev, ldw, z0_inout_dev, ldz0, n, d, k, dtype, workspace_dev,
                               workspace_bytes, stream))
    return s;
  // :9  tol = tol * code_dim
  if (int s = lasso_cd_run(n, d, k, alpha, tol * (double)k, maxiter, n_active_out, max_steps_out,
                           workspace_dev, workspace_bytes, stream))
    return s;
  return lasso_cd_finish(z_out_dev, ldz, z0_inout_dev, ldz0, n, d, k, alpha, workspace_dev,
                         workspace_bytes, stream);
}

// ---- convolutional ISTA/FISTA: lasso/conv2d/ista.py:7-49, lip_const.py:96-135 ----------
size_t lasso_conv_ista_workspace_bytes(int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz,
                                       int64_t Wz, int kh, int kw, int sh, int sw, int ph, int pw) {
  const ConvGeom g = make_geom(N, C, H, W, K, Hz, Wz, kh, kw, sh, sw, ph, pw);
  if (check_conv(g, LASSO_F32)) return 0;
  return carve_conv(nullptr, g).bytes;
}

const char* lasso_conv_ista_kernel_name(int64_t N, int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz, int64_t Wz,
                                        int kh, int kw, int sh, int sw, int ph, int pw) {
  const ConvGeom g = make_geom(N, C, H, W, K, Hz, Wz, kh, kw, sh, sw, ph, pw);
  if (N <= 0 || check_conv(g, LASSO_F32)) return "";
  const int cus = std::max(device_cus(), 1);
  if (const char* fused = conv_fused_kernel_name(g, cus)) return fused;
  // the two-kernel form: what the launchers themselves say about this geometry (their dry runs: no duplicate of the
  // eligibility rules here -- VERDICT r05 / tests/test_bench_gpu.py: a name must occur in its own dispatch)
  static const float kAligned[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};
  bool synth = false, few = false;
  int gp = 0;
  (void)launch_conv_synth(kAligned, kAligned, kAligned, nullptr, g, cus, &synth, nullptr, 1);
  if (!synth) (void)launch_conv_synth_few(kAligned, kAligned, kAligned, nullptr, g, cus, &few, nullptr, 1);
  (void)launch_conv_grad_prox(nullptr, nullptr, 0, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, kGenGrid, g, cus, &gp, nullptr, 1);
  static thread_local char cname[192];
  snprintf(cname, sizeof(cname), "%s + %s",
           synth ? "lasso::conv_synth_kernel" : few ? "lasso::conv_synth_few_kernel"
                                                    : "lasso::gemm_nt_kernel + lasso::conv_residual_kernel",
           gp > 0 ? "lasso::conv_grad_prox_kernel"
                  : "lasso::conv_patches_kernel + lasso::gemm_nt_kernel + lasso::generic_prox_kernel");
  return cname;
}

int lasso_conv_ista_solve(const void* x_dev, const void* w_dev, const void* z0_dev, void* z_out_dev, int64_t N,
                          int64_t C, int64_t H, int64_t W, int64_t K, int64_t Hz, int64_t Wz, int kh, int kw,
                          int sh, int sw, int ph, int pw, int dtype, double alpha, double lr, int fast,
                          int maxiter, double tol, int32_t* iters_out, float* last_delta_out,
                          void* workspace_dev, size_t workspace_bytes, void* stream) {
  const ConvGeom g = make_geom(N, C, H, W, K, Hz, Wz, kh, kw, sh, sw, ph, pw);
  if (int s = check_conv(g, dtype)) return s;
  if (!w_dev || !workspace_dev || (N > 0 && (!x_dev || !z_out_dev))) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (maxiter < 0 || !(lr > 0.0) || !(alpha >= 0.0)) return fail(LASSO_ERR_BAD_ARG, "maxiter=%d lr=%g alpha=%g", maxiter, lr, alpha);
  ConvWorkspace ws = carve_conv(workspace_dev, g);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (iters_out) *iters_out = 0;
  if (last_delta_out) *last_delta_out = NAN;
  if (N == 0) return LASSO_OK;
  const int ckk = g.C * g.kh * g.kw, P = g.Hz * g.Wz;
  const int64_t M = (int64_t)g.N * P;
  const int ldr = (ckk + 3) / 4 * 4;
  const int cus = std::max(device_cus(), 1);
  const float* const conv_w = (const float*)w_dev;
  bool fused = false;                     // conv_fused.hip: whole iterations in one launch, a workgroup per image
  LASSO_HIP_TRY(launch_conv_fused_pack(conv_w, ws.Wf, g, cus, &fused, st));
  if (!fused) LASSO_HIP_TRY(launch_conv_pack_w(conv_w, ws.Wt, ws.Wp, g.K, ckk, ldr, st));
  if (z0_dev) {
    LASSO_HIP_TRY(launch_conv_relayout((const float*)z0_dev, ws.Zm, ws.Ym, g.N, g.K, P, 1, st));     // y0 = z0 in the same pass
  } else {
    LASSO_HIP_TRY(hipMemsetAsync(ws.Zm, 0, (size_t)M * g.K * 4, st));
    LASSO_HIP_TRY(hipMemsetAsync(ws.Ym, 0, (size_t)M * g.K * 4, st));
  }
  const float budget = (float)((double)M * (double)g.K * tol);     // ista.py:16, compared in fp32
  const float lr_f = (float)lr, lam = (float)(alpha * lr);
  double t_mom = 1.0;
  float last = NAN;
  int it = 0;
  // conv_fused.hip takes up to 64 iterations per launch: `iterate` only queues them (momentum factor + where the sum
  // goes; a chunk's slots are consecutive), `flush` launches what is queued
  float q_coef[64];
  int queued = 0;
  float* q_slot0 = nullptr;
  const int q_max = fused ? std::min(64, conv_fused_max_iters(g, cus)) : 0;
  const bool two_y = fused && conv_fused_two_y_buffers(g, cus);     // bands: y ping-pongs between Ym and G
  auto flush = [&]() -> int {
    if (queued == 0) return LASSO_OK;
    LASSO_HIP_TRY(launch_conv_fused(ws.Wf, (const float*)x_dev, ws.Zm, ws.Ym, two_y ? ws.G : ws.Ym, lr_f, lam, q_coef,
                                    queued, ws.dpart, kConvDpart, q_slot0, g, cus, st));            // :19-20,:29,:42,:44
    if (two_y) std::swap(ws.Ym, ws.G);           // (save / restore below copy whatever ws.Ym is at the time)
    queued = 0;
    q_slot0 = nullptr;
    return LASSO_OK;
  };
  // one iteration on the stream; the sum |z - z_next| of the iteration (ista.py:44) goes to *delta_slot when given
  auto iterate = [&](float* delta_slot) -> int {
    const double t_next = (1.0 + sqrt(1.0 + 4.0 * t_mom * t_mom)) / 2.0;           // :41
    const float coef = fast ? (float)((t_mom - 1.0) / t_next) : 0.0f;               // :42
    if (fused) {
      const bool fits = queued < q_max &&
                        (delta_slot ? (q_slot0 && delta_slot == q_slot0 + queued) : q_slot0 == nullptr);
      if (queued > 0 && !fits)
        if (int s = flush()) return s;
      if (queued == 0) q_slot0 = delta_slot;
      q_coef[queued++] = coef;
      t_mom = t_next;
      return LASSO_OK;
    }
    int dcount = 0;
    LASSO_HIP_TRY(launch_conv_residual(ws.Ym, ws.Wt, conv_w, (const float*)x_dev, ws.PT, ws.R, g, cus, st));   // :19
    // gradient + prox: the fused implicit-GEMM kernel when the geometry fits, else patches + GEMM + prox
    LASSO_HIP_TRY(launch_conv_grad_prox(ws.R, ws.Wp, ldr, ws.Zm, ws.Ym, lr_f, lam, coef, ws.dpart, kGenGrid, g, cus,
                                        &dcount, st));                                            // :20,:29,:42,:44
    if (dcount == 0) {
      LASSO_HIP_TRY(launch_conv_gradient(ws.R, ws.Wp, ws.PT, ldr, ws.G, g, st));                  // :20
      LASSO_HIP_TRY(launch_generic_prox(ws.Zm, g.K, ws.Ym, ws.G, (int)M, g.K, lr_f, lam, coef, ws.dpart,
                                        kGenGrid, st));                                           // :29,:42,:44
      dcount = kGenGrid;
    }
    t_mom = t_next;
    if (delta_slot) {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, ws.dpart, dcount, delta_slot);
      LASSO_HIP_TRY(hipGetLastError());
    }
    return LASSO_OK;
  };
  if (!(tol > 0.0)) {
    for (; it < maxiter; ++it)
      if (int s = iterate(nullptr)) return s;
    if (int s = flush()) return s;
  } else {
    const int64_t code_words = M * g.K;
    const int copy_grid = (int)std::min<int64_t>((code_words / 4 + 255) / 256 + 1, (int64_t)cus * 16);
    auto save = [&]() -> int {
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Zm, ws.Zc, code_words);
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Ym, ws.Yc, code_words);
      LASSO_HIP_TRY(hipGetLastError());
      return LASSO_OK;
    };
    auto restore = [&]() -> int {
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Zc, ws.Zm, code_words);
      hipLaunchKernelGGL(copy_words_kernel, dim3(copy_grid), dim3(256), 0, st, ws.Yc, ws.Ym, code_words);
      LASSO_HIP_TRY(hipGetLastError());
      return LASSO_OK;
    };
    if (int s = speculate_stop_rule(maxiter, budget, ws.delta, st, &t_mom, iterate, save, restore, flush, &it, &last,
                                    "lasso_conv_ista_solve"))
      return s;
  }
  LASSO_HIP_TRY(launch_conv_relayout(ws.Zm, (float*)z_out_dev, nullptr, g.N, g.K, P, 0, st));
  if (iters_out) *iters_out = it;
  if (last_delta_out) *last_delta_out = last;
  return LASSO_OK;
}

// (0.5*||x - conv_transpose2d(z)||^2 + alpha*||z||_1) / N   (ista.py:23-26) -> loss_dev
int lasso_conv_objective(const void* x_dev, const void* w_dev, const void* z_dev, int64_t N, int64_t C, int64_t H,
                         int64_t W, int64_t K, int64_t Hz, int64_t Wz, int kh, int kw, int sh, int sw, int ph,
                         int pw, int dtype, double alpha, float* loss_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream) {
  const ConvGeom g = make_geom(N, C, H, W, K, Hz, Wz, kh, kw, sh, sw, ph, pw);
  if (int s = check_conv(g, dtype)) return s;
  if (!x_dev || !w_dev || !z_dev || !loss_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (N == 0) return fail(LASSO_ERR_BAD_ARG, "empty batch");
  ConvWorkspace ws = carve_conv(workspace_dev, g);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ckk = g.C * g.kh * g.kw, P = g.Hz * g.Wz;
  const int64_t M = (int64_t)g.N * P;
  LASSO_HIP_TRY(launch_conv_pack_w((const float*)w_dev, ws.Wt, ws.Wp, g.K, ckk, (ckk + 3) / 4 * 4, st));
  LASSO_HIP_TRY(launch_conv_relayout((const float*)z_dev, ws.Zm, nullptr, g.N, g.K, P, 1, st));
  LASSO_HIP_TRY(launch_conv_residual(ws.Zm, ws.Wt, (const float*)w_dev, (const float*)x_dev, ws.PT, ws.R, g,
                                     std::max(device_cus(), 1), st));
  LASSO_HIP_TRY(launch_objective_reduce(ws.R, (int64_t)g.N * g.C * g.H * g.W, ws.Zm, g.K, (int)M, g.K, ws.dpart,
                                        kGenGrid, alpha, (double)g.N, ws.sums, loss_dev, st));
  return LASSO_OK;
}

size_t lasso_conv_lip_workspace_bytes(int64_t K, int64_t C, int ksize, int sample) {
  (void)ksize;
  if (K <= 0 || C <= 0 || sample <= 0) return 0;
  return align_up((size_t)sample * 4) + align_up((size_t)std::min(K, C) * 4) + 256;
}

int lasso_conv_lip_bound(const void* w_dev, int64_t K, int64_t C, int ksize, int padding, int sample,
                         int take_sqrt, double* l_out, void* workspace_dev, size_t workspace_bytes,
                         void* stream) {
  if (!w_dev || !workspace_dev || K <= 0 || C <= 0 || ksize <= 0 || sample < 2)
    return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (ksize % 2 != 1) return fail(LASSO_ERR_BAD_ARG, "The dimension of the kernel must be odd.");   // :101-102
  if (workspace_bytes < lasso_conv_lip_workspace_bytes(K, C, ksize, sample))
    return fail(LASSO_ERR_WORKSPACE, "need %zu bytes", lasso_conv_lip_workspace_bytes(K, C, ksize, sample));
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = (char*)workspace_dev;
  float* freq = (float*)base;
  float* maxes = (float*)(base + align_up((size_t)sample * 4));
  double* out = (double*)((char*)maxes + align_up((size_t)std::min(K, C) * 4));
  // torch.linspace(0, 2*pi, sample) in fp32 (:110): start + i*step below the midpoint,
  // end - (sample-1-i)*step above it
  std::vector<float> f(sample);
  const float lo = 0.0f, hi = (float)(2.0 * M_PI);
  const float step = (hi - lo) / (float)(sample - 1);
  for (int i = 0; i < sample; ++i) f[i] = i < sample / 2 ? lo + step * (float)i : hi - step * (float)(sample - 1 - i);
  LASSO_HIP_TRY(hipMemcpyAsync(freq, f.data(), (size_t)sample * 4, hipMemcpyHostToDevice, st));
  LASSO_HIP_TRY(hipStreamSynchronize(st));            // f is a stack-lifetime staging buffer
  const int T = ksize * ksize;
  // the smaller channel dimension is summed last (:106-107)
  const bool swap = K > C;
  const int O = (int)(swap ? C : K), I = (int)(swap ? K : C);
  const int64_t so = swap ? T : (int64_t)C * T, si = swap ? (int64_t)C * T : T;
  if ((size_t)I * T * 4 > 64 * 1024) return fail(LASSO_ERR_UNSUPPORTED, "kernel too large for the bound kernel");
  LASSO_HIP_TRY(launch_conv_lip((const float*)w_dev, O, I, so, si, ksize, padding, freq, sample, take_sqrt, maxes,
                                out, st));
  if (l_out) {
    LASSO_HIP_TRY(hipMemcpyAsync(l_out, out, sizeof(double), hipMemcpyDeviceToHost, st));
    LASSO_HIP_TRY(hipStreamSynchronize(st));
  }
  return LASSO_OK;
}

// ---- reverse-mode derivative of the unrolled fixed-step solve (autograd.hip) --------------
namespace {
struct BwWorkspace { float* Wt; float* zbA; float* zbB; float* yb; float* ub; float* gb; float* y; float* nr; float* rb; float* T1; float* T2; float* scratch; size_t bytes; };
BwWorkspace carve_bw(void* base, int64_t n, int64_t d, int64_t k) {
  BwWorkspace w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += align_up(std::max<size_t>(bytes, 4));
    return reinterpret_cast<float*>(r);
  };
  const size_t nk = (size_t)n * k * 4, nd = (size_t)n * d * 4, dk = (size_t)d * k * 4;
  w.Wt = take(dk);
  w.zbA = take(nk); w.zbB = take(nk); w.yb = take(nk); w.ub = take(nk); w.gb = take(nk); w.y = take(nk);
  w.nr = take(nd); w.rb = take(nd);
  w.T1 = take(dk); w.T2 = take(dk);
  w.scratch = take(16 * dk);
  w.bytes = off;
  return w;
}
}  // namespace

size_t lasso_fista_backward_workspace_bytes(int64_t n, int64_t d, int64_t k) {
  if (n < 0 || d <= 0 || k <= 0) return 0;
  return carve_bw(nullptr, n, d, k).bytes;
}

int lasso_fista_backward(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* trace_dev,
                         const void* grad_z_dev, int64_t n, int64_t d, int64_t k, int dtype, double lr, int fast,
                         int iterations, void* grad_x_dev, void* grad_w_dev, void* grad_z0_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream) {
  return lasso_fista_backward_steps(x_dev, ldx, w_dev, ldw, trace_dev, grad_z_dev, n, d, k, dtype, lr, nullptr, fast,
                                    iterations, grad_x_dev, grad_w_dev, grad_z0_dev, workspace_dev, workspace_bytes,
                                    stream);
}

int lasso_fista_backward_steps(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* trace_dev,
                               const void* grad_z_dev, int64_t n, int64_t d, int64_t k, int dtype, double lr,
                               const float* lr_steps_host, int fast, int iterations, void* grad_x_dev,
                               void* grad_w_dev, void* grad_z0_dev, void* workspace_dev, size_t workspace_bytes,
                               void* stream) {
  if (int s = check_common(n, d, k, dtype, /*allow_large=*/true)) return s;
  if (!x_dev || !w_dev || !trace_dev || !grad_z_dev || !workspace_dev) return fail(LASSO_ERR_BAD_ARG, "null pointer");
  if (ldx < d || ldw < k || iterations < 0) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  if (d > INT32_MAX / 2 || k > INT32_MAX / 2) return fail(LASSO_ERR_UNSUPPORTED, "shape too large");
  BwWorkspace ws = carve_bw(workspace_dev, n, d, k);
  if (workspace_bytes < ws.bytes) return fail(LASSO_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* X = (const float*)x_dev;
  const float* W = (const float*)w_dev;
  const float* trace = (const float*)trace_dev;
  const int64_t nk = n * k;
  float* gx = (float*)grad_x_dev;
  float* gw = (float*)grad_w_dev;
  float* gz0 = (float*)grad_z0_dev;
  if (gx && n > 0) LASSO_HIP_TRY(hipMemsetAsync(gx, 0, (size_t)n * d * 4, st));
  if (gw) LASSO_HIP_TRY(hipMemsetAsync(gw, 0, (size_t)d * k * 4, st));
  if (n == 0) return LASSO_OK;
  const int cus = device_cus();
  // momentum coefficients c_i = (t_i - 1)/t_{i+1}  (ista.py:98-99)
  std::vector<float> coef((size_t)std::max(iterations, 1));
  {
    double t = 1.0;
    for (int i = 0; i < iterations; ++i) {
      const double tn = (1.0 + sqrt(1.0 + 4.0 * t * t)) / 2.0;
      coef[i] = fast ? (float)((t - 1.0) / tn) : 0.0f;
      t = tn;
    }
  }
  LASSO_HIP_TRY(launch_transpose_pad(W, ldw, (int)d, (int)k, ws.Wt, d, (int)k, (int)d, st));
  float* zb_next = ws.zbA;
  float* zb_cur = ws.zbB;
  LASSO_HIP_TRY(hipMemcpyAsync(zb_next, grad_z_dev, (size_t)nk * 4, hipMemcpyDeviceToDevice, st));
  LASSO_HIP_TRY(hipMemsetAsync(ws.yb, 0, (size_t)nk * 4, st));
  const int splits = gram_splits((int)d, (int)k, (int)n, 0, cus);
  for (int i = iterations - 1; i >= 0; --i) {
    const float* z_next = trace + (int64_t)(i + 1) * nk;
    const float* z_i = trace + (int64_t)i * nk;
    // c_i formed y_{i+1}; for the last iteration y_{i+1} was never used (yb == 0)
    LASSO_HIP_TRY(launch_bw_prox(zb_next, zb_cur, ws.yb, z_next, ws.ub, ws.gb, nk, coef[i],
                                 lr_steps_host ? lr_steps_host[i] : (float)lr, st));
    // the point of iteration i:  y_i = z_i + c_{i-1} (z_i - z_{i-1}),  y_0 = z_0
    LASSO_HIP_TRY(launch_bw_point(z_i, (i > 0 && fast) ? trace + (int64_t)(i - 1) * nk : nullptr, ws.y, nk,
                                  i > 0 ? coef[i - 1] : 0.0f, st));
    // nr = x - y W^T (= -r_i);  rb = gb W^T;  yb_i = ub + rb W
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.y, k, W, ldw, X, ldx, ws.nr, d, (int)n, (int)d, (int)k, st, 0));
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.gb, k, W, ldw, nullptr, 0, ws.rb, d, (int)n, (int)d, (int)k, st, 1));
    LASSO_HIP_TRY(launch_gemm_nt_sub(ws.rb, d, ws.Wt, d, ws.ub, k, ws.yb, k, (int)n, (int)k, (int)d, st, 1));
    if (gw) {
      // Wb += r_i^T gb + rb^T y_i = -(nr^T gb) + rb^T y_i
      LASSO_HIP_TRY(launch_gram_tn(ws.nr, d, (int)d, ws.gb, k, (int)k, (int)n, ws.T1, k, 0, ws.scratch, splits, st));
      LASSO_HIP_TRY(launch_gram_tn(ws.rb, d, (int)d, ws.y, k, (int)k, (int)n, ws.T2, k, 0, ws.scratch, splits, st));
      LASSO_HIP_TRY(launch_bw_axpy(gw, ws.T2, 1.0f, ws.T1, -1.0f, d * k, st));
    }
    if (gx) LASSO_HIP_TRY(launch_bw_axpy(gx, ws.rb, -1.0f, nullptr, 0.0f, n * d, st));
    std::swap(zb_next, zb_cur);       // zb_cur (adjoint of z_i so far) becomes the next "zb_{i+1}"
  }
  if (gz0) {
    // z0b = zb_0 + yb_0  (y_0 = z_0)
    LASSO_HIP_TRY(hipMemcpyAsync(gz0, zb_next, (size_t)nk * 4, hipMemcpyDeviceToDevice, st));
    LASSO_HIP_TRY(launch_bw_axpy(gz0, ws.yb, 1.0f, nullptr, 0.0f, nk, st));
  }
  return LASSO_OK;
}

// ---- patch front end (conv.hip) --------------------------------------------------------
namespace {
int patch_geom(ConvGeom& g, int64_t N, int64_t C, int64_t H, int64_t W, int ph, int pw, int sh, int sw) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0 || ph > H || pw > W)
    return fail(LASSO_ERR_BAD_ARG, "bad patch geometry");
  g = make_geom(N, C, H, W, 1, (H - ph) / sh + 1, (W - pw) / sw + 1, ph, pw, sh, sw, 0, 0);
  if ((int64_t)g.N * g.Hz * g.Wz > INT32_MAX / 2) return fail(LASSO_ERR_UNSUPPORTED, "too many patches");
  return LASSO_OK;
}
}  // namespace

int lasso_patches_extract(const void* img_dev, void* patches_dev, int64_t ld, float* means_dev, int64_t N,
                          int64_t C, int64_t H, int64_t W, int ph, int pw, int sh, int sw, int center,
                          void* stream) {
  ConvGeom g;
  if (int s = patch_geom(g, N, C, H, W, ph, pw, sh, sw)) return s;
  if ((N > 0 && (!img_dev || !patches_dev)) || ld < (int64_t)C * ph * pw) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  LASSO_HIP_TRY(launch_patches_extract((const float*)img_dev, (float*)patches_dev, ld, means_dev, g, center,
                                       static_cast<hipStream_t>(stream)));
  return LASSO_OK;
}

int lasso_patches_reconstruct(const void* patches_dev, int64_t ld, const float* means_dev, void* img_out_dev,
                              int64_t N, int64_t C, int64_t H, int64_t W, int ph, int pw, int sh, int sw,
                              void* stream) {
  ConvGeom g;
  if (int s = patch_geom(g, N, C, H, W, ph, pw, sh, sw)) return s;
  if ((N > 0 && (!img_out_dev || !patches_dev)) || ld < (int64_t)C * ph * pw) return fail(LASSO_ERR_BAD_ARG, "bad argument");
  LASSO_HIP_TRY(launch_patches_reconstruct((const float*)patches_dev, ld, means_dev, (float*)img_out_dev, g,
                                           static_cast<hipStream_t>(stream)));
  return LASSO_OK;
}

}  // extern "C"
