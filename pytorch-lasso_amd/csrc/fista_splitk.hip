// Small-batch variant of the fused persistent FISTA kernel: a 16-row tile is owned by a
// GROUP of C = K/128 workgroups (C = 8 at K = 1024), each on its own CU, so that a batch of
// n rows keeps n/16 * C compute units busy instead of n/16 (BASELINE's n = 4096 on 8 GPUs is
// 512 rows = 32 tiles per GPU: 32 CUs with the one-workgroup-per-tile kernel, 256 here).
//
// Member j of a group owns the dictionary columns (atoms) [128j, 128j+128):
//   * its slice of W never moves: the B operands of both GEMMs live in REGISTERS for the
//     whole solve (per wave 64 VGPRs of W[d-block of the wave][slice] for GEMM-1 and 64 VGPRs
//     of W[:, 16 atoms of the wave] for GEMM-2 -- together the 256 x 128 slice twice, 256 KiB
//     of the CU's 512 KiB register file); nothing is streamed from L2 inside the loop;
//   * GEMM-1 contracts over the member's own atoms only: p_j = y[:, slice] W[:, slice]^T, a
//     16 x 256 PARTIAL residual.  The members exchange their partials through L2 (16-byte
//     stores + one flag per producing wave; consumers poll the flags and read with
//     L1-bypassing loads) and every member forms the same r = p_0 + p_1 + ... in the same
//     fixed order (p_0's MFMA chain starts from -x);
//   * GEMM-2 + prox + momentum act on the member's own atoms: g[:, slice] = r W[:, slice],
//     so z, y of the slice never leave the CU (z in registers, y in an 8 KiB LDS tile).
// A group works on T tiles at once (T = 1, 2, 4): per iteration GEMM-1 of all T tiles, then
// per tile {wait for the peers' partials, sum, GEMM-2, prox}.  The hand-off latency of tile t
// (~2 us) is hidden behind the GEMMs of the other tiles; with T = 1 it is exposed.
//
// Arithmetic is IDENTICAL to fista_tile_sp.hip, operation by operation: that kernel sums
// GEMM-1 in the same 128-atom slices in the same order, GEMM-2's chain and the epilogue are
// the same instruction sequence -- a row's code does not depend on which kernel (or how many
// GPUs) computed it (tests/test_splitk_gpu.py, bitwise).
//
// Hand-off safety: tags are a per-launch epoch counter (never 0; flags zeroed by the host
// before every launch); payload buffers alternate by epoch parity -- a wave can publish epoch
// e+1 only after it consumed epoch e, which needs every peer's epoch-e publish, which each
// peer issues after ITS epoch e-1 loads returned.  Every spin is bounded: on a timeout (a peer
// is not resident) the whole grid aborts without touching z_out and the host re-runs the work
// on the one-workgroup-per-tile kernel.
#include "tile_device.hpp"

#ifdef LASSO_SPLITK_TIMING
// debug build (tools/splitk_timeline.py): wall-clock stamps of ONE iteration, 32 per workgroup
__device__ unsigned long long lasso_splitk_stamps[2048 * 32];
extern "C" int lasso_debug_splitk_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lasso_splitk_stamps), sizeof(lasso_splitk_stamps));
}
#define SK_STAMP(slot) do { if (it == LASSO_SPLITK_TIMING && tid == 0) lasso_splitk_stamps[blockIdx.x * 32 + (slot)] = wall_clock64(); } while (0)
#else
#define SK_STAMP(slot) do { } while (0)
#endif

namespace lasso {
namespace splitk {

constexpr int kSlice = 128;           // atoms per member
constexpr int kPartBytes = kTileM * kFistaD * 4;   // one partial residual tile: 16 KiB
constexpr int kSplitMaxParts = kSplitkMaxParts;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 as_f32x4(u32x4 v) {
  f32x4 r;
  r[0] = __uint_as_float(v[0]); r[1] = __uint_as_float(v[1]);
  r[2] = __uint_as_float(v[2]); r[3] = __uint_as_float(v[3]);
  return r;
}
__device__ __forceinline__ u32x4 as_u32x4(f32x4 v) {
  u32x4 r;
  r[0] = __float_as_uint(v[0]); r[1] = __float_as_uint(v[1]);
  r[2] = __float_as_uint(v[2]); r[3] = __float_as_uint(v[3]);
  return r;
}

constexpr size_t lds_bytes(int T) {
  return (size_t)T * (kTileM * kSlice * 4) + (size_t)T * (kTileM * kFistaD * 4) + 2 * (size_t)(kTileM * kFistaD * 4) + 128;
}

// STOP: compile the in-kernel global stop rule in (needs every tile in a resident group slot).
// Descriptor of `rows` rows of a row-major matrix from row0 on (null base or rows <= 0: nothing, every read 0) and a
// 4-byte load at (r, cc) through it -- r >= rows is out of range by the record count, `col_ok` false by the offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rows_rsrc(const float* base, int64_t ld, int row0, int rows) {
  const int64_t bytes = (base && rows > 0) ? (int64_t)rows * ld * 4 : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + (int64_t)row0 * ld : base), 0,
                                           (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
}
__device__ __forceinline__ float tile_rows_load(const __amdgpu_buffer_rsrc_t rs, int64_t ld, int r, int cc, bool col_ok) {
  unsigned o = col_ok ? (unsigned)(r * (int)ld + cc) * 4u : 0xfffffff0u;
  asm volatile("" : "+v"(o));
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
}

template <int K, int T, bool STOP>
__global__ __launch_bounds__(kFistaThreads, 2) void fista_splitk_kernel(const FistaTileParams p) {
  // step size and threshold: launch arguments, or device memory (lr = LASSO_LR_AUTO)
  const float lr_ = p.lr_dev ? p.lr_dev[0] : p.lr, lam_ = p.lr_dev ? p.lr_dev[1] : p.lam;
  constexpr int C = K / kSlice;
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int YT_BYTES = kTileM * kSlice * 4;     // 8 KiB per tile slot
  constexpr int RT_BYTES = kTileM * D * 4;          // 16 KiB
  // partial loads in flight per wave: all C peers at T = 1 (latency), half of them when other
  // tiles cover the latency anyway (registers)
  constexpr int PB = (T == 1 || C <= 4) ? C : C / 2;
  static_assert(C >= 2 && C <= 8 && K % kSlice == 0 && (T == 1 || T == 2 || T == 4), "geometry");

  // block -> (group, member): the C members of a group sit on blocks that are congruent
  // mod 8, i.e. -- as dispatch is observed to go -- on ONE XCD
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mem = idx % C;                              // member = atom slice
  const int grp = (idx / C) * 8 + xcd;
  if (grp >= p.groups) return;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const yt = (lds_char*)smem;                 // [T] y slices
  lds_char* const xt = yt + T * YT_BYTES;               // [T] -x in the GEMM-1 accumulator layout, [wave][cb][lane] x 16 B
  lds_char* const rt = xt + T * RT_BYTES;               // [2] residual tiles (ping-pong over the tiles of an iteration)
  lds_f32* const red = (lds_f32*)(rt + 2 * RT_BYTES);   // [NW] delta sums, [NW] verdict, [NW+1] total, [NW+2] abort, [NW+3] one-XCD

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;

  // ---- the member's slice of W, resident in registers for the whole solve ----------------
  // GEMM-1 (wave -> residual columns [32w, 32w+32)): B[k][dcol] = Wp[dcol][128 mem + k]
  f32x4 b1[4][2][2];      // [step][k-half][col-block], component jj: k = 32 s + 16 ss + 4 q + jj
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        b1[s][ss][cb] = *reinterpret_cast<const f32x4*>(p.Wp + (size_t)(32 * wid + 16 * cb + n) * K +
                                                         kSlice * mem + 32 * s + 16 * ss + 4 * q);
  // GEMM-2 (wave -> atoms [16w, 16w+16) of the slice): B[d][atom] = Wtp[128 mem + 16 w + n][d]
  f32x4 b2[D / 32][2];    // [d-chunk][half], component jj: d = 32 t + 16 ss + 4 q + jj
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
      b2[t][ss] = *reinterpret_cast<const f32x4*>(p.Wtp + (size_t)(kSlice * mem + 16 * wid + n) * D + 32 * t +
                                                   16 * ss + 4 * q);

  // exchange buffers: payload [parity][group][slot][member][wave][cb][lane] x 16 B,
  // flags [group][member][wave][slot]
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, 2 * p.groups * T * C * kPartBytes, 0x00020000);
  const unsigned my_part = (unsigned)((grp * T * C + mem) * kPartBytes + wid * 2048 + lane * 16);   // + slot * C * kPartBytes
  const unsigned peer_part0 = (unsigned)(grp * T * C * kPartBytes + wid * 2048 + lane * 16);
  const unsigned parity_stride = (unsigned)(p.groups * T * C * kPartBytes);
  unsigned* const my_flag = p.xflags + ((size_t)(grp * C + mem) * NW + wid) * T;                     // + slot
  const unsigned* const peer_flag = p.xflags + ((size_t)(grp * C) * NW + wid) * T;                   // + peer * NW * T + slot
  const int col0 = kSlice * mem + 16 * wid;             // first atom of this wave's z / y block

  // ---- are all members of this group on ONE XCD? ------------------------------------------
  // The hand-off is correct on any placement (write-through stores, L1-bypassing loads).
  // When the C members share an XCD -- the dispatcher is observed to put block b on XCD b % 8,
  // and the block -> (group, member) map above follows that -- they also share its L2, which is
  // the coherence point of an XCD: plain stores (acknowledged by L2 at vmcnt(0), L1 is
  // write-through) are then visible to the peers' L1-bypassing loads without crossing the
  // fabric, and the lines stay in that L2 for the next iteration.  That is decided at run time
  // from the hardware's XCC id, exchanged once per launch with the placement-independent form:
  // a different placement only selects the slower path.
  bool local;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* const xid =
        reinterpret_cast<unsigned long long*>(p.xflags + (size_t)kSplitMaxParts * NW * 4) + (size_t)grp * C;
    if (wid == 0) {
      if (lane == 0)
        __hip_atomic_store(xid + mem, (1ull << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      bool ok, same;
      do {
        unsigned long long v = (1ull << 32) | xcc;
        if (lane < C) v = __hip_atomic_load(xid + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = __all((unsigned)(v >> 32) == 1u);
        same = __all((unsigned)v == xcc);
        if (!ok) {
          __builtin_amdgcn_s_sleep(2);
          if ((spins & 63) == 63 &&
              __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            break;
        }
      } while (!ok && ++spins < kStopSpinLimit);
      if (lane == 0) {
        red[NW + 3] = (ok && same) ? 1.0f : 0.0f;
        red[NW + 2] = ok ? 0.0f : 1.0f;
        if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    local = red[NW + 3] != 0.0f;
    if (red[NW + 2] != 0.0f) return;                    // a peer never showed up: the host re-runs the work
    __builtin_amdgcn_s_barrier();                        // red[] is rewritten below
  }
  const __amdgpu_buffer_rsrc_t frsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.xflags, 0, kSplitMaxParts * NW * 4 * 4, 0x00020000);
  const unsigned my_flag_off = (unsigned)((((grp * C + mem) * NW + wid) * T) * 4);

  unsigned epoch = 0;
  bool aborted = false;
  const int nparts = p.groups * C;                      // one |dz| granule / partial per workgroup and iteration
  // group g owns tiles g + groups * (T * round + slot)
  for (int round = 0; grp + p.groups * T * round < p.ntiles; ++round) {
    int nt = 0;                                         // active slots of this round (uniform)
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (grp + p.groups * (T * round + t) < p.ntiles) nt = t + 1;
    // ---- state of the tiles: z (registers), y (LDS slices), -x (member 0's GEMM-1 start) ---
    f32x4 zreg[T];
    static_for<T>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      const int row0 = (grp + p.groups * (T * round + t)) * kTileM;
      const float* ysrc = p.y_in ? p.y_in : p.z_in;
      const int64_t ldy = p.y_in ? p.ldy_in : p.ldz_in;
      // (through buffer descriptors of the tile's rows, offset out of range where there is nothing to read, offsets
      // opaque: as `in ? load : 0` every one of these 16 loads sat in a branch of its own with an s_waitcnt vmcnt(0)
      // behind it -- 16 to 24 memory round trips in a row at the head of every tile round; round 5)
      const int trows = (t < nt) ? min((int)kTileM, p.n - row0) : 0;
      const __amdgpu_buffer_rsrc_t zsrc = tile_rows_rsrc(p.z_in, p.ldz_in, row0, trows);
      const __amdgpu_buffer_rsrc_t ysc = tile_rows_rsrc(ysrc, ldy, row0, trows);
      const __amdgpu_buffer_rsrc_t xsc = tile_rows_rsrc(mem == 0 ? p.X : nullptr, p.ldx, row0, trows);
      float yv[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = col0 + n;
        zreg[t][rg] = tile_rows_load(zsrc, p.ldz_in, r, cc, cc < p.k);
        yv[rg] = tile_rows_load(ysc, ldy, r, cc, cc < p.k);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) *(lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(4 * q + rg, 16 * wid + n)) = yv[rg];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        f32x4 xn;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
          const float v = tile_rows_load(xsc, p.ldx, r, cc, cc < p.d);
          xn[rg] = -v;                                   // members > 0 start their chain at 0
        }
        *(lds_f32x4*)(xt + t * RT_BYTES + wid * 2048 + cb * 1024 + lane * 16) = xn;
      }
    });
    if (tid == 0) red[NW + 2] = 0.0f;
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    bool stopped = false;
    for (int it = 0; it < p.iters; ++it) {
      const float coef = p.coef[it];
      ++epoch;
      const unsigned par_off = (epoch & 1u) * parity_stride;

      // ============ GEMM-1 of every tile: p_mem = y[:, slice] W[:, slice]^T (- x), published ====
      SK_STAMP(0);
      int pending = -1;                                  // tile whose partial is stored but not yet flagged
      auto raise_flag = [&](int tp) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in the XCD's L2 / written through
        if (lane == 0) {
          if (local) __builtin_amdgcn_raw_buffer_store_b32(epoch, frsrc, my_flag_off + tp * 4, 0, 0);
          else __hip_atomic_store(my_flag + tp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      static_for<T>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if (t < nt) {
          f32x4 acc[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[cb] = *(const lds_f32x4*)(xt + t * RT_BYTES + wid * 2048 + cb * 1024 + lane * 16);
          const lds_char* const yrow = yt + t * YT_BYTES + n * (kSlice * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            f32x4 a[2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
              a[ss] = *(const lds_f32x4*)(yrow + (s >> 1) * 256 + (((8 * (s & 1) + 4 * ss + q) ^ n) << 4));
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                  acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][jj], b1[s][ss][cb][jj], acc[cb], 0, 0, 0);
          }
          // publish: two 16-byte stores per lane; they drain under the NEXT tile's MFMAs, only then the wave's
          // flag for this slot goes out (the drain of the last tile is the only one that is waited for in the open)
          SK_STAMP(1 + t);                               // tile t: GEMM-1 issued
          if (pending >= 0) raise_flag(pending);
      SK_STAMP(5);                                       // all partials flagged
          const unsigned dst = my_part + par_off + t * (C * kPartBytes);
          if (local) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc[cb]), xrsrc, dst + cb * 1024, 0, 0);
          } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc[cb]), xrsrc, dst + cb * 1024, 0, 16);
          }
          pending = t;
        }
      });
      if (pending >= 0) raise_flag(pending);

      // in-kernel stop rule: fetch the previous iteration's |dz| granules (one per workgroup)
      // now, look at them once the first tile's partials have arrived
      unsigned long long gr[4] = {0ull, 0ull, 0ull, 0ull};
      const bool check = STOP && p.stop_on && it > 0;
      const unsigned long long* const grow =
          p.stop_gran ? p.stop_gran + (size_t)((it - 1) & (kStopRing - 1)) * nparts : nullptr;
      if (STOP && check && wid == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (lane + 64 * e < nparts)
            gr[e] = __hip_atomic_load(grow + lane + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }

      float dsum = 0.0f;
      bool leave = false;
      // ============ per tile: gather the partials, r, GEMM-2 on the wave's 16 atoms, prox ========
      // ONE poll for the flags of all T tiles (lane = peer + C * tile: the same wave of every peer, L1-bypassing):
      // a poll is a round trip of 0.8 us even when the flags are long set (tools/splitk_timeline.py).
      {
        int spins = 0;
        bool ok;
        const int fpeer = lane & (C - 1), ftile = lane / C;
        do {
          unsigned v = epoch;
          if (ftile < nt && fpeer != mem)
            v = __hip_atomic_load(peer_flag + (size_t)fpeer * NW * T + ftile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = __all(v == epoch);
          if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 63 &&
                __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
              break;
          }
        } while (!ok && ++spins < kStopSpinLimit * 4);
        if (!ok && lane == 0) {       // a peer is not resident: the whole grid gives up
          __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          red[NW + 2] = 1.0f;
        }
      }
      SK_STAMP(6);                                         // flags of every tile seen
      static_for<T>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if (t < nt && !leave) {
          if (t == 0 && STOP && check && wid == 0) {
            const unsigned want = (unsigned)it;
            float partsum = 0.0f;
            int spins = 0;
            bool ok;
            do {
              ok = true;
              partsum = 0.0f;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (lane + 64 * e < nparts) {
                  if ((unsigned)(gr[e] >> 32) != want) {
                    gr[e] = __hip_atomic_load(grow + lane + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && ((unsigned)(gr[e] >> 32) == want);
                  }
                  partsum += __uint_as_float((unsigned)gr[e]);
                }
              ok = __all(ok);
              if (!ok) {
                __builtin_amdgcn_s_sleep(8);
                if ((spins & 63) == 63 &&
                    __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                  break;
              }
            } while (!ok && ++spins < kStopSpinLimit);
            const float total = wave_sum(partsum);
            if (lane == 0) {
              red[NW] = !ok ? 2.0f : (total <= p.stop_budget ? 1.0f : 0.0f);      // ista.py:93
              red[NW + 1] = total;
              if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          // r = p_0 + p_1 + ... + p_{C-1}, left to right (the own partial is read back like the
          // others: it sits in L2, and the accumulator registers are free meanwhile)
          const unsigned src = peer_part0 + par_off + t * (C * kPartBytes);
          f32x4 rsum[2];
          static_for<C / PB>([&](auto h_c) {
            constexpr int h = decltype(h_c)::value;
            f32x4 part[PB][2];
#pragma unroll
            for (int pe = 0; pe < PB; ++pe)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                part[pe][cb] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(
                    xrsrc, src + (h * PB + pe) * kPartBytes + cb * 1024, 0, 16));
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if constexpr (h == 0) rsum[cb] = part[0][cb];
#pragma unroll
              for (int pe = (h == 0 ? 1 : 0); pe < PB; ++pe)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) rsum[cb][rg] = __fadd_rn(rsum[cb][rg], part[pe][cb][rg]);
            }
          });
          SK_STAMP(7 + 5 * t);                           // partials summed
          lds_char* const rbuf = rt + (t & 1) * RT_BYTES;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              *(lds_f32*)(rbuf + tile_off<D>(4 * q + rg, 32 * wid + 16 * cb + n)) = rsum[cb][rg];
          LASSO_WAIT_LGKM0();
          __builtin_amdgcn_s_barrier();                  // r tile complete (and, at t = 0, the verdicts)
          SK_STAMP(8 + 5 * t);
          if (red[NW + 2] != 0.0f || (t == 0 && STOP && check && red[NW] == 2.0f)) {
            aborted = true;
            leave = true;
          } else if (t == 0 && STOP && check && red[NW] == 1.0f) {   // iteration it-1 met the rule: z is its z_next
            if (blockIdx.x == 0 && tid == 0) {
              p.stop_out[0] = it;
              p.stop_out[1] = __float_as_int(red[NW + 1]);
            }
            stopped = true;
            leave = true;
          } else {
            f32x4 g2 = {0.f, 0.f, 0.f, 0.f};
            const lds_char* const rrow = rbuf + n * (D * 4);
#pragma unroll
            for (int tt = 0; tt < D / 32; ++tt) {
              f32x4 a[2];
#pragma unroll
              for (int ss = 0; ss < 2; ++ss)
                a[ss] = *(const lds_f32x4*)(rrow + (tt >> 1) * 256 + (((8 * (tt & 1) + 4 * ss + q) ^ n) << 4));
#pragma unroll
              for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][jj], b2[tt][ss][jj], g2, 0, 0, 0);
            }
            SK_STAMP(9 + 5 * t);                         // GEMM-2 issued
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              lds_f32* const yp = (lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(4 * q + rg, 16 * wid + n));
              const float zo = zreg[t][rg];
              const float stp = __fmul_rn(lr_, g2[rg]);                         // lr * grad
              const float zn = soft_threshold(__fsub_rn(*yp, stp), lam_);
              dsum += __builtin_fabsf(__fsub_rn(zo, zn));                         // |z - z_next|
              const float mom = __fmul_rn(coef, __fsub_rn(zn, zo));               // c (z_next - z)
              *yp = __fadd_rn(zn, mom);
              zreg[t][rg] = zn;
            }
          }
        }
      });
      SK_STAMP(26);
      if (leave) break;
      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();                      // y slices complete; red[] complete
      if ((p.partials || (STOP && p.stop_on)) && tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += red[w];
        if (p.partials) p.partials[((int64_t)it * p.part_stride) + (int64_t)round * nparts + grp * C + mem] = tsum;
        if (STOP && p.stop_on)
          __hip_atomic_store(p.stop_gran + (size_t)(it & (kStopRing - 1)) * nparts + grp * C + mem,
                             ((unsigned long long)(unsigned)(it + 1) << 32) | __float_as_uint(tsum),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (aborted) break;                                  // z_out untouched: the host re-runs the work
    if (STOP && p.stop_on && !stopped && blockIdx.x == 0 && wid == 0 && p.iters > 0) {
      // ran to maxiter: report the last iteration's global delta (does not change z)
      const unsigned want = (unsigned)p.iters;
      const unsigned long long* const lrow = p.stop_gran + (size_t)((p.iters - 1) & (kStopRing - 1)) * nparts;
      float partsum = 0.0f;
      int spins = 0;
      bool ok;
      do {
        ok = true;
        partsum = 0.0f;
        for (int e = lane; e < nparts; e += 64) {
          const unsigned long long gv = __hip_atomic_load(lrow + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(gv >> 32) == want);
          partsum += __uint_as_float((unsigned)gv);
        }
        ok = __all(ok);
        if (!ok) __builtin_amdgcn_s_sleep(8);
      } while (!ok && ++spins < kStopSpinLimit);
      const float total = wave_sum(partsum);
      if (lane == 0) {
        p.stop_out[0] = p.iters;
        p.stop_out[1] = __float_as_int(total);
        if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    static_for<T>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      const int row0 = (grp + p.groups * (T * round + t)) * kTileM;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = col0 + n;
        if (t < nt && (row0 + r) < p.n && cc < p.k) {
          p.z_out[(int64_t)(row0 + r) * p.ldz_out + cc] = zreg[t][rg];
          if (p.y_out)
            p.y_out[(int64_t)(row0 + r) * p.ldy_out + cc] =
                *(const lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(r, 16 * wid + n));
        }
      }
    });
    __builtin_amdgcn_s_barrier();                        // yt / red reuse by the next round
  }
  // rounds in which this group had no tile left: its slots of the per-iteration sums are zero
  if (p.partials && !aborted && tid < p.iters) {
    const int total_rounds = p.part_stride / nparts;
    int mine = 0;
    while (grp + p.groups * T * mine < p.ntiles) ++mine;
    for (int r = mine; r < total_rounds; ++r)
      for (int it = tid; it < p.iters; it += kFistaThreads)
        p.partials[(int64_t)it * p.part_stride + (int64_t)r * nparts + grp * C + mem] = 0.0f;
  }
}


// =================================================================================================
// Reduce-scatter form (K = 1024: C = 8 members = 8 waves = 8 column blocks of r; no in-kernel stop rule).
// In the kernel above every member pulls ALL C partials of a tile (128 KiB per tile and compute unit), in two
// dependent batches, before its GEMM-2 can start -- and a compute unit reads data another one has just written at
// ~110 GB/s: >1 us per tile that nothing hides.  Here the C partials of a tile are summed ONCE:
//   * member j reduces column block j (the 2 KiB blocks that wave j of every member produced), in the canonical
//     left-to-right order, and publishes the reduced block (wave pair 2t, 2t+1 of the member takes tile slot t, one
//     16-byte column half each: every wave has at most one such job per iteration, all jobs of an iteration run
//     side by side);
//   * every member then fetches the eight reduced blocks of a tile: 16 KiB instead of 128, two 16-byte loads per
//     lane, issued a tile ahead (8 registers) -- the fetch of tile t+1 flies under GEMM-2 of tile t.
// 36 KiB cross a compute unit's L2 port per tile instead of 128; the price is a second hop (flag2) per iteration,
// paid once for all T tiles.  Same sums in the same order: bitwise the same code as every other kernel.
template <int T>
__global__ __launch_bounds__(kFistaThreads, 2) void fista_splitk_rs_kernel(const FistaTileParams p) {
  const float lr_ = p.lr_dev ? p.lr_dev[0] : p.lr, lam_ = p.lr_dev ? p.lr_dev[1] : p.lam;
  constexpr int K = 1024;
  constexpr int C = K / kSlice;
  constexpr int D = kFistaD;
  constexpr int NW = kFistaWaves;
  constexpr int YT_BYTES = kTileM * kSlice * 4;     // 8 KiB per tile slot
  constexpr int RT_BYTES = kTileM * D * 4;          // 16 KiB
  static_assert(C == NW && (T == 1 || T == 2 || T == 4), "geometry");

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int mem = idx % C;
  const int grp = (idx / C) * 8 + xcd;
  if (grp >= p.groups) return;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  lds_char* const yt = (lds_char*)smem;                 // [T] y slices
  lds_char* const xt = yt + T * YT_BYTES;               // [T] -x in the GEMM-1 accumulator layout
  lds_char* const rt = xt + T * RT_BYTES;               // [2] residual tiles
  lds_f32* const red = (lds_f32*)(rt + 2 * RT_BYTES);   // [NW] delta sums, [NW+2] abort, [NW+3] one-XCD

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;

  f32x4 b1[4][2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        b1[s][ss][cb] = *reinterpret_cast<const f32x4*>(p.Wp + (size_t)(32 * wid + 16 * cb + n) * K +
                                                         kSlice * mem + 32 * s + 16 * ss + 4 * q);
  f32x4 b2[D / 32][2];
#pragma unroll
  for (int t = 0; t < D / 32; ++t)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
      b2[t][ss] = *reinterpret_cast<const f32x4*>(p.Wtp + (size_t)(kSlice * mem + 16 * wid + n) * D + 32 * t +
                                                   16 * ss + 4 * q);

  // exchange buffers: partials [parity][group][slot][member][wave][cb][lane] x 16 B, then the reduced blocks
  // [parity][group][slot][block][cb][lane] x 16 B; flags1 [group][member][wave][slot], flags2 [group][member][cb][slot]
  const unsigned part_bytes = (unsigned)(2 * p.groups * T * C * kPartBytes);
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, part_bytes + 2 * p.groups * T * kPartBytes, 0x00020000);
  const unsigned my_part = (unsigned)((grp * T * C + mem) * kPartBytes + wid * 2048 + lane * 16);   // + slot * C * kPartBytes
  const unsigned parity_stride = (unsigned)(p.groups * T * C * kPartBytes);
  const unsigned rparity_stride = (unsigned)(p.groups * T * kPartBytes);
  unsigned* const my_flag = p.xflags + ((size_t)(grp * C + mem) * NW + wid) * T;
  const int col0 = kSlice * mem + 16 * wid;
  // this wave's reduction job: tile slot jt, column half jcb of column block `mem`
  const int jt = wid >> 1, jcb = wid & 1;
  unsigned* const flags2 = p.xflags + (size_t)kSplitMaxParts * NW * 4 + 2 * (size_t)kSplitMaxParts;
  unsigned* const my_flag2 = flags2 + ((size_t)(grp * C + mem) * 2 + jcb) * T + jt;

  bool local;
  {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* const xid =
        reinterpret_cast<unsigned long long*>(p.xflags + (size_t)kSplitMaxParts * NW * 4) + (size_t)grp * C;
    if (wid == 0) {
      if (lane == 0)
        __hip_atomic_store(xid + mem, (1ull << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      bool ok, same;
      do {
        unsigned long long v = (1ull << 32) | xcc;
        if (lane < C) v = __hip_atomic_load(xid + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = __all((unsigned)(v >> 32) == 1u);
        same = __all((unsigned)v == xcc);
        if (!ok) {
          __builtin_amdgcn_s_sleep(2);
          if ((spins & 63) == 63 &&
              __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            break;
        }
      } while (!ok && ++spins < kStopSpinLimit);
      if (lane == 0) {
        red[NW + 3] = (ok && same) ? 1.0f : 0.0f;
        red[NW + 2] = ok ? 0.0f : 1.0f;
        if (!ok) __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    local = red[NW + 3] != 0.0f;
    if (red[NW + 2] != 0.0f) return;
    __builtin_amdgcn_s_barrier();
  }

  const __amdgpu_buffer_rsrc_t frsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.xflags, 0, (int)kSplitkFlagBytes, 0x00020000);
  const unsigned my_flag_off = (unsigned)((((grp * C + mem) * NW + wid) * T) * 4);
  const unsigned my_flag2_off = (unsigned)((kSplitMaxParts * NW * 4 + 2 * kSplitMaxParts + ((grp * C + mem) * 2 + jcb) * T + jt) * 4);
  // bounded poll: `want` in every lane that takes part (lanes with `mine` false pass)
  auto poll = [&](const unsigned* addr, bool mine, unsigned want, int limit) __attribute__((always_inline)) -> bool {
    int spins = 0;
    bool ok;
    do {
      unsigned v = want;
      if (mine) v = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = __all(v == want);
      if (!ok) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 63 && __hip_atomic_load(p.stop_out + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
          break;
      }
    } while (!ok && ++spins < limit);
    if (!ok && lane == 0) {           // a peer is not resident: the whole grid gives up
      __hip_atomic_store(p.stop_out + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      red[NW + 2] = 1.0f;
    }
    return ok;
  };

  unsigned epoch = 0;
  bool aborted = false;
  const int nparts = p.groups * C;
  for (int round = 0; grp + p.groups * T * round < p.ntiles; ++round) {
    int nt = 0;
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (grp + p.groups * (T * round + t) < p.ntiles) nt = t + 1;
    f32x4 zreg[T];
    static_for<T>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      const int row0 = (grp + p.groups * (T * round + t)) * kTileM;
      const float* ysrc = p.y_in ? p.y_in : p.z_in;
      const int64_t ldy = p.y_in ? p.ldy_in : p.ldz_in;
      // (through buffer descriptors of the tile's rows, offset out of range where there is nothing to read, offsets
      // opaque: as `in ? load : 0` every one of these 16 loads sat in a branch of its own with an s_waitcnt vmcnt(0)
      // behind it -- 16 to 24 memory round trips in a row at the head of every tile round; round 5)
      const int trows = (t < nt) ? min((int)kTileM, p.n - row0) : 0;
      const __amdgpu_buffer_rsrc_t zsrc = tile_rows_rsrc(p.z_in, p.ldz_in, row0, trows);
      const __amdgpu_buffer_rsrc_t ysc = tile_rows_rsrc(ysrc, ldy, row0, trows);
      const __amdgpu_buffer_rsrc_t xsc = tile_rows_rsrc(mem == 0 ? p.X : nullptr, p.ldx, row0, trows);
      float yv[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = col0 + n;
        zreg[t][rg] = tile_rows_load(zsrc, p.ldz_in, r, cc, cc < p.k);
        yv[rg] = tile_rows_load(ysc, ldy, r, cc, cc < p.k);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) *(lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(4 * q + rg, 16 * wid + n)) = yv[rg];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        f32x4 xn;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int r = 4 * q + rg, cc = 32 * wid + 16 * cb + n;
          const float v = tile_rows_load(xsc, p.ldx, r, cc, cc < p.d);
          xn[rg] = -v;
        }
        *(lds_f32x4*)(xt + t * RT_BYTES + wid * 2048 + cb * 1024 + lane * 16) = xn;
      }
    });
    if (tid == 0) red[NW + 2] = 0.0f;
    LASSO_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();

    for (int it = 0; it < p.iters; ++it) {
      const float coef = p.coef[it];
      ++epoch;
      const unsigned par_off = (epoch & 1u) * parity_stride;
      const unsigned rpar_off = part_bytes + (epoch & 1u) * rparity_stride + (unsigned)(grp * T * kPartBytes);

      // ============ GEMM-1 of every tile: p_mem = y[:, slice] W[:, slice]^T (- x), published ====
      SK_STAMP(0);
      int pending = -1;
      auto raise_flag = [&](int tp) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in the XCD's L2 / written through
        if (lane == 0) {
          if (local) __builtin_amdgcn_raw_buffer_store_b32(epoch, frsrc, my_flag_off + tp * 4, 0, 0);
          else __hip_atomic_store(my_flag + tp, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      static_for<T>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if (t < nt) {
          f32x4 acc[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[cb] = *(const lds_f32x4*)(xt + t * RT_BYTES + wid * 2048 + cb * 1024 + lane * 16);
          const lds_char* const yrow = yt + t * YT_BYTES + n * (kSlice * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            f32x4 a[2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
              a[ss] = *(const lds_f32x4*)(yrow + (s >> 1) * 256 + (((8 * (s & 1) + 4 * ss + q) ^ n) << 4));
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                  acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][jj], b1[s][ss][cb][jj], acc[cb], 0, 0, 0);
          }
          SK_STAMP(1 + t);
          if (pending >= 0) raise_flag(pending);
          const unsigned dst = my_part + par_off + t * (C * kPartBytes);
          if (local) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc[cb]), xrsrc, dst + cb * 1024, 0, 0);
          } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc[cb]), xrsrc, dst + cb * 1024, 0, 16);
          }
          pending = t;
        }
      });
      if (pending >= 0) raise_flag(pending);
      SK_STAMP(5);

      // ============ reduce: this wave's job -- column half jcb of block `mem` of tile slot jt ====================
      bool bad = false;
      if (jt < nt) {
        // wave `mem` of every member wrote the blocks (lane m polls member m's flag)
        const unsigned* const f1 = p.xflags + ((size_t)(grp * C + (lane & (C - 1))) * NW + mem) * T + jt;
        bad = !poll(f1, lane < C, epoch, kStopSpinLimit * 4);
        if (!bad) {
          const unsigned src = par_off + (unsigned)((grp * T + jt) * C * kPartBytes) + (unsigned)(mem * 2048 + jcb * 1024 + lane * 16);
          f32x4 part[C];
#pragma unroll
          for (int m = 0; m < C; ++m)
            part[m] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, src + m * kPartBytes, 0, 16));
          f32x4 rs = part[0];
#pragma unroll
          for (int m = 1; m < C; ++m)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) rs[rg] = __fadd_rn(rs[rg], part[m][rg]);      // r = ((p_0 + p_1) + p_2) + ...
          const unsigned dst = rpar_off + (unsigned)(jt * kPartBytes + mem * 2048 + jcb * 1024 + lane * 16);
          if (local) __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(rs), xrsrc, dst, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(rs), xrsrc, dst, 0, 16);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) {
            if (local) __builtin_amdgcn_raw_buffer_store_b32(epoch, frsrc, my_flag2_off, 0, 0);
            else __hip_atomic_store(my_flag2, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      SK_STAMP(6);
      // ============ the reduced blocks of every tile: block `wid` comes from member `wid` (lane = cb + 2 tile) =====
      if (!bad) {
        const unsigned* const f2 = flags2 + ((size_t)(grp * C + wid) * 2 + (lane & 1)) * T + (lane >> 1);
        bad = !poll(f2, lane < 2 * nt, epoch, kStopSpinLimit * 4);
      }
      auto fetch_r = [&](int t, f32x4 (&rb)[2]) __attribute__((always_inline)) {
        const unsigned src = rpar_off + (unsigned)(t * kPartBytes + wid * 2048 + lane * 16);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          rb[cb] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, src + cb * 1024, 0, 16));
      };
      f32x4 rnext[2];
      if (!bad) fetch_r(0, rnext);

      float dsum = 0.0f;
      bool leave = false;
      static_for<T>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        if (t < nt && !leave) {
          const f32x4 rsum[2] = {rnext[0], rnext[1]};
          if (t + 1 < T && t + 1 < nt && !bad) fetch_r(t + 1, rnext);      // flies under this tile's GEMM-2
          SK_STAMP(7 + 5 * t);
          lds_char* const rbuf = rt + (t & 1) * RT_BYTES;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              *(lds_f32*)(rbuf + tile_off<D>(4 * q + rg, 32 * wid + 16 * cb + n)) = rsum[cb][rg];
          LASSO_WAIT_LGKM0();
          __builtin_amdgcn_s_barrier();                  // r tile complete
          SK_STAMP(8 + 5 * t);
          if (red[NW + 2] != 0.0f) {
            aborted = true;
            leave = true;
          } else {
            f32x4 g2 = {0.f, 0.f, 0.f, 0.f};
            const lds_char* const rrow = rbuf + n * (D * 4);
#pragma unroll
            for (int tt = 0; tt < D / 32; ++tt) {
              f32x4 a[2];
#pragma unroll
              for (int ss = 0; ss < 2; ++ss)
                a[ss] = *(const lds_f32x4*)(rrow + (tt >> 1) * 256 + (((8 * (tt & 1) + 4 * ss + q) ^ n) << 4));
#pragma unroll
              for (int ss = 0; ss < 2; ++ss)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ss][jj], b2[tt][ss][jj], g2, 0, 0, 0);
            }
            SK_STAMP(9 + 5 * t);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              lds_f32* const yp = (lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(4 * q + rg, 16 * wid + n));
              const float zo = zreg[t][rg];
              const float stp = __fmul_rn(lr_, g2[rg]);                         // lr * grad
              const float zn = soft_threshold(__fsub_rn(*yp, stp), lam_);
              dsum += __builtin_fabsf(__fsub_rn(zo, zn));                         // |z - z_next|
              const float mom = __fmul_rn(coef, __fsub_rn(zn, zo));               // c (z_next - z)
              *yp = __fadd_rn(zn, mom);
              zreg[t][rg] = zn;
            }
          }
        }
      });
      SK_STAMP(26);
      if (leave) break;
      dsum = wave_sum(dsum);
      if (lane == 0) red[wid] = dsum;
      LASSO_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();                      // y slices complete; red[] complete
      if (p.partials && tid == 0) {
        float tsum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tsum += red[w];
        p.partials[((int64_t)it * p.part_stride) + (int64_t)round * nparts + grp * C + mem] = tsum;
      }
    }
    if (aborted) break;

    static_for<T>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      const int row0 = (grp + p.groups * (T * round + t)) * kTileM;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int r = 4 * q + rg, cc = col0 + n;
        if (t < nt && (row0 + r) < p.n && cc < p.k) {
          p.z_out[(int64_t)(row0 + r) * p.ldz_out + cc] = zreg[t][rg];
          if (p.y_out)
            p.y_out[(int64_t)(row0 + r) * p.ldy_out + cc] =
                *(const lds_f32*)(yt + t * YT_BYTES + tile_off<kSlice>(r, 16 * wid + n));
        }
      }
    });
    __builtin_amdgcn_s_barrier();
  }
  if (p.partials && !aborted && tid < p.iters) {
    const int total_rounds = p.part_stride / nparts;
    int mine = 0;
    while (grp + p.groups * T * mine < p.ntiles) ++mine;
    for (int r = mine; r < total_rounds; ++r)
      for (int it = tid; it < p.iters; it += kFistaThreads)
        p.partials[(int64_t)it * p.part_stride + (int64_t)r * nparts + grp * C + mem] = 0.0f;
  }
}

template <int T>
static hipError_t launch_rs(const FistaTileParams& p, hipStream_t stream) {
  constexpr int C = 8;
  const size_t lds = lds_bytes(T);
  const void* fn = reinterpret_cast<const void*>(&fista_splitk_rs_kernel<T>);
  if (lds > 64 * 1024)
    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  const int grid = (p.groups + 7) / 8 * 8 * C;
  hipLaunchKernelGGL((fista_splitk_rs_kernel<T>), dim3(grid), dim3(kFistaThreads), lds, stream, p);
  return hipGetLastError();
}

template <int K, int T, bool STOP>
static hipError_t launch_kts(const FistaTileParams& p, hipStream_t stream) {
  constexpr int C = K / kSlice;
  const size_t lds = lds_bytes(T);
  const void* fn = reinterpret_cast<const void*>(&fista_splitk_kernel<K, T, STOP>);
  if (lds > 64 * 1024)
    if (hipError_t e = ensure_dynamic_lds(fn, lds); e != hipSuccess) return e;
  const int grid = (p.groups + 7) / 8 * 8 * C;
  hipLaunchKernelGGL((fista_splitk_kernel<K, T, STOP>), dim3(grid), dim3(kFistaThreads), lds, stream, p);
  return hipGetLastError();
}

template <int K, int T>
static hipError_t launch_kt(const FistaTileParams& p, hipStream_t stream) {
  if constexpr (K == 1024)
    if (!p.stop_on && ((T >= 2 && p.variant == 0) || p.variant == 2)) return launch_rs<T>(p, stream);   // reduce-scatter form
  return p.stop_on ? launch_kts<K, T, true>(p, stream) : launch_kts<K, T, false>(p, stream);
}

template <int K>
static hipError_t launch_k(const FistaTileParams& p, int tiles, hipStream_t stream) {
  switch (tiles) {
    case 1: return launch_kt<K, 1>(p, stream);
    case 2: return launch_kt<K, 2>(p, stream);
    case 4: return launch_kt<K, 4>(p, stream);
  }
  return hipErrorInvalidValue;
}

template <int K>
static hipError_t occupancy_k(int* blocks_per_cu) {
  const void* fn = reinterpret_cast<const void*>(&fista_splitk_kernel<K, 1, true>);
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, kFistaThreads, lds_bytes(1));
}

}  // namespace splitk

int fista_splitk_members(int kpad) { return kpad / splitk::kSlice; }

// payload bytes for `groups` groups working on `tiles` tiles at once
size_t fista_splitk_exchange_bytes(int kpad, int groups, int tiles) {
  // partials of every member (two parities) + the reduced blocks of the reduce-scatter form
  return (size_t)2 * groups * tiles * (kpad / splitk::kSlice) * splitk::kPartBytes + (size_t)2 * groups * tiles * splitk::kPartBytes;
}

hipError_t fista_splitk_occupancy(int kpad, int* blocks_per_cu) {
  switch (kpad) {
    case 256: return splitk::occupancy_k<256>(blocks_per_cu);
    case 512: return splitk::occupancy_k<512>(blocks_per_cu);
    case 1024: return splitk::occupancy_k<1024>(blocks_per_cu);
  }
  return hipErrorInvalidValue;
}

// `tiles` (1, 2 or 4): tiles a group works on at once
hipError_t launch_fista_splitk(const FistaTileParams& p, int kpad, int tiles, hipStream_t stream) {
  switch (kpad) {
    case 256: return splitk::launch_k<256>(p, tiles, stream);
    case 512: return splitk::launch_k<512>(p, tiles, stream);
    case 1024: return splitk::launch_k<1024>(p, tiles, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace lasso
