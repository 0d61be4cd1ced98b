// Bound GEMM + filter of a gated_i8 index with PERSISTENT workgroups (round 4; DHR_PARAM_GEMM_VARIANT = 6).  Same tile, same arithmetic and
// the same operand images as gemm_g8.hip; what changes is everything around the stage loops, because a per-tile timeline of that kernel
// (tools/g8_trace.py, -DG8_TRACE=1; profiles/r04_gemm_timeline.txt) showed where a 256 x 256 tile's ~58 k cycles go with an open filter: 24.6 k of
// matrix instructions, and
//   * 2.0 k between a workgroup's exit and its successor's start on the CU (3.7 k with an open filter), 1.9-2.1 k of prologue (the first
//     stage pair's round trip to L2) -- 160 KiB of LDS and 2 x 248 registers per SIMD allow ONE workgroup per CU, so nothing hides either;
//   * 13.0 k of filter epilogue (3.0 k with a closed filter): private hit stacks that fill the whole staging ring, two list reservations
//     (returning global atomics) per lane and their round trips, per-lane flush loops;
//   * ~4 k of waiting at the gated pairs' barriers for LDS-DMA pieces issued one block (~500 cycles) earlier -- less than an L2 round trip.
// Here a workgroup takes tile after tile from per-XCD counters, and
//   1. the LDS-DMA stream never stops: the ring is free from a tile's LAST pair barrier on (its last fragments are in registers by then), so
//      pair 0 / pair 1 (+ the constants) of the NEXT tile are simply virtual pairs npairs / npairs + 1 of the current one.  They land during
//      the epilogue; the next tile's "prologue" is one barrier;
//   2. every piece of a pair is issued in the block right behind the pair barrier (a whole block of slack before the next barrier);
//   3. the epilogue leaves the ring alone: a hit goes to the thread's private stack BEHIND the ring (12 four-byte entries: the local row + the
//      sum rounded up to a multiple of 256; the branch-free push of gemm_g8.hip) and the stacks are flushed to the queries' lists INSIDE THE
//      NEXT TILE'S STAGE LOOPS: the list reservations (returning atomics) behind its first pair barrier, the stores behind the second -- both
//      round trips run under matrix work.  (A wave-private queue filled with v_cmp -> s_bcnt1 / v_mbcnt ranks was tried first: ~100 cycles per
//      pushed element, the scan went from 13 k to 22 k cycles.  Stacks of 5 eight-byte entries overflow in 33 % of the (wave, tile) scans of a
//      real search, 12 in 3 %: tools/g8p_stat.py.  A thread with more hits sends its surplus straight to the lists in a second, cold scan.)
//   4. tile indices come from the counter two tiles ahead (one returning atomic per tile by one lane, published through the LDS).
// The XCD a workgroup runs on is read from HW_REG_XCC_ID and is an AFFINITY: XCD v's sweep is the corpus tile groups g = v (mod 8) against all
// query tiles in the order of gemm_wg_tile (dhr_internal.h), so the L2 behaviour is that of the 3-D grid -- and a workgroup whose own sweep is
// used up goes on with the other XCDs', so every tile is computed whatever the placement (a 3-workgroup launch need not touch XCD 0).
// Results do not depend on which workgroup computes which tile: the lists are unordered sets.
// MEASURED: ~10 % fewer cycles per tile and the SAME time as gemm_g8.hip (closed filter 18.7 / 18.8 ms per 2 M rows, open 21.1 / 21.2; bench step
// 124.6 / 126.7 ms): the launch runs at the 1 400 W package cap with either kernel and the clock is what the power budget leaves (DESIGN.md
// section 4b).  Not the default; kept because it is the measurement that shows the bound.
#include "gemm_g8.h"
#include <atomic>
#include <mutex>
#include <type_traits>

#ifndef G8_TRACE
#define G8_TRACE 0
#endif
namespace dhr {

constexpr int G8P_DEPTH = 12;                             // private hit-stack slots per thread (+ one slot that takes the pushes of a full stack); over a
                                                          // search of 2 M rows 3 % of the (wave, tile) scans hold a lane with more hits (33 % with 5)
constexpr int G8P_QUEUE = G8_META + 4096;                 // the stacks: slot j of thread t = the dword at G8P_QUEUE + (j * 512 + t) * 4, behind the ring
constexpr int G8P_MAIL = G8P_QUEUE + (G8P_DEPTH + 1) * G8_NT * 4;    // one word: the tile index published for the tile after next
constexpr int G8P_LDS = G8P_MAIL + 64;
static_assert(G8P_LDS <= 163840, "LDS of a CU");
constexpr int G8P_CTR_SLOTS = 4096;                       // launches in flight share nothing: each takes the next slot of 8 counters
__device__ unsigned int g8p_ctr[G8P_CTR_SLOTS * 8];

#if G8_TRACE
constexpr int G8P_TRACE_SLOTS = 1 << 18;
__device__ unsigned long long g8p_trace_buf[G8P_TRACE_SLOTS * 8];
#define G8P_T(i) do { if (threadIdx.x == 0 && tslot < (unsigned)G8P_TRACE_SLOTS) g8p_trace_buf[(size_t)tslot * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long g8p_stat[16];        // [0] wave-tiles scanned, [1 + i] of them with a lane that holds more than G8P_STAT_D[i] hits, [10] hits
__device__ constexpr int G8P_STAT_D[8] = {3, 5, 6, 8, 10, 13, 16, 32};
#else
#define G8P_T(i) do { } while (0)
#endif

// Everything that is needed once per tile (the tile map's parameters, the list pointers, the constants' arrays) is read from the kernel
// argument segment WHERE it is used, through a pointer the compiler cannot see through: as plain `p.field` uses all of it is loaded at
// entry and kept in ~60 scalar registers across the stage loops, which then spill.
typedef const __attribute__((address_space(4))) GemmArgs* KArgs;
__device__ __forceinline__ KArgs g8p_kargs() {
  KArgs k = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}

// One query column (NI_) of the lane's accumulators against its threshold (two levels: the maximum of a group of four, then the four
// elements -- gemm_g8.hip).  A hit pushes (local row, integer sum) onto the thread's private stack; a full stack (G8P_DEPTH hits of one
// thread in one tile: hot queries only) just counts on, and its surplus goes straight to the query's list in a second, cold scan that only
// waves with such a lane run (SURPLUS).  A stack entry is ONE dword: the local row in the low byte, the sum rounded UP to a multiple of 256
// above it (the list carries an upper bound of the bound: 256 units are ~1e-5 of a candidate's sum) -- twice the depth in the same LDS.
__device__ __forceinline__ uint32_t g8p_entry(int v, int rl) { return ((uint32_t)(v + 255) & 0xffffff00u) | (uint32_t)rl; }
template <bool SURPLUS, int NI_, bool CHECK_ROWS>
__device__ __forceinline__ void g8p_scan_half(KArgs kp, floatx16 (&acc)[4][2], const int rbase, const int rows_valid, const uint32_t row0, const int q,
                                              uint32_t* stack, const int t, const float mul, uint32_t& j) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const floatx16& a = acc[mi][NI_];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gm = max(max(__float_as_int(a[4 * g]), __float_as_int(a[4 * g + 1])), max(__float_as_int(a[4 * g + 2]), __float_as_int(a[4 * g + 3])));
      if (gm >= t) {
        asm volatile("");
        if constexpr (!SURPLUS && !CHECK_ROWS) {
          // full tile, stack pass: BRANCH-FREE pushes -- every lane of the active group writes all four (row, sum) pairs to its current slot
          // and advances only on a hit; a miss is overwritten by the lane's next write
#pragma unroll
          for (int e = 4 * g; e < 4 * g + 4; ++e) {
            const int v = __float_as_int(a[e]);
            const int rl = rbase + mi * 32 + (e & 3) + 8 * (e >> 2);
            const uint32_t slot = j < (uint32_t)G8P_DEPTH ? j : (uint32_t)G8P_DEPTH;
            stack[slot * G8_NT] = g8p_entry(v, rl);
            j += v >= t ? 1u : 0u;
          }
        } else
#pragma unroll
        for (int e = 4 * g; e < 4 * g + 4; ++e) {
          const int v = __float_as_int(a[e]);
          const int rl = rbase + mi * 32 + (e & 3) + 8 * (e >> 2);
          if (v >= t && (!CHECK_ROWS || rl < rows_valid)) {
            asm volatile("");
            if constexpr (SURPLUS) {
              if (j >= (uint32_t)G8P_DEPTH) {
                const uint32_t cap = kp->cap;
                const uint32_t slot = atomicAdd(kp->cnt + q, 1u);
                const uint2 ev = make_uint2(row0 + (uint32_t)rl, __float_as_uint(g8_score(v, mul)));
                if (slot < cap) kp->cand[(int64_t)q * cap + slot] = ev;
                else if (const ListTier* lt = kp->tier) { const uint32_t o = slot - cap; if (o < lt->ovf_cap[q]) lt->ovf[(size_t)lt->ovf_off[q] + o] = ev; }      // second tier (GemmArgs::tier)
              }
            } else if (j < (uint32_t)G8P_DEPTH) stack[j * G8_NT] = g8p_entry(v, rl);
            ++j;
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(G8_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_filter_g8p_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ts = p.ts, td = p.td;
  const int nsp = ts >> 1, npairs = (ts + td) >> 1;      // the launcher selects this kernel only when ts and td are even and ts >= 4
  const int nq = p.n_qtiles;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const uint32_t per_group = (uint32_t)(DOC_GROUP * nq);
  uint32_t fv = xcc;                              // the XCD whose sweep this workgroup is taking tiles from: its own, then -- when that is used up -- the next ones

  // linear index L of XCD v's sweep -> tile: group z of the XCD = group 8 z + v of the launch; inside a group the query tile moves slowest
  // over DOC_GROUP corpus tiles (the launch's last group may hold fewer).  The first index that is not a tile means the sweep is used up.
  auto decode = [&](uint32_t L, uint32_t v, int64_t& dt_o, int& qt_o, bool& ok_o, bool& dead_o) __attribute__((always_inline)) {
    int64_t dt = 0;
    int qt = 0, ok = 0, dead = 0;
    const int64_t groups_v = p.p_groups > (int64_t)v ? (p.p_groups - (int64_t)v + 7) >> 3 : 0;
    if ((int64_t)L < groups_v * (int64_t)per_group) {          // < 2^31 (launcher)
      const uint32_t z = L / per_group, r = L - z * per_group;
      const int64_t grp = (int64_t)z * 8 + (int64_t)v;
      KArgs k = g8p_kargs();
      const int64_t left = (k->seq_hi - k->seq_lo) - grp * DOC_GROUP;
      uint32_t dl;
      if (left >= DOC_GROUP) { qt = (int)(r / DOC_GROUP); dl = r % DOC_GROUP; }
      else { const uint32_t nd = (uint32_t)left; qt = (int)(r / nd); dl = r - (uint32_t)qt * nd; }
      if (qt < nq) {
        const int64_t seq = k->seq_lo + grp * DOC_GROUP + (int64_t)dl;
        dt = seq_to_tile_fast(seq, k->map_mode, k->period, k->head, k->perm_mul, k->perm_n, k->inv_perm_n, k->inv_pm1);
        if (dt >= k->n_tiles) { dt = k->n_tiles - 1; dead = 1; }      // cannot happen for the callers' sequences; computed with a closed filter
        ok = 1;
      }
    }
    // (the divisions and the double-precision tile map run on the vector ALU: without the readfirstlanes everything derived from a tile --
    // the DMA descriptors, the piece counts -- stays in vector registers and every LDS-DMA piece is issued from a waterfall loop)
    dt_o = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)dt >> 32)) << 32) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)dt));
    qt_o = __builtin_amdgcn_readfirstlane(qt);
    ok_o = __builtin_amdgcn_readfirstlane(ok) != 0;
    dead_o = __builtin_amdgcn_readfirstlane(dead) != 0;
  };
  // Which workgroup runs on which XCD is the hardware's business (a launch of three workgroups need not touch XCD 0 at all), so the XCD is an
  // affinity, not a partition: a workgroup whose own sweep is used up goes on with the other XCDs' sweeps, one after the other -- every tile
  // of the launch is taken by somebody whatever the placement.  Taking a tile from a new sweep is synchronous (one atomic round trip; it
  // happens at most eight times in a workgroup's life).
#define G8P_ACQUIRE_SYNC(dt_, qt_, ok_, dead_)                                                                                        \
  do {                                                                                                                               \
    for (int tries_ = 0; !(ok_) && tries_ < 8; ++tries_) {                                                                          \
      __syncthreads();                                                                                                               \
      if (threadIdx.x == 0)                                                                                                          \
        *(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL) = atomicAdd(g8p_kargs()->p_ctr + fv, 1u);   \
      __syncthreads();                                                                                                               \
      const uint32_t La_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL)); \
      decode(La_, fv, dt_, qt_, ok_, dead_);                                                                                         \
      if (!(ok_)) fv = (fv + 1u) & 7u;                                                                                               \
    }                                                                                                                                \
  } while (0)

  // ---- first two tiles of this workgroup
  int64_t cur_dt = 0, nxt_dt = 0;
  int cur_qt = 0, nxt_qt = 0;
  bool cur_ok = false, nxt_ok = false, cur_dead = false, nxt_dead = false;
  G8P_ACQUIRE_SYNC(cur_dt, cur_qt, cur_ok, cur_dead);
  if (!cur_ok) return;
  G8P_ACQUIRE_SYNC(nxt_dt, nxt_qt, nxt_ok, nxt_dead);
#if G8_TRACE
  uint32_t wg_tile = 0;                           // trace slot of a tile: 1024 x the workgroup + its tile count
#endif

  // ---- LDS-DMA, fixed roles: wave w streams half (w & 1) of image ((w >> 1) & 1 ? query : corpus) of stage 2g + (w >> 2) of every pair g
  const bool dma_b = ((wave >> 1) & 1) != 0;
  const int dma_s = wave >> 2;
  const int dma_h = wave & 1;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t smem_u = (uint32_t)(uintptr_t)LDS_PTR(smem);
  auto role_base = [&](int64_t t_dt, int t_qt) __attribute__((always_inline)) -> const char* {
    KArgs k = g8p_kargs();
    return dma_b ? (const char*)k->b_tiles + (int64_t)t_qt * ((int64_t)ts * SP_STAGE_B + (int64_t)td * SP_DENSE)
                 : (const char*)k->a_tiles + t_dt * ((int64_t)ts * S8_STAGE_A + (int64_t)td * SP_DENSE);
  };
  const char* base_cur = role_base(cur_dt, cur_qt);
  const char* base_nxt = role_base(nxt_dt, nxt_qt);
  int soff = 0;                                   // ring slot of the current tile's stage 0 (the stream's slots run on across tiles)
  int dma_soff = 0, dma_n = 0, nx_soff = 0, nx_n = 0;
  uint32_t dma_lds = 0, nx_lds = 0;
  const char* nx_base = base_cur;
  bool nx_consts = false;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base_cur, (short)0, 0x7fffffff, 0x00020000);
#define G8P_ISSUE_CONSTS(t_dt, t_qt)                                                                                                     \
  do {                                                                                                                                 \
    if (wave < 4) {                                                                                                                    \
      KArgs kc_ = g8p_kargs();                                                                                                         \
      const char* src_ = wave == 0 ? (const char*)(kc_->g8_rsum + (t_dt) * TILE_ROWS) : wave == 1 ? (const char*)(kc_->i8_mul + (t_qt) * TILE_ROWS) \
                       : wave == 2 ? (const char*)(kc_->thr + (t_qt) * TILE_ROWS) : (const char*)(kc_->g8_shift + (t_qt) * TILE_ROWS);  \
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src_ + lane_off), LDS_PTR(smem + G8_META + wave * 1024), 16, 0, 0);                  \
    }                                                                                                                                  \
  } while (0)
  // virtual pair g of the current tile's stream: g >= npairs is pair g - npairs of the NEXT tile.  Prepared in ascending order, one pair per
  // call (the scalar work sits in the stage loops): the source offset advances by two stages, the LDS address toggles between the role's
  // two ring slots; only the first ungated pair and the first pair of the next tile are set up from scratch.
  const int st_sp = dma_b ? SP_STAGE_B : S8_STAGE_A;                 // this role's bytes per gated stage
  const int half_sp = dma_h * (dma_b ? 8192 : S8_STAGE_A / 2), half_dn = dma_h * 8192;
  const int n_sp = dma_b ? 8 : 5;
  const uint32_t lds_role = smem_u + (uint32_t)(dma_b ? G8_QOFF : 0);
#define G8P_DMA_PREPARE(g_)                                                                                                            \
  do {                                                                                                                                 \
    const int gp_ = (g_);                                                                                                              \
    if (gp_ == npairs) {                                   /* pair 0 of the next tile */                                                \
      nx_soff = dma_s * st_sp + half_sp;                                                                                               \
      nx_n = nxt_ok ? n_sp : 0;                                                                                                        \
      nx_base = base_nxt;                                                                                                              \
      nx_consts = nxt_ok;                                                                                                              \
    } else if (gp_ == nsp) {                               /* first ungated pair */                                                     \
      nx_soff = ts * st_sp + dma_s * SP_DENSE + half_dn;                                                                               \
      nx_n = 8;                                                                                                                        \
      nx_consts = false;                                                                                                               \
    } else {                                                                                                                           \
      nx_soff += gp_ < nsp || gp_ > npairs ? 2 * st_sp : 2 * SP_DENSE;                                                                 \
      nx_consts = false;                                                                                                               \
    }                                                                                                                                  \
    nx_lds = lds_role + (uint32_t)(((soff + 2 * gp_ + dma_s) & 3) * G8_SLOT + ((gp_ < nsp || gp_ >= npairs) ? half_sp : half_dn));     \
  } while (0)
  auto dma_commit = [&]() __attribute__((always_inline)) {
    dma_soff = nx_soff; dma_lds = nx_lds; dma_n = nx_n;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)nx_base, (short)0, 0x7fffffff, 0x00020000);
    if (nx_consts) G8P_ISSUE_CONSTS(nxt_dt, nxt_qt);
  };
  auto dma_piece = [&](int j) __attribute__((always_inline)) {
    if (j < dma_n) {
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(uintptr_t)(dma_lds + (uint32_t)(j >> 2) * 4096u);
      const int so = dma_soff + (j >> 2) * 4096;
      switch (j & 3) {      // the immediate must be a literal
        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)lane_off, so, 0, 0); break;
        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)lane_off, so, 1024, 0); break;
        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)lane_off, so, 2048, 0); break;
        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)lane_off, so, 3072, 0); break;
      }
    }
  };

  // ---- flush of a tile's hit stacks, inside the NEXT tile's stage loops: behind its first pair barrier every lane reserves its hits in
  // its two queries' lists (returning atomics), behind the second one it stores them -- both round trips run under matrix instructions
  uint32_t* const stack = (uint32_t*)(smem + G8P_QUEUE) + threadIdx.x;
  int fl_stage = 0;                               // 0 nothing, 1 reservations to issue, 2 stores to issue (wave-uniform)
  bool mail_pending = false;
  uint32_t fl_row0 = 0, fl_s = 0;                 // fl_s = hits of the lane's first query column | hits of both << 8
  int fl_q = 0;                                   // the lane's first query (global index; the second is 32 further)
  uint32_t fl_base0 = 0, fl_base1 = 0, pend = 0;
  float fl_mul0 = 0.f, fl_mul1 = 0.f;
  auto flush_step = [&]() __attribute__((always_inline)) {
    if (mail_pending) {                       // the tile index asked for at the top of this tile has arrived (every pair barrier waits for vmcnt(0))
      if (threadIdx.x == 0) *(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL) = pend;
      mail_pending = false;
    }
    if (fl_stage == 2) {
      KArgs k = g8p_kargs();
      const uint32_t cap = k->cap;
      uint2* const cand = k->cand;
      const uint32_t s0 = fl_s & 255u, s1 = fl_s >> 8;
      // four entries per round: their LDS reads go out together, one wait, then the stores (few lanes hold more than four hits)
      for (uint32_t i0 = 0; __builtin_amdgcn_ballot_w64(i0 < s1) != 0; i0 += 4) {
        uint32_t en[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = stack[(i0 + u < (uint32_t)G8P_DEPTH ? i0 + u : (uint32_t)G8P_DEPTH) * G8_NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t i = i0 + u;
          if (i < s1) {
            const bool first = i < s0;
            const uint32_t slot = first ? fl_base0 + i : fl_base1 + (i - s0);
            const int q = first ? fl_q : fl_q + 32;
            const uint2 ev = make_uint2(fl_row0 + (en[u] & 255u), __float_as_uint(g8_score((int)(en[u] & 0xffffff00u), first ? fl_mul0 : fl_mul1)));
            if (slot < cap) cand[(int64_t)q * cap + slot] = ev;
            else if (const ListTier* lt = k->tier) { const uint32_t o = slot - cap; if (o < lt->ovf_cap[q]) lt->ovf[(size_t)lt->ovf_off[q] + o] = ev; }      // second tier (GemmArgs::tier), cold
          }
        }
      }
      fl_stage = 0;
    } else if (fl_stage == 1) {
      KArgs k = g8p_kargs();
      uint32_t* const cnt = k->cnt;
      const uint32_t s0 = fl_s & 255u, s1 = fl_s >> 8;
      if (s0 > 0) fl_base0 = atomicAdd(cnt + fl_q, s0);
      if (s1 > s0) fl_base1 = atomicAdd(cnt + fl_q + 32, s1 - s0);
      fl_stage = 2;
    }
  };
#define G8P_FLUSH_ALL()                                                  \
  while (fl_stage != 0) {                                                \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
    flush_step();                                                        \
  }

  // ---- the first tile's operands: the only DMA issue outside a stage loop
  G8P_ISSUE_CONSTS(cur_dt, cur_qt);
  nx_soff = dma_s * st_sp + half_sp - 2 * st_sp; nx_n = n_sp; nx_base = base_cur;     // so that dma_prepare(0) advances onto pair 0
  G8P_DMA_PREPARE(0); dma_commit();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(j);
  G8P_DMA_PREPARE(1); dma_commit();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(j);
  dma_n = 0;


  // One 256 x 256 tile.  PARTIAL: the batch's LAST query tile when at most 128 of its 256 queries are real; the waves are then numbered so
  // that the wave columns holding real queries sit on different SIMDs, the others only stream their share of the LDS-DMA (gemm_g8.hip).
  for (;;) {
    const bool PARTIAL = p.partial_wn > 0 && cur_qt == nq - 1;
#if G8_TRACE
    const unsigned tslot = wg_tile < 1024u && blockIdx.x < (unsigned)(G8P_TRACE_SLOTS / 1024) ? blockIdx.x * 1024u + wg_tile : (unsigned)G8P_TRACE_SLOTS;
    ++wg_tile;
    if (threadIdx.x == 0 && tslot < (unsigned)G8P_TRACE_SLOTS) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      g8p_trace_buf[(size_t)tslot * 8 + 0] = __builtin_amdgcn_s_memrealtime();
      g8p_trace_buf[(size_t)tslot * 8 + 7] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32) | ((unsigned long long)cur_qt << 40) | ((unsigned long long)(cur_dt & 0xffff) << 48);
    }
    G8P_T(1);
#endif
    const int wm = PARTIAL ? (wave & 1) : (wave >> 2), wn = PARTIAL ? (wave >> 1) : (wave & 3);
    const bool active = !PARTIAL || wn < p.partial_wn;
    const int64_t dt = cur_dt;
    const int qt = cur_qt;
    // per-lane LDS offsets inside a ring slot
    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    const int swz4 = (frow >> 2) & 3;
    const int c0 = (fhalf ^ swz4) << 4, c1 = ((2 + fhalf) ^ swz4) << 4;
    const int a8_off = ((wm * 8 + fhalf) * 32 + frow) * 16;                     // gated corpus values, + mi * 1024
    const int p8_off = S8_A_BYTES + ((wm * 2 + fhalf) * 32 + frow) * 16;        // the lane's four position words
    const int ad_row = (wm * 128 + frow) * 64;                                  // ungated corpus rows, + mi * 2048
    const int q_row = G8_QOFF + (wn * 64 + frow) * 64;                          // query rows (both kinds), + ni * 2048
#define G8P_SLOT_OF(u) (smem + ((soff + (u)) & 3) * G8_SLOT)

    floatx16 acc[4][2];
    // fragment read number g of a gated stage, in the order the matrix instructions need them: position words + first query column (two
    // halves), the four corpus blocks, the second query column (needed from the block's fifth instruction on)
    auto read_s8 = [&](G8Frag& f, const char* sl, int g) __attribute__((always_inline)) {
      if (g == 0) f.pw = *(const intx4*)(sl + p8_off);
      if (g < 2) f.b[0].h[g] = *(const intx4*)(sl + q_row + (g ? c1 : c0));
      else if (g < 6) f.a[g - 2] = *(const intx4*)(sl + a8_off + (g - 2) * 1024);
      else if (g < 8) f.b[1].h[g - 6] = *(const intx4*)(sl + q_row + 2048 + ((g - 6) ? c1 : c0));
    };
    // ... of ungated block t (stage t >> 1 of the ungated part, 32-column half t & 1)
    auto read_dn = [&](G8Frag& f, const char* sl, int cc, int g) __attribute__((always_inline)) {
      if (g == 0) f.b[0].h[0] = *(const intx4*)(sl + q_row + cc);
      else if (g < 5) f.a[g - 1] = *(const intx4*)(sl + ad_row + (g - 1) * 2048 + cc);
      else if (g == 5) f.b[1].h[0] = *(const intx4*)(sl + q_row + 2048 + cc);
    };
    // gated block: computes the stage `fc` came from; reads the fragments of gated stage un into fn; DMA: one piece of the pair just
    // committed behind every matrix instruction.  (Macros, not lambdas: closures that refer to other closures are not always promoted to
    // registers in a function of this size -- the whole persistent state then lives in scratch memory.)
#define G8P_BLK_S8(fc, fn, un, LOAD, DMA)                                                                                              \
    do {                                                                                                                               \
      const char* sl_ = G8P_SLOT_OF(un);                                                                                               \
      if (active) {                                                                                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                                             \
          const int ni_ = g_ >> 2, mi_ = g_ & 3;                                                                                       \
          g8_smfmac(acc[mi_][ni_], fc.a[mi_], fc.b[ni_].v, (uint32_t)fc.pw[mi_]);                                                      \
          if (LOAD) read_s8(fn, sl_, g_);                                                                                              \
          if (DMA) dma_piece(g_);                                                                                                      \
          __builtin_amdgcn_sched_barrier(0);      /* keep the reads / pieces BETWEEN the matrix instructions (the asm is not volatile) */ \
        }                                                                                                                              \
      } else if (DMA) {                                                                                                                \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) dma_piece(g_);                                                                \
      }                                                                                                                                \
    } while (0)
#define G8P_BLK_DN(fc, fn, tn, LOAD, DMA)                                                                                              \
    do {                                                                                                                               \
      const char* sl_ = G8P_SLOT_OF(ts + ((tn) >> 1));                                                                                 \
      const int cc_ = ((tn) & 1) ? c1 : c0;                                                                                            \
      if (active) {                                                                                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                                             \
          const int ni_ = g_ >> 2, mi_ = g_ & 3;                                                                                       \
          g8_mfma(acc[mi_][ni_], fc.a[mi_], fc.b[ni_].h[0]);                                                                           \
          if (LOAD) read_dn(fn, sl_, cc_, g_);                                                                                         \
          if (DMA) dma_piece(g_);                                                                                                      \
          __builtin_amdgcn_sched_barrier(0);                                                                                           \
        }                                                                                                                              \
      } else if (DMA) {                                                                                                                \
        _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) dma_piece(g_);                                                                \
      }                                                                                                                                \
    } while (0)
    // pair barrier: this wave's pieces have landed, every wave is done with the pair's slots; then the next virtual pair is committed and
    // the flush of the previous tile's hits moves one step
#define G8P_PAIR_SYNC(g)                                                                                                               \
    do {                                                                                                                               \
      G8P_DMA_PREPARE((g) + 2);                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);          /* or the compiler sinks the block above below the wait */                            \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                                      \
      __builtin_amdgcn_s_barrier();                                                                                                    \
      dma_commit();                                                                                                                    \
      flush_step();                                                                                                                    \
    } while (0)

    // ---- the tile's first pair and constants are in the LDS (issued by the previous tile's last pairs, or above for the first tile)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    G8P_T(2);
    if (threadIdx.x == 0) pend = atomicAdd(g8p_kargs()->p_ctr + fv, 1u);      // the tile after next (from the sweep in use)
    mail_pending = true;
    {
      const int32_t* rs = (const int32_t*)(smem + G8_META) + wm * 128 + 4 * (lane >> 5);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const intx4 v = *(const intx4*)(rs + mi * 32 + 8 * g4);
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) acc[mi][0][4 * g4 + i4] = acc[mi][1][4 * g4 + i4] = __int_as_float(v[i4]);
        }
    }
    int sh_r[2];
    float mul_r[2], thr_f[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int ql = wn * 64 + ni * 32 + (lane & 31);
      mul_r[ni] = ((const float*)(smem + G8_META + 1024))[ql];
      thr_f[ni] = ((const float*)(smem + G8_META + 2048))[ql];
      sh_r[ni] = ((const int*)(smem + G8_META + 3072))[ql];
      asm volatile("" : "+v"(mul_r[ni]), "+v"(thr_f[ni]), "+v"(sh_r[ni]));      // in registers from here on: the next tile's constants land in the same bytes
    }
    G8Frag f0, f1;
    {
      const char* sl = G8P_SLOT_OF(0);
#pragma unroll
      for (int g = 0; g < 8; ++g) read_s8(f0, sl, g);
    }
    // (the last pair of each half is peeled so that "read the next fragments" is a compile-time fact in both bodies)
#pragma unroll 1
    for (int g = 0; g + 1 < nsp; ++g) {
      G8P_BLK_S8(f0, f1, 2 * g + 1, true, false);
      G8P_PAIR_SYNC(g);
      G8P_BLK_S8(f1, f0, 2 * g + 2, true, true);
    }
    {
      const int g = nsp - 1;
      G8P_BLK_S8(f0, f1, 2 * g + 1, true, false);
      G8P_PAIR_SYNC(g);
      G8P_BLK_S8(f1, f0, 2 * g + 2, false, true);
    }
    G8P_T(3);
    if (td > 0) {
      // gated sums -> ungated units: every accumulator shifted left by its query's shift, then the first ungated fragments
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");     // the last matrix results are in the accumulators
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] = __uint_as_float(__float_as_uint(acc[mi][ni][e]) << sh_r[ni]);
      __builtin_amdgcn_sched_barrier(0);
      {
        const char* sl = G8P_SLOT_OF(ts);
#pragma unroll
        for (int g = 0; g < 6; ++g) read_dn(f0, sl, c0, g);
      }
      asm volatile("s_nop 4" ::: "memory");      // VALU write -> matrix read of the accumulators
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
      for (int g = nsp; g + 1 < npairs; ++g) {
        const int t0 = 4 * (g - nsp);
        G8P_BLK_DN(f0, f1, t0 + 1, true, false);
        G8P_BLK_DN(f1, f0, t0 + 2, true, false);
        G8P_BLK_DN(f0, f1, t0 + 3, true, false);
        G8P_PAIR_SYNC(g);
        G8P_BLK_DN(f1, f0, t0 + 4, true, true);
      }
      {
        const int g = npairs - 1;
        const int t0 = 4 * (g - nsp);
        G8P_BLK_DN(f0, f1, t0 + 1, true, false);
        G8P_BLK_DN(f1, f0, t0 + 2, true, false);
        G8P_BLK_DN(f0, f1, t0 + 3, true, false);
        G8P_PAIR_SYNC(g);
        G8P_BLK_DN(f1, f0, t0 + 4, false, true);
      }
    }
    G8P_T(4);
    // the tile after next, published behind this tile's first pair barrier
    const uint32_t Lnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL));
    G8P_FLUSH_ALL();                                       // (nothing left unless the previous tile queued more than 64 entries per pair of this one)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last matrix results are in the accumulators
    G8P_T(5);
    // ---- filter epilogue: hits -> the threads' stacks behind the ring, which already receives the next tile
    fl_stage = 0;
    if (active) {
      int thr_r[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        asm volatile("" : "+v"(thr_f[ni]));
        thr_r[ni] = cur_dead ? INT_MAX : g8_thr_units(thr_f[ni], mul_r[ni]);
      }
      KArgs k = g8p_kargs();
      const uint32_t row0 = (uint32_t)(dt * TILE_ROWS);
      const int64_t n_rows = k->n_rows;
      const int rows_valid = (int)(n_rows - dt * TILE_ROWS < TILE_ROWS ? n_rows - dt * TILE_ROWS : TILE_ROWS);
      int rbase = wm * 128 + 4 * fhalf;
      asm volatile("" : "+v"(rbase));          // (or the 128 per-element rows derived from it are hoisted out of the tile loop and live -- spilled -- across the stage loops)
      const int q0 = qt * TILE_ROWS + wn * 64 + (lane & 31);
      const bool full = rows_valid == TILE_ROWS;
      uint32_t j = 0;
      if (full) g8p_scan_half<false, 0, false>(k, acc, rbase, rows_valid, row0, q0, stack, thr_r[0], mul_r[0], j);
      else g8p_scan_half<false, 0, true>(k, acc, rbase, rows_valid, row0, q0, stack, thr_r[0], mul_r[0], j);
      const uint32_t s0 = j < (uint32_t)G8P_DEPTH ? j : (uint32_t)G8P_DEPTH;
      if (full) g8p_scan_half<false, 1, false>(k, acc, rbase, rows_valid, row0, q0 + 32, stack, thr_r[1], mul_r[1], j);
      else g8p_scan_half<false, 1, true>(k, acc, rbase, rows_valid, row0, q0 + 32, stack, thr_r[1], mul_r[1], j);
      const uint32_t s1 = j < (uint32_t)G8P_DEPTH ? j : (uint32_t)G8P_DEPTH;
#if G8_TRACE
      {
        const unsigned long long hits = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(j > 0u));
        unsigned long long tot = j;
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (lane == 0) { atomicAdd(&g8p_stat[0], 1ull); atomicAdd(&g8p_stat[10], tot); atomicAdd(&g8p_stat[11], hits); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (__builtin_amdgcn_ballot_w64(j > (uint32_t)G8P_STAT_D[i]) != 0 && lane == 0) atomicAdd(&g8p_stat[1 + i], 1ull);
      }
#endif
      if (__builtin_amdgcn_ballot_w64(j > (uint32_t)G8P_DEPTH) != 0) {      // some thread's stack is full: its surplus goes straight to the lists
        uint32_t js = 0;
        g8p_scan_half<true, 0, true>(k, acc, rbase, rows_valid, row0, q0, stack, thr_r[0], mul_r[0], js);
        g8p_scan_half<true, 1, true>(k, acc, rbase, rows_valid, row0, q0 + 32, stack, thr_r[1], mul_r[1], js);
      }
      if (__builtin_amdgcn_ballot_w64(s1 > 0u) != 0) {
        fl_stage = 1;
        fl_s = s0 | (s1 << 8);
        fl_q = q0;
        fl_mul0 = mul_r[0]; fl_mul1 = mul_r[1];
        fl_row0 = row0;
      }
    }
    G8P_T(6);
    // ---- on to the next tile of the stream
    cur_dt = nxt_dt; cur_qt = nxt_qt; cur_ok = nxt_ok; cur_dead = nxt_dead;
    decode(Lnn, fv, nxt_dt, nxt_qt, nxt_ok, nxt_dead);
    if (!nxt_ok && cur_ok) {                     // that sweep is used up: on to the next XCD's (the barriers inside are the tile's only extra ones)
      fv = (fv + 1u) & 7u;
      G8P_ACQUIRE_SYNC(nxt_dt, nxt_qt, nxt_ok, nxt_dead);
    }
    base_cur = base_nxt;
    base_nxt = role_base(nxt_dt, nxt_qt);
    soff = (soff + 2 * npairs) & 3;
    if (!cur_ok) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  flush_step();
  G8P_FLUSH_ALL();
}

hipError_t launch_gemm_g8p(const GemmArgs& a, hipStream_t s) {
  static std::mutex attr_mu;                       // per-device, under a lock (handles on different devices / host threads)
  static bool attr_set_dev[64] = {};
  static unsigned* ctr_dev[64] = {};
  static int n_cu[64] = {};
  static std::atomic<unsigned> next_slot{0};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  {
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (!attr_set_dev[dev_]) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_filter_g8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G8P_LDS);
      if (e != hipSuccess) return e;
      void* ptr = nullptr;
      e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(g8p_ctr));
      if (e != hipSuccess) return e;
      ctr_dev[dev_] = (unsigned*)ptr;
      e = hipDeviceGetAttribute(&n_cu[dev_], hipDeviceAttributeMultiprocessorCount, dev_);
      if (e != hipSuccess) return e;
      attr_set_dev[dev_] = true;
    }
  }
  GemmArgs b = a;
  const int64_t n_seq = a.seq_hi - a.seq_lo;
  b.p_groups = (n_seq + DOC_GROUP - 1) / DOC_GROUP;
  b.inv_perm_n = 1.0 / (double)(a.perm_n > 0 ? a.perm_n : 1);
  b.inv_pm1 = 1.0 / (double)(a.period > 1 ? a.period - 1 : 1);
  static const int partial_on = getenv("DHR_G8_PARTIAL") ? atoi(getenv("DHR_G8_PARTIAL")) : 1;
  const int valid_last = a.n_queries - (a.n_qtiles - 1) * TILE_ROWS;          // real queries of the batch's last query tile
  b.partial_wn = (partial_on && valid_last > 0 && valid_last <= 128) ? (valid_last + 63) / 64 : 0;
  unsigned* ctr = ctr_dev[dev_] + (size_t)(next_slot.fetch_add(1u) % G8P_CTR_SLOTS) * 8;
  hipError_t e = hipMemsetAsync(ctr, 0, 32, s);
  if (e != hipSuccess) return e;
  b.p_ctr = ctr;
  const int64_t tiles = n_seq * a.n_qtiles;
  const unsigned wgs = (unsigned)std::min<int64_t>(tiles, n_cu[dev_] > 0 ? n_cu[dev_] : 256);
  hipLaunchKernelGGL(gemm_filter_g8p_kernel, dim3(wgs), dim3(G8_NT), G8P_LDS, s, b);
  return hipGetLastError();
}

// can this launch run on the persistent kernel?  (stage pairs on both halves, at least two pairs, the XCD's linear tile index in 31 bits)
bool gemm_g8p_ok(const GemmArgs& a) {
  const int64_t n_seq = a.seq_hi - a.seq_lo;
  const int64_t groups = (n_seq + DOC_GROUP - 1) / DOC_GROUP;
  return !a.dump && a.ts >= 4 && !(a.ts & 1) && !(a.td & 1) && a.ts_q == a.ts &&
         ((groups + 7) / 8) * (int64_t)DOC_GROUP * a.n_qtiles < (int64_t)1 << 31;
}

}  // namespace dhr

#if G8_TRACE
// tuning hook of the trace build: copies the trace records (8 x u64 per tile, slot = 1024 x workgroup + the workgroup's tile count) to the host
extern "C" int dhr_debug_g8p_stat(unsigned long long* out16) try {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(dhr::g8p_stat), 128) != hipSuccess) return -1;
  unsigned long long z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(dhr::g8p_stat), z, 128) == hipSuccess ? 0 : -1;
} DHR_CATCH_STATUS
extern "C" int dhr_debug_g8p_trace(unsigned long long* out, int max_slots) try {
  const int m = max_slots < dhr::G8P_TRACE_SLOTS ? max_slots : dhr::G8P_TRACE_SLOTS;
  if (out && m > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(dhr::g8p_trace_buf), (size_t)m * 64) != hipSuccess) return -1;
  void* ptr = nullptr;
  if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(dhr::g8p_trace_buf)) != hipSuccess) return -1;
  if (hipMemset(ptr, 0, (size_t)dhr::G8P_TRACE_SLOTS * 64) != hipSuccess) return -1;
  return 0;
} DHR_CATCH_STATUS
#endif
