// Bound GEMM + filter of a gated_i8 index with PERSISTENT workgroups (round 4).  Same tile, same arithmetic and the same operand images as
// gemm_g8.hip; what changes is everything around the stage loops, because a per-tile timeline of that kernel (tools/g8_trace.py, -DG8_TRACE=1)
// showed where a 256 x 256 tile's ~58 k cycles went with an open filter: 24.6 k of matrix instructions, and
//   * 2.0 k between a workgroup's exit and its successor's start on the CU (3.5 k with an open filter), 1.9-2.1 k of prologue (the first
//     stage pair's round trip to L2) -- 160 KiB of LDS and 2 x 248 registers per SIMD allow ONE workgroup per CU, so nothing hides either;
//   * 12.8 k of filter epilogue (3.0 k with a closed filter): private hit stacks that fill the whole staging ring, two list reservations
//     (returning global atomics) per lane and their round trips, per-lane flush loops;
//   * ~4 k of waiting at the gated pairs' barriers for LDS-DMA pieces issued one block (~500 cycles) earlier -- less than an L2 round trip.
// Here a workgroup takes tile after tile from a per-XCD counter, and
//   1. the LDS-DMA stream never stops: the ring is free from a tile's LAST pair barrier on (its last fragments are in registers by then), so
//      pair 0 / pair 1 (+ the constants) of the NEXT tile are simply virtual pairs npairs / npairs + 1 of the current one.  They land during
//      the epilogue; the next tile's "prologue" is one barrier;
//   2. every piece of a pair is issued in the block right behind the pair barrier (a whole block of slack before the next barrier);
//   3. the epilogue leaves the ring alone: a hit goes to a small WAVE-PRIVATE queue (376 entries of (row, query, integer sum) behind the ring;
//      position = the wave's count + the lane's rank among the hits of that compare: v_cmp -> s_bcnt1 / v_mbcnt, no atomics, no branches on
//      the push) and the queue is flushed to the queries' lists INSIDE THE NEXT TILE'S STAGE LOOPS: one entry per lane, the list
//      reservation (+ the query's unit) issued behind one pair barrier, the store behind the next -- both round trips under matrix work;
//   4. tile indices come from the counter two tiles ahead (one returning atomic per tile by one lane, published through the LDS).
// The XCD a workgroup runs on is read from HW_REG_XCC_ID; every XCD owns the corpus tile groups g = xcc (mod 8) and sweeps them against all
// query tiles in the order of gemm_wg_tile (dhr_internal.h), so the L2 behaviour is that of the 3-D grid.  Results do not depend on which
// workgroup computes which tile: the lists are unordered sets.
#include "gemm_g8.h"
#include <atomic>
#include <mutex>
#include <type_traits>

#ifndef G8_TRACE
#define G8_TRACE 0
#endif
namespace dhr {

constexpr int G8P_NQW = 376;                              // entries of a wave's hit queue (its 128 x 64 part of the tile: 4.6 % of the accumulators)
constexpr int G8P_QW_BYTES = (G8P_NQW + 64) * 8;          // + 64 scratch entries behind it: lanes of a push that hold no hit write there
constexpr int G8P_QUEUE = G8_META + 4096;
constexpr int G8P_MAIL = G8P_QUEUE + 8 * G8P_QW_BYTES;    // one word: the tile index published for the tile after next
constexpr int G8P_LDS = G8P_MAIL + 64;
static_assert(G8P_LDS <= 163840, "LDS of a CU");
constexpr int G8P_CTR_SLOTS = 4096;                       // launches in flight share nothing: each takes the next slot of 8 counters
__device__ unsigned int g8p_ctr[G8P_CTR_SLOTS * 8];

#if G8_TRACE
constexpr int G8P_TRACE_SLOTS = 1 << 18;
__device__ unsigned long long g8p_trace_buf[G8P_TRACE_SLOTS * 8];
#define G8P_T(i) do { if (threadIdx.x == 0 && tslot < (unsigned)G8P_TRACE_SLOTS) g8p_trace_buf[(size_t)tslot * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define G8P_T(i) do { } while (0)
#endif

// One query column (NI_) of the lane's accumulators against the threshold t; hits are appended to the wave's queue.
template <int NI_, bool CHECK_ROWS>
__device__ __forceinline__ void g8p_scan_half(const GemmArgs& p, floatx16 (&acc)[4][2], char* smem, const int t, const float mul, const int rows_valid,
                                              const uint32_t q_lds, const uint32_t scratch_addr, const uint32_t lane_meta, int& wc, const uint32_t row0,
                                              const int q_glob) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const floatx16& a = acc[mi][NI_];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gm = max(max(__float_as_int(a[4 * g]), __float_as_int(a[4 * g + 1])), max(__float_as_int(a[4 * g + 2]), __float_as_int(a[4 * g + 3])));
      if (__builtin_amdgcn_ballot_w64(gm >= t) != 0) {            // wave-uniform: some lane of the wave holds a hit in this group of four
#pragma unroll
        for (int e = 4 * g; e < 4 * g + 4; ++e) {
          const int v = __float_as_int(a[e]);
          const uint32_t cm = (uint32_t)(mi * 32 + (e & 3) + 8 * (e >> 2)) | ((uint32_t)(NI_ * 32) << 8);
          const uint32_t meta = lane_meta + cm;                     // local row | query of the tile << 8 (no carries: both stay below 256)
          bool hit = v >= t;
          if (CHECK_ROWS) hit = hit && (int)(meta & 255u) < rows_valid;
          const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
          const int n = __builtin_popcountll(m);
          if (wc + n <= G8P_NQW) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t addr = hit ? q_lds + (uint32_t)(wc + (int)rank) * 8u : scratch_addr;
            *(uint2*)(smem + addr) = make_uint2(meta, (uint32_t)v);
            wc += n;
          } else if (hit) {                                         // the wave's queue is full (a tile of the hottest queries): straight to the list
            const uint32_t slot = atomicAdd(p.cnt + q_glob, 1u);
            if (slot < p.cap) p.cand[(int64_t)q_glob * p.cap + slot] = make_uint2(row0 + (meta & 255u), __float_as_uint(g8_score(v, mul)));
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
  const int nsp = ts >> 1, npairs = (ts + td) >> 1;      // the launcher selects this kernel only when ts and td are even and npairs >= 2
  // Everything that is needed once per tile (the tile map's parameters, the list pointers, the constants' arrays) is read from the kernel
  // argument segment WHERE it is used, through a pointer the compiler cannot see through: as plain `p.field` uses all of it is loaded at
  // entry and kept in ~60 scalar registers across the stage loops, which then spill.
  typedef const __attribute__((address_space(4))) GemmArgs* KArgs;
  auto kargs = [&]() __attribute__((always_inline)) -> KArgs {
    KArgs k = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
  };
  const int nq = p.n_qtiles;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const int64_t groups_x = p.p_groups > (int64_t)xcc ? (p.p_groups - (int64_t)xcc + 7) >> 3 : 0;
  const uint32_t per_group = (uint32_t)(DOC_GROUP * nq);
  const uint32_t limit = (uint32_t)groups_x * per_group;          // < 2^31 (launcher)

  // linear index of the XCD's sweep -> tile: group z of the XCD = group 8 z + xcc of the launch; inside a group the query tile moves slowest
  // over DOC_GROUP corpus tiles (the launch's last group may hold fewer).  The first index that is not a tile ends the XCD's stream.
  auto decode = [&](uint32_t L, int64_t& dt_o, int& qt_o, bool& ok_o, bool& dead_o) __attribute__((always_inline)) {
    int64_t dt = 0;
    int qt = 0, ok = 0, dead = 0;
    if (L < limit) {
      const uint32_t z = L / per_group, r = L - z * per_group;
      const int64_t grp = (int64_t)z * 8 + (int64_t)xcc;
      KArgs k = kargs();
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

  // ---- first two tile indices of this workgroup
  if (threadIdx.x == 0) *(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL) = atomicAdd(p.p_ctr + xcc, 2u);
  __syncthreads();
  const uint32_t L0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL));
  int64_t cur_dt, nxt_dt;
  int cur_qt, nxt_qt;
  bool cur_ok, nxt_ok, cur_dead, nxt_dead;
  decode(L0, cur_dt, cur_qt, cur_ok, cur_dead);
  decode(L0 + 1u, nxt_dt, nxt_qt, nxt_ok, nxt_dead);
  if (!cur_ok) return;
#if G8_TRACE
  uint32_t Lcur = L0, Lnxt = L0 + 1u;
#endif

  // ---- LDS-DMA, fixed roles: wave w streams half (w & 1) of image ((w >> 1) & 1 ? query : corpus) of stage 2g + (w >> 2) of every pair g
  const bool dma_b = ((wave >> 1) & 1) != 0;
  const int dma_s = wave >> 2;
  const int dma_h = wave & 1;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t smem_u = (uint32_t)(uintptr_t)LDS_PTR(smem);
  auto role_base = [&](int64_t t_dt, int t_qt) __attribute__((always_inline)) -> const char* {
    KArgs k = kargs();
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
  auto issue_consts = [&](int64_t t_dt, int t_qt) __attribute__((always_inline)) {
    if (wave < 4) {
      KArgs k = kargs();
      const char* src = wave == 0 ? (const char*)(k->g8_rsum + t_dt * TILE_ROWS) : wave == 1 ? (const char*)(k->i8_mul + t_qt * TILE_ROWS)
                      : wave == 2 ? (const char*)(k->thr + t_qt * TILE_ROWS) : (const char*)(k->g8_shift + t_qt * TILE_ROWS);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + lane_off), LDS_PTR(smem + G8_META + wave * 1024), 16, 0, 0);
    }
  };
  // virtual pair g of the current tile's stream: g >= npairs is pair g - npairs of the NEXT tile.  Prepared in ascending order, one pair per
  // call (the scalar work sits in the stage loops): the source offset advances by two stages, the LDS address toggles between the role's
  // two ring slots; only the first ungated pair and the first pair of the next tile are set up from scratch.
  const int st_sp = dma_b ? SP_STAGE_B : S8_STAGE_A;                 // this role's bytes per gated stage
  const int half_sp = dma_h * (dma_b ? 8192 : S8_STAGE_A / 2), half_dn = dma_h * 8192;
  const int n_sp = dma_b ? 8 : 5;
  const uint32_t lds_role = smem_u + (uint32_t)(dma_b ? G8_QOFF : 0);
  auto dma_prepare = [&](int g) __attribute__((always_inline)) {
    if (g == npairs) {                                   // pair 0 of the next tile
      nx_soff = dma_s * st_sp + half_sp;
      nx_n = nxt_ok ? n_sp : 0;
      nx_base = base_nxt;
      nx_consts = nxt_ok;
    } else if (g == nsp) {                               // first ungated pair
      nx_soff = ts * st_sp + dma_s * SP_DENSE + half_dn;
      nx_n = 8;
      nx_consts = false;
    } else {
      nx_soff += g < nsp || g > npairs ? 2 * st_sp : 2 * SP_DENSE;
      nx_consts = false;
    }
    nx_lds = lds_role + (uint32_t)(((soff + 2 * g + dma_s) & 3) * G8_SLOT + ((g < nsp || g >= npairs) ? half_sp : half_dn));
  };
  auto dma_commit = [&]() __attribute__((always_inline)) {
    dma_soff = nx_soff; dma_lds = nx_lds; dma_n = nx_n;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)nx_base, (short)0, 0x7fffffff, 0x00020000);
    if (nx_consts) issue_consts(nxt_dt, nxt_qt);
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

  // ---- the wave's hit queue and its flush (one entry per lane and round, inside the next tile's stage loops)
  const uint32_t q_lds = (uint32_t)(G8P_QUEUE + wave * G8P_QW_BYTES);
  const uint32_t scratch_addr = q_lds + (uint32_t)(G8P_NQW + lane) * 8u;
  int fl_wc = 0, fl_round = 0;
  bool fl_pending = false, mail_pending = false;
  uint32_t fl_row0 = 0;
  int fl_q0 = 0;
  uint32_t fl_slot = 0, pend = 0;
  float fl_mul = 0.f;
  auto flush_step = [&]() __attribute__((always_inline)) {
    if (mail_pending) {                       // the tile index asked for at the top of this tile has arrived (every pair barrier waits for vmcnt(0))
      if (threadIdx.x == 0) *(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL) = pend;
      mail_pending = false;
    }
    if (fl_pending) {
      const int idx = fl_round * 64 + lane;
      if (idx < fl_wc) {
        const uint2 e = *(const uint2*)(smem + q_lds + (uint32_t)idx * 8u);
        const int q = fl_q0 + (int)((e.x >> 8) & 255u);
        KArgs k = kargs();
        const uint32_t cap = k->cap;
        if (fl_slot < cap) k->cand[(int64_t)q * cap + fl_slot] = make_uint2(fl_row0 + (e.x & 255u), __float_as_uint(g8_score((int)e.y, fl_mul)));
      }
      ++fl_round;
      fl_pending = false;
    }
    if (fl_round * 64 < fl_wc) {
      const int idx = fl_round * 64 + lane;
      if (idx < fl_wc) {
        const uint2 e = *(const uint2*)(smem + q_lds + (uint32_t)idx * 8u);
        const int q = fl_q0 + (int)((e.x >> 8) & 255u);
        KArgs k = kargs();
        fl_slot = atomicAdd(k->cnt + q, 1u);
        fl_mul = k->i8_mul[q];
      }
      fl_pending = true;
    }
  };
  auto flush_all = [&]() __attribute__((always_inline)) {
    while (fl_pending || fl_round * 64 < fl_wc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      flush_step();
    }
  };

  // ---- the first tile's operands: the only DMA issue outside a stage loop
  issue_consts(cur_dt, cur_qt);
  nx_soff = dma_s * st_sp + half_sp - 2 * st_sp; nx_n = n_sp; nx_base = base_cur;     // so that dma_prepare(0) advances onto pair 0
  dma_prepare(0); dma_commit();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(j);
  dma_prepare(1); dma_commit();
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(j);
  dma_n = 0;

  constexpr std::integral_constant<bool, true> YES{};
  constexpr std::integral_constant<bool, false> NO{};

  // One 256 x 256 tile.  PARTIAL: the batch's LAST query tile when at most 128 of its 256 queries are real; the waves are then numbered so
  // that the wave columns holding real queries sit on different SIMDs, the others only stream their share of the LDS-DMA (gemm_g8.hip).
  auto run_tile = [&](auto partial_c) __attribute__((always_inline)) {
    constexpr bool PARTIAL = decltype(partial_c)::value;
#if G8_TRACE
    const unsigned tslot = xcc * (unsigned)(G8P_TRACE_SLOTS / 8) + Lcur;
    if (threadIdx.x == 0 && Lcur < (unsigned)(G8P_TRACE_SLOTS / 8)) {
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
    auto slot_of = [&](int u) __attribute__((always_inline)) -> const char* { return smem + ((soff + u) & 3) * G8_SLOT; };

    floatx16 acc[4][2];
    auto read_s8 = [&](G8Frag& f, const char* sl, int g) __attribute__((always_inline)) {
      if (g < 4) f.b[g >> 1].h[g & 1] = *(const intx4*)(sl + q_row + (g >> 1) * 2048 + ((g & 1) ? c1 : c0));
      else if (g < 8) f.a[g - 4] = *(const intx4*)(sl + a8_off + (g - 4) * 1024);
      else if (g == 8) f.pw = *(const intx4*)(sl + p8_off);
    };
    auto read_dn = [&](G8Frag& f, const char* sl, int cc, int g) __attribute__((always_inline)) {
      if (g < 2) f.b[g].h[0] = *(const intx4*)(sl + q_row + g * 2048 + cc);
      else if (g < 6) f.a[g - 2] = *(const intx4*)(sl + ad_row + (g - 2) * 2048 + cc);
    };
    // gated block: computes the stage `fc` came from; reads the fragments of gated stage un into fn; DMA: one piece of the pair just
    // committed behind every matrix instruction
    auto blk_s8 = [&](const G8Frag& fc, G8Frag& fn, int un, auto load_c, auto dma_c) __attribute__((always_inline)) {
      constexpr bool LOAD = decltype(load_c)::value, DMA = decltype(dma_c)::value;
      const char* sl = slot_of(un);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int ni = g >> 2, mi = g & 3;
        if (!PARTIAL || active) {
          if (LOAD) {
            read_s8(fn, sl, g);
            if (g == 0) read_s8(fn, sl, 8);
          }
          g8_smfmac(acc[mi][ni], fc.a[mi], fc.b[ni].v, (uint32_t)fc.pw[mi]);
        }
        if (DMA) dma_piece(g);
      }
    };
    auto blk_dn = [&](const G8Frag& fc, G8Frag& fn, int tn, auto load_c, auto dma_c) __attribute__((always_inline)) {
      constexpr bool LOAD = decltype(load_c)::value, DMA = decltype(dma_c)::value;
      const char* sl = slot_of(ts + (tn >> 1));
      const int cc = (tn & 1) ? c1 : c0;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int ni = g >> 2, mi = g & 3;
        if (!PARTIAL || active) {
          if (LOAD) read_dn(fn, sl, cc, g);
          g8_mfma(acc[mi][ni], fc.a[mi], fc.b[ni].h[0]);
        }
        if (DMA) dma_piece(g);
      }
    };
    // pair barrier: this wave's pieces have landed, every wave is done with the pair's slots; then the next virtual pair is committed and
    // the flush of the previous tile's hits moves one step
    auto pair_sync = [&](int g) __attribute__((always_inline)) {
      dma_prepare(g + 2);
      __builtin_amdgcn_sched_barrier(0);                    // the matrix asm is not volatile: without this the compiler sinks the block above below the wait
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma_commit();
      flush_step();
    };

    // ---- the tile's first pair and constants are in the LDS (issued by the previous tile's last pairs, or above for the first tile)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    G8P_T(2);
    if (threadIdx.x == 0) pend = atomicAdd(kargs()->p_ctr + xcc, 1u);      // the tile after next
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
      const char* sl = slot_of(0);
#pragma unroll
      for (int g = 0; g < 9; ++g) read_s8(f0, sl, g);
    }
    // (the last pair of each half is peeled so that "read the next fragments" is a compile-time fact in both bodies)
#pragma unroll 1
    for (int g = 0; g + 1 < nsp; ++g) {
      blk_s8(f0, f1, 2 * g + 1, YES, NO);
      pair_sync(g);
      blk_s8(f1, f0, 2 * g + 2, YES, YES);
    }
    {
      const int g = nsp - 1;
      blk_s8(f0, f1, 2 * g + 1, YES, NO);
      pair_sync(g);
      blk_s8(f1, f0, 2 * g + 2, NO, YES);
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
        const char* sl = slot_of(ts);
#pragma unroll
        for (int g = 0; g < 6; ++g) read_dn(f0, sl, c0, g);
      }
      asm volatile("s_nop 4" ::: "memory");      // VALU write -> matrix read of the accumulators
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
      for (int g = nsp; g + 1 < npairs; ++g) {
        const int t0 = 4 * (g - nsp);
        blk_dn(f0, f1, t0 + 1, YES, NO);
        blk_dn(f1, f0, t0 + 2, YES, NO);
        blk_dn(f0, f1, t0 + 3, YES, NO);
        pair_sync(g);
        blk_dn(f1, f0, t0 + 4, YES, YES);
      }
      {
        const int g = npairs - 1;
        const int t0 = 4 * (g - nsp);
        blk_dn(f0, f1, t0 + 1, YES, NO);
        blk_dn(f1, f0, t0 + 2, YES, NO);
        blk_dn(f0, f1, t0 + 3, YES, NO);
        pair_sync(g);
        blk_dn(f1, f0, t0 + 4, NO, YES);
      }
    }
    G8P_T(4);
    // the tile after next, published behind this tile's first pair barrier
    const uint32_t Lnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(volatile __attribute__((address_space(3))) uint32_t*)LDS_PTR(smem + G8P_MAIL));
    flush_all();                                       // (nothing left unless the previous tile queued more than 64 entries per pair of this one)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last matrix results are in the accumulators
    G8P_T(5);
    // ---- filter epilogue: hits -> the wave's queue; the ring already receives the next tile
    int wc = 0;
    if (!PARTIAL || active) {
      int thr_r[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        asm volatile("" : "+v"(thr_f[ni]));
        thr_r[ni] = cur_dead ? INT_MAX : g8_thr_units(thr_f[ni], mul_r[ni]);
      }
      const uint32_t row0 = (uint32_t)(dt * TILE_ROWS);
      const int64_t n_rows = kargs()->n_rows;
      const int rows_valid = (int)(n_rows - dt * TILE_ROWS < TILE_ROWS ? n_rows - dt * TILE_ROWS : TILE_ROWS);
      uint32_t lane_meta = (uint32_t)(wm * 128 + 4 * fhalf) | ((uint32_t)(wn * 64 + (lane & 31)) << 8);
      asm volatile("" : "+v"(lane_meta));          // (or the 128 per-element values derived from it are hoisted out of the tile loop and live -- spilled -- across the stage loops)
      const int q0 = qt * TILE_ROWS + wn * 64 + (lane & 31);
      if (rows_valid == TILE_ROWS) {
        g8p_scan_half<0, false>(p, acc, smem, thr_r[0], mul_r[0], rows_valid, q_lds, scratch_addr, lane_meta, wc, row0, q0);
        g8p_scan_half<1, false>(p, acc, smem, thr_r[1], mul_r[1], rows_valid, q_lds, scratch_addr, lane_meta, wc, row0, q0 + 32);
      } else {
        g8p_scan_half<0, true>(p, acc, smem, thr_r[0], mul_r[0], rows_valid, q_lds, scratch_addr, lane_meta, wc, row0, q0);
        g8p_scan_half<1, true>(p, acc, smem, thr_r[1], mul_r[1], rows_valid, q_lds, scratch_addr, lane_meta, wc, row0, q0 + 32);
      }
    }
    fl_wc = wc; fl_round = 0; fl_pending = false;
    fl_row0 = (uint32_t)(dt * TILE_ROWS);
    fl_q0 = qt * TILE_ROWS;
    G8P_T(6);
    // ---- on to the next tile of the stream
    cur_dt = nxt_dt; cur_qt = nxt_qt; cur_ok = nxt_ok; cur_dead = nxt_dead;
    decode(Lnn, nxt_dt, nxt_qt, nxt_ok, nxt_dead);
#if G8_TRACE
    Lcur = Lnxt; Lnxt = Lnn;
#endif
    base_cur = base_nxt;
    base_nxt = role_base(nxt_dt, nxt_qt);
    soff = (soff + 2 * npairs) & 3;
  };

  for (;;) {
    if (p.partial_wn > 0 && cur_qt == nq - 1) run_tile(YES);
    else run_tile(NO);
    if (!cur_ok) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  flush_step();
  flush_all();
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
// tuning hook of the trace build: copies the trace records (8 x u64 per tile, slot = xcc * 32768 + the tile's index on its XCD) to the host
extern "C" int dhr_debug_g8p_trace(unsigned long long* out, int max_slots) {
  const int m = max_slots < dhr::G8P_TRACE_SLOTS ? max_slots : dhr::G8P_TRACE_SLOTS;
  if (out && m > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(dhr::g8p_trace_buf), (size_t)m * 64) != hipSuccess) return -1;
  void* ptr = nullptr;
  if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(dhr::g8p_trace_buf)) != hipSuccess) return -1;
  if (hipMemset(ptr, 0, (size_t)dhr::G8P_TRACE_SLOTS * 64) != hipSuccess) return -1;
  return 0;
}
#endif
