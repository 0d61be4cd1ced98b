// Bound GEMM + filter of a gated_i8 index: BOTH halves of the bound are integer sums.
//   gated half  : v_smfmac_i32_32x32x64_i8 -- the 2:4 instruction on int8 operands, 64 logical columns (= 32 slices x 2 buckets) per
//                 issue; measured under the package power cap on DLR-shaped operands (tools/probe/smfmac8_probe.hip,
//                 profiles/r03_smfmac8_probe.txt): 21.8 ns per instruction and SIMD against 27.9 ns for the fp16 form, which covers
//                 only 16 slices -- the gated half of a tile costs 24 x 21.8 instead of 48 x 27.9 ns of matrix time per 32 x 32 block;
//   ungated half: v_mfma_i32_32x32x32_i8 on the int8 image of the ungated columns, as in gemm_w4.hip.
// One accumulator set of int32 (started at 128 x the row's sum of gated values: the query's gated operand is stored as level - 128,
// which gives it 8 bits): the gated stages run first, the sums are shifted left by the query's `shift` (its gated unit is
// 2^shift ungated units, query_prep_kernel), the ungated stages accumulate on top, and the filter compares integers.
// Register layouts of the 2:4 int8 instruction (measured, the probe validates them on random operands):
//   A: lane l = (row l & 31, half hA = l >> 5) holds 16 stored bytes E = 0..15; bytes 2g, 2g+1 are the two non-zeros of group g,
//      position of byte E in its group of four logical columns at idx[2E+1 : 2E];
//   B: group g of half hA multiplies bytes 16 hA + 4 (g & 3) .. + 3 of lane (column, half g >> 2).
// With slices 16 hA + E in A's bytes and position = 2 (slice & 1) + bucket, lane (n, hB) of B holds the EXPANDED slices
// (bucket-0 column, bucket-1 column) 8 hB .. 8 hB + 7 in its first 16 bytes and 16 + 8 hB .. in its second: chunks hB and 2 + hB of
// the query's 64-byte stage row in natural order -- no register expansion at all.
// Structure: the 8-wave form of gemm_w4.hip (every wave computes and issues its share of the LDS-DMA, stages handed over in
// pairs, one workgroup barrier per pair); a gated stage is ONE block of 8 matrix instructions per wave (K = 64 logical columns).
#include "gemm_g8.h"
#include <mutex>
#include <type_traits>

#ifndef G8_TRACE
#define G8_TRACE 0    // 1: thread 0 of every workgroup records clock values at the tile's phase boundaries (tools/g8_trace.py; timing only)
#endif
#ifndef G8_ABL
#define G8_ABL 0      // timing ablations (wrong results): 4 = no DMA pieces in the loop, 8 = 4 + no fragment reads in the loop, 16 = no list flush, 32 = no epilogue
#endif
namespace dhr {

#if G8_TRACE
constexpr int G8_TRACE_SLOTS = 1 << 18;
__device__ unsigned long long g8_trace_buf[G8_TRACE_SLOTS * 8];
__device__ unsigned int g8_trace_n;
#define G8_T(i) do { if (threadIdx.x == 0 && tslot < G8_TRACE_SLOTS) g8_trace_buf[(size_t)tslot * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define G8_T(i) do { } while (0)
#endif
__device__ __forceinline__ void g8_dump_tile(const GemmArgs& p, floatx16 (&acc)[4][2], int64_t dt, int qt, int wm, int wn, int lane,
                                             const float (&mul_r)[2]) {
  const int fhalf = lane >> 5;
  const int64_t row_base = dt * TILE_ROWS + wm * 128;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int q = qt * TILE_ROWS + wn * 64 + ni * 32 + (lane & 31);
    if (q < p.n_queries) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = row_base + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
          if (row < p.n_rows && row >= p.dump_row0 && row - p.dump_row0 < p.dump_ld)
            p.dump[(int64_t)q * p.dump_ld + (row - p.dump_row0)] = g8_score(__float_as_int(acc[mi][ni][e]), mul_r[ni]);
        }
    }
  }
}

// The 128 accumulators of a lane can only be addressed by unrolled code; in the first cut every one of the 128 copies carried the score
// conversion and the stack-overflow path (a global atomic + store) -- ~40 KB of code that every tile walked through.
// One query column (ni) of the lane's accumulators.  Two levels: the maximum of a group of four accumulators against the threshold,
// then the four elements.  (A third level over the 16 accumulators of a 32 x 32 block was dropped in round 3: at the bench's hit
// rate of 0.23 % per accumulator, 91 % of the blocks hold a hit in SOME lane of the wave, so the test was nearly always taken and
// only cost its 15 maxima.)
template <bool SURPLUS, int NI_, bool CHECK_ROWS = true>
__device__ __forceinline__ void g8_scan_half(const GemmArgs& p, floatx16 (&acc)[4][2], int qt, int wm, int wn, int lane, int rows_valid, int64_t row0,
                                             uint2* stack, const int t, const float mul, uint32_t& j) {
  const int fhalf = lane >> 5;
  const int rbase = wm * 128 + 4 * fhalf;
  const int q = qt * TILE_ROWS + wn * 64 + NI_ * 32 + (lane & 31);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const floatx16& a = acc[mi][NI_];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int gm = max(max(__float_as_int(a[4 * g]), __float_as_int(a[4 * g + 1])), max(__float_as_int(a[4 * g + 2]), __float_as_int(a[4 * g + 3])));
      if (gm >= t) {
        asm volatile("");
        if constexpr (!SURPLUS && !CHECK_ROWS) {
          // full tile, stack pass: BRANCH-FREE pushes.  Every lane of the active group writes all four (row, sum) pairs to its current stack
          // slot and advances only on a hit -- a miss is overwritten by the lane's next write; slot EPI_STACK (full stack) is this thread's 8
          // bytes of the constants' area behind the ring, idle by now.  (A push used to be a serial v_cmp -> s_and_saveexec -> s_cbranch
          // chain of ~35 cycles, ~58 of them per tile with nothing to overlap.)
#pragma unroll
          for (int e = 4 * g; e < 4 * g + 4; ++e) {
            const int v = __float_as_int(a[e]);
            const int rl = rbase + mi * 32 + (e & 3) + 8 * (e >> 2);
            const uint32_t slot = j < (uint32_t)EPI_STACK ? j : (uint32_t)EPI_STACK;
            stack[slot * G8_NT] = make_uint2((uint32_t)rl, (uint32_t)v);
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
              if (j >= EPI_STACK) {
                const uint32_t slot = atomicAdd(p.cnt + q, 1u);
                cand_store(p, q, slot, make_uint2((uint32_t)(row0 + rl), __float_as_uint(g8_score(v, mul))));
              }
            } else if (j < EPI_STACK) stack[j * G8_NT] = make_uint2((uint32_t)rl, (uint32_t)v);
            ++j;
          }
        }
      }
    }
  }
}
__device__ __forceinline__ void g8_epilogue_surplus(const GemmArgs& p, floatx16 (&acc)[4][2], int qt, int wm, int wn, int lane, int rows_valid, int64_t row0,
                                                 const int (&thr_r)[2], const float (&mul_r)[2]) {
  uint32_t j = 0;
  g8_scan_half<true, 0>(p, acc, qt, wm, wn, lane, rows_valid, row0, nullptr, thr_r[0], mul_r[0], j);
  g8_scan_half<true, 1>(p, acc, qt, wm, wn, lane, rows_valid, row0, nullptr, thr_r[1], mul_r[1], j);
}
// Filter epilogue.  Scheme of gemm_epilogue_w (private hit stacks in the idle ring, one global atomic per lane and query), on
// integers, with a SMALL hot path: a hit pushes (row, raw integer sum) and nothing else; conversion happens in the flush loop, and
// a lane whose stack is full (EPI_STACK hits in one tile: the hottest queries only) just counts on -- its surplus is appended by a
// second, cold scan (g8_epilogue_surplus) that only waves with such a lane run.  The list reservation of the first query column goes
// out BEFORE the second column is scanned, so that its L2 round trip (~1.5 us, 5 % of an open-filter tile when both were waited for
// behind the scan) runs under the second half of the scan; the second one runs under the flush of the first column's hits.
__device__ __forceinline__ void g8_epilogue(const GemmArgs& p, floatx16 (&acc)[4][2], int64_t dt, int qt, int wm, int wn, int tid, int lane,
                                            char* smem, const int (&thr_r)[2], const float (&mul_r)[2]) {
  __syncthreads();                       // every wave is done with the staging ring
  const int64_t row0 = dt * TILE_ROWS;
  const int rows_valid = (int)(p.n_rows - row0 < TILE_ROWS ? p.n_rows - row0 : TILE_ROWS);
  uint2* stack = (uint2*)smem + tid;                       // slot j at stack[j * G8_NT]
  const int q0 = qt * TILE_ROWS + wn * 64 + (lane & 31);
  // every tile but a shard's last one is full: its copy of the scan has no per-element row test (one compare and one scalar AND less in
  // each of the ~90 serial test chains of a tile)
  const bool full = rows_valid == TILE_ROWS;
  uint32_t j = 0;
  if (full) g8_scan_half<false, 0, false>(p, acc, qt, wm, wn, lane, rows_valid, row0, stack, thr_r[0], mul_r[0], j);
  else g8_scan_half<false, 0, true>(p, acc, qt, wm, wn, lane, rows_valid, row0, stack, thr_r[0], mul_r[0], j);
  const uint32_t s0 = j < (uint32_t)EPI_STACK ? j : (uint32_t)EPI_STACK;
#if G8_ABL != 16
  uint32_t base0 = 0u, base1 = 0u;
  if (s0 > 0) base0 = atomicAdd(p.cnt + q0, s0);
#endif
  if (full) g8_scan_half<false, 1, false>(p, acc, qt, wm, wn, lane, rows_valid, row0, stack, thr_r[1], mul_r[1], j);
  else g8_scan_half<false, 1, true>(p, acc, qt, wm, wn, lane, rows_valid, row0, stack, thr_r[1], mul_r[1], j);
  const uint32_t s1 = j < (uint32_t)EPI_STACK ? j : (uint32_t)EPI_STACK;
#if G8_ABL == 16       // timing only: hits are found and stacked, never flushed to the lists
  return;
#else
  if (s1 > s0) base1 = atomicAdd(p.cnt + q0 + 32, s1 - s0);
  if (__builtin_amdgcn_ballot_w64(j > (uint32_t)EPI_STACK) != 0) g8_epilogue_surplus(p, acc, qt, wm, wn, lane, rows_valid, row0, thr_r, mul_r);
  for (uint32_t i = 0; i < s0; ++i) {
    const uint2 en = stack[i * G8_NT];
    const uint32_t slot = base0 + i;
    cand_store(p, q0, slot, make_uint2((uint32_t)row0 + en.x, __float_as_uint(g8_score((int)en.y, mul_r[0]))));
  }
  for (uint32_t i = s0; i < s1; ++i) {
    const uint2 en = stack[i * G8_NT];
    const uint32_t slot = base1 + (i - s0);
    cand_store(p, q0 + 32, slot, make_uint2((uint32_t)row0 + en.x, __float_as_uint(g8_score((int)en.y, mul_r[1]))));
  }
#endif
}

#if G8_ABL == 2       // timing only: the pair barrier does not wait for this wave's DMA
#define G8_PAIR_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#elif G8_ABL == 3     // timing only: neither the DMA wait nor the barrier
#define G8_PAIR_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define G8_PAIR_SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#endif

// DMA piece (of this wave's <= 8 per stage pair) issued behind matrix instruction g of the block in phase ph: 0 = the block behind
// the pair barrier, 1 = the next pair's first block
#ifndef G8_DMA_SCHED
#define G8_DMA_SCHED 1    // 0 = the schedule of rounds 3-4a (four pieces behind the barrier, four in the next block)
#endif
#if G8_DMA_SCHED == 1      // every piece of the next pair right behind the pair barrier (one per matrix instruction of that block)
__host__ __device__ constexpr int g8_dma_piece(int ph, int g) { return ph == 0 ? g : -1; }
#elif G8_DMA_SCHED == 2    // pieces 0-5 behind the barrier, 6-7 in the following block
__host__ __device__ constexpr int g8_dma_piece(int ph, int g) { return ph == 0 ? (g >= 2 ? g - 2 : -1) : (ph == 1 && g < 2) ? 6 + g : -1; }
#else
__host__ __device__ constexpr int g8_dma_piece(int ph, int g) { return (ph <= 1 && (g & 1)) ? ph * 4 + (g >> 1) : -1; }
#endif

// One 256 x 256 tile.  PARTIAL: the batch's LAST query tile when at most 128 of its 256 queries are real (6 980 queries = 27 tiles + 68: the
// 28th tile used to cost a full tile for 1 % of the queries, 2.6 % of the launch).  The waves are then numbered so that the wave columns that
// hold real queries (wn < partial_wn) sit on DIFFERENT SIMDs (a workgroup's waves go to the SIMDs in cyclic order: waves w and w + 4 share
// one), the others neither read fragments nor issue matrix instructions -- they only stream their share of the LDS-DMA and keep the
// barriers; each SIMD then runs ONE computing wave with the matrix pipe to itself and the tile takes about half the time.
template <bool DUMP, bool PARTIAL>
__device__ __forceinline__ void g8_tile(const GemmArgs& p, char* smem, const int64_t dt, const int qt, const unsigned tslot = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ts = p.ts, td = p.td;
  const int nsp = ts >> 1, npairs = (ts + td) >> 1;      // the launcher selects this kernel only when ts and td are even
  const char* a_src = (const char*)p.a_tiles + dt * ((int64_t)ts * S8_STAGE_A + (int64_t)td * SP_DENSE);
  const char* b_src = (const char*)p.b_tiles + (int64_t)qt * ((int64_t)ts * SP_STAGE_B + (int64_t)td * SP_DENSE);
  const int wm = PARTIAL ? (wave & 1) : (wave >> 2), wn = PARTIAL ? (wave >> 1) : (wave & 3);
  const bool active = !PARTIAL || wn < p.partial_wn;

  // ---- LDS-DMA, fixed roles: wave w streams half (w & 1) of image ((w >> 1) & 1 ? query : corpus) of stage 2g + (w >> 2)
  const bool dma_b = ((wave >> 1) & 1) != 0;
  const int dma_s = wave >> 2;
  const int dma_h = wave & 1;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t smem_u = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const __amdgpu_buffer_rsrc_t role_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(dma_b ? b_src : a_src), (short)0, 0x7fffffff, 0x00020000);
  int dma_soff = 0, dma_n = 0, nx_soff = 0, nx_n = 0;
  uint32_t dma_lds = 0, nx_lds = 0;
  auto dma_prepare = [&](int g) __attribute__((always_inline)) {
    const int u = 2 * g + dma_s;
    const bool sp = u < ts;
    const int half_bytes = dma_h * ((!dma_b && sp) ? S8_STAGE_A / 2 : 8192);
    if (dma_b) nx_soff = (sp ? u * SP_STAGE_B : ts * SP_STAGE_B + (u - ts) * SP_DENSE) + half_bytes;
    else nx_soff = (sp ? u * S8_STAGE_A : ts * S8_STAGE_A + (u - ts) * SP_DENSE) + half_bytes;
    nx_lds = smem_u + (uint32_t)((u & 3) * G8_SLOT + (dma_b ? G8_QOFF : 0) + half_bytes);
    nx_n = g < npairs ? ((!dma_b && sp) ? 5 : 8) : 0;
    if ((G8_ABL == 4 || G8_ABL == 8) && g >= 2) nx_n = 0;
  };
  auto dma_commit = [&]() __attribute__((always_inline)) { dma_soff = nx_soff; dma_lds = nx_lds; dma_n = nx_n; };
  auto dma_piece = [&](int j) __attribute__((always_inline)) {
    if (j < dma_n) {
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(uintptr_t)(dma_lds + (uint32_t)(j >> 2) * 4096u);
      const int so = dma_soff + (j >> 2) * 4096;
      switch (j & 3) {      // the immediate must be a literal
        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 0, 0); break;
        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 1024, 0); break;
        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 2048, 0); break;
        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 3072, 0); break;
      }
    }
  };
  auto issue_pair = [&](int g) {
    dma_prepare(g);
    dma_commit();
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_piece(j);
  };

  floatx16 acc[4][2];          // set in the prologue below (row sums of the gated image)
  // per-lane LDS offsets inside a ring slot
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int swz4 = (frow >> 2) & 3;
  const int c0 = (fhalf ^ swz4) << 4, c1 = ((2 + fhalf) ^ swz4) << 4;
  const int a8_off = ((wm * 8 + fhalf) * 32 + frow) * 16;                     // gated corpus values, + mi * 1024
  const int p8_off = S8_A_BYTES + ((wm * 2 + fhalf) * 32 + frow) * 16;        // the lane's four position words
  const int ad_row = (wm * 128 + frow) * 64;                                  // ungated corpus rows, + mi * 2048
  const int q_row = G8_QOFF + (wn * 64 + frow) * 64;                          // query rows (both kinds), + ni * 2048

  // fragment read number g (0..8) of the gated stage in ring slot sl: 2 x 2 query chunks, 4 corpus value reads, the 4 position words
  auto read_s8 = [&](G8Frag& f, const char* sl, int g) __attribute__((always_inline)) {
    if (g < 4) f.b[g >> 1].h[g & 1] = *(const intx4*)(sl + q_row + (g >> 1) * 2048 + ((g & 1) ? c1 : c0));
    else if (g < 8) f.a[g - 4] = *(const intx4*)(sl + a8_off + (g - 4) * 1024);
    else if (g == 8) f.pw = *(const intx4*)(sl + p8_off);
  };
  // fragment read number g (0..5) of ungated block t (stage t >> 1 of the ungated part, 32-column half t & 1)
  auto read_dn = [&](G8Frag& f, const char* sl, int cc, int g) __attribute__((always_inline)) {
    if (g < 2) f.b[g].h[0] = *(const intx4*)(sl + q_row + g * 2048 + cc);
    else if (g < 6) f.a[g - 2] = *(const intx4*)(sl + ad_row + (g - 2) * 2048 + cc);
  };
  // gated block: computes stage `fc` came from; reads the fragments of gated stage un into fn (12 reads spread over the 8 issue groups)
  auto blk_s8 = [&](const G8Frag& fc, G8Frag& fn, int un, bool do_load, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    const char* sl = smem + (un & 3) * G8_SLOT;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (!PARTIAL || active) {
        if (do_load && G8_ABL != 8) {
          read_s8(fn, sl, g);
          if (g == 0) read_s8(fn, sl, 8);
        }
        g8_smfmac(acc[mi][ni], fc.a[mi], fc.b[ni].v, (uint32_t)fc.pw[mi]);
      }
      if (g8_dma_piece(PH, g) >= 0) dma_piece(g8_dma_piece(PH, g));
      __builtin_amdgcn_sched_barrier(0);       // keep the reads between the matrix instructions (blocks without DMA pieces were clustered: 9 reads, one wait, 8 instructions)
    }
  };
  auto blk_dn = [&](const G8Frag& fc, G8Frag& fn, int tn, bool do_load, auto ph_c) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    const char* sl = smem + ((ts + (tn >> 1)) & 3) * G8_SLOT;
    const int cc = (tn & 1) ? c1 : c0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (!PARTIAL || active) {
        if (do_load && G8_ABL != 8) read_dn(fn, sl, cc, g);
        g8_mfma(acc[mi][ni], fc.a[mi], fc.b[ni].h[0]);
      }
      if (g8_dma_piece(PH, g) >= 0) dma_piece(g8_dma_piece(PH, g));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue.  ONE memory round trip before the first matrix instruction, and nothing the compiler's wait-count model has to
  // guess about: the tile's per-row and per-query constants (accumulator start values; unit, threshold and shift of the queries) travel
  // as four 1 KiB LDS-DMA pieces ahead of the first stage pair and are read from the LDS behind the first barrier.  (Until round 3
  // they were per-lane global loads, each pinned by an empty asm -- four serialised L2 round trips before the first DMA piece was
  // even issued, behind ~400 scalar instructions of 64-bit division: together most of the 2.5 us "per-tile constant" that neither a
  // persistent workgroup nor a skipped first-pair wait had removed in rounds 1-2.)
  if (wave < 4 && (ts > 0 || wave == 1 || wave == 2)) {      // (a dense_i8 index runs here with ts = 0: no row sums, no shifts)
    const char* src = wave == 0 ? (const char*)(p.g8_rsum + dt * TILE_ROWS) : wave == 1 ? (const char*)(p.i8_mul + qt * TILE_ROWS)
                    : wave == 2 ? (const char*)(p.thr + qt * TILE_ROWS) : (const char*)(p.g8_shift + qt * TILE_ROWS);
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + lane_off), LDS_PTR(smem + G8_META + wave * 1024), 16, 0, 0);
  }
  issue_pair(0);
  if (npairs > 1) {            // the constants and pair 0 must have landed; pair 1 (this wave's 5 or 8 pieces, loads return in order) may stay in flight
    issue_pair(1);
    if (dma_n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  G8_T(2);
  // The query's gated operand has 8 bits: level L in [0, 255] is stored as L - 128 (a column that carries nothing as -128), so
  //   sum_cols (stored + 128) * d8 = sum_cols stored * d8 + 128 * (sum of the row's gated int8 values),
  // and the second term is a constant of the ROW: the accumulators start there (g8_rsum, built with the index).
  if (ts > 0) {
    const int32_t* rs = (const int32_t*)(smem + G8_META) + wm * 128 + 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const intx4 v = *(const intx4*)(rs + mi * 32 + 8 * g4);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) acc[mi][0][4 * g4 + i4] = acc[mi][1][4 * g4 + i4] = __int_as_float(v[i4]);
      }
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][0][e] = acc[mi][1][e] = 0.f;
  }
  int sh_r[2];
  float mul_r[2], thr_f[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int ql = wn * 64 + ni * 32 + (lane & 31);
    mul_r[ni] = ((const float*)(smem + G8_META + 1024))[ql];
    thr_f[ni] = ((const float*)(smem + G8_META + 2048))[ql];
    sh_r[ni] = ts > 0 ? ((const int*)(smem + G8_META + 3072))[ql] : 0;
    asm volatile("" : "+v"(mul_r[ni]), "+v"(thr_f[ni]), "+v"(sh_r[ni]));      // in registers from here on: the epilogue reuses the LDS
  }
  G8Frag f0, f1;
  if (ts > 0) {
#pragma unroll
    for (int g = 0; g < 9; ++g) read_s8(f0, smem, g);
  }
  if (G8_ABL == 8) f1 = f0;
  constexpr std::integral_constant<int, 0> PH0{};
  constexpr std::integral_constant<int, 1> PH1{};
  constexpr std::integral_constant<int, 2> PH2{};
  dma_n = 0;             // nothing pending during pair 0's first block (pairs 0 and 1 went out above)
  // (the last pair is peeled so that "read the next fragments" is a compile-time fact in both bodies: as a run-time flag it put a scalar
  // branch and an address computation in front of each of the 12 reads of every pair's second block)
#pragma unroll 1
  for (int g = 0; g + 1 < nsp; ++g) {
    blk_s8(f0, f1, 2 * g + 1, true, PH1);
    dma_prepare(g + 2);
    __builtin_amdgcn_sched_barrier(0);                    // the matrix asm is not volatile: without this the compiler sinks the block above below the wait
    G8_PAIR_SYNC();
    dma_commit();                                         // pair g+2 goes into the ring half this pair just left
    blk_s8(f1, f0, 2 * g + 2, true, PH0);
  }
  if (nsp > 0) {
    const int g = nsp - 1;
    blk_s8(f0, f1, 2 * g + 1, true, PH1);
    dma_prepare(g + 2);
    __builtin_amdgcn_sched_barrier(0);
    G8_PAIR_SYNC();
    dma_commit();
    blk_s8(f1, f0, 2 * g + 2, false, PH0);
  }
  G8_T(3);
  if (td > 0) {
    // gated sums -> ungated units: every accumulator shifted left by its query's shift, then the first ungated fragments
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");     // the last matrix results are in the accumulators
    __builtin_amdgcn_sched_barrier(0);
    if (ts > 0)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = __uint_as_float(__float_as_uint(acc[mi][ni][e]) << sh_r[ni]);
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* sl = smem + (ts & 3) * G8_SLOT;
#pragma unroll
      for (int g = 0; g < 6; ++g) read_dn(f0, sl, c0, g);
    }
    asm volatile("s_nop 4" ::: "memory");      // VALU write -> matrix read of the accumulators
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int g = nsp; g + 1 < npairs; ++g) {
      const int t0 = 4 * (g - nsp);
      blk_dn(f0, f1, t0 + 1, true, PH1);
      blk_dn(f1, f0, t0 + 2, true, PH2);
      blk_dn(f0, f1, t0 + 3, true, PH2);
      dma_prepare(g + 2);
      __builtin_amdgcn_sched_barrier(0);
      G8_PAIR_SYNC();
      dma_commit();
      blk_dn(f1, f0, t0 + 4, true, PH0);
    }
    {
      const int g = npairs - 1;
      const int t0 = 4 * (g - nsp);
      blk_dn(f0, f1, t0 + 1, true, PH1);
      blk_dn(f1, f0, t0 + 2, true, PH2);
      blk_dn(f0, f1, t0 + 3, true, PH2);
      dma_prepare(g + 2);
      __builtin_amdgcn_sched_barrier(0);
      G8_PAIR_SYNC();
      dma_commit();
      blk_dn(f1, f0, t0 + 4, false, PH0);
    }
  }
  G8_T(4);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last matrix results are in the accumulators
  G8_T(5);
  if (DUMP) { g8_dump_tile(p, acc, dt, qt, wm, wn, lane, mul_r); return; }
  int thr_r[2];                // thresholds in accumulator units: a division per query, off the start-up path
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    asm volatile("" : "+v"(thr_f[ni]));
    thr_r[ni] = g8_thr_units(thr_f[ni], mul_r[ni]);
  }
#if G8_ABL == 32      // timing only: no filter epilogue at all
  if (p.n_queries >= 0) { if (__float_as_int(acc[0][0][0]) == 0x7fffffff) p.cnt[0] = 1; return; }
#endif
  if (PARTIAL && !active) { __syncthreads(); return; }      // (the barrier g8_epilogue opens with)
  g8_epilogue(p, acc, dt, qt, wm, wn, (int)threadIdx.x, lane, smem, thr_r, mul_r);
}

// (Until round 6 a build flag G8_VGPR_CAP capped the kernel at 2 x 224 registers per SIMD for the "gathers resident beside the GEMM" experiment of
// DESIGN.md section 4c.  Round 6 re-ran it with the gather launches cut to one workgroup per CU: a capped build of today's kernel no longer returns
// the right candidates (scratch inside the asm-pinned stage loops), and the gathers reach 1.0 TB/s from one wave per SIMD where the step needs 5 --
// docs/experiments.md; the flag is gone.)
#define G8_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
template <bool DUMP>
__global__ void __launch_bounds__(G8_NT) G8_KERNEL_ATTR gemm_filter_g8_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t dt;
  int qt;
  if (!gemm_wg_tile(p, dt, qt)) return;
#if G8_TRACE
  unsigned tslot = 0;
  if (threadIdx.x == 0) tslot = atomicAdd(&g8_trace_n, 1u);
  tslot = __builtin_amdgcn_readfirstlane(tslot);
  if (threadIdx.x == 0 && tslot < G8_TRACE_SLOTS) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g8_trace_buf[(size_t)tslot * 8 + 0] = __builtin_amdgcn_s_memrealtime();
    g8_trace_buf[(size_t)tslot * 8 + 7] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xf) << 32) | ((unsigned long long)qt << 40) | ((unsigned long long)(dt & 0xffff) << 48);
  }
  G8_T(1);
  if (!DUMP && p.partial_wn > 0 && qt == p.n_qtiles - 1) g8_tile<false, true>(p, smem, dt, qt, tslot);
  else g8_tile<DUMP, false>(p, smem, dt, qt, tslot);
  G8_T(6);
#else
  if (!DUMP && p.partial_wn > 0 && qt == p.n_qtiles - 1) g8_tile<false, true>(p, smem, dt, qt);
  else g8_tile<DUMP, false>(p, smem, dt, qt);
#endif
}
#if G8_TRACE
}  // namespace dhr
// tuning hook of the trace build: copies the first `max_slots` trace records (8 x u64 each) to the host and resets the record counter
extern "C" int dhr_debug_g8_trace(unsigned long long* out, int max_slots, unsigned* n_out) try {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(dhr::g8_trace_n), 4) != hipSuccess) return -1;
  if (n_out) *n_out = n;
  const unsigned m = n < (unsigned)max_slots ? n : (unsigned)max_slots;
  const unsigned mm = m < (unsigned)dhr::G8_TRACE_SLOTS ? m : (unsigned)dhr::G8_TRACE_SLOTS;
  if (out && mm && hipMemcpyFromSymbol(out, HIP_SYMBOL(dhr::g8_trace_buf), (size_t)mm * 64) != hipSuccess) return -1;
  const unsigned zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(dhr::g8_trace_n), &zero, 4) != hipSuccess) return -1;
  return 0;
} DHR_CATCH_STATUS
namespace dhr {
#endif

hipError_t launch_gemm_g8(const GemmArgs& a, dim3 grid, hipStream_t s) {
  static std::mutex attr_mu;                       // per-device, under a lock (handles on different devices / host threads)
  static bool attr_set_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  {
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    bool& attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_filter_g8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_RING_LDS);
      if (e != hipSuccess) return e;
      e = hipFuncSetAttribute((const void*)gemm_filter_g8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_RING_LDS);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
  }
  if (a.dump) { hipLaunchKernelGGL(gemm_filter_g8_kernel<true>, grid, dim3(G8_NT), G8_RING_LDS, s, a); return hipGetLastError(); }
  GemmArgs b = a;
  constexpr int partial_on = 1;
  const int valid_last = a.n_queries - (a.n_qtiles - 1) * TILE_ROWS;          // real queries of the batch's last query tile
  b.partial_wn = (partial_on && valid_last > 0 && valid_last <= 128) ? (valid_last + 63) / 64 : 0;
  hipLaunchKernelGGL(gemm_filter_g8_kernel<false>, grid, dim3(G8_NT), G8_RING_LDS, s, b);
  return hipGetLastError();
}

}  // namespace dhr
