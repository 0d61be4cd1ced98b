// Bound GEMM + filter of the 2:4 layout without producer waves: variant 4 (four waves, one per SIMD, 128 x 128 wave tiles)
// and variant 5 (eight waves, two per SIMD, 128 x 64 wave tiles) -- one templated kernel, see the comment below.
#include "gemm_common.h"
#include <mutex>
#include <type_traits>

namespace dhr {

// ------------------------------------------------------------------------------------------ bound GEMM, every wave computes
// Stage images of the operands, 256 x 256 tiles; no producer waves with parked registers (the 12-wave producer / consumer kernel these
// replaced was retired in round 6): EVERY wave holds accumulators and issues its share of the LDS-DMA.
//   NI = 4: four waves, one per SIMD with the SIMD's whole 512-entry register file, 128 x 128 wave tiles (256 accumulators):
//           a third less LDS fragment traffic per multiply-add.  The wave is its SIMD's only instruction stream.
//   NI = 2: eight waves, two per SIMD at 256 registers, 128 x 64 wave tiles: while one wave of a SIMD sits in a DMA issue
//           (an LDS-DMA instruction holds its wave ~60-80 cycles) the other one feeds the matrix pipe.
// Stages are handed over in PAIRS (halves of the 4 x 34 KiB ring): one workgroup barrier per 64 columns.  The DMA of pair
// g+2 goes out behind the barrier of pair g (which frees its ring half) and lands while pair g+1 is computed; the barrier
// sits before the pair's last block of matrix instructions, whose fragments are in registers, and the first fragments of
// the next pair are read under them.  A block (16 columns deep: 4 x NI matrix instructions) is written as 4 * NI issue
// groups "fragment read | 2 expansion VALU | matrix instruction | DMA piece": with one or two instruction streams per SIMD,
// whatever sits between two matrix instructions beyond the ~32 cycles the first one executes idles the matrix pipe
// (measured on the first cut of NI = 4: 8 expansions + 2 DMA pieces in a row between groups of 4 matrix instructions cost
// a third of the pipe's time).
constexpr float I8_MAGIC = 12582912.f;      // 2^23 + 2^22 = 0x4B400000: int32 adds of |sum| < 2^22 on these bits are exact fp32 adds
template <int NI, bool I8>
__device__ __forceinline__ void gemm_dump_tile_w(const GemmArgs& p, floatx16 (&acc)[4][NI], int64_t dt, int qt, int wm, int wn, int lane,
                                                 const float (&mul_r)[NI]) {
  const int fhalf = lane >> 5;
  const int64_t row_base = dt * TILE_ROWS + wm * 128;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31);
    if (q < p.n_queries) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = row_base + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
          if (row < p.n_rows && row >= p.dump_row0 && row - p.dump_row0 < p.dump_ld)
            p.dump[(int64_t)q * p.dump_ld + (row - p.dump_row0)] = I8 ? (acc[mi][ni][e] - I8_MAGIC) * mul_r[ni] : acc[mi][ni][e];
        }
    }
  }
}
// Filter epilogue, same scheme as gemm_epilogue (private hit stacks in the idle ring, one global atomic per thread and
// query), for NT consumer threads holding 4 x NI blocks each.
template <int NI, int NT, bool I8>
__device__ __forceinline__ void gemm_epilogue_w(const GemmArgs& p, floatx16 (&acc)[4][NI], int64_t dt, int qt, int wm, int wn,
                                                int tid, int lane, char* smem, const float (&thr_r)[NI], const float (&mul_r)[NI]) {
  __syncthreads();                       // every wave is done with the staging ring
  const int fhalf = lane >> 5;
  const int64_t row0 = dt * TILE_ROWS;
  const int rows_valid = (int)(p.n_rows - row0 < TILE_ROWS ? p.n_rows - row0 : TILE_ROWS);
  uint2* stack = (uint2*)smem + tid;                       // slot j at stack[j * NT]
  uint32_t j = 0, jn[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31);
    const float t = thr_r[ni];                 // loaded at kernel start: a global load here is ~1 us of exposed latency per tile
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      // Two-level test before the per-element branches: at the bench's hit rate (0.14 % of the accumulators) three of four
      // 32 x 32 blocks hold a hit SOMEWHERE in the wave, so a block-level test alone sends nearly every block through 16 element
      // branches (measured: +10 % on the kernel with an open filter).  The 16 values of a lane are 4 groups of 4 consecutive rows:
      // one v_max3 + v_max per group, element branches only inside a group that holds a hit (30 % of the groups).
      const floatx16& a = acc[mi][ni];
      float gm[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) gm[g] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a[4 * g], a[4 * g + 1]), a[4 * g + 2]), a[4 * g + 3]);
      const float mx = __builtin_fmaxf(__builtin_fmaxf(gm[0], gm[1]), __builtin_fmaxf(gm[2], gm[3]));
      if (mx >= t) {
        asm volatile("");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (gm[g] >= t) {
            asm volatile("");
#pragma unroll
            for (int e = 4 * g; e < 4 * g + 4; ++e) {
              const float v = a[e];
              if (v >= t) {
                asm volatile("");
                const int rl = wm * 128 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
                if (rl < rows_valid) {
                  const float vs = I8 ? (v - I8_MAGIC) * mul_r[ni] : v;          // the candidate lists carry score units
                  if (j < EPI_STACK) stack[j * NT] = make_uint2((uint32_t)rl, __float_as_uint(vs));
                  else {
                    const uint32_t slot = atomicAdd(p.cnt + q, 1u);
                    cand_store<NI != 4>(p, q, slot, make_uint2((uint32_t)(row0 + rl), __float_as_uint(vs)));
                  }
                  ++j;
                }
              }
            }
          }
        }
      }
    }
    jn[ni] = j < EPI_STACK ? j : EPI_STACK;
  }
  if (j == 0) return;
  // all of a lane's list reservations go out before the first one is waited for: one L2 round trip per tile, not NI
  // (A/B on one box, open filter, 2 M rows: 31.58 vs 31.81 ms)
  uint32_t base[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t lo = ni ? jn[ni - 1] : 0u, hi = jn[ni];
    base[ni] = 0u;
    if (hi > lo) base[ni] = atomicAdd(p.cnt + (qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31)), hi - lo);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t lo = ni ? jn[ni - 1] : 0u, hi = jn[ni];
    const int q = qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31);
    for (uint32_t i = lo; i < hi; ++i) {
      const uint2 en = stack[i * NT];
      const uint32_t slot = base[ni] + (i - lo);
      cand_store<NI != 4>(p, q, slot, make_uint2((uint32_t)row0 + en.x, en.y));
    }
  }
}

// Matrix instructions as inline asm with the accumulator pinned to the AGPR half of the register file (with every AGPR holding
// an accumulator the compiler's own allocation of the builtins splits tuples into VGPRs and spills around the loops).  In
// __device__ functions: the host pass of the compiler rejects the constraints inside a kernel body.  Hazards the compiler
// cannot see through the asm: VALU write -> matrix read (the expansion instructions sit BEFORE a matrix instruction that
// does not read them; expand_bucket_columns ends with s_nop 1), matrix write -> VALU read (s_nop padding before the epilogue).
// AV: the accumulator lives in the architectural registers ("v") instead of the accumulation registers ("a").  Two waves per SIMD
// (NI = 2) have 256 registers each and the whole kernel fits the 256 architectural ones: no v_accvgpr_read in front of the
// filter epilogue and 36 registers fewer (218 instead of 126 + 128); NI = 4 needs more than 256 registers, i.e. the "a" file.
template <bool AV>
__device__ __forceinline__ void mfma_f16(floatx16& c, const half8& a, const half8& b) {
  if constexpr (AV) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// int8 stages of a dense_i8 index: the same 16-byte fragments, 32 columns deep; the accumulator registers hold int32 sums until
// the conversion between the dense and the gated stages
template <bool AV>
__device__ __forceinline__ void mfma_i8(floatx16& c, const half8& a, const half8& b) {
  if constexpr (AV) asm("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// Issue group of a sparse block: the two expansion instructions of one register of the NEXT query block's compressed fragment
// (value v, bucket in the sign bit -> the two bucket columns (max(v,0), max(-v,0))), then one matrix instruction of the current one.
#ifndef W4_ABL
#define W4_ABL 0
#endif
// (the constraint letter of the accumulator is the only difference between the AV and the non-AV form)
#define DHR_SM_UNIT_ASM(ACC, ABID)                                                                                              \
  asm("v_pk_max_f16 %1, %6, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"                                                    \
      "v_pk_max_f16 %2, %6, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"                                                    \
      "v_smfmac_f32_32x32x32_f16 %0, %3, %4, %5" ABID                                                                           \
      : ACC(c), "=&v"(o_lo), "=&v"(o_hi) : "v"(a), "v"(b), "v"(idx), "v"(raw))
#define DHR_SM_ONLY_ASM(ACC, ABID) asm("v_smfmac_f32_32x32x32_f16 %0, %1, %2, %3" ABID : ACC(c) : "v"(a), "v"(b), "v"(idx))
template <int KB, bool AV>
__device__ __forceinline__ void sm_unit(floatx16& c, uint32_t& o_lo, uint32_t& o_hi, const half8& a, const half16& b, uint32_t idx, uint32_t raw) {
#if W4_ABL == 7 || W4_ABL == 9      // timing only: no expansion instructions (the expanded blocks keep their prologue contents)
  if constexpr (KB == 0) { if constexpr (AV) DHR_SM_ONLY_ASM("+v", ""); else DHR_SM_ONLY_ASM("+a", ""); }
  else { if constexpr (AV) DHR_SM_ONLY_ASM("+v", " abid:1"); else DHR_SM_ONLY_ASM("+a", " abid:1"); }
  return;
#endif
  if constexpr (KB == 0) { if constexpr (AV) DHR_SM_UNIT_ASM("+v", ""); else DHR_SM_UNIT_ASM("+a", ""); }
  else { if constexpr (AV) DHR_SM_UNIT_ASM("+v", " abid:1"); else DHR_SM_UNIT_ASM("+a", " abid:1"); }
}
#undef DHR_SM_UNIT_ASM
#undef DHR_SM_ONLY_ASM

// timing ablations (wrong results): W4_ABL 1 = no pair barrier, 2 = no DMA wait before it, 3 = neither, 4 = no DMA pieces in the loop,
// 7 = 4 + no expansion instructions, 8 = 4 + no fragment reads in the loop, 9 = 4 + 7 + 8
#if W4_ABL == 1
#define W4_PAIR_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#elif W4_ABL == 2
#define W4_PAIR_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#elif W4_ABL == 3
#define W4_PAIR_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define W4_PAIR_SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#endif

template <int NI>
struct WFrag {             // fragments of one 16-deep block: 4 corpus row blocks, NI query blocks (compressed in sparse stages)
  half8 a[4];
  union { half8 h; uint32_t w[4]; } b[NI];
};
union WExp { half16 h; uint32_t w[8]; };   // one expanded query block

// DMA piece (of this wave's share of a stage pair) issued behind matrix instruction g of the block in phase ph (0: the block
// behind the pair barrier, 1 / 2: the next pair's first two blocks, 3: its third), or -1.
//   NI = 4 (16 instructions per block, <= 18 pieces per wave): 6 + 6 + 6 behind instructions 8-13 (0-7 carry the fragment reads);
//   NI = 2 ( 8 instructions per block, <=  9 pieces per wave): 3 + 3 + 3 behind instructions 2, 4, 6.
#ifndef W5_PARTNER_SHIFT
#define W5_PARTNER_SHIFT 0
#endif
// SH (NI = 2 only, experiment W5_PARTNER_SHIFT): the two waves of a SIMD (waves w and w + 4) issue their pieces in different blocks
// -- SH 0: 5 behind the pair barrier (phase 0) + 4 in phase 1; SH 1: 4 late in phase 1 + 5 in phase 2.
template <int NI, int SH = -1>
__host__ __device__ constexpr int wx_dma_piece(int ph, int g) {
  if (NI == 2 && SH == 0) return ph == 0 ? (g >= 1 && g <= 5 ? g - 1 : -1) : ph == 1 ? (g <= 3 ? 5 + g : -1) : -1;
  if (NI == 2 && SH == 1) return ph == 1 ? (g >= 4 ? g - 4 : -1) : ph == 2 ? (g <= 4 ? 4 + g : -1) : -1;
  if (ph > 2) return -1;
  if (NI == 4) return (g >= 8 && g < 14) ? ph * 6 + g - 8 : -1;
  return (g == 2 || g == 4 || g == 6) ? ph * 3 + (g >> 1) - 1 : -1;
}

// I8 (dense_i8 indexes): the td ungated stages are int8 images (64 columns per stage) and run FIRST, on v_mfma_i32_32x32x32_i8 at
// twice the fp16 instruction's columns per issue; then every accumulator is converted once, fp32(sum) * i8_mul[query], and the
// gated stages accumulate on top in fp32.  Stage unit u of the tile: u < td dense, else sparse stage u - td.
template <bool DUMP, int NI, bool I8>
__global__ void __launch_bounds__(1024 / NI) __attribute__((amdgpu_waves_per_eu(4 / NI, 4 / NI))) gemm_filter_wx_kernel(GemmArgs p) {
  constexpr int NWAVES = 16 / NI;            // 4 or 8
  constexpr int NT = 64 * NWAVES;
  constexpr int NG = 4 * NI;                 // matrix instructions per block
  constexpr int WN = NWAVES / 2;             // waves along the queries
  constexpr int MAXP = NI == 4 ? 18 : 9;     // DMA pieces per wave and stage pair
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t dt;
  int qt;
  if (!gemm_wg_tile(p, dt, qt)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ts = p.ts, tsq = p.ts_q, td = p.td, nst = tsq + td;
  const char* a_src = (const char*)p.a_tiles + dt * ((int64_t)ts * SP_STAGE_A + (int64_t)td * SP_DENSE);
  const char* b_src = (const char*)p.b_tiles + (int64_t)qt * ((int64_t)tsq * SP_STAGE_B + (int64_t)td * SP_DENSE);
  const int wm = wave / WN, wn = wave % WN;
  const int npairs = nst >> 1, nsp = tsq >> 1;     // pairs are homogeneous: the launcher selects this kernel only when tsq and td are even
  const int u_sp0 = I8 ? td : 0;                   // first sparse stage unit; dense units are [tsq, nst) or, I8, [0, td)
  const int sp_lo = u_sp0 >> 1, sp_hi = sp_lo + nsp, dn_lo = I8 ? 0 : nsp, dn_hi = I8 ? (td >> 1) : npairs;

  // ---- LDS-DMA, fixed roles.  NI = 4: wave w streams the whole image (w & 1 ? query : corpus) of stage 2g + (w >> 1);
  // NI = 2: wave w streams half (w & 1) of image ((w >> 1) & 1 ? query : corpus) of stage 2g + (w >> 2).  Buffer form of the
  // instruction: scalar base (resource) + scalar offset + one constant lane-offset register + immediate; the immediate is
  // added to the global AND to the LDS address (M0 + immediate + 16 * lane), so M0 and the scalar offset change every 4th piece.
  const bool dma_b = NI == 4 ? (wave & 1) != 0 : ((wave >> 1) & 1) != 0;
  const int dma_s = NI == 4 ? wave >> 1 : wave >> 2;
  const int dma_h = NI == 4 ? 0 : (wave & 1);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const int role_sp = dma_b ? SP_STAGE_B : SP_STAGE_A;
  const int role_dense0 = dma_b ? tsq * SP_STAGE_B : ts * SP_STAGE_A;
  const uint32_t smem_u = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const __amdgpu_buffer_rsrc_t role_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(dma_b ? b_src : a_src), (short)0, 0x7fffffff, 0x00020000);
  int dma_soff = 0, dma_n = 0, nx_soff = 0, nx_n = 0;
  uint32_t dma_lds = 0, nx_lds = 0;
  auto dma_prepare = [&](int g) __attribute__((always_inline)) {      // this wave's source / destination / piece count of pair g
    const int u = 2 * g + dma_s;
    const bool sp = u >= u_sp0 && u < u_sp0 + tsq;
    const int usp = u - u_sp0;
    const int us = (!dma_b && usp >= ts) ? usp - ts : usp;             // an ungated batch runs the corpus's sparse stages twice
    const int half_bytes = NI == 4 ? 0 : dma_h * ((!dma_b && sp) ? SP_STAGE_A / 2 : 8192);
    nx_soff = (sp ? us * role_sp : role_dense0 + (I8 ? u : u - tsq) * SP_DENSE) + half_bytes;
    nx_lds = smem_u + (uint32_t)((u & 3) * SP_SLOT + (dma_b ? SP_STAGE_A : 0) + half_bytes);
    nx_n = g < npairs ? ((!dma_b && sp) ? MAXP : MAXP - MAXP / 9) : 0;    // 18 / 16 or 9 / 8 KiB
    if ((W4_ABL == 4 || W4_ABL >= 7) && g >= 2) nx_n = 0;
  };
  auto dma_commit = [&]() __attribute__((always_inline)) { dma_soff = nx_soff; dma_lds = nx_lds; dma_n = nx_n; };
  auto dma_piece = [&](int j) __attribute__((always_inline)) {
#if W4_ABL == 5      // timing only: the same bytes as plain loads into registers that nobody reads (no LDS write)
    if (j < dma_n) { typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(role_rsrc, (int)lane_off + (j & 3) * 1024, dma_soff + (j >> 2) * 4096, 0); asm volatile("" :: "v"(x)); }
    return;
#endif
    if (j < dma_n) {
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(uintptr_t)(dma_lds + (uint32_t)(j >> 2) * 4096u);
      const int so = W4_ABL == 6 ? 0 : dma_soff + (j >> 2) * 4096;      // 6 (timing only): always the tile's first 4 KiB, L1-resident
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
    for (int j = 0; j < MAXP; ++j) dma_piece(j);
  };

  floatx16 acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = I8 ? I8_MAGIC : 0.f;
  float thr_r[NI];                                // this lane's query thresholds for the filter epilogue and (I8) accumulator units ->
  float mul_r[NI];                                // score units of its queries: read from the LDS behind the first barrier (prologue below)
  // Per-lane LDS offsets inside a ring slot.  All three images (corpus values, query values, dense columns of either side) have
  // 64-byte rows with the 16-byte chunk c stored at c ^ ((row>>2)&3); 16-slice / 16-column block kb is chunk kb*2 + fhalf.
  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int swz4 = (frow >> 2) & 3;
  const int a_row = (wm * 128 + frow) * 64;                         // + mi*2048
  const int q_row = SP_STAGE_A + (wn * (32 * NI) + frow) * 64;      // + ni*2048
  const int c0 = (fhalf ^ swz4) << 4, c1 = ((2 + fhalf) ^ swz4) << 4;
  const int p_off = SP_A_BYTES + ((wm * 8 + fhalf) * 32 + frow) * 4; // position words, + mi*256

  // Position words of a sparse stage (one u32 per 32-row block: low half = 16-slice block 0, high half = block 1), two sets:
  // even / odd stage of a pair.
  uint32_t pwx[4] = {0u, 0u, 0u, 0u}, pwy[4] = {0u, 0u, 0u, 0u};
  auto load_pw = [&](uint32_t (&pw)[4], int u) __attribute__((always_inline)) {
    const char* sl = smem + (u & 3) * SP_SLOT;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) pw[mi] = *(const uint32_t*)(sl + p_off + mi * 256);
  };
  // block t of the tile = (stage t >> 1, 16-deep half t & 1); fragment read number g of the NI + 4 of a block (query blocks first:
  // the last group of the block before already expands query block 0)
  auto frag_read = [&](WFrag<NI>& f, const char* sl, int g) __attribute__((always_inline)) {
    if (g < NI) f.b[g].h = *(const half8*)(sl + q_row + g * 2048);
    else if (g < NI + 4) f.a[g - NI] = *(const half8*)(sl + a_row + (g - NI) * 2048);
  };
  WExp bfa, bfb;       // expanded query block: bfa holds the CURRENT block's ni = 0 expansion on entry of a sparse block
  auto blk_sparse = [&](const WFrag<NI>& fc, WFrag<NI>& fn, int tn, bool do_load, auto kb_c, auto ph_c, const uint32_t (&pw)[4], auto sh_c) __attribute__((always_inline)) {
    constexpr int SH = decltype(sh_c)::value;
    constexpr int KB = decltype(kb_c)::value;
    constexpr int PH = decltype(ph_c)::value;
    const char* sl = smem + ((tn >> 1) & 3) * SP_SLOT + ((tn & 1) ? c1 : c0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (do_load && W4_ABL != 8 && W4_ABL != 9) frag_read(fn, sl, g);
      WExp& cur = (ni & 1) ? bfb : bfa;
      WExp& oth = (ni & 1) ? bfa : bfb;
      const uint32_t rw = ni < NI - 1 ? fc.b[ni < NI - 1 ? ni + 1 : 0].w[mi] : fn.b[0].w[mi];
      sm_unit<KB, NI == 2>(acc[mi][ni], oth.w[2 * mi], oth.w[2 * mi + 1], fc.a[mi], cur.h, pw[mi], rw);
      if (wx_dma_piece<NI, SH>(PH, g) >= 0) dma_piece(wx_dma_piece<NI, SH>(PH, g));
    }
  };
  auto blk_dense = [&](const WFrag<NI>& fc, WFrag<NI>& fn, int tn, bool do_load, auto ph_c, auto sh_c) __attribute__((always_inline)) {
    constexpr int SH = decltype(sh_c)::value;
    constexpr int PH = decltype(ph_c)::value;
    const char* sl = smem + ((tn >> 1) & 3) * SP_SLOT + ((tn & 1) ? c1 : c0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (do_load && W4_ABL != 8 && W4_ABL != 9) frag_read(fn, sl, g);
      if constexpr (I8) mfma_i8<NI == 2>(acc[mi][ni], fc.a[mi], fc.b[ni].h); else mfma_f16<NI == 2>(acc[mi][ni], fc.a[mi], fc.b[ni].h);
      if (wx_dma_piece<NI, SH>(PH, g) >= 0) dma_piece(wx_dma_piece<NI, SH>(PH, g));
    }
  };

  // Prologue: ONE memory round trip before the first matrix instruction.  The per-query constants of the tile (thresholds; I8: units)
  // travel as 1 KiB LDS-DMA pieces ahead of the first stage pair and are read from the LDS behind the first barrier -- until round 3
  // they were per-lane global loads pinned by an empty asm, i.e. serialised L2 round trips in front of the first DMA piece.
  if (wave < (I8 ? 2 : 1)) {
    const char* src = wave == 0 ? (const char*)(p.thr + qt * TILE_ROWS) : (const char*)(p.i8_mul + qt * TILE_ROWS);
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + lane_off), LDS_PTR(smem + WX_META + wave * 1024), 16, 0, 0);
  }
  issue_pair(0);
  if (npairs > 1) {            // the constants and pair 0 landed; pair 1 (MAXP - MAXP/9 or MAXP pieces per wave) may stay in flight
    issue_pair(1);
    if constexpr (NI == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int ql = wn * (32 * NI) + ni * 32 + (lane & 31);
    thr_r[ni] = ((const float*)(smem + WX_META))[ql];
    if constexpr (I8) {
      // the accumulators run in units of mul (corpus scale x query scale) above I8_MAGIC; the roundings here and in the fp32
      // accumulation on top of the 2^23-sized offset are paid by the filter margin (query_prep_kernel)
      mul_r[ni] = ((const float*)(smem + WX_META + 1024))[ql];
      thr_r[ni] = fmaf(thr_r[ni], 1.f / mul_r[ni], I8_MAGIC);
      asm volatile("" : "+v"(mul_r[ni]));
    } else mul_r[ni] = 1.f;
    asm volatile("" : "+v"(thr_r[ni]));          // in registers from here on: the epilogue reuses the LDS
  }
  WFrag<NI> f0, f1;
  if (nsp > 0 && sp_lo == 0) load_pw(pwx, 0);
  {
    const char* sl = smem + c0;
#pragma unroll
    for (int g = 0; g < NI + 4; ++g) frag_read(f0, sl, g);
  }
  if (nsp > 0 && sp_lo == 0) expand_bucket_columns(f0.b[0].w, bfa.w);
  if (W4_ABL >= 7) { if (nsp > 0) expand_bucket_columns(f0.b[1].w, bfb.w); f1 = f0; }     // the ablations reuse these registers for the whole tile
  constexpr std::integral_constant<int, 0> KB0{};
  constexpr std::integral_constant<int, 1> KB1{};
  constexpr std::integral_constant<int, 0> PH0{};
  constexpr std::integral_constant<int, 1> PH1{};
  constexpr std::integral_constant<int, 2> PH2{};
  constexpr std::integral_constant<int, 3> PH3{};
  dma_n = 0;             // nothing pending during pair 0's first blocks (pairs 0 and 1 went out above)
  // The tile's LAST pair is peeled out of its loop, so that "read the next block's fragments" is a compile-time fact in both copies of
  // the body: as a run-time flag it put a scalar branch and an address computation in front of every read of each pair's last block
  // (gemm_g8.hip, measured there: -4 % per launch).
  constexpr std::integral_constant<bool, true> MORE{};
  constexpr std::integral_constant<bool, false> LAST{};
  auto run_sparse = [&](auto sh_c) __attribute__((always_inline)) {
    auto pair = [&](int g, auto more_c) __attribute__((always_inline)) {
      constexpr bool more = decltype(more_c)::value;
      const int t0 = 4 * g;
      blk_sparse(f0, f1, t0 + 1, true, KB0, PH1, pwx, sh_c);
      blk_sparse(f1, f0, t0 + 2, true, KB1, PH2, pwx, sh_c);
      load_pw(pwy, 2 * g + 1);
      blk_sparse(f0, f1, t0 + 3, true, KB0, PH3, pwy, sh_c);
      // every read of this pair's ring half has been issued; the next pair must have landed before anybody reads it
      dma_prepare(g + 2);
      __builtin_amdgcn_sched_barrier(0);                    // the matrix asm is not volatile: without this the compiler sinks the block above below the wait
      W4_PAIR_SYNC();
      dma_commit();                                         // pair g+2 goes into the ring half this pair just left
      if (g + 1 < sp_hi) load_pw(pwx, 2 * (g + 1));
      blk_sparse(f1, f0, t0 + 4, more, KB1, PH0, pwy, sh_c);
    };
    const bool tail = sp_hi == npairs && sp_hi > sp_lo;
#pragma unroll 1
    for (int g = sp_lo; g < sp_hi - (tail ? 1 : 0); ++g) pair(g, MORE);
    if (tail) pair(sp_hi - 1, LAST);
  };
  auto run_dense = [&](auto sh_c) __attribute__((always_inline)) {
    auto pair = [&](int g, auto more_c) __attribute__((always_inline)) {
      constexpr bool more = decltype(more_c)::value;
      const int t0 = 4 * g;
      blk_dense(f0, f1, t0 + 1, true, PH1, sh_c);
      blk_dense(f1, f0, t0 + 2, true, PH2, sh_c);
      blk_dense(f0, f1, t0 + 3, true, PH3, sh_c);
      dma_prepare(g + 2);
      __builtin_amdgcn_sched_barrier(0);
      W4_PAIR_SYNC();
      dma_commit();
      blk_dense(f1, f0, t0 + 4, more, PH0, sh_c);
    };
    const bool tail = dn_hi == npairs && dn_hi > dn_lo;
#pragma unroll 1
    for (int g = dn_lo; g < dn_hi - (tail ? 1 : 0); ++g) pair(g, MORE);
    if (tail) pair(dn_hi - 1, LAST);
  };
  auto run_loops = [&](auto sh_c) __attribute__((always_inline)) {
    if constexpr (I8) {
      run_dense(sh_c);
      // No conversion: the accumulators were started at the bit pattern of 2^23 + 2^22 (I8_MAGIC), and while |sum| < 2^22 an int32
      // add on those bits IS the fp32 number I8_MAGIC + sum -- the gated stages go on accumulating in fp32 on the same registers.
      // f0 already holds the first sparse block's fragments (read under the last dense block).
      if (nsp > 0) {
        load_pw(pwx, u_sp0);
        expand_bucket_columns(f0.b[0].w, bfa.w);
      }
      run_sparse(sh_c);
    } else {
      run_sparse(sh_c);
      run_dense(sh_c);
    }
  };
  if constexpr (NI == 2 && W5_PARTNER_SHIFT) {
    if (wave < 4) run_loops(std::integral_constant<int, 0>{}); else run_loops(std::integral_constant<int, 1>{});
  } else run_loops(std::integral_constant<int, -1>{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last matrix results are in the accumulators
  if (DUMP) { gemm_dump_tile_w<NI, I8>(p, acc, dt, qt, wm, wn, lane, mul_r); return; }
  gemm_epilogue_w<NI, NT, I8>(p, acc, dt, qt, wm, wn, (int)threadIdx.x, lane, smem, thr_r, mul_r);
}

hipError_t launch_gemm_wx(const GemmArgs& a, dim3 grid, int variant, hipStream_t s) {
  static std::mutex attr_mu;                       // per-device, under a lock (handles on different devices / host threads)
  static bool attr_set_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  bool& attr_set = attr_set_dev[dev_ & 63];
  if (!attr_set) {
    const void* fns[8] = {(const void*)gemm_filter_wx_kernel<false, 4, false>, (const void*)gemm_filter_wx_kernel<true, 4, false>,
                          (const void*)gemm_filter_wx_kernel<false, 2, false>, (const void*)gemm_filter_wx_kernel<true, 2, false>,
                          (const void*)gemm_filter_wx_kernel<false, 4, true>, (const void*)gemm_filter_wx_kernel<true, 4, true>,
                          (const void*)gemm_filter_wx_kernel<false, 2, true>, (const void*)gemm_filter_wx_kernel<true, 2, true>};
    for (const void* f : fns) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_RING_LDS);
      if (e != hipSuccess) return e;
    }
    attr_set = true;
  }
#define WX_LAUNCH(NI_, NT_) do { \
    if (a.i8_mul) { if (a.dump) hipLaunchKernelGGL((gemm_filter_wx_kernel<true, NI_, true>), grid, dim3(NT_), GEMM_RING_LDS, s, a); \
                    else hipLaunchKernelGGL((gemm_filter_wx_kernel<false, NI_, true>), grid, dim3(NT_), GEMM_RING_LDS, s, a); } \
    else { if (a.dump) hipLaunchKernelGGL((gemm_filter_wx_kernel<true, NI_, false>), grid, dim3(NT_), GEMM_RING_LDS, s, a); \
           else hipLaunchKernelGGL((gemm_filter_wx_kernel<false, NI_, false>), grid, dim3(NT_), GEMM_RING_LDS, s, a); } } while (0)
  if (variant == 4) WX_LAUNCH(4, 256); else WX_LAUNCH(2, 512);
#undef WX_LAUNCH
  return hipGetLastError();
}

}  // namespace dhr
