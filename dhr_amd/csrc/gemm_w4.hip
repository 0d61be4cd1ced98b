// Bound GEMM + filter, 4-wave variant (one wave per SIMD, 128 x 128 wave tiles) -- see the comment below.
#include "gemm_common.h"
#include <type_traits>

namespace dhr {

// ------------------------------------------------------------------------------------------ 4-wave bound GEMM (variant 4)
// Same operand images, same 256 x 256 tile, but FOUR waves per workgroup -- one per SIMD, each with the SIMD's whole
// 512-entry register file: a 128 x 128 wave tile (256 fp32 accumulators) and two sets of fragment registers.  What it
// buys over the 12-wave producer / consumer kernel above: a third less LDS fragment traffic per multiply-add (the wave
// tile is square), no parked producer registers, and one workgroup barrier per PAIR of 32-column stages (64 matrix
// instructions per wave between barriers instead of 16).  The stage ring is the same 4 x 34 KiB; stages are handed over
// in pairs (ring halves), the LDS-DMA of pair g+1 is issued by the four waves with fixed roles (wave w: stage w>>1 of the
// pair, corpus image for even w, query image for odd w) right after the barrier that frees its ring half, and lands while
// pair g is computed.  The barrier of pair g sits before the pair's last 16 matrix instructions, and the first fragments of
// pair g+1 are read under them.
constexpr int GEMM_W4_THREADS = 256;
template <int NI>
__device__ __forceinline__ void gemm_dump_tile_w(const GemmArgs& p, floatx16 (&acc)[4][NI], int64_t dt, int qt, int wm, int wn, int lane) {
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
            p.dump[(int64_t)q * p.dump_ld + (row - p.dump_row0)] = acc[mi][ni][e];
        }
    }
  }
}
// Filter epilogue, same scheme as gemm_epilogue (private hit stacks in the idle ring, one global atomic per thread and
// query), for NT consumer threads holding 4 x NI blocks each.
template <int NI, int NT>
__device__ __forceinline__ void gemm_epilogue_w(const GemmArgs& p, floatx16 (&acc)[4][NI], int64_t dt, int qt, int wm, int wn,
                                                int tid, int lane, char* smem) {
  __syncthreads();                       // every wave is done with the staging ring
  const int fhalf = lane >> 5;
  const int64_t row0 = dt * TILE_ROWS;
  const int rows_valid = (int)(p.n_rows - row0 < TILE_ROWS ? p.n_rows - row0 : TILE_ROWS);
  uint2* stack = (uint2*)smem + tid;                       // slot j at stack[j * NT]
  uint32_t j = 0, jn[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int q = qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31);
    const float t = p.thr[q];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const floatx16& a = acc[mi][ni];
      const float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
      const float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
      const float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
      const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3)), __builtin_fmaxf(m4, a[15]));
      if (mx >= t) {
        asm volatile("");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = a[e];
          if (v >= t) {
            asm volatile("");
            const int rl = wm * 128 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
            if (rl < rows_valid) {
              if (j < EPI_STACK) stack[j * NT] = make_uint2((uint32_t)rl, __float_as_uint(v));
              else {
                const uint32_t slot = atomicAdd(p.cnt + q, 1u);
                if (slot < p.cap) p.cand[(int64_t)q * p.cap + slot] = make_uint2((uint32_t)(row0 + rl), __float_as_uint(v));
              }
              ++j;
            }
          }
        }
      }
    }
    jn[ni] = j < EPI_STACK ? j : EPI_STACK;
  }
  if (j == 0) return;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const uint32_t lo = ni ? jn[ni - 1] : 0u, hi = jn[ni];
    if (hi > lo) {
      const int q = qt * TILE_ROWS + wn * (32 * NI) + ni * 32 + (lane & 31);
      const uint32_t base = atomicAdd(p.cnt + q, hi - lo);
      for (uint32_t i = lo; i < hi; ++i) {
        const uint2 en = stack[i * NT];
        const uint32_t slot = base + (i - lo);
        if (slot < p.cap) p.cand[(int64_t)q * p.cap + slot] = make_uint2((uint32_t)row0 + en.x, en.y);
      }
    }
  }
}

// Matrix instructions as inline asm with the accumulator pinned to the AGPR half of the register file: with all 256 AGPRs
// holding accumulators the compiler's own allocation of the builtins splits tuples into VGPRs and spills around the loops.
// Hazards the compiler cannot see through the asm: VALU write -> matrix read (expand_bucket_columns ends with s_nop 1),
// matrix write -> VALU read of the accumulators (s_nop padding before the epilogue).
__device__ __forceinline__ void smfmac_kb0(floatx16& c, const half8& a, const half16& b, uint32_t idx) {
  asm("v_smfmac_f32_32x32x32_f16 %0, %1, %2, %3" : "+a"(c) : "v"(a), "v"(b), "v"(idx));
}
__device__ __forceinline__ void smfmac_kb1(floatx16& c, const half8& a, const half16& b, uint32_t idx) {
  asm("v_smfmac_f32_32x32x32_f16 %0, %1, %2, %3 abid:1" : "+a"(c) : "v"(a), "v"(b), "v"(idx));
}
__device__ __forceinline__ void mfma_f16(floatx16& c, const half8& a, const half8& b) {
  asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// timing ablations (wrong results): W4_ABL 1 = no pair barrier, 2 = no DMA wait before it, 3 = neither, 4 = no DMA pieces in the loop
#ifndef W4_ABL
#define W4_ABL 0
#endif
#if W4_ABL == 1
#define W4_PAIR_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#elif W4_ABL == 2
#define W4_PAIR_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#elif W4_ABL == 3
#define W4_PAIR_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define W4_PAIR_SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#endif
#define W4_PREP(g) do { if constexpr (W4_DMA_MODE == 3) { nx_off = role_goff((g) + 3, nx_18); } else dma_prepare((g) + 2); } while (0)
#define W4_COMMIT(g) do { if constexpr (W4_DMA_MODE == 3) { \
    wr_on = (g) + 2 < npairs; wr_off = (uint32_t)(((2 * ((g) + 2) + dma_s) & 3) * SP_SLOT + (dma_b ? SP_STAGE_A : 0)) + lane_off; \
    wr_18 = wr_on && ld_18; ld_gp = role_src + nx_off; ld_18 = nx_18; } else dma_commit(); } while (0)
#ifndef W4_BUFFER_DMA
#define W4_BUFFER_DMA 1
#endif
#ifndef W4_DMA_MODE
#define W4_DMA_MODE 0
#endif
// DMA piece issued behind matrix instruction g (0..15) of the block with phase PH (0: the block right behind the pair barrier,
// 1 / 2: the next pair's first two blocks), or -1.  Mode 0: 6 + 6 + 6 in groups 8-13; mode 1: 16 in phase 0 (one per group), 2 in phase 1.
__host__ __device__ constexpr int w4_dma_piece(int ph, int g) {
#if W4_DMA_MODE >= 2
  return -1;    // staggered schedule: see w4_dma_gap below
#elif W4_DMA_MODE == 0
  return (g >= 8 && g < 14) ? ph * 6 + g - 8 : -1;
#else
  return ph == 0 ? g : (ph == 1 && g >= 8 && g < 10 ? 16 + g - 8 : -1);
#endif
}
// Issue group of a sparse block: the two expansion instructions of one register of the NEXT query block's compressed fragment
// (value v, bucket in the sign bit -> the two bucket columns (max(v,0), max(-v,0))), then one matrix instruction of the current
// one.  (In a __device__ function: the host pass of the compiler rejects the constraints inside the kernel body itself.)
template <int KB>
__device__ __forceinline__ void sm_unit(floatx16& c, uint32_t& o_lo, uint32_t& o_hi, const half8& a, const half16& b, uint32_t idx, uint32_t raw) {
  if constexpr (KB == 0)
    asm("v_pk_max_f16 %1, %6, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
        "v_pk_max_f16 %2, %6, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
        "v_smfmac_f32_32x32x32_f16 %0, %3, %4, %5"
        : "+a"(c), "=&v"(o_lo), "=&v"(o_hi) : "v"(a), "v"(b), "v"(idx), "v"(raw));
  else
    asm("v_pk_max_f16 %1, %6, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
        "v_pk_max_f16 %2, %6, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
        "v_smfmac_f32_32x32x32_f16 %0, %3, %4, %5 abid:1"
        : "+a"(c), "=&v"(o_lo), "=&v"(o_hi) : "v"(a), "v"(b), "v"(idx), "v"(raw));
}

struct W4Frag {            // fragments of one 16-deep block: 4 corpus row blocks, 4 query blocks (compressed in sparse stages)
  half8 a[4];
  union { half8 h; uint32_t w[4]; } b[4];
};

template <bool DUMP>
__global__ void __launch_bounds__(GEMM_W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_filter_w4_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t b = blockIdx.x;
  const int xcd = (int)(b & 7);
  const int64_t i = b >> 3;
  const int per_group = DOC_GROUP * p.n_qtiles;
  const int64_t g_local = i / per_group;
  const int r = (int)(i - g_local * per_group);
  const int qt = r / DOC_GROUP;
  const int dl = r - qt * DOC_GROUP;
  const int64_t seq = p.seq_lo + (g_local * 8 + xcd) * DOC_GROUP + dl;
  if (seq >= p.seq_hi) return;
  const int64_t dt = seq_to_tile(seq, p.map_mode, p.period, p.head);
  if (dt >= p.n_tiles) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ts = p.ts, tsq = p.ts_q, td = p.td, nst = tsq + td;
  const char* a_src = (const char*)p.a_tiles + dt * ((int64_t)ts * SP_STAGE_A + (int64_t)td * SP_DENSE);
  const char* b_src = (const char*)p.b_tiles + (int64_t)qt * ((int64_t)tsq * SP_STAGE_B + (int64_t)td * SP_DENSE);
  const char* a_dense = a_src + (int64_t)ts * SP_STAGE_A;
  const char* b_dense = b_src + (int64_t)tsq * SP_STAGE_B;
  const int wm = wave >> 1, wn = wave & 1;
  const int npairs_ = nst >> 1;

  // ---- LDS-DMA, fixed roles: this wave streams image (wave & 1 ? query : corpus) of stage 2*g + (wave >> 1) of pair g
  const bool dma_b = (wave & 1) != 0;
  const int dma_s = wave >> 1;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // dma_setup(g) fixes this wave's source / destination of pair g; dma_piece(j) issues the j-th KiB (three instructions:
  // address add, M0, the load).  The pieces are spread over the matrix instructions of three blocks (6 + 6 + 6): a burst
  // of 18 DMA instructions would hold the wave's only instruction stream -- and with it the SIMD's matrix pipe.
  const char* dma_gp = nullptr;
  uint32_t dma_lds = 0;
  int dma_n = 0;
  const char* nx_gp = nullptr;        // the state of the NEXT pair to go out is computed ahead of the barrier that frees its ring half
  uint32_t nx_lds = 0;
  int nx_n = 0;
  int dma_soff = 0, nx_soff = 0;
  const char* const role_src = (dma_b ? b_src : a_src) + lane_off;
  const int64_t role_sp = dma_b ? SP_STAGE_B : SP_STAGE_A;
  const int64_t role_dense0 = dma_b ? (int64_t)tsq * SP_STAGE_B : (int64_t)ts * SP_STAGE_A;
  const uint32_t smem_u = (uint32_t)(uintptr_t)LDS_PTR(smem);
  auto dma_prepare = [&](int g) __attribute__((always_inline)) {
    const int u = 2 * g + dma_s;
    const bool sp = u < tsq;
    const int us = (!dma_b && u >= ts) ? u - ts : u;          // an ungated batch runs the corpus's sparse stages twice
    const int64_t off = sp ? (int64_t)us * role_sp : role_dense0 + (int64_t)(u - tsq) * SP_DENSE;
    nx_gp = role_src + off;
    nx_soff = (int)off;
    nx_lds = smem_u + (uint32_t)((u & 3) * SP_SLOT + (dma_b ? SP_STAGE_A : 0));
    nx_n = g < npairs_ ? ((!dma_b && sp) ? 18 : 16) : 0;
  };
  int dma_res = -1;
  auto dma_commit = [&]() __attribute__((always_inline)) { dma_gp = nx_gp; dma_soff = nx_soff; dma_lds = nx_lds; dma_n = W4_ABL == 4 ? 0 : nx_n; dma_res = dma_n > 0 ? wave % 3 : -1; };
  auto dma_setup = [&](int g) __attribute__((always_inline)) { dma_prepare(g); dma_commit(); };
  // buffer form of the LDS-DMA: scalar base (resource) + scalar pair offset + constant lane offset register + immediate --
  // no per-piece vector address arithmetic, one address register per lane instead of two
  const __amdgpu_buffer_rsrc_t role_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(dma_b ? b_src : a_src), (short)0, 0x7fffffff, 0x00020000);
  auto dma_piece = [&](int j) __attribute__((always_inline)) {
#if W4_BUFFER_DMA
    if (j < dma_n) {
      // the instruction's immediate offset is added to the global AND to the LDS address (M0 + offset + 16 * lane)
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(uintptr_t)(dma_lds + (uint32_t)(j >> 2) * 4096u);
      const int so = dma_soff + (j >> 2) * 4096;
      switch (j & 3) {      // the immediate must be a literal
        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 0, 0); break;
        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 1024, 0); break;
        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 2048, 0); break;
        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(role_rsrc, l, 16, (int)lane_off, so, 3072, 0); break;
      }
    }
#else
    if (j < dma_n) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(dma_gp + j * 1024), (__attribute__((address_space(3))) void*)(uintptr_t)(dma_lds + (uint32_t)j * 1024u), 16, 0, 0);
#endif
  };
  auto issue_pair = [&](int g) {
    dma_setup(g);
#pragma unroll
    for (int j = 0; j < 18; ++j) dma_piece(j);
  };

  // ---- mode 3: register-staged operand stream.  An LDS-DMA instruction holds the issuing wave ~60-80 cycles (measured: with
  // all of them removed from the loop the kernel runs in 23.8 instead of 38.5 ms), which a wave that is its SIMD's only
  // matrix-instruction stream cannot hide.  A plain global_load_dwordx4 retires from the issue port at once and a
  // ds_write_b128 in ~13 cycles, so the stream goes global -> 18 x 4 staging registers -> LDS: the loads of pair g+3 are
  // issued during pair g+1 (behind the stores that free their registers), the stores of pair g+2 behind the barrier of pair g.
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 stg[18];
  const char* ld_gp = role_src;       // source of the pair being loaded into the staging registers
  bool ld_18 = false;
  uint32_t wr_off = 0;                // LDS byte offset (from smem) of the pair being stored, this lane
  bool wr_on = false, wr_18 = false;
  int64_t nx_off = 0;
  bool nx_18 = false;
  auto role_goff = [&](int g, bool& is18) __attribute__((always_inline)) -> int64_t {
    int u = 2 * g + dma_s;
    if (u > nst - 1) u = nst - 2 + dma_s;                       // past the end: a valid address (the data is never stored)
    const bool sp = u < tsq;
    const int us = (!dma_b && u >= ts) ? u - ts : u;
    is18 = !dma_b && sp;
    return sp ? (int64_t)us * role_sp : role_dense0 + (int64_t)(u - tsq) * SP_DENSE;
  };
  auto stg_load = [&](int j) __attribute__((always_inline)) {
    // roles with 16 pieces re-read a valid address for the last two (unconditional loads: a branch around a load costs a full drain)
    stg[j] = *(const u32x4*)(ld_gp + ((j >= 16 && !ld_18) ? 15 : j) * 1024);
  };
  auto stg_store = [&](int j) __attribute__((always_inline)) {
    if (j < 16) { if (wr_on) *(u32x4*)(smem + wr_off + j * 1024) = stg[j]; }
    else if (wr_18) *(u32x4*)(smem + wr_off + j * 1024) = stg[j];
  };
  // phase PH (0: the block behind the pair barrier, 1-3: the next pair's blocks), matrix instruction g -> staging work
  auto xfer = [&](auto ph_c, int g) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;
    if constexpr (PH == 0) { if (g & 1) stg_store(g >> 1); }                       // stores 0-7
    else if constexpr (PH == 1) { if (g < 10) stg_store(8 + g); if (g >= 8) stg_load(g - 8); }   // stores 8-17, loads 0-7
    else if constexpr (PH == 2) { if (g < 10) stg_load(8 + g); }                  // loads 8-17
  };

  floatx16 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int swz4 = (frow >> 2) & 3;
  const int a_row = (wm * 128 + frow) * 64;                         // + mi*2048
  const int q_row = SP_STAGE_A + (wn * 128 + frow) * 64;            // + ni*2048
  const int c0 = (fhalf ^ swz4) << 4, c1 = ((2 + fhalf) ^ swz4) << 4;
  const int p_off = SP_A_BYTES + ((wm * 8 + fhalf) * 32 + frow) * 4; // position words, + mi*256

  // block t of the tile = (stage t >> 1, 16-deep half t & 1)
  auto load_frag = [&](W4Frag& f, int t) __attribute__((always_inline)) {
    const char* sl = smem + ((t >> 1) & 3) * SP_SLOT;
    const int c = (t & 1) ? c1 : c0;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.b[ni].h = *(const half8*)(sl + q_row + c + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) f.a[mi] = *(const half8*)(sl + a_row + c + mi * 2048);
  };
  // Position words of a sparse stage (one u32 per 32-row block: low half = 16-slice block 0, high half = block 1), two sets:
  // even / odd stage of a pair.
  uint32_t pwx[4] = {0u, 0u, 0u, 0u}, pwy[4] = {0u, 0u, 0u, 0u};
  auto load_pw = [&](uint32_t (&pw)[4], int u) __attribute__((always_inline)) {
    const char* sl = smem + (u & 3) * SP_SLOT;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) pw[mi] = *(const uint32_t*)(sl + p_off + mi * 256);
  };
  // One 16-instruction block, written as 16 issue groups "fragment read | 2 expansion VALU | matrix instruction | DMA piece":
  // the wave is the SIMD's only instruction stream, so whatever sits between two matrix instructions beyond the ~32 cycles the
  // first one executes leaves the matrix pipe idle (measured on the first version of this kernel: 8 expansions + 2 DMA pieces
  // in a row between groups of 4 matrix instructions cost a third of the pipe's time).  Per group: the g-th of the 8 fragment
  // reads of the NEXT block (groups 0-7), the two v_pk_max_f16 that expand one register of the next query block's compressed
  // fragment (sparse stages), one matrix instruction, and in groups 8-13 one DMA piece of the pair that is going out.
  union BF { half16 h; uint32_t w[8]; };
  BF bfa, bfb;       // expanded query block: bfa holds the CURRENT block's ni = 0 expansion on entry of a sparse block
  auto blk_sparse = [&](const W4Frag& fc, W4Frag& fn, int tn, bool do_load, auto kb_c, auto dma_c, const uint32_t (&pw)[4]) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_c)::value;
    constexpr int DMA0 = decltype(dma_c)::value;
    const char* sl = smem + ((tn >> 1) & 3) * SP_SLOT + ((tn & 1) ? c1 : c0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (do_load) {
        if (g < 4) fn.b[g].h = *(const half8*)(sl + q_row + g * 2048);
        else if (g < 8) fn.a[g - 4] = *(const half8*)(sl + a_row + (g - 4) * 2048);
      }
      BF& cur = (ni & 1) ? bfb : bfa;
      BF& oth = (ni & 1) ? bfa : bfb;
      const uint32_t rw = ni < 3 ? fc.b[ni + 1].w[mi] : fn.b[0].w[mi];
      sm_unit<KB>(acc[mi][ni], oth.w[2 * mi], oth.w[2 * mi + 1], fc.a[mi], cur.h, pw[mi], rw);
      if constexpr (DMA0 >= 0) if (w4_dma_piece(DMA0, g) >= 0) dma_piece(w4_dma_piece(DMA0, g));
      if constexpr (W4_DMA_MODE == 3 && DMA0 >= 0) xfer(dma_c, g);
      if constexpr (W4_DMA_MODE == 2 && DMA0 >= 0) {
        // staggered: the four waves share one address path (64 B / clk: a KiB piece holds it ~16 cycles); in lockstep they all issue
        // at once and each piece waits for the others'.  Gap G = 16 * phase + g of the 64 behind the pair barrier carries piece G / 3 of
        // the wave with wave % 3 == G % 3 (pieces 0-17 in gaps 0-53).
        const int G = 16 * DMA0 + g;
        if (G < 54 && (G % 3) == dma_res) dma_piece(G / 3);
      }
    }
  };
  auto blk_dense = [&](const W4Frag& fc, W4Frag& fn, int tn, bool do_load, auto dma_c) __attribute__((always_inline)) {
    constexpr int DMA0 = decltype(dma_c)::value;
    const char* sl = smem + ((tn >> 1) & 3) * SP_SLOT + ((tn & 1) ? c1 : c0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int ni = g >> 2, mi = g & 3;
      if (do_load) {
        if (g < 4) fn.b[g].h = *(const half8*)(sl + q_row + g * 2048);
        else if (g < 8) fn.a[g - 4] = *(const half8*)(sl + a_row + (g - 4) * 2048);
      }
      mfma_f16(acc[mi][ni], fc.a[mi], fc.b[ni].h);
      if constexpr (DMA0 >= 0) if (w4_dma_piece(DMA0, g) >= 0) dma_piece(w4_dma_piece(DMA0, g));
      if constexpr (W4_DMA_MODE == 3 && DMA0 >= 0) xfer(dma_c, g);
      if constexpr (W4_DMA_MODE == 2 && DMA0 >= 0) {
        // staggered: the four waves share one address path (64 B / clk: a KiB piece holds it ~16 cycles); in lockstep they all issue
        // at once and each piece waits for the others'.  Gap G = 16 * phase + g of the 64 behind the pair barrier carries piece G / 3 of
        // the wave with wave % 3 == G % 3 (pieces 0-17 in gaps 0-53).
        const int G = 16 * DMA0 + g;
        if (G < 54 && (G % 3) == dma_res) dma_piece(G / 3);
      }
    }
  };

  // Pairs are homogeneous (the launcher only selects this kernel when tsq and td are even): a loop over the sparse
  // pairs, then one over the dense pairs, both straight-line in the accumulators.
  const int npairs = nst >> 1, nsp = tsq >> 1;
  issue_pair(0);
  if (npairs > 1) { issue_pair(1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }   // pair 0 landed; pair 1 (16 or 18 pieces per wave) in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (W4_DMA_MODE == 3) { nx_off = role_goff(2, nx_18); ld_gp = role_src + nx_off; ld_18 = nx_18; }   // pair 2 is staged during pair 0
  W4Frag f0, f1;
  if (nsp > 0) load_pw(pwx, 0);
  load_frag(f0, 0);
  if (nsp > 0) expand_bucket_columns(f0.b[0].w, bfa.w);
  constexpr std::integral_constant<int, 0> KB0{};
  constexpr std::integral_constant<int, 1> KB1{};
  constexpr std::integral_constant<int, -1> NODMA{};
  constexpr std::integral_constant<int, 0> DMA_A{};
  constexpr std::integral_constant<int, 1> DMA_B{};
  constexpr std::integral_constant<int, 2> DMA_C{};
  constexpr std::integral_constant<int, (W4_DMA_MODE >= 2 ? 3 : -1)> DMA_D{};
  dma_n = 0; dma_res = -1;   // nothing pending during pair 0's first two blocks (pairs 0 and 1 went out in the prologue)
  // Schedule of the DMA of pair g+2 (ring half of pair g): block 3 of pair g (behind the barrier that frees the half)
  // carries pieces 0-5, blocks 0 and 1 of pair g+1 pieces 6-11 and 12-17; the barrier of pair g+1 waits for them.
#pragma unroll 1
  for (int g = 0; g < nsp; ++g) {
    const int t0 = 4 * g;
    blk_sparse(f0, f1, t0 + 1, true, KB0, DMA_B, pwx);
    blk_sparse(f1, f0, t0 + 2, true, KB1, DMA_C, pwx);
    load_pw(pwy, 2 * g + 1);
    blk_sparse(f0, f1, t0 + 3, true, KB0, DMA_D, pwy);
    // every read of this pair's ring half has been issued; the next pair must have landed before anybody reads it
    W4_PREP(g);
    __builtin_amdgcn_sched_barrier(0);                    // the matrix asm is not volatile: without this the compiler sinks block 2 below the wait
    W4_PAIR_SYNC();
    W4_COMMIT(g);
    if (g + 1 < nsp) load_pw(pwx, 2 * (g + 1));
    blk_sparse(f1, f0, t0 + 4, g + 1 < npairs, KB1, DMA_A, pwy);
  }
#pragma unroll 1
  for (int g = nsp; g < npairs; ++g) {
    const int t0 = 4 * g;
    blk_dense(f0, f1, t0 + 1, true, DMA_B);
    blk_dense(f1, f0, t0 + 2, true, DMA_C);
    blk_dense(f0, f1, t0 + 3, true, DMA_D);
    W4_PREP(g);
    __builtin_amdgcn_sched_barrier(0);
    W4_PAIR_SYNC();
    W4_COMMIT(g);
    blk_dense(f1, f0, t0 + 4, g + 1 < npairs, DMA_A);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // the last matrix results are in the accumulators
  if (DUMP) { gemm_dump_tile_w<4>(p, acc, dt, qt, wm, wn, lane); return; }
  gemm_epilogue_w<4, GEMM_W4_THREADS>(p, acc, dt, qt, wm, wn, (int)threadIdx.x, lane, smem);
}


hipError_t launch_gemm_w4(const GemmArgs& a, dim3 grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_filter_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_RING_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)gemm_filter_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_RING_LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (a.dump) hipLaunchKernelGGL(gemm_filter_w4_kernel<true>, grid, dim3(GEMM_W4_THREADS), GEMM_RING_LDS, s, a);
  else hipLaunchKernelGGL(gemm_filter_w4_kernel<false>, grid, dim3(GEMM_W4_THREADS), GEMM_RING_LDS, s, a);
  return hipGetLastError();
}

}  // namespace dhr
