// Internal declarations shared by the kernels (kernels.hip, gemm_*.hip) and the host side (dhr_state.h: index_build.hip, search_core.hip, api.hip; sharded.hip).
// gfx950 only: 64-wide wavefronts, v_mfma_f32_32x32x16_f16, global_load_lds (16 B), 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dhr_hip.h"
#include "abi_guard.h"

// library-internal helpers with C linkage (abi.cpp, index_build.hip, api.hip), used by the other translation units
extern "C" int dhr_set_error_message(int code, const char* msg);   // records the calling thread's last error, returns code
extern "C" int dhr_index_device(const dhr_index* ix);
extern "C" void dhr_internal_index_arena(dhr_index* ix, void*** base, size_t** bytes);   // the handle's grow-only scratch for dhr_search_sharded_local
// dhr_search_begin / dhr_search_finish without their final stream synchronisation (sharded.hip: the shards of a one-process search work
// concurrently until the collective layer synchronises; over RCCL the whole call enqueues up to its one host read)
extern "C" int dhr_internal_search_begin_async(dhr_index* ix, const dhr_query_batch* qb, int32_t k, float* out_sample_scores_dev, void* stream);
extern "C" void dhr_internal_search_abort(dhr_index* ix);      // drops the state a dhr_search_begin left behind (error exits of the sharded step)
extern "C" int dhr_internal_search_pre_async(dhr_index* ix, const dhr_query_batch* qb, int32_t k, int32_t r_local, float* out_scores_dev, void* stream);
extern "C" int dhr_internal_search_begin_rest_async(dhr_index* ix, const float* tau_dev, float* out_sample_scores_dev, void* stream);
extern "C" int dhr_internal_search_mid_async(dhr_index* ix, const float* tau_hat_dev, int32_t r_local, float* out_scores_dev, void* stream);
extern "C" int dhr_internal_search_finish_async(dhr_index* ix, const float* tau_hat_dev, float* out_scores, int64_t* out_rows, int32_t* out_count_dev,
                                                int32_t out_mem_kind, void* stream);

namespace dhr {

// ---------------------------------------------------------------------------------------------
// Operand tiles.  Corpus rows and query rows are both stored as the *LDS image* the bound GEMM
// stages: [tile][kstep][256 rows][8 chunks][8 fp16], chunk position XOR-swizzled by (row>>1)&7 so
// that the four 16-lane groups of a ds_read_b128 fragment read hit 16 distinct 16-B slots
// (cdna_hip_programming.md T2).  A K-step tile is one contiguous 32 KiB block in HBM, so the
// global->LDS DMA is perfectly linear on both sides.
// ---------------------------------------------------------------------------------------------
constexpr int TILE_ROWS = 256;
constexpr int TILE_K = 64;
constexpr int CHUNK = 8;                                   // fp16 per 16-byte chunk
constexpr int TILE_HALVES = TILE_ROWS * TILE_K;            // 16384 fp16 = 32 KiB
constexpr int SP_A_BYTES = 16384;      // sparse stage: 256 rows x 32 stored slice values
constexpr int SP_IDX_BYTES = 2048;     //               256 rows x [lane half][block] u16 position bits
constexpr int SP_STAGE_A = SP_A_BYTES + SP_IDX_BYTES;   // corpus bytes per sparse stage (18 KiB)
constexpr int SP_STAGE_B = 16384;      // query bytes per sparse stage: 256 rows x 32 slice values, bucket in the sign bit (expanded to the two bucket columns in registers)
constexpr int SP_DENSE = 16384;        // 2:4 layout, ungated columns: 32-column stages, 256 rows x 64 bytes, for corpus and queries
constexpr int SP_SLOT = SP_STAGE_A + SP_STAGE_B;   // LDS ring slot (34 KiB): corpus part at +0, query part at +SP_STAGE_A
// gated_i8 indexes (gemm_g8.hip): the gated half as int8 on v_smfmac_i32_32x32x64_i8.  A stage is still 32 slices:
//   corpus: [8 blocks of 32 rows][lane half h][32 rows][16 stored bytes = slices 16h .. 16h+15] + [2 groups of 128 rows][h][32 rows][4 blocks]
//           u32 position words (2 bits per stored value: 2 * (slice & 1) + bucket) = 8 + 2 KiB;
//   query : [256 queries][64 bytes = 32 slices x (bucket-0 column, bucket-1 column)], 16-byte chunks swizzled like every other image.
constexpr int S8_A_BYTES = 8192;
constexpr int S8_STAGE_A = S8_A_BYTES + 2048;
#ifndef DHR_HEAVY
#define DHR_HEAVY 64
#endif
constexpr int HEAVY = DHR_HEAVY;        // per-row list of the largest gated values used by the refine step (64, or 32 in A/B builds: 8 lanes x HEAVY / 8 entries)
// ... stored as ONE 6 x HEAVY-byte record per row (384 B): HEAVY u32 keys, then HEAVY fp16 values (a candidate's refine read is one contiguous
// segment instead of a 256-byte and a 128-byte one in two arrays): heavy_key = record base, heavy_val = base + 4 x HEAVY bytes
constexpr int HEAVY_KEY_STRIDE = HEAVY * 6 / 4;    // u32 per record
constexpr int HEAVY_VAL_STRIDE = HEAVY * 6 / 2;    // fp16 per record
#ifndef DHR_DOC_GROUP
#define DHR_DOC_GROUP 4
#endif
constexpr int DOC_GROUP = DHR_DOC_GROUP;                   // doc tiles that share one XCD sweep (A/B builds: -DDHR_DOC_GROUP=n; 3 / 4 / 6 / 8 measure alike)

__host__ __device__ inline int64_t tiled_chunk_offset(int64_t row, int chunk, int ksteps) {
  const int64_t tile = row >> 8;
  const int rl = (int)(row & 255);
  const int ks = chunk >> 3, cc = chunk & 7;
  const int phys = cc ^ ((rl >> 1) & 7);
  return ((tile * ksteps + ks) * TILE_ROWS + rl) * (int64_t)TILE_K + phys * CHUNK;   // in fp16 elements
}

// Candidate keys: (order-preserving bits of the fp32 score) << 32 | (0xFFFFFFFF - local row), so a
// plain descending u64 sort is "score desc, row asc".  0 = empty slot.  -0.0 takes +0.0's pattern: the two compare equal as floats
// (torch.topk, Python's sorted), so they must tie here too and fall to "row asc" (the shard reduces take any caller's scores; found by
// tools/stress_modes.py in round 6 -- until then +0.0 sorted before -0.0); a key decodes to +0.0.
__host__ __device__ inline uint32_t f32_ordered(float f) {
  union { float f; uint32_t u; } v; v.f = f;
  if ((v.u << 1) == 0u) return 0x80000000u;
  return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
__host__ __device__ inline float ordered_f32(uint32_t o) {
  union { float f; uint32_t u; } v;
  v.u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return v.f;
}
__host__ __device__ inline int64_t seq_to_tile(int64_t i, int map_mode, int period, int64_t head, int64_t perm_mul = 1, int64_t perm_n = 1) {
  if (map_mode == 0) return i;
  if (map_mode == 1) return head + i * period;
  if (map_mode == 3) i = (i * perm_mul) % perm_n;        // the non-sample tiles in a scattered order (perm_mul coprime to perm_n, both < 2^24)
  return head + (i / (period - 1)) * period + i % (period - 1) + 1;
}
// The same map without integer division (the bound-GEMM kernels call it once per workgroup, before anything else can start: the
// 64-bit  %  and  /  above expand to ~400 dependent scalar instructions, ~1 us of every 26 us tile).  All operands are below 2^24
// (tile counts of a shard), so the product i * perm_mul < 2^48 is exact in a double and a quotient estimated with the
// precomputed reciprocal is off by at most one: one correction step makes it exact (checked against the integer form on the host:
// dhr_debug_seq_to_tile, tests/test_host_logic.py).
__host__ __device__ inline int64_t divmod24(int64_t x, int64_t n, double inv_n, int64_t& rem) {      // 0 <= x < 2^48, x / n < 2^24
  const double xd = (double)(int32_t)(x >> 24) * 16777216.0 + (double)(int32_t)(x & 0xFFFFFF);
  int64_t q = (int64_t)(int32_t)(xd * inv_n);
  int64_t r = x - q * n;
  if (r < 0) { r += n; --q; } else if (r >= n) { r -= n; ++q; }
  rem = r;
  return q;
}
__host__ __device__ inline int64_t seq_to_tile_fast(int64_t i, int map_mode, int period, int64_t head, int64_t perm_mul, int64_t perm_n,
                                                    double inv_perm_n, double inv_pm1) {
  if (map_mode == 0) return i;
  if (map_mode == 1) return head + i * period;
  if (map_mode == 3) (void)divmod24(i * perm_mul, perm_n, inv_perm_n, i);
  int64_t rem;
  const int64_t q = divmod24(i, period - 1, inv_pm1, rem);
  return head + q * period + rem + 1;
}
__host__ __device__ inline uint64_t make_key(float score, uint32_t row) {
  return ((uint64_t)f32_ordered(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

struct GemmArgs {
  const __half* a_tiles;      // corpus operand tiles
  const __half* b_tiles;      // query operand tiles
  int ksteps;                 // K_pad / 64
  int k_split;                // K-steps [0,k_split) are the gated (DLR) half; informational
  int ts, td;                 // 2:4 layout only (ts > 0 selects it): ts 32-slice sparse stages, then td 32-column dense stages
  int ts_q;                   // sparse stages on the query side: ts, or 2*ts for an ungated batch (stage u pairs with corpus stage u % ts)
  // corpus tiles of this launch: sequence positions [seq_lo, seq_hi) mapped to tile ids by
  //   map_mode 0: tile = i                 (contiguous)
  //   map_mode 1: tile = head + i*period   (the strided sample)
  //   map_mode 2: tile = head + (i/(period-1))*period + i%(period-1) + 1   (everything but the sample)
  //   map_mode 3: as 2 with i -> (i * perm_mul) % perm_n: every range of positions is a scattered subset of the non-sample tiles
  int64_t seq_lo, seq_hi;
  int map_mode, period;
  int64_t perm_mul, perm_n;
  double inv_perm_n, inv_pm1; // 1 / perm_n and 1 / (period - 1): filled in by launch_gemm_filter (seq_to_tile_fast)
  int64_t head, n_tiles;
  int n_qtiles;               // Q_pad / 256
  int64_t n_rows;             // valid corpus rows (rows >= n_rows are zero padding)
  const float* thr;           // [Q_pad] tau - margin (+inf for padded queries)
  uint2* cand;                // [Q_pad][cap] (local row, bound score bits)
  uint32_t* cnt;              // [Q_pad]
  uint32_t cap;
  float* dump;                // debug: [n_queries][dump_ld] bound scores, or nullptr
  int64_t dump_ld;
  int64_t dump_row0;
  int n_queries;
  int variant;                // fp16-gated kernel: 4 (4 waves) / 5 (8 waves), 0 = the library default (g_gemm_variant); 6: A/B builds, persistent workgroups
  const int32_t* g8_shift;    // [Q_pad] or null.  Non-null: gated_i8 index (gemm_g8.hip) -- the ts gated stages are int8 2:4 images and run FIRST into int32
                              // accumulators, which are then shifted left by this per-query amount (gated unit = 2^shift x ungated unit) before the td
                              // int8 stages of the ungated columns accumulate on top; i8_mul is the final unit (score = sum * i8_mul)
  const int32_t* g8_rsum;     // gated_i8: [n_tiles * 256] 128 x (sum of the row's gated int8 values): the accumulators START there, which pays for the
                              // query operand being stored as level - 128 (8 bits of query resolution instead of 7)
  int partial_wn;             // gemm_g8.hip: > 0 = the batch's last query tile holds real queries in its first partial_wn (1 or 2) 64-query wave columns only (set by launch_gemm_g8)
  uint32_t* p_ctr;            // gemm_g8p.hip (persistent workgroups): this launch's 8 per-XCD tile counters, zeroed on the stream before the launch (set by launch_gemm_g8p)
  int64_t p_groups;           //   corpus tile groups of the launch, ceil((seq_hi - seq_lo) / DOC_GROUP)
  const float* i8_mul;        // [Q_pad] or null.  Non-null: the td dense stages hold int8 columns (64 per stage); the kernel runs them
                              // FIRST on v_mfma_i32_32x32x32_i8 and turns the integer sums into fp32 with this per-query factor
                              // (corpus scale x query scale) before the gated stages accumulate on top
  // Two-tier candidate lists (round 5): every query owns `cap` slots of the uniform array `cand`; a query the controller expects to need
  // more (a few per cent of a batch pass 10-100 x the average through the filter) also owns ovf_cap[q] slots at ovf[ovf_off[q]] in a shared
  // arena, planned on the device from the previous launch's list lengths (plan_overflow_kernel).  Slot s of query q lives at
  // cand[q * cap + s] for s < cap and at ovf[ovf_off[q] + s - cap] behind it.  tier null: uniform lists only.
  // (ONE pointer to a three-pointer record in device memory, read on the cold path only: three more pointer arguments cost the 4-wave kernel,
  // which sits at the scalar-register limit, 1 KB of scratch per lane)
  const struct ListTier* tier;
};
struct ListTier { uint2* ovf; const uint32_t* ovf_off; const uint32_t* ovf_cap; };
#if defined(__HIPCC__)
// TIER = false: the 4-wave kernel (gemm_filter_wx_kernel<NI = 4>, DHR_PARAM_GEMM_VARIANT 4) holds 509 registers per lane and 256 unrolled copies
// of this store; the second tier in each of them costs it 1 KB of scratch per lane, so that kernel writes the uniform part only and
// the controller does not plan a second tier for it (ensure_ws).
template <bool TIER = true>
__device__ __forceinline__ void cand_store(const GemmArgs& p, int q, uint32_t slot, uint2 v) {
  if (slot < p.cap) p.cand[(int64_t)q * p.cap + slot] = v;
  else if (TIER && p.tier) {                  // cold: only the queries with an arena segment ever get here
    const ListTier* t = p.tier;
    const uint32_t o = slot - p.cap;
    if (o < t->ovf_cap[q]) t->ovf[(size_t)t->ovf_off[q] + o] = v;
  }
}
#endif

// gated_i8: x >= 0 in units of a step, rounded UP to [0, 127].  inv_step carries a relative 1e-6 of head room, so that the fp32
// rounding of the product can never make  step * q(x) < x ; the SAME expression in the tile builder, query_prep and the refine step
__host__ __device__ inline int quant_up_i8(float x, float inv_step) {
  float r = ceilf(x * inv_step);
  r = r < 0.f ? 0.f : (r > 127.f ? 127.f : r);
  return (int)r;
}
// the query side of a gated_i8 index has 8 bits: levels 0 .. 255, stored as level - 128 (see gemm_g8.hip)
__host__ __device__ inline int quant_up_u8(float x, float inv_step) {
  float r = ceilf(x * inv_step);
  r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
  return (int)r;
}
// int8 image of the ungated columns (dense_i8 indexes): q(v) = clamp(rint(v * inv_scale), -127, 127), the SAME expression in the
// corpus tile kernel, the row-error kernel and query_prep
__host__ __device__ inline int quant_i8(float v, float inv_scale) {
  float r = rintf(v * inv_scale);
  r = r < -127.f ? -127.f : (r > 127.f ? 127.f : r);
  return (int)r;
}

struct RescoreArgs {
  const __half* vals_rm;      // corpus values, row-major fp16 [n_rows][k_rm]
  const void* c_idx;          // [n_rows_pad][d_dlr] (1 or 2 bytes per entry) or null
  int c_idx_dtype;            // dhr_idx_dtype
  const float* q32;           // [Q_pad][K_pad]
  const int16_t* q_idx;       // [Q_pad][d_dlr]
  // fast path (queries exactly representable in fp16, 1-byte index dtype): fp16 copy of the queries (gated values whose index
  // cannot equal any corpus index byte are zeroed), the query indices as bytes, and the batch flag that disables it
  const __half* q16; const uint8_t* q_idx8; const uint32_t* q_inexact;
  int d_dlr, k_rm;
  int gate;                   // 0: ungated inner product over all columns (--IP stage 1)
  int64_t n_rows;
  // candidate source: explicit list (cand != null) or the implicit range [row0, row0+count_all)
  const uint2* cand; const uint32_t* cnt; uint32_t cap;
  const uint32_t* rows32;     // alternative explicit list [n_queries][ld_rows] of local rows (score_rows)
  int64_t ld_rows;
  int64_t row0; uint32_t count_all;
  const float* thr_exact;     // optional [Q_pad]: candidates whose refined bound < thr are dropped (unused v1)
  uint64_t* out_keys;         // [Q_pad][ld_keys] or null
  int64_t ld_keys;
  float* out_scores;          // [n_queries][ld_scores] or null
  int64_t ld_scores;
  int n_queries;
  uint32_t max_count;         // grid.x * CANDS_PER_WG covers this many
  // flat launch (candidate-list mode): workgroup b serves the query q with blk_off[q] <= b < blk_off[q+1]; grid = flat_blocks
  const uint32_t* blk_off; uint32_t flat_blocks;
  int split;                  // set by launch_rescore: 1 = rescore_fast_kernel was launched beside this kernel and takes the batches it can
};

struct SelectArgs {
  uint64_t* topk_keys;        // [Q_pad][kp] sorted descending, 0 = empty
  const uint64_t* in_keys;    // [Q_pad][ld_keys]
  int64_t ld_keys;
  const uint32_t* cnt;        // per query count (null -> count_all)
  uint32_t count_all, cap;
  int k, kp, sort_n;
  int k_keep;                 // entries the running list keeps (0 = k).  The sampled run publishes the r-th best (k = r) but keeps the k best seen:
                              // rows of the sample that tie with the final k-th score must not be lost to the rank that only defines the threshold
  int kps;                    // slots of the running list actually in use (power of two >= k, <= kp): sizes the LDS sort
  int monotone;               // 1: the published threshold never falls (tau = max(old tau, k-th best)): sampled runs whose rank grows with the fraction seen
  const float* margin;        // [Q_pad]
  float* tau;                 // [Q_pad] exact k-th best so far (-inf until k results exist)
  float* thr;                 // [Q_pad] tau - margin
  int n_queries;
};

// launchers (kernels.hip)
hipError_t launch_scan_rows(const __half* src, int64_t ld, int64_t n_rows, int d_dlr, int k, uint32_t* max_sq_bits,
                            uint32_t* neg_flag, hipStream_t s);       // max_sq_bits[2] receives max |ungated value| (float bits)
// dense_i8: per-row quantisation error of the ungated columns -> out_bits[0] = max_r ||d - scale*q(d)||^2, [1] = max_r ||scale*q(d)||^2
hipError_t launch_i8_row_err(const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls, float scale, const float* col_scale,
                             uint32_t* out_bits, hipStream_t s);
hipError_t launch_col_absmax(const __half* vals_rm, int k_rm, int64_t n_rows, int col0, int n_cols, uint32_t* colmax_bits, hipStream_t s);   // columns [col0, col0 + n_cols)
hipError_t launch_tile_rows_sparse(const __half* src, int64_t ld, int64_t row_lo, int64_t n_rows_src, int64_t n_rows_fill,
                                   int d_dlr, int d_cls, int ts, int td, const void* idx, int idx_dtype, const uint8_t* map,
                                   bool abs_dlr, char* tiles, float i8_inv_scale /* 0: fp16 dense stages */, const float* col_scale,
                                   const float* g8_inv_cs /* non-null: gated_i8 stages */, hipStream_t s);
hipError_t launch_g8_row_sum(const __half* vals_rm, int k_rm, int64_t n_rows, int64_t n_rows_fill, int d_dlr, bool abs_dlr, const float* g8_inv_cs,
                             int32_t* rsum128, hipStream_t s);
hipError_t launch_copy_rows(const __half* src, int64_t ld, int64_t n_rows, int k, int k_rm, __half* dst, hipStream_t s);
hipError_t launch_idx_hist(const uint8_t* idx, const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, float* hist, hipStream_t s);
// gated_i8 indexes: what query_prep needs to build the int8 image of the gated half (all null / 0 otherwise)
struct G8Prep {
  const float* inv_cs;   // [d_dlr] 1 / step of gated corpus column j (with 1e-6 of head room); non-null selects the gated_i8 path
  const float* w;        // [d_dlr] query weight of column j: step_j / s_ref (rounded up)
  float s_ref;           // reference step: one gated operand product counts step_query * s_ref score units
  int max_shift;         // largest left shift of the gated sums that cannot overflow int32
  uint8_t* q8;           // [Q_pad][d_dlr] out: the query's gated int8 operand values (refine step)
  int32_t* shift;        // [Q_pad] out
  float* unit;           // [Q_pad] out: score units of one gated product (= i8_mul * 2^shift)
  // dense-only int8 index with a residual image (RefineArgs::resid8): by how much the refine level may raise the filter threshold
  float* thr_raise;      // [Q_pad] out, or null
  float resid_ec2;       // >= || what the residual image itself loses ||  in the weighted space of i8_ec
};
hipError_t launch_query_prep(const void* src, int src_is_f32, int64_t ld, const void* idx, int idx_dtype, int64_t ld_idx,
                             int n_queries, int q_pad, int d_dlr, int d_cls, int k_rm, int n_buckets, int kt,
                             const uint8_t* map, bool abs_dlr, float dmax, __half* q_tiles, float* q32, int16_t* q_idx,
                             float* margin, float* tau, float* thr, int ts, int td, uint32_t* q_pack, __half* q16, uint8_t* q_idx8,
                             uint32_t* q_inexact, int c_idx_dtype,
                             float i8_scale /* 0: fp16 dense stages */, float i8_ec, float i8_nc, float* i8_mul, const float* col_scale,
                             const G8Prep& g8, hipStream_t s);
#if defined(__HIPCC__)
// Workgroup -> (corpus tile dt, query tile qt) of a bound-GEMM launch.  Grid = (8, DOC_GROUP * n_qtiles, tile groups per XCD): workgroups
// are dispatched x-fastest and dealt round-robin to the 8 XCDs, so blockIdx.x is the XCD, and each XCD sweeps DOC_GROUP corpus tiles
// against all query tiles before it moves on (its L2 holds those corpus tiles; the query tile set streams from the Infinity Cache) --
// the linear order of the former 1-D grid, without its division by DOC_GROUP * n_qtiles.
constexpr int MAP_SPREAD_Q = 0x100;        // GemmArgs::map_mode flag (set by launch_gemm_filter): the launch has fewer than 8 tile groups -- see gemm_wg_tile
__device__ __forceinline__ bool gemm_wg_tile(const GemmArgs& p, int64_t& dt, int& qt) {
  const int r = (int)blockIdx.y;
  const int mode = p.map_mode;
  int64_t seq;
  if (mode & MAP_SPREAD_Q) {
    // Small launches (threshold bootstrap: 1 tile; first phases of a sampled run: 4, 13 tiles): with fewer than 8 tile groups the XCD-major
    // map above leaves XCDs idle -- the 112 workgroups of a 4-tile launch all landed on ONE XCD's 32 CUs, 3.5 rounds, 214 us for a launch whose
    // tiles take ~55 us each (found in round 6 on a shard's kernel timeline: the begin of a 1/8 shard is made of such launches).  Here the
    // XCD (blockIdx.x) takes every 8th QUERY tile of every corpus tile instead: grid = (8, DOC_GROUP x ceil(n_qtiles / 8), tile groups).
    const int qg = r / DOC_GROUP;
    const int dl = r - qg * DOC_GROUP;
    qt = qg * 8 + (int)blockIdx.x;
    if (qt >= p.n_qtiles) return false;
    seq = p.seq_lo + (int64_t)blockIdx.z * DOC_GROUP + dl;
  } else {
    qt = r / DOC_GROUP;
    const int dl = r - qt * DOC_GROUP;
    seq = p.seq_lo + ((int64_t)blockIdx.z * 8 + (int64_t)blockIdx.x) * DOC_GROUP + dl;
  }
  if (seq >= p.seq_hi) return false;
  dt = seq_to_tile_fast(seq, mode & 0xff, p.period, p.head, p.perm_mul, p.perm_n, p.inv_perm_n, p.inv_pm1);
  return dt < p.n_tiles;
}
#endif
inline int sparse_query_stages(int ts, bool gated, bool g8 = false) { return ts > 0 ? ((gated || g8) ? ts : 2 * ts) : 0; }
hipError_t launch_gemm_filter(const GemmArgs& a, hipStream_t s);
hipError_t launch_heavy_build(const __half* vals_rm, int k_rm, const void* idx, int idx_dtype, int64_t n_rows, int d_dlr,
                              const uint8_t* map, int n_buckets, uint32_t* heavy_key, __half* heavy_val,
                              const float* g8_inv_cs /* gated_i8 indexes: the key also carries the entry's int8 level */, int abs_mode, hipStream_t s);
#ifndef REFINE_PER_WG_N
#define REFINE_PER_WG_N 512   // round 5: 256 -> 512 takes 1.0 ms off the refine level of a config-3 step (11.1 -> 10.05 ms alone; 1 024: 9.8)
#endif
constexpr int REFINE_PER_WG = REFINE_PER_WG_N;       // candidates of ONE query per refine workgroup (its operand words are staged in LDS once); a multiple of 32
struct RefineArgs {
  const uint2* cand; const uint32_t* cnt; uint32_t cap;      // bound candidates (row, U bits)
  const uint2* ovf; const uint32_t* ovf_off; const uint32_t* ovf_cap;   // ... their second tier (GemmArgs), or null
  const uint32_t* heavy_key; const __half* heavy_val;       // [n_rows][HEAVY]
  const uint32_t* q_pack; int d_dlr;                         // [Q_pad][d_dlr]: fp16 value | bucket | idx low bits
  const float* thr;                                          // [Q_pad]
  uint2* out; uint32_t* out_cnt; uint32_t out_cap;           // survivors (row, refined bound), per-query count, list capacity
  int n_queries; uint32_t max_count;
  const uint32_t* blk_off; uint32_t flat_blocks;             // flat launch, as in RescoreArgs
  // gated_i8 indexes: the bound counted listed entry (slice j, value d) as q8[j] * d8(d, j) * unit[q]; the refine step takes ALL of
  // a row's listed same-bucket entries off the bound and puts the real-valued product back where the index values agree
  const uint8_t* g8_q8;                                      // [Q_pad][d_dlr] or null
  const float* g8_inv_cs;                                    // [d_dlr] 1 / step of gated column j (the tile builder's factor)
  const float* g8_unit;                                      // [Q_pad] score units of one gated operand product
  int abs_mode;
  int ungated;                                               // gated_i8, plain inner product batch: every listed entry counts in both directions (no bucket / index test)
  // dense-only int8 index (round 4): the refine level is the RESIDUAL image of the corpus -- nibble c of a row = 8 + rint((d - cs_c d8) * 14 / cs_c),
  // what the int8 image of column c lost, in 1/14 of its step -- so that  U + sum_c q_c (cs_c / 14) (nibble - 8)  leaves only the QUERY's
  // rounding (and 1/15 of the corpus') to the margin: the threshold of this level is thr + thr_raise (query_prep_kernel)
  const uint8_t* resid8; int resid_ld;                       // [n_rows][resid_ld] or null
  const float* q32; int64_t q32_ld;                          // fp32 queries (row-major copies of the batch)
  const float* col_scale; int d_cls;                         // cs_c
  const float* thr_raise;                                    // [Q_pad]
};
hipError_t launch_resid_build(const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls, const float* col_scale, uint8_t* resid8, int resid_ld, hipStream_t s);
hipError_t launch_refine(const RefineArgs& a, hipStream_t s);
hipError_t launch_rescore(const RescoreArgs& a, hipStream_t s);
hipError_t launch_select(const SelectArgs& a, hipStream_t s);
hipError_t launch_select_global(const SelectArgs& a, hipStream_t s);    // k > 16384 (select_global.hip)
hipError_t launch_merge_topk_global(int n_queries, int n_in, const float* in_scores, const int64_t* in_rows, int k_out, float* out_scores,
                                    int64_t* out_rows, hipStream_t s);      // dhr_merge_topk with n_in > 16384
hipError_t launch_emit(const uint64_t* topk_keys, int kp, int n_queries, int k, int64_t row_offset, float* out_scores,
                       int64_t* out_rows, hipStream_t s);
hipError_t launch_keys_to_rows(const uint64_t* topk_keys, int kp, int n_queries, int k, uint32_t* rows, hipStream_t s);
hipError_t launch_bound_topm(const float* bound /*[n_queries][256]*/, int n_rows, int n_queries, int m, uint32_t* rows /*[n_queries][m]*/, hipStream_t s);
hipError_t launch_densify(const void* lex, int in_is_f32, int64_t ld, int64_t batch, int remove, int dims, int n_groups, void* out_val,
                          int val_is_f32, int64_t ld_val, void* out_idx, int idx_is_i16, int64_t ld_idx, hipStream_t s);
hipError_t launch_pq_init(const __half* vals, int64_t ld, int64_t n, int64_t stride, int dsub, int M, float* cb, int ksub, hipStream_t s);
hipError_t launch_pq_assign(const __half* vals, int64_t ld, int64_t n, int64_t stride, int dsub, int M, const float* cb, uint8_t* codes,
                            int64_t ld_codes, float* sums, uint32_t* counts, float* err, int ksub, hipStream_t s);
hipError_t launch_pq_update(float* cb, const float* sums, const uint32_t* counts, int dsub, int M, int ksub, hipStream_t s);
hipError_t launch_pq_decode(const uint8_t* codes, int64_t ld_codes, int64_t n, int M, int dsub, const float* cb, __half* out, int64_t ld_out,
                            int ksub, hipStream_t s);
// exclusive scan of ceil(min(cnt[q], cap) / per) over the queries -> offs[0 .. n_queries] (one workgroup)
hipError_t launch_block_offsets(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t per, uint32_t* offs, hipStream_t s);
// second tier of the bound lists for the NEXT launch, from the list lengths of the previous one: a query whose list is expected to outgrow
// its `cap` uniform slots -- 2 x cnt_prev x rows_next / rows_prev + 2048 entries -- gets the difference (rounded up to 256, at most
// max_extra) as a segment of the arena; segments are handed out in query order until the arena is used up (ovf_off / ovf_cap [n_queries];
// cnt_prev null: no segments)
hipError_t launch_plan_overflow(const uint32_t* cnt_prev, double rows_ratio, uint32_t cap, uint32_t max_extra, uint32_t arena_entries, int n_queries,
                                uint32_t* ovf_off, uint32_t* ovf_cap, hipStream_t s);
hipError_t launch_max_u32(const uint32_t* v, int n, uint32_t* out_max, unsigned long long* out_sum, hipStream_t s);
hipError_t launch_lists_ready(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t per, uint32_t* offs, uint32_t* out_max, uint32_t* out_max2,
                              unsigned long long* out_sum, unsigned long long* out_sum2, uint32_t* fail_flags, uint32_t* zero, int n_zero,
                              hipStream_t s, const uint32_t* ovf_cap = nullptr);
hipError_t launch_rows_to_local(const int64_t* rows, int64_t n, int64_t row_offset, int64_t n_rows, uint32_t* out,
                                hipStream_t s);
hipError_t launch_mark_overflow(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t* fail_flags, hipStream_t s);
hipError_t launch_verify(const uint64_t* topk_keys, int kp, int k, const float* tau_hat, int n_queries, uint32_t* fail_flags,
                         uint32_t* n_fail, hipStream_t s);
hipError_t launch_gather_queries(const float* q32, const int16_t* q_idx, int k_rm, int d_dlr, const int32_t* ids, int n,
                                 float* out32, int16_t* out_idx, hipStream_t s);
hipError_t launch_scatter_keys(const uint64_t* src, uint64_t* dst, int kp, const int32_t* ids, int n, hipStream_t s);
hipError_t launch_emit_scores(const uint64_t* topk_keys, int kp, int n_queries, int r, float* out, hipStream_t s);
hipError_t launch_make_thr(const float* tau, const float* margin, int n_queries, int q_pad, float* thr, hipStream_t s);
hipError_t launch_raise_thr(float* thr_hat, const float* thr_run, int n_queries, hipStream_t s);
hipError_t launch_flag_tau_above(const float* tau_own, const float* tau_ext, int n_queries, uint32_t* fail_flags, hipStream_t s);
hipError_t launch_raise_thr_rank(float* thr_hat, float* tau_hat, const uint64_t* topk_keys, int kp, int r, const float* margin, int n_queries, hipStream_t s);
hipError_t launch_count_ge(const uint64_t* topk_keys, int kp, int k, const float* tau, const uint32_t* fail_flags, int n_queries,
                           int32_t* out, hipStream_t s);
hipError_t launch_merge_topk(int n_queries, int n_in, const float* in_scores, const int64_t* in_rows, int k_out,
                             float* out_scores, int64_t* out_rows, hipStream_t s);
// stride_s / stride_r: elements between the [Q, list_len] sections of consecutive lists in in_scores / in_rows (0: dense [n_lists, Q, list_len]);
// the sharded search gathers [counts | scores | rows | status] blocks, so a rank's sections lie a whole block apart
hipError_t launch_merge_lists(int n_queries, int n_lists, int list_len, const float* in_scores, const int64_t* in_rows, int k_out,
                              float* out_scores, int64_t* out_rows, hipStream_t s, int64_t stride_s = 0, int64_t stride_r = 0);

extern int g_gemm_variant;
extern thread_local int g_last_gemm_kernel;      // kernels.hip: the bound-GEMM kernel of the calling thread's latest launch_gemm_filter
constexpr int RESCORE_CANDS_PER_WG = 32;
constexpr uint32_t FLAT_GRID_MAX = 1u << 20;      // flat launches: at most this many workgroups, the rest of the block list by grid stride
constexpr uint32_t FLAT_GRID_ASYNC = 16384;       // ... and this many when the host does not know the block count (controller without read-backs)
constexpr int SELECT_THREADS = 1024;

}  // namespace dhr
