// HIP kernels of the MI355X dense-hybrid retrieval path (gfx950 only).
//
//   scan_rows / tile_rows   corpus upload: norms + sign scan, row-major fp16 -> MFMA operand tiles
//   query_prep              queries -> operand tiles (fp16), exact fp32 copy, int16 idx, filter margins
//   gemm_filter             bound GEMM  U = Q x D^T  (v_mfma_f32_32x32x16_f16, LDS-DMA staged tiles)
//                           fused with the per-query threshold filter: scores never reach HBM
//   rescore                 exact gated inner product of the surviving (query,row) pairs (fp64 sum)
//   select                  per-query running top-k (bitonic in LDS) and threshold update
//   emit / merge_topk       result formatting; k-way reduce of per-shard lists
//
// Reference semantics: /root/reference/retrieval/gip_retrieval.py:119-125 (gated IP + topk),
// :74-75 (plain IP + argsort), retrieval/merge.result.py:22-42 (shard reduce).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "dhr_internal.h"
#include "gemm_common.h"
#include <type_traits>

namespace dhr {

typedef short short8 __attribute__((ext_vector_type(8)));

// hipFuncAttributeMaxDynamicSharedMemorySize caches: per device and under a lock -- handles on different devices may be used
// concurrently from different host threads (dhr_hip.h)
static hipError_t ensure_lds_attr(const void* fn, int bytes, int (&have)[64]) {
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  if (bytes <= have[dev & 63]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have[dev & 63] = bytes;
  return e;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------ scan_rows
// One wave per corpus row: max_d ||d||^2 (for the rounding margin of the bound filter) and "any
// negative DLR value" (switches the bound to |q|.|d| on the gated half).
__global__ void __launch_bounds__(256) scan_rows_kernel(const __half* __restrict__ src, int64_t ld, int64_t n_rows,
                                                        int d_dlr, int k, uint32_t* max_sq_bits, uint32_t* neg_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const bool vec = ((k & 7) == 0) && ((ld & 7) == 0) && ((((uintptr_t)src) & 15) == 0);
  float best = 0.f, amax = 0.f, gmax = 0.f;
  bool neg = false;
  for (int64_t row = wave0; row < n_rows; row += nwaves) {
    const __half* r = src + row * ld;
    float s = 0.f;
    if (vec) {
      for (int c = lane; c * 8 < k; c += 64) {
        const half8 v = *(const half8*)(r + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          s += f * f;
          neg |= (c * 8 + e < d_dlr) && (f < 0.f);
          if (fabsf(f) <= 65504.f) { if (c * 8 + e >= d_dlr) amax = fmaxf(amax, fabsf(f)); else gmax = fmaxf(gmax, fabsf(f)); }
        }
      }
    } else {
      for (int j = lane; j < k; j += 64) {
        const float f = __half2float(r[j]);
        s += f * f;
        neg |= (j < d_dlr) && (f < 0.f);
        if (fabsf(f) <= 65504.f) { if (j >= d_dlr) amax = fmaxf(amax, fabsf(f)); else gmax = fmaxf(gmax, fabsf(f)); }
      }
    }
    s = wave_sum(s);
    best = fmaxf(best, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o, 64)); gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64)); }
  if (lane == 0 && best > 0.f) atomicMax(max_sq_bits, __float_as_uint(best));
  if (lane == 0 && amax > 0.f) atomicMax(max_sq_bits + 2, __float_as_uint(amax));     // scale of the int8 image of the ungated columns
  if (lane == 0 && gmax > 0.f) atomicMax(max_sq_bits + 3, __float_as_uint(gmax));     // largest finite |gated value|: it must fit fp16 in units of that scale
  if (__any(neg) && lane == 0) atomicOr(neg_flag, 1u);
}

// dense_i8 indexes: one wave per row over the ungated columns of the row-major copy.  The bound GEMM multiplies the int8
// images; what it loses per row is  <q, d - scale*q8(d)>  <= ||q|| * ||d - scale*q8(d)||  and  <q - sq*q8(q), scale*q8(d)>
// <= ||q - sq*q8(q)|| * ||scale*q8(d)||  (query_prep turns the two corpus-wide maxima into the filter margin).
__global__ void __launch_bounds__(256) i8_row_err_kernel(const __half* __restrict__ vals_rm, int k_rm, int64_t n_rows, int d_dlr,
                                                         int d_cls, float scale, const float* __restrict__ col_scale, uint32_t* out_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float be = 0.f, bn = 0.f;
  for (int64_t row = wave0; row < n_rows; row += nwaves) {
    const __half* r = vals_rm + row * k_rm + d_dlr;
    float se = 0.f, sn = 0.f;
    for (int j = lane; j < d_cls; j += 64) {
      // column j lives in units of col_scale[j]; the query side multiplies by w_j = col_scale[j] / scale, so the row is measured in
      // the space d'_j = d_j / w_j, where every column has the step `scale`
      const float cs = col_scale[j];
      const float f = __half2float(r[j]);
      const float q8 = (float)quant_i8(f, 1.f / cs);
      const float e = (f - cs * q8) * (scale / cs);
      const float back = scale * q8;
      se += e * e;
      sn += back * back;
    }
    be = fmaxf(be, wave_sum(se));
    bn = fmaxf(bn, wave_sum(sn));
  }
  if (lane == 0 && !(be >= 0.f)) be = INFINITY;      // a NaN / inf row: the margin becomes infinite, the filter passes everything (still exact)
  if (lane == 0 && !(bn >= 0.f)) bn = INFINITY;
  if (lane == 0) { atomicMax(out_bits, __float_as_uint(be)); atomicMax(out_bits + 1, __float_as_uint(bn)); }
}
hipError_t launch_i8_row_err(const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls, float scale, const float* col_scale,
                             uint32_t* out_bits, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t blocks = (n_rows + 3) / 4;
  hipLaunchKernelGGL(i8_row_err_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, vals_rm, k_rm, n_rows,
                     d_dlr, d_cls, scale, col_scale, out_bits);
  return hipGetLastError();
}
// Largest finite |value| of every column of a range (the ungated columns: steps of their int8 image; the gated columns of a gated_i8
// index likewise) (thread = column, the threads of a wave read 128 contiguous bytes of a row).
__global__ void __launch_bounds__(256) col_absmax_kernel(const __half* __restrict__ vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls,
                                                         int64_t rows_per_block, uint32_t* __restrict__ colmax_bits) {
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= d_cls) return;
  const int64_t lo = (int64_t)blockIdx.x * rows_per_block, hi = lo + rows_per_block < n_rows ? lo + rows_per_block : n_rows;
  float m = 0.f;
  for (int64_t row = lo; row < hi; ++row) {
    const float f = fabsf(__half2float(vals_rm[row * k_rm + d_dlr + j]));
    if (f <= 65504.f) m = fmaxf(m, f);
  }
  if (m > 0.f) atomicMax(colmax_bits + j, __float_as_uint(m));
}
hipError_t launch_col_absmax(const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr /* first column */, int d_cls /* columns */, uint32_t* colmax_bits, hipStream_t s) {
  if (n_rows <= 0 || d_cls <= 0) return hipSuccess;
  const int64_t rpb = 2048;
  hipLaunchKernelGGL(col_absmax_kernel, dim3((unsigned)((n_rows + rpb - 1) / rpb), (unsigned)((d_cls + 255) / 256)), dim3(256), 0, s, vals_rm, k_rm,
                     n_rows, d_dlr, d_cls, rpb, colmax_bits);
  return hipGetLastError();
}
hipError_t launch_scan_rows(const __half* src, int64_t ld, int64_t n_rows, int d_dlr, int k, uint32_t* max_sq_bits,
                            uint32_t* neg_flag, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t blocks = (n_rows + 3) / 4;
  hipLaunchKernelGGL(scan_rows_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, src, ld,
                     n_rows, d_dlr, k, max_sq_bits, neg_flag);
  return hipGetLastError();
}

// x >= 0 scaled into the units of a dense_i8 index and rounded UP to fp16 (the gated operands stay an upper bound)
__device__ __forceinline__ _Float16 half_up(float x) {
  _Float16 h = (_Float16)x;
  if ((float)h < x) { union { _Float16 h; uint16_t u; } v; v.h = h; v.u += 1; h = v.h; }     // x > 0 here: next representable value (inf past 65504)
  return h;
}
// ------------------------------------------------------------------------------------------ tile_rows
// Bucket of a slice index value: per-slice 256-entry table (8-bit index dtypes, balanced by corpus
// frequency at build time) or value % n_buckets (int16 indices, whole-word BM25 vocabularies).
__device__ __forceinline__ int bucket_of(int idx_value, int j, const uint8_t* __restrict__ map, int n_buckets) {
  if (map) return map[j * 256 + (idx_value & 255)];
  return (int)((uint32_t)(idx_value & 0xFFFF) % (uint32_t)n_buckets);
}
__device__ __forceinline__ int load_idx(const void* __restrict__ idx, int dtype, int64_t off) {
  if (dtype == DHR_IDX_I16) return ((const int16_t*)idx)[off];
  if (dtype == DHR_IDX_I8) return ((const int8_t*)idx)[off];
  return ((const uint8_t*)idx)[off];
}

// ---- 2:4 layout (two index buckets, sparse matrix cores).  Logical gated columns are (slice, bucket)
// pairs, k = 2*slice + bucket: every group of four logical columns = two slices holds exactly two
// values, which is the structured sparsity v_smfmac_f32_32x32x32_f16 wants on its A operand.  So the
// corpus side stores just the ORIGINAL slice values (32 per stage) plus 2 position bits per value
// (position = 2*(slice&1) + bucket); only the query side is expanded to the two bucket columns.
// Register layouts were measured on gfx950 (tools/probe/smfmac_probe.hip, gpurun_out/smfmac_probe.txt):
//   A: lane l holds row l&31, stored elements E=0..7 = slices 8*(l>>5)+E of the 16-slice block,
//      position bits of element E at idx[2E+1:2E] (low 16 bits with abid=0, high 16 with abid=1);
//   B: lane l holds column l&31, element e: group g=e>>2 (position e&3) where the logical group
//      G = 4*(g>>1) + 2*(l>>5) + (g&1) covers slices 2G, 2G+1 of the block.
// Stage image of the corpus (18 KiB): [256 rows][4 chunks of 8 slices, chunk ^ ((row>>2)&3)] fp16, then
// [8 blocks of 32 rows][lane half][32 rows] u32 position words (low 16 bits: block 0 of the stage, high: block 1).  Tile = ts sparse stages, then td dense stages of 32 columns
// ([256 rows][4 chunks, chunk ^ ((row>>2)&3)] fp16 = 16 KiB, the same image for corpus and queries).
__global__ void __launch_bounds__(256) tile_rows_sparse_kernel(const __half* __restrict__ src, int64_t ld, int64_t row_lo,
                                                               int64_t n_rows_src, int64_t n_rows_fill, int d_dlr,
                                                               int d_cls, int ts, int td, const void* __restrict__ idx,
                                                               int idx_dtype, const uint8_t* __restrict__ map, int abs_dlr,
                                                               char* __restrict__ tiles, float i8_inv, const float* __restrict__ col_scale,
                                                               const float* __restrict__ g8_inv_cs) {
  const int k = d_dlr + d_cls;
  const bool g8 = g8_inv_cs != nullptr;
  const int sp_chunks = g8 ? ts * 2 : ts * 4, dn_chunks = td * 4;
  const int cpr = sp_chunks + dn_chunks;
  const int sp_stage = g8 ? S8_STAGE_A : SP_STAGE_A;
  const int64_t tile_bytes = (int64_t)ts * sp_stage + (int64_t)td * SP_DENSE;
  const int64_t total = n_rows_fill * cpr;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rl = g / cpr;
    const int c = (int)(g - rl * cpr);
    const int64_t row = row_lo + rl;
    const int r = (int)(row & 255);
    char* tile = tiles + (row >> 8) * tile_bytes;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
    if (c < sp_chunks && g8) {
      // gated_i8: chunk = the 16 stored bytes of lane half h of stage st (slices 16h .. 16h+15 of the stage's 32), each value in units
      // of its column's step and rounded UP, + the lane's position word (element E at bits [2E+1:2E]: 2 * (slice & 1) + bucket)
      const int st = c >> 1, h = c & 1, j0 = st * 32 + h * 16;
      union { half8 h8; int8_t b[16]; } o;
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int bucket = 0, q8 = 0;
        if (rl < n_rows_src && j0 + e < d_dlr) {
          float x = __half2float(src[rl * ld + j0 + e]);
          if (abs_dlr) x = fabsf(x);
          q8 = x > 0.f ? quant_up_i8(x, g8_inv_cs[j0 + e]) : 0;          // NaN / inf -> the row's bound is wrong, exactly as with the fp16 image
          bucket = bucket_of(load_idx(idx, idx_dtype, row * d_dlr + j0 + e), j0 + e, map, 2);
        }
        o.b[e] = (int8_t)q8;
        bits |= (uint32_t)(((e & 1) << 1) | bucket) << (2 * e);
      }
      char* stg = tile + (int64_t)st * S8_STAGE_A;
      const int slot = ((r >> 5) * 2 + h) * 32 + (r & 31);
      *(half8*)(stg + slot * 16) = o.h8;
      // position words: the four 32-row blocks of a 128-row wave tile side by side, so that a lane reads its four words with ONE ds_read_b128
      *(uint32_t*)(stg + S8_A_BYTES + ((((r >> 7) * 2 + h) * 32 + (r & 31)) * 4 + ((r >> 5) & 3)) * 4) = bits;
    } else if (c < sp_chunks) {
      const int st = c >> 2, cc = c & 3, j0 = c * 8;
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int bucket = 0;
        if (rl < n_rows_src && j0 + e < d_dlr) {
          _Float16 x = (_Float16)__half2float(src[rl * ld + j0 + e]);
          if (abs_dlr && x < (_Float16)0.f) x = -x;
          if (i8_inv > 0.f && x > (_Float16)0.f) x = half_up((float)x * i8_inv);     // dense_i8: accumulator units (1 / corpus scale on this side)
          v[e] = x;
          bucket = bucket_of(load_idx(idx, idx_dtype, row * d_dlr + j0 + e), j0 + e, map, 2);
        }
        bits |= (uint32_t)(((e & 1) << 1) | bucket) << (2 * e);
      }
      char* stg = tile + (int64_t)st * SP_STAGE_A;
      *(half8*)(stg + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = v;
      // position words: [32-row block][lane half][row in block] u32 = (block-1 bits << 16 | block-0 bits), so that the 32 lanes
      // of a half read 32 consecutive words (the row-major [row][half] order cost a 2-way bank conflict on every read)
      *(uint16_t*)(stg + SP_A_BYTES + ((((r >> 5) * 2 + (cc & 1)) * 32 + (r & 31)) * 4) + (cc >> 1) * 2) = (uint16_t)bits;
    } else if (i8_inv > 0.f) {
      // int8 stage: 64 columns per 64-byte row, chunk cc = columns 16*cc .. 16*cc+15 (a lane half of v_mfma_i32_32x32x32_i8
      // takes 16 consecutive bytes; the order of the columns inside a 32-deep block is irrelevant as long as both operands agree)
      const int dc = c - sp_chunks, st = dc >> 2, cc = dc & 3, j0 = d_dlr + dc * 16;
      union { half8 h; int8_t b[16]; } o;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        o.b[e] = (rl < n_rows_src && j0 + e < k) ? (int8_t)quant_i8(__half2float(src[rl * ld + j0 + e]), 1.f / col_scale[j0 + e - d_dlr]) : (int8_t)0;
      char* stg = tile + (int64_t)ts * sp_stage + (int64_t)st * SP_DENSE;
      *(half8*)(stg + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = o.h;
    } else {
      const int dc = c - sp_chunks, st = dc >> 2, cc = dc & 3, j0 = d_dlr + dc * 8;
      if (rl < n_rows_src)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (j0 + e < k) v[e] = (_Float16)__half2float(src[rl * ld + j0 + e]);
      char* stg = tile + (int64_t)ts * sp_stage + (int64_t)st * SP_DENSE;
      *(half8*)(stg + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = v;
    }
  }
}
hipError_t launch_tile_rows_sparse(const __half* src, int64_t ld, int64_t row_lo, int64_t n_rows_src, int64_t n_rows_fill,
                                   int d_dlr, int d_cls, int ts, int td, const void* idx, int idx_dtype, const uint8_t* map,
                                   bool abs_dlr, char* tiles, float i8_inv_scale, const float* col_scale, const float* g8_inv_cs, hipStream_t s) {
  if (n_rows_fill <= 0) return hipSuccess;
  const int64_t total = n_rows_fill * ((g8_inv_cs ? ts * 2 : ts * 4) + td * 4);
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(tile_rows_sparse_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, src, ld,
                     row_lo, n_rows_src, n_rows_fill, d_dlr, d_cls, ts, td, idx, idx_dtype, map, abs_dlr ? 1 : 0, tiles, i8_inv_scale, col_scale, g8_inv_cs);
  return hipGetLastError();
}

// gated_i8: 128 x (sum of the row's gated int8 values), one wave per row -- the bound GEMM's accumulators start there (gemm_g8.hip)
__global__ void __launch_bounds__(256) g8_row_sum_kernel(const __half* __restrict__ vals_rm, int k_rm, int64_t n_rows, int64_t n_rows_fill,
                                                         int d_dlr, int abs_dlr, const float* __restrict__ inv_cs, int32_t* __restrict__ rsum128) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < n_rows_fill; row += nwaves) {
    int sum = 0;
    if (row < n_rows)
      for (int j = lane; j < d_dlr; j += 64) {
        float x = __half2float(vals_rm[row * k_rm + j]);
        if (abs_dlr) x = fabsf(x);
        if (x > 0.f) sum += quant_up_i8(x, inv_cs[j]);
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) rsum128[row] = 128 * sum;
  }
}
hipError_t launch_g8_row_sum(const __half* vals_rm, int k_rm, int64_t n_rows, int64_t n_rows_fill, int d_dlr, bool abs_dlr, const float* g8_inv_cs,
                             int32_t* rsum128, hipStream_t s) {
  if (n_rows_fill <= 0) return hipSuccess;
  const int64_t blocks = (n_rows_fill + 3) / 4;
  hipLaunchKernelGGL(g8_row_sum_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, vals_rm, k_rm, n_rows, n_rows_fill, d_dlr,
                     abs_dlr ? 1 : 0, g8_inv_cs, rsum128);
  return hipGetLastError();
}

// Row-major fp16 copy [n][k_rm] (zero padded) that the exact rescoring reads: one contiguous row per pair.
__global__ void __launch_bounds__(256) copy_rows_kernel(const __half* __restrict__ src, int64_t ld, int64_t n_rows, int k,
                                                        int k_rm, __half* __restrict__ dst) {
  const int cpr = k_rm >> 3;
  const int64_t total = n_rows * cpr;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rl = g / cpr;
    const int c = (int)(g - rl * cpr);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c * 8 + e < k) ? (_Float16)__half2float(src[rl * ld + c * 8 + e]) : (_Float16)0.f;
    *(half8*)(dst + rl * k_rm + c * 8) = v;
  }
}
hipError_t launch_copy_rows(const __half* src, int64_t ld, int64_t n_rows, int k, int k_rm, __half* dst, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t blocks = (n_rows * (k_rm >> 3) + 255) / 256;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, src, ld, n_rows,
                     k, k_rm, dst);
  return hipGetLastError();
}

// Value MASS per (slice, 8-bit index value): hist[j][v] = sum of |value| over the rows whose index in slice j is v.
// The bucket maps balance this mass: under independence the expected bound slack of a slice is
// sum_b Mq_b * Md_b - (matches), which for similar query / corpus distributions is least when the buckets carry equal mass
// (balancing the COUNTS instead put the few index values of the tiny-valued background slices on one side and nearly
// all large entries on the other).
__global__ void __launch_bounds__(256) idx_hist_kernel(const uint8_t* __restrict__ idx, const __half* __restrict__ vals_rm, int k_rm,
                                                       int64_t n_rows, int d_dlr, float* __restrict__ hist) {
  // block = (slice group of 64 slices) x (row stripe); LDS histogram 64 x 256
  __shared__ float h[64 * 256];
  for (int i = threadIdx.x; i < 64 * 256; i += 256) h[i] = 0.f;
  __syncthreads();
  const int j0 = blockIdx.x * 64;
  const int lane_j = threadIdx.x & 63;
  const int sub = threadIdx.x >> 6;
  if (j0 + lane_j < d_dlr)
    for (int64_t r = (int64_t)blockIdx.y * 4 + sub; r < n_rows; r += (int64_t)gridDim.y * 4) {
      const float v = fabsf(__half2float(vals_rm[r * k_rm + j0 + lane_j]));
      if (v > 0.f) atomicAdd(&h[lane_j * 256 + idx[r * d_dlr + j0 + lane_j]], v);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 256; i += 256)
    if (h[i] > 0.f && j0 + (i >> 8) < d_dlr) atomicAdd(&hist[(int64_t)(j0 + (i >> 8)) * 256 + (i & 255)], h[i]);
}
hipError_t launch_idx_hist(const uint8_t* idx, const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, float* hist, hipStream_t s) {
  const int64_t stripes = std::min<int64_t>(256, (n_rows + 1023) / 1024);
  hipLaunchKernelGGL(idx_hist_kernel, dim3((d_dlr + 63) / 64, (unsigned)(stripes < 1 ? 1 : stripes)), dim3(256), 0, s, idx, vals_rm, k_rm,
                     n_rows, d_dlr, hist);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ query_prep
// One wave per (padded) query row.  Produces the fp16 operand tile image (|.| on the gated half in
// abs mode), the exact fp32 query used by the rescoring, the int16 index row, and the filter margin
//   margin = 1.05*K_pad*2^-24*||q16||*dmax   (fp32 accumulation error of the MFMA chain, products exact)
//          + ||q32 - q16||*dmax              (fp16 rounding of a query that is not fp16-representable)
// so that  U_computed >= S_exact - margin  for every row (Cauchy-Schwarz on sum |q_j d_j|).
__global__ void __launch_bounds__(256) query_prep_kernel(const void* __restrict__ src, int src_is_f32, int64_t ld,
                                                         const void* __restrict__ idx, int idx_dtype, int64_t ld_idx,
                                                         int n_queries, int q_pad, int d_dlr, int d_cls, int k_rm,
                                                         int n_buckets, int kt, const uint8_t* __restrict__ map,
                                                         int abs_dlr, float dmax, __half* __restrict__ q_tiles,
                                                         float* __restrict__ q32, int16_t* __restrict__ q_idx,
                                                         float* __restrict__ margin, float* __restrict__ tau,
                                                         float* __restrict__ thr, int ts, int td,
                                                         uint32_t* __restrict__ q_pack, __half* __restrict__ q16,
                                                         uint8_t* __restrict__ q_idx8, uint32_t* __restrict__ q_inexact,
                                                         int c_idx_dtype, float i8_scale, float i8_ec, float i8_nc,
                                                         float* __restrict__ i8_mul, const float* __restrict__ col_scale, G8Prep g8) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= q_pad) return;
  const int k = d_dlr + d_cls;
  const int ksteps = kt >> 6;
  const int dlr_cols = n_buckets * d_dlr;
  const bool real = q < n_queries;
  auto qval = [&](int j) -> float {
    if (!real || j >= k) return 0.f;
    return src_is_f32 ? ((const float*)src)[(int64_t)q * ld + j] : __half2float(((const __half*)src)[(int64_t)q * ld + j]);
  };
  auto qidx = [&](int j) -> int { return (real && idx) ? load_idx(idx, idx_dtype, (int64_t)q * ld_idx + j) : 0; };
  // exact fp32 copy (row-major, what the rescoring multiplies with) + norms of q16 and of the fp16 residual
  float s16 = 0.f, sr = 0.f;
  bool inexact = false, zero_chunk = false;
  for (int j = lane; j < k_rm; j += 64) {
    const float v = qval(j);
    q32[(int64_t)q * k_rm + j] = v;
    const float back = (float)(_Float16)v;
    s16 += back * back;
    sr += (v - back) * (v - back);
    inexact |= (back != v) && (v == v);
    {   // a chunk of eight zero columns (8 consecutive lanes): the rescoring kernel then builds per-query chunk masks (q_inexact[1])
      const uint64_t nzb = __ballot(v != 0.f);
      const int c0 = (j - lane) >> 3;
      for (int g = 0; g < 8; ++g)
        if ((c0 + g) * 8 < k && ((nzb >> (8 * g)) & 0xFFull) == 0ull) zero_chunk = true;
    }
    if (q16) {
      // fast rescoring path: a gated value whose index no corpus byte can equal contributes nothing -> store 0
      bool dead = false;
      if (j < d_dlr && real && idx) {
        const int iv = qidx(j);
        dead = (c_idx_dtype == DHR_IDX_U8 && (iv < 0 || iv > 255)) || (c_idx_dtype == DHR_IDX_I8 && (iv < -128 || iv > 127));
      }
      q16[(int64_t)q * k_rm + j] = __float2half(dead ? 0.f : back);
    }
  }
  if (inexact && q_inexact) atomicOr(q_inexact, 1u);
  if (zero_chunk && real && q_inexact && lane == 0) atomicOr(q_inexact + 1, 1u);
  if (q_idx8)
    for (int j = lane; j < d_dlr; j += 64) q_idx8[(int64_t)q * d_dlr + j] = (uint8_t)(qidx(j) & 0xFF);
  for (int j = lane; j < d_dlr; j += 64) {
    const int iv = qidx(j);
    q_idx[(int64_t)q * d_dlr + j] = (int16_t)iv;
    if (q_pack) {      // refine step: fp16 bound-operand value << 16 | bucket << 12 | idx low 12 bits
      union { _Float16 h; uint16_t u; } cv; cv.h = (_Float16)(abs_dlr ? fabsf(qval(j)) : fmaxf(qval(j), 0.f));   // the bound operand
      if (g8.inv_cs && (float)cv.h > 0.f) cv.h = half_up(abs_dlr ? fabsf(qval(j)) : fmaxf(qval(j), 0.f));        // gated_i8: the refine step puts this value's product back, it must not fall below the real one
      const uint32_t bk = (n_buckets > 1 && idx) ? (uint32_t)bucket_of(iv, j, map, n_buckets) : 0u;
      q_pack[(int64_t)q * d_dlr + j] = ((uint32_t)cv.u << 16) | (bk << 12) | ((uint32_t)iv & 0xFFFu);
    }
  }
  // dense_i8 index.  The query's int8 scale sq: max |ungated value| / 127, coarsened (a) until the largest gated value fits fp16
  // in units of sq and (b) until ||q8|| * max_r ||d8_r|| < 2^22 - the integer sums must stay inside the exact range of the
  // accumulators' 2^23 + 2^22 offset (gemm_w4.hip).  Then the norm of the ungated part and of what the int8 image loses: the margin.
  float i8_inv_q = 0.f, i8_sq = 0.f, i8_qn = 0.f, i8_qe = 0.f;
  bool i8_zero = false;
  // gated_i8 index: both halves of the bound are integer sums.  Gated operand of the query: qop_j = max(q_j, 0) (|q_j| in abs mode),
  // weighted by w_j = (step of corpus column j) / s_ref and rounded UP to [0, 255] in the query's own step (8 bits: the operand is
  // stored as level - 128 and the kernel's accumulators start at 128 x the row's sum of gated values); one gated product then
  // counts  q8_j * d8_j  units of  u = step * s_ref >= qop_j * d_j.  The ungated int8 sums are in units of sc * sq, and the kernel
  // shifts the gated sums left by `shift` bits before it adds them: u = 2^shift * u_f with u_f = sc * sq.  The coarser side gives:
  // the gated unit is usually ~50x the ungated one, so shift = floor(log2(ratio)) and sq grows by less than 2x (or, the other way
  // round, the gated step grows to sc * sq / s_ref).
  float g8_inv_sqg = 0.f, g8_uf = 0.f;
  int g8_sh = 0;
  if (g8.inv_cs) {
    const float inv_sc = i8_scale > 0.f ? 1.f / i8_scale : 0.f;
    float am = 0.f, gm = 0.f;
    for (int j = lane; j < k; j += 64) {
      float v = qval(j);
      if (j < d_dlr) v = (abs_dlr ? fabsf(v) : fmaxf(v, 0.f)) * g8.w[j]; else v = fabsf(v * (col_scale[j - d_dlr] * inv_sc));
      if (v <= 3.0e38f) { if (j >= d_dlr) am = fmaxf(am, v); else gm = fmaxf(gm, v); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { am = fmaxf(am, __shfl_xor(am, o, 64)); gm = fmaxf(gm, __shfl_xor(gm, o, 64)); }
    const float u_nat = fmaxf(gm * (1.00001f / 255.f) * g8.s_ref * 1.000001f, 1e-30f);
    if (d_cls > 0 && am > 0.f && i8_scale > 0.f) {
      const float v_nat = i8_scale * (am / 127.f);
      if (u_nat >= v_nat) {
        g8_sh = min(g8.max_shift, max(0, ilogbf(u_nat / v_nat)));
        g8_uf = ldexpf(u_nat, -g8_sh);
      } else g8_uf = v_nat * 1.000001f;
      i8_sq = g8_uf * inv_sc;
    } else {
      g8_uf = u_nat;
      i8_sq = i8_scale > 0.f ? g8_uf * inv_sc : 1.f;
    }
    g8_inv_sqg = g8.s_ref / ldexpf(g8_uf, g8_sh) * 1.000003f;
    for (int j = lane; j < d_dlr; j += 64) {
      const float v = (abs_dlr ? fabsf(qval(j)) : fmaxf(qval(j), 0.f)) * g8.w[j];
      g8.q8[(int64_t)q * d_dlr + j] = (uint8_t)(v > 0.f ? quant_up_u8(v, g8_inv_sqg) : 0);
    }
    if (i8_scale > 0.f) {        // norm of the ungated part and of what its int8 image loses, at the scale chosen above
      i8_inv_q = 1.f / i8_sq;
      float sn = 0.f, se = 0.f;
      for (int j = d_dlr + lane; j < k; j += 64) {
        const float v = qval(j) * (col_scale[j - d_dlr] * inv_sc);
        const float e = v - i8_sq * (float)quant_i8(v, i8_inv_q);
        sn += v * v;
        se += e * e;
      }
      i8_qn = sqrtf(wave_sum(sn));
      i8_qe = sqrtf(wave_sum(se));
    }
  } else
  if (i8_scale > 0.f) {
    // ungated query values enter weighted, q'_j = q_j * col_scale[j] / scale: the corpus side divided column j by the same factor
    // (its own int8 step), <q, d> = <q', d'>
    const float inv_sc = 1.f / i8_scale;
    auto qw = [&](int j) -> float { return qval(j) * (col_scale[j - d_dlr] * inv_sc); };
    float am = 0.f, gm = 0.f;
    for (int j = lane; j < k; j += 64) {
      const float v = fabsf(j >= d_dlr ? qw(j) : qval(j));
      if (v <= 3.0e38f) { if (j >= d_dlr) am = fmaxf(am, v); else gm = fmaxf(gm, v); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { am = fmaxf(am, __shfl_xor(am, o, 64)); gm = fmaxf(gm, __shfl_xor(gm, o, 64)); }
    i8_sq = fmaxf(am > 0.f ? am / 127.f : 1.f, gm / 60000.f);
    const float n8_cap = 4.0e6f / fmaxf(i8_nc / i8_scale, 1.f);         // ||q8|| must stay below this
    for (int it = 0; it < 4; ++it) {
      i8_inv_q = 1.f / i8_sq;
      i8_zero = it == 3;              // last resort (never seen): an all-zero int8 image, the margin pays the whole ungated part
      float sn = 0.f, se = 0.f, s8 = 0.f;
      for (int j = d_dlr + lane; j < k; j += 64) {
        const float v = qw(j);
        const float q8 = i8_zero ? 0.f : (float)quant_i8(v, i8_inv_q);
        const float e = v - i8_sq * q8;
        sn += v * v;
        se += e * e;
        s8 += q8 * q8;
      }
      i8_qn = sqrtf(wave_sum(sn));
      i8_qe = sqrtf(wave_sum(se));
      const float n8 = sqrtf(wave_sum(s8));
      if (n8 <= n8_cap) break;
      if (it < 2) i8_sq *= 1.05f * n8 / n8_cap;
    }
  }
  if (ts + td > 0) {
    // stage layout (2:4).  A sparse stage holds the row's 32 slice values once, the bucket in the sign bit (the bound
    // operand is >= 0): the GEMM expands a value v into its two bucket columns (max(v,0), max(-v,0)) in
    // registers.  Order inside a 16-slice block: [0-3, 8-11 | 4-7, 12-15], the two 16-byte chunks the two
    // lane halves of the smfmac B operand read.  An ungated batch (plain inner product over a gated index)
    // carries the sparse stages twice, all values in bucket 0, then all in bucket 1.  Then 32-column dense stages.
    const int ts_q = (idx || g8.inv_cs) ? ts : 2 * ts;
    char* tile = (char*)q_tiles + (int64_t)(q >> 8) * ((int64_t)ts_q * SP_STAGE_B + (int64_t)td * SP_DENSE);
    const int r = q & 255;
    for (int c = lane; c < ts_q * 4 + td * 4; c += 64) {
      half8 h8;
      if (c < ts_q * 4 && g8.inv_cs) {
        // gated_i8: 64-byte row = the stage's 32 slices x (bucket-0 column, bucket-1 column), natural order; an ungated batch
        // (plain inner product over a gated index) carries the value in both columns
        const int st = c >> 2, cc = c & 3;
        union { half8 h; uint8_t b[16]; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int j = st * 32 + cc * 8 + e;
          int v8 = 0, bk = 0;
          if (j < d_dlr) {
            const float v = (abs_dlr ? fabsf(qval(j)) : fmaxf(qval(j), 0.f)) * g8.w[j];
            v8 = v > 0.f ? quant_up_u8(v, g8_inv_sqg) : 0;
            bk = idx ? bucket_of(qidx(j), j, map, 2) : 2;
          }
          o.b[2 * e] = (uint8_t)((bk != 1 ? v8 : 0) - 128);        // level - 128 as int8; a column that carries nothing holds -128
          o.b[2 * e + 1] = (uint8_t)((bk != 0 ? v8 : 0) - 128);
        }
        *(half8*)(tile + (int64_t)st * SP_STAGE_B + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = o.h;
      } else if (c < ts_q * 4) {
        const int st = c >> 2, cc = c & 3;
        const int kb = cc >> 1, hh = cc & 1;
        const int cst = st < ts ? st : st - ts;
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
          float v = 0.f;
          const int j = cst * 32 + kb * 16 + 8 * (e8 >> 2) + 4 * hh + (e8 & 3);
          bool neg_b = false;
          if (j < d_dlr) {
            v = qval(j);
            v = abs_dlr ? fabsf(v) : fmaxf(v, 0.f);
            neg_b = (idx ? bucket_of(qidx(j), j, map, 2) : (st >= ts)) != 0;
          }
          _Float16 hv = (i8_scale > 0.f && v > 0.f) ? half_up(v * i8_inv_q) : (_Float16)v;     // dense_i8: accumulator units, rounded up
          h8[e8] = neg_b ? -hv : hv;
        }
        *(half8*)(tile + (int64_t)st * SP_STAGE_B + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = h8;
      } else if (i8_scale > 0.f) {
        const int dc = c - ts_q * 4, st = dc >> 2, cc = dc & 3;
        union { half8 h; int8_t b[16]; } o;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = d_dlr + dc * 16 + e;
          o.b[e] = (i8_zero || j >= k) ? (int8_t)0 : (int8_t)quant_i8(qval(j) * (col_scale[j - d_dlr] * (1.f / i8_scale)), i8_inv_q);
        }
        *(half8*)(tile + (int64_t)ts_q * SP_STAGE_B + (int64_t)st * SP_DENSE + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = o.h;
      } else {
        const int dc = c - ts_q * 4, st = dc >> 2, cc = dc & 3;
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) h8[e8] = (_Float16)qval(d_dlr + dc * 8 + e8);
        *(half8*)(tile + (int64_t)ts_q * SP_STAGE_B + (int64_t)st * SP_DENSE + r * 64 + ((cc ^ ((r >> 2) & 3)) * 16)) = h8;
      }
    }
  }
  s16 = wave_sum(s16);
  sr = wave_sum(sr);
  if (lane == 0) {
    float m = 1.05f * (float)(kt + 256) * 5.9604645e-8f * sqrtf(s16) * dmax + 1.0001f * sqrtf(sr) * dmax;
    if (g8.inv_cs) {
      // integer sums: nothing is rounded in the accumulation, the gated operands are rounded UP from the fp32 query, and what the
      // int8 image of the ungated columns loses is measured against the fp32 query: the margin is the ungated Cauchy-Schwarz term
      // (below) + the fp32 conversion of the unit
      m = 1.001f * (i8_qn * i8_ec + i8_qe * i8_nc) + 1.2e-6f * (i8_qn + i8_qe) * i8_nc + 2.f * g8_uf;
      if (!(m >= 0.f)) m = INFINITY;
      i8_mul[q] = real ? g8_uf : 0.f;
      g8.shift[q] = g8_sh;
      g8.unit[q] = ldexpf(g8_uf, g8_sh);
    } else
    if (i8_scale > 0.f) {
      // int8 image of the ungated columns: <q,d> - mul * <q8,d8>  =  <q, d - sc*d8>  +  <q - sq*q8, sc*d8>  +  (sc*sq - mul) <q8,d8>
      // with the corpus-wide maxima ec >= ||d - sc*d8||, nc >= ||sc*d8|| (Cauchy-Schwarz on the first two, fp32 rounding of the
      // factor and of the converted sum on the third)
      // The factor is rounded UP (the gated products up(q/sq) * up(d/sc) * mul must not fall below q*d, which the refine step takes
      // off again), and the fp32 accumulation of the gated stages happens on top of the 2^23-sized offset: <= 1 unit (ulp of
      // [2^23, 2^24), 2 above) per matrix instruction of the chain plus the threshold conversion in the kernel -> 4 * (chain + 8) units.
      const float mul = __uint_as_float(__float_as_uint(i8_scale * i8_sq) + 8u);     // +2^-20: above every rounding of the scaled operands
      m += 1.001f * (i8_qn * i8_ec + i8_qe * i8_nc) + 1.2e-6f * (i8_qn + i8_qe) * i8_nc + 4.f * (float)(2 * (idx ? ts : 2 * ts) + 8) * mul;
      if (!(m >= 0.f)) m = INFINITY;                     // NaN somewhere: filter with -inf thresholds (everything is rescored exactly)
      i8_mul[q] = real ? mul : 0.f;
      if (g8.thr_raise) {
        // residual refine level: the corpus term ||q'|| ec of the margin is replaced by the measured  <q, d - sc*d8>  (up to what the
        // residual image loses, ||q'|| ec2, the fp32 rounding of that sum and of the residuals themselves): the level's threshold is
        // thr + (what the margin no longer has to pay)
        const float gone = 1.001f * i8_qn * i8_ec;
        const float left = 1.01f * i8_qn * g8.resid_ec2 + 1e-3f * gone + 4e-6f * (i8_qn + i8_qe) * i8_nc;
        const float up = gone - left;
        g8.thr_raise[q] = (up > 0.f && m < INFINITY) ? up : 0.f;
      }
    }
    margin[q] = m;
    tau[q] = -INFINITY;
    thr[q] = real ? -INFINITY : INFINITY;              // padded queries never pass the filter
  }
}

hipError_t launch_query_prep(const void* src, int src_is_f32, int64_t ld, const void* idx, int idx_dtype, int64_t ld_idx,
                             int n_queries, int q_pad, int d_dlr, int d_cls, int k_rm, int n_buckets, int kt,
                             const uint8_t* map, bool abs_dlr, float dmax, __half* q_tiles, float* q32, int16_t* q_idx,
                             float* margin, float* tau, float* thr, int ts, int td, uint32_t* q_pack, __half* q16, uint8_t* q_idx8,
                             uint32_t* q_inexact, int c_idx_dtype, float i8_scale, float i8_ec, float i8_nc, float* i8_mul,
                             const float* col_scale, const G8Prep& g8, hipStream_t s) {
  hipLaunchKernelGGL(query_prep_kernel, dim3((q_pad + 3) / 4), dim3(256), 0, s, src, src_is_f32, ld, idx, idx_dtype,
                     ld_idx, n_queries, q_pad, d_dlr, d_cls, k_rm, n_buckets, kt, map, abs_dlr ? 1 : 0, dmax, q_tiles, q32,
                     q_idx, margin, tau, thr, ts, td, q_pack, q16, q_idx8, q_inexact, c_idx_dtype, i8_scale, i8_ec, i8_nc, i8_mul, col_scale, g8);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ gemm_filter
// The bound GEMM U = Q x D^T fused with the per-query threshold filter: 256 corpus rows x 256 queries per workgroup over the STAGE images
// (32 gated slices on the 2:4 sparse matrix instructions / 32 or 64 ungated columns on the dense ones per stage, every stage one contiguous
// LDS image: tile_rows_sparse_kernel, query_prep_kernel).  The kernels live in their own translation units:
//   gemm_g8.hip   gemm_filter_g8_kernel      integer operands (gated_i8 indexes; dense-only int8 indexes with no gated stage): the default
//                                            of every index of a million rows or more
//   gemm_w4.hip   gemm_filter_wx_kernel<NI>  fp16 gated stages (optionally int8 ungated stages), 8 waves (NI = 2) or 4 waves (NI = 4)
// Both take any even number of gated and of ungated stages (dhr_index_create rounds the stage counts up with all-zero stages), so there is
// no other layout any more: the K-step tile layout with its kernel (gemm_filter_v3_kernel: bucket counts other than two, gated widths that
// are no multiple of 32) and the 12-wave producer / consumer kernel (gemm_filter_sparse_kernel: odd stage counts) were retired in round 6
// together with the persistent-workgroup form of the integer kernel (tools/ab/gemm_g8p.hip, built only with DHR_AB_VARIANTS=1).
//
// Workgroup -> tile map (XCD aware): grid = (8 XCDs, DOC_GROUP corpus tiles x query tiles, tile groups): dhr_internal.h gemm_wg_tile.

int g_gemm_variant = 5;   // DHR_PARAM_GEMM_VARIANT of the fp16-gated kernel: 4 (4 waves, 128 x 128 wave tiles) or 5 (8 waves, 128 x 64; default: 2 % faster in the bench)

// which bound-GEMM kernel the calling thread's latest launch_gemm_filter ran (dhr_index_get_info(DHR_INFO_GEMM_KERNEL): a report names the
// kernel that really ran instead of re-deriving the dispatch below): 3 / 4 gemm_filter_wx_kernel<2> / <4>, 5 gemm_filter_g8_kernel,
// 6 gemm_filter_g8p_kernel (A/B builds only); 1 and 2 were the kernels retired in round 6
thread_local int g_last_gemm_kernel = 0;
static hipError_t launch_gemm_filter_grid(const GemmArgs& a, dim3 grid, hipStream_t s);
hipError_t launch_gemm_filter(const GemmArgs& a_in, hipStream_t s) {
  static bool env_read = false;
  if (!env_read) {
    if (const char* e = getenv("DHR_GEMM_VARIANT")) g_gemm_variant = atoi(e) == 4 ? 4 : 5;     // tuning only (the torch-free PMC driver)
    env_read = true;
  }
  const int64_t n_tiles = a_in.seq_hi - a_in.seq_lo;
  if (n_tiles <= 0) return hipSuccess;
  const int64_t groups = (n_tiles + DOC_GROUP - 1) / DOC_GROUP;
  const int64_t groups_per_xcd = (groups + 7) / 8;
  // limits of the 3-D grid and of the division-free tile map; check_queries (search_core.hip) rejects the batches that would hit the first with a
  // proper message, the others cannot be reached through the C ABI (n_rows < 2^32, sample period <= 256)
  if ((int64_t)DOC_GROUP * a_in.n_qtiles > 65535 || a_in.perm_n >= (1 << 24) || a_in.period > (1 << 20)) return hipErrorInvalidValue;
  if (a_in.map_mode >= 2 && a_in.period < 2) return hipErrorInvalidValue;      // "everything but the sample" needs a sample: divmod24 by period - 1
#ifdef DHR_AB_VARIANTS
  // A/B builds (tools/ab_build.sh): the integer kernel with persistent workgroups over the whole launch (tools/ab/gemm_g8p.hip) for
  // DHR_PARAM_GEMM_VARIANT = 6 / DHR_G8_PERSIST=1 -- the measurement that shows the kernel is held by the package power cap
  static const int g8_persist = getenv("DHR_G8_PERSIST") ? atoi(getenv("DHR_G8_PERSIST")) : 0;
  if (a_in.g8_shift && (g8_persist || a_in.variant == 6) && gemm_g8p_ok(a_in)) { g_last_gemm_kernel = 6; return launch_gemm_g8p(a_in, s); }
#endif
  GemmArgs a = a_in;
  a.inv_perm_n = 1.0 / (double)(a.perm_n > 0 ? a.perm_n : 1);
  a.inv_pm1 = 1.0 / (double)(a.period > 1 ? a.period - 1 : 1);
  // grid = (XCD, corpus tile of the group x query tile, tile group of the XCD): gemm_wg_tile (dhr_internal.h); gridDim.z <= 65535, so a
  // launch over more than 65535 x 32 corpus tiles (537 M rows) is cut into several
  if (groups < 8) {      // fewer tile groups than XCDs: the query tiles are dealt over the XCDs instead (gemm_wg_tile, MAP_SPREAD_Q)
    a.map_mode |= MAP_SPREAD_Q;
    const dim3 grid(8u, (unsigned)(DOC_GROUP * ((a.n_qtiles + 7) / 8)), (unsigned)groups);
    return launch_gemm_filter_grid(a, grid, s);
  }
  for (int64_t z0 = 0; z0 < groups_per_xcd; z0 += 65535) {
    GemmArgs c = a;
    c.seq_lo = a.seq_lo + z0 * 8 * DOC_GROUP;
    const dim3 grid(8u, (unsigned)(DOC_GROUP * a.n_qtiles), (unsigned)std::min<int64_t>(65535, groups_per_xcd - z0));
    const hipError_t e = launch_gemm_filter_grid(c, grid, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
static hipError_t launch_gemm_filter_grid(const GemmArgs& a, dim3 grid, hipStream_t s) {
  const int variant = a.variant == 4 ? 4 : a.variant ? 5 : g_gemm_variant;        // per handle (dhr_index_set_param), else the library default
  // every index is a stage image with an even number of stages of either kind (dhr_index_create pads with all-zero stages)
  if (a.ts + a.td <= 0 || (a.ts & 1) || (a.td & 1) || (a.ts_q & 1)) return hipErrorInvalidValue;
  if (a.g8_shift) { g_last_gemm_kernel = 5; return (a.ts > 0 && a.ts_q == a.ts) ? launch_gemm_g8(a, grid, s) : hipErrorInvalidValue; }
  // a dense-only int8 index (no gated stage on either side): the integer kernel of the gated_i8 indexes with ts = 0 -- the same operand
  // images, integer filter epilogue; config 2: 2.78 -> 2.70 ms per launch alone, 76.4 -> 75.1 ms per step (DHR_DENSE_G8=0: gemm_filter_wx_kernel)
  static const int dense_g8 = getenv("DHR_DENSE_G8") ? atoi(getenv("DHR_DENSE_G8")) : 1;
  if (a.i8_mul && dense_g8 && a.ts == 0 && a.ts_q == 0 && a.td > 0) { g_last_gemm_kernel = 5; return launch_gemm_g8(a, grid, s); }
  g_last_gemm_kernel = variant == 4 ? 4 : 3;
  return launch_gemm_wx(a, grid, variant, s);      // fp16 gated stages; ungated stages fp16 or (i8_mul) int8
}

// Flat launches of the per-candidate kernels.  Their natural grid is (blocks of the LONGEST list) x (queries), which
// is mostly empty workgroups when the list lengths are skewed (measured: 0.5-1.2 ms per launch for a few hundred
// thousand candidates).  Instead: offs = exclusive scan of every query's block count, one workgroup per real block,
// which finds its query by binary search.
__global__ void __launch_bounds__(1024) block_offsets_kernel(const uint32_t* __restrict__ cnt, uint32_t cap, int n_queries, uint32_t per,
                                                             uint32_t* __restrict__ offs) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int chunk = (n_queries + 1023) / 1024;
  const int lo = tid * chunk, hi = min(lo + chunk, n_queries);
  uint32_t s = 0;
  for (int q = lo; q < hi; ++q) { uint32_t c = cnt[q]; if (c > cap) c = cap; s += (c + per - 1) / per; }
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0u;
  for (int q = lo; q < hi; ++q) { offs[q] = run; uint32_t c = cnt[q]; if (c > cap) c = cap; run += (c + per - 1) / per; }
  if (tid == 1023) offs[n_queries] = part[1023];
}
hipError_t launch_block_offsets(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t per, uint32_t* offs, hipStream_t s) {
  hipLaunchKernelGGL(block_offsets_kernel, dim3(1), dim3(1024), 0, s, cnt, cap, n_queries, per, offs);
  return hipGetLastError();
}
// The same scan with the rest of the bookkeeping a list set needs before a per-candidate kernel walks it, in ONE launch (the controller
// without read-backs used four: max_u32 + mark_overflow + block_offsets + a fill; at ~8 us each they were ~40 us of every phase of a
// sampled run and of every chunk of a shard's main pass): fullest list -> out_max / out_max2 (stored: one list set per launch), entries ->
// atomicAdd(out_sum / out_sum2) (accumulated over the phases of a search), overflowed lists flag their query, and `zero` (the next level's
// counters) is cleared.
__global__ void __launch_bounds__(1024) lists_ready_kernel(const uint32_t* __restrict__ cnt, uint32_t cap, int n_queries, uint32_t per,
                                                           uint32_t* __restrict__ offs, uint32_t* __restrict__ out_max, uint32_t* __restrict__ out_max2,
                                                           unsigned long long* __restrict__ out_sum, unsigned long long* __restrict__ out_sum2,
                                                           uint32_t* __restrict__ fail_flags, uint32_t* __restrict__ zero, int n_zero,
                                                           const uint32_t* __restrict__ ovf_cap) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t wmax[16];
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x;
  const int chunk = (n_queries + 1023) / 1024;
  const int lo = tid * chunk, hi = min(lo + chunk, n_queries);
  uint32_t s = 0, m = 0;
  unsigned long long tot = 0;
  for (int q = lo; q < hi; ++q) {
    uint32_t c = cnt[q];
    m = c > m ? c : m;
    tot += c;
    const uint32_t cq = ovf_cap ? cap + ovf_cap[q] : cap;          // two-tier lists: the query's own capacity
    if (c > cq) { c = cq; if (fail_flags) fail_flags[q] = 1u; }
    s += (c + per - 1) / per;
  }
  part[tid] = s;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t om = __shfl_xor(m, o, 64);
    m = om > m ? om : m;
    tot += __shfl_xor(tot, o, 64);
  }
  if ((tid & 63) == 0) { wmax[tid >> 6] = m; wsum[tid >> 6] = tot; }
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0u;
  for (int q = lo; q < hi; ++q) { offs[q] = run; uint32_t c = cnt[q]; const uint32_t cq = ovf_cap ? cap + ovf_cap[q] : cap; if (c > cq) c = cq; run += (c + per - 1) / per; }
  if (tid == 1023) offs[n_queries] = part[1023];
  if (tid == 0) {
    for (int i = 1; i < 16; ++i) { m = wmax[i] > m ? wmax[i] : m; tot += wsum[i]; }
    if (out_max) *out_max = m;
    if (out_max2) *out_max2 = m;
    if (out_sum) atomicAdd(out_sum, tot);
    if (out_sum2) atomicAdd(out_sum2, tot);
  }
  for (int j = tid; j < n_zero; j += 1024) zero[j] = 0u;
}
hipError_t launch_lists_ready(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t per, uint32_t* offs, uint32_t* out_max, uint32_t* out_max2,
                              unsigned long long* out_sum, unsigned long long* out_sum2, uint32_t* fail_flags, uint32_t* zero, int n_zero,
                              hipStream_t s, const uint32_t* ovf_cap) {
  hipLaunchKernelGGL(lists_ready_kernel, dim3(1), dim3(1024), 0, s, cnt, cap, n_queries, per, offs, out_max, out_max2, out_sum, out_sum2, fail_flags,
                     zero, n_zero, ovf_cap);
  return hipGetLastError();
}
// Second tier of the bound lists (GemmArgs::ovf): one workgroup; thread t plans a contiguous range of queries, an exclusive scan of the
// segment sizes gives the offsets, and what does not fit into the arena any more is cut (those lists overflow and flag their query).
__global__ void __launch_bounds__(1024) plan_overflow_kernel(const uint32_t* __restrict__ cnt_prev, double rows_ratio, uint32_t cap, uint32_t max_extra,
                                                             uint32_t arena_entries, int n_queries, uint32_t* __restrict__ ovf_off,
                                                             uint32_t* __restrict__ ovf_cap) {
  __shared__ unsigned long long part[1024];
  __shared__ double scale_s;
  const int tid = threadIdx.x;
  const int chunk = (n_queries + 1023) / 1024;
  const int lo = tid * chunk, hi = min(lo + chunk, n_queries);
  auto want = [&](int q) -> uint32_t {          // what the query's list is expected to need beyond its uniform slots, with 2 x head room
    if (!cnt_prev) return 0u;
    const double need = 2.0 * (double)cnt_prev[q] * rows_ratio + 2048.0;
    if (!(need > (double)cap)) return 0u;
    const double extra = need - (double)cap;
    return extra >= (double)max_extra ? max_extra : (uint32_t)extra;
  };
  // pass 1: the batch's total.  If it exceeds the arena EVERY segment shrinks by the same factor (down to half, a segment still holds what its
  // query is predicted to need; handing the arena out in query order instead left the later half of a batch without any -- found at 4
  // shards, where the first chunk of a shard's main pass is 3/4 of the shard: 40 queries per step overflowed and took the repair path)
  unsigned long long s = 0;
  for (int q = lo; q < hi; ++q) s += want(q);
  part[tid] = s;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) {
    if (tid < d) part[tid] += part[tid + d];
    __syncthreads();
  }
  if (tid == 0) {
    const unsigned long long total = part[0] + 256ull * (unsigned long long)n_queries;      // (+ the rounding of every segment up to 256)
    scale_s = total > arena_entries ? (double)arena_entries / (double)total : 1.0;
  }
  __syncthreads();
  const double scale = scale_s;
  auto seg = [&](int q) -> uint32_t {
    const uint32_t w = want(q);
    if (w == 0u) return 0u;
    const uint32_t v = scale < 1.0 ? (uint32_t)((double)w * scale) : w;
    return (v + 255u) & ~255u;
  };
  __syncthreads();
  // pass 2: exclusive scan of the segment sizes
  s = 0;
  for (int q = lo; q < hi; ++q) s += seg(q);
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned long long run = tid ? part[tid - 1] : 0ull;
  for (int q = lo; q < hi; ++q) {
    uint32_t w = seg(q);
    if (run >= arena_entries) w = 0u;
    else if (run + w > arena_entries) w = (uint32_t)(arena_entries - run);
    ovf_off[q] = (uint32_t)(run < arena_entries ? run : arena_entries);
    ovf_cap[q] = w;
    run += seg(q);
  }
}
hipError_t launch_plan_overflow(const uint32_t* cnt_prev, double rows_ratio, uint32_t cap, uint32_t max_extra, uint32_t arena_entries, int n_queries,
                                uint32_t* ovf_off, uint32_t* ovf_cap, hipStream_t s) {
  hipLaunchKernelGGL(plan_overflow_kernel, dim3(1), dim3(1024), 0, s, cnt_prev, rows_ratio, cap, max_extra, arena_entries, n_queries, ovf_off, ovf_cap);
  return hipGetLastError();
}
__device__ __forceinline__ bool flat_block(const uint32_t* __restrict__ offs, int n_queries, uint32_t b, int& q, uint32_t& blk) {
  if (b >= offs[n_queries]) return false;
  int lo = 0, hi = n_queries;              // largest q with offs[q] <= b (offs[q+1] > b picks the non-empty one)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offs[mid] <= b) lo = mid; else hi = mid;
  }
  q = lo; blk = b - offs[lo];
  return true;
}

// ------------------------------------------------------------------------------------------ heavy lists + refine
// Build (once per index): per row the HEAVY largest-magnitude gated entries, as
//   key = slice << 20 | bucket << 16 | index value (16 bits),  val = the fp16 value;  unused slots key = ~0.
// One wave per row, HEAVY rounds of wave-wide arg-max over a register copy of the row (d_dlr <= 1024).
template <int SL>      // key registers per lane: 16 (d_dlr <= 1024) or 64 (<= 4096)
__global__ void __launch_bounds__(256) heavy_build_kernel(const __half* __restrict__ vals_rm, int k_rm,
                                                          const void* __restrict__ idx, int idx_dtype, int64_t n_rows,
                                                          int d_dlr, const uint8_t* __restrict__ map, int n_buckets,
                                                          uint32_t* __restrict__ heavy_key, __half* __restrict__ heavy_val,
                                                          const float* __restrict__ g8_inv_cs, int abs_mode) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < n_rows; row += nwaves) {
    constexpr int MAXJ = 64 * SL - 1;
    uint32_t key[SL];        // (|fp16| bits << 16) | (MAXJ - slice): larger value first, lower slice on ties
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      const int j = lane + 64 * sl;
      key[sl] = 0u;
      if (j < d_dlr) {
        union { _Float16 h; uint16_t u; } cv; cv.h = (_Float16)__half2float(vals_rm[row * k_rm + j]);
        const uint32_t mag = cv.u & 0x7FFFu;
        if (mag) key[sl] = (mag << 16) | (uint32_t)(MAXJ - j);
      }
    }
    for (int r = 0; r < HEAVY; ++r) {
      uint32_t best = 0u;
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) best = key[sl] > best ? key[sl] : best;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t ob = __shfl_xor(best, o, 64); best = ob > best ? ob : best; }
      if (best == 0u) {                       // fewer than HEAVY non-zero entries: pad
        if (lane == 0) for (int rr = r; rr < HEAVY; ++rr) { heavy_key[row * HEAVY_KEY_STRIDE + rr] = 0xFFFFFFFFu; heavy_val[row * HEAVY_VAL_STRIDE + rr] = __float2half(0.f); }
        break;
      }
      const int j = MAXJ - (int)(best & 0xFFFFu);
      if ((j & 63) == lane) {                 // owner lane writes the entry and retires it
        const int iv = load_idx(idx, idx_dtype, row * d_dlr + j);
        const uint32_t bk = n_buckets > 1 ? (uint32_t)bucket_of(iv, j, map, n_buckets) : 0u;
        uint32_t key_out = ((uint32_t)j << 20) | (bk << 16) | ((uint32_t)iv & 0xFFFFu);
        if (g8_inv_cs) {
          // gated_i8 index (two buckets: one bucket bit; the refine step compares 12 index bits): the entry's int8 operand level -- what the
          // bound GEMM counted for it -- travels in the key's idle bits [19:17] and [15:12], so that the refine step neither looks the
          // column's step up nor repeats the quantisation per candidate (quant_up_i8: the tile builder's expression)
          float d = __half2float(vals_rm[row * k_rm + j]);
          if (abs_mode) d = fabsf(d);
          const uint32_t lvl = d > 0.f ? (uint32_t)quant_up_i8(d, g8_inv_cs[j]) : 0u;
          key_out = ((uint32_t)j << 20) | ((lvl >> 4) << 17) | ((bk & 1u) << 16) | ((lvl & 0xFu) << 12) | ((uint32_t)iv & 0xFFFu);
        }
        heavy_key[row * HEAVY_KEY_STRIDE + r] = key_out;
        heavy_val[row * HEAVY_VAL_STRIDE + r] = vals_rm[row * k_rm + j];
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) if (sl == (j >> 6)) key[sl] = 0u;
      }
    }
  }
}
hipError_t launch_heavy_build(const __half* vals_rm, int k_rm, const void* idx, int idx_dtype, int64_t n_rows, int d_dlr,
                              const uint8_t* map, int n_buckets, uint32_t* heavy_key, __half* heavy_val, const float* g8_inv_cs, int abs_mode, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t blocks = (n_rows + 3) / 4;
  if (d_dlr <= 1024)
    hipLaunchKernelGGL(heavy_build_kernel<16>, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, vals_rm, k_rm,
                       idx, idx_dtype, n_rows, d_dlr, map, n_buckets, heavy_key, heavy_val, g8_inv_cs, abs_mode);
  else
    hipLaunchKernelGGL(heavy_build_kernel<64>, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, vals_rm, k_rm,
                       idx, idx_dtype, n_rows, d_dlr, map, n_buckets, heavy_key, heavy_val, g8_inv_cs, abs_mode);
  return hipGetLastError();
}

// Refine: the bound U of a candidate counted every gated slice whose BUCKETS agree; for the row's heavy
// entries we can afford to look at the real index: same bucket but different index value contributes
// nothing to the exact score, so |q_j d_j| is taken off the bound.  U2 = U - sum(corr) is still an
// upper bound of the exact score (only certain mismatches are removed; the query side keeps 12 index
// bits, an alias there only makes the bound looser).  8 lanes per candidate, 8 heavy entries per lane.
// (Round 4: marking the row gathers of the refine / rescoring kernels non-temporal -- they are read once, the idea being that they should
// not evict the bound GEMM's operand tiles from L2 when the two overlap -- made BOTH kernels slower, with and without overlap: refine
// 12.0 -> 14.1 ms, rescoring 15.8 -> 16.6 ms per config-3 step, step 121.7 -> 124.5 ms.  Plain loads stay.)
static __device__ __forceinline__ uint4 gather16(const void* p) { return *(const uint4*)p; }
static __device__ __forceinline__ uint2 gather8(const void* p) { return *(const uint2*)p; }
// (Round 4: refine in two levels -- the record as four blocks of 16 entries, each its keys then its values; the candidate's first two lanes
// read block 0 = the 16 heaviest entries = one 128-byte line, the candidate is re-tested, and only the survivors' other six lanes read blocks
// 1-3 -- prunes 60 % of the candidates after one line instead of three, and was SLOWER: refine 11.3 -> 15.9 ms per config-3 step.  The kernel
// is bound by the round trips a wave waits for, not by bytes: nearly every wave holds a survivor among its 8 candidates and then pays two
// dependent gathers per iteration.  It would take a compacting pass between the levels (a third list set); not built.)
#ifndef REFINE_PREFETCH
#define REFINE_PREFETCH 2   // rounds of a lane group in flight: 0 = entry -> record -> lookups in sequence, 1 = the next entries ahead, 2 = + the next record
#endif
#ifndef REFINE_WPE
#define REFINE_WPE 8      // waves per SIMD the register allocation aims at: 8 = 64 registers (8 bytes of scratch); the compiler's own choice (66: 7 waves) runs 9.95 instead of 8.85 ms per config-3 step
#endif
#if REFINE_WPE > 0
#define REFINE_ATTR __attribute__((amdgpu_waves_per_eu(REFINE_WPE)))
#else
#define REFINE_ATTR
#endif
template <bool G8>
__global__ void __launch_bounds__(256) REFINE_ATTR refine_kernel(RefineArgs p) {
  extern __shared__ uint32_t qw[];             // [d_dlr] query words (up to 4096 slices: the slice id has 12 bits in a heavy-list key);
                                               // G8: [d_dlr] pairs {query word, the query's int8 operand level}: one 8-byte LDS read per listed entry
  int q = blockIdx.y;
  uint32_t blk = blockIdx.x;
  for (uint32_t fb = blockIdx.x;; fb += gridDim.x) {       // flat launches: grid stride over the block list (see rescore_kernel); else one pass
  if (p.blk_off && !flat_block(p.blk_off, p.n_queries, fb, q, blk)) return;
  uint32_t count = p.cnt[q];
  const uint32_t cap_q = p.ovf_cap ? p.cap + p.ovf_cap[q] : p.cap;      // two-tier lists (GemmArgs::ovf)
  if (count > cap_q) count = cap_q;
  const uint32_t base = blk * REFINE_PER_WG;
  if (base >= count) { if (p.blk_off) continue; return; }
  __syncthreads();                                           // the previous block's readers are done with the staged query words
  for (int j = threadIdx.x; j < p.d_dlr; j += 256) {
    if constexpr (G8) { qw[2 * j] = p.q_pack[(int64_t)q * p.d_dlr + j]; qw[2 * j + 1] = p.g8_q8[(int64_t)q * p.d_dlr + j]; }
    else qw[j] = p.q_pack[(int64_t)q * p.d_dlr + j];
  }
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const float t = p.thr[q];
  const double unit = G8 ? (double)p.g8_unit[q] : 0.0;
  // (A variant that issued the loads of all 8 candidates of a lane group up front ran 30 % SLOWER: 4x the gathers
  // in flight per CU only thrash the memory system; the dependent chain below at 8 waves per SIMD is the sweet spot.)
  // the list entry of the NEXT round is loaded before the current one's record is gathered: one dependent round trip per round instead of two
  auto entry = [&](uint32_t i) __attribute__((always_inline)) -> uint2 {
    if (i >= count || i >= base + REFINE_PER_WG) return make_uint2(0u, 0u);
    return i < p.cap ? p.cand[(int64_t)q * p.cap + i] : p.ovf[(size_t)p.ovf_off[q] + (i - p.cap)];
  };
  // ... and so is the next round's RECORD: two rounds of a lane group are in flight (entry of round r + 2, record of round r + 1) while
  // round r is looked up.  (All 8 records of a lane group up front thrash the memory system, see above; one ahead does not.)
  constexpr int EPL = HEAVY / 8;              // entries per lane: 8 (two 16-byte key loads + one 16-byte value load) or 4 (one + an 8-byte one)
  static_assert(EPL == 8 || EPL == 4, "8 lanes per candidate read 8 or 4 entries each");
  struct Rec { uint4 k0, k1, hv; };
  auto record = [&](uint32_t i, const uint2 c, Rec& r) __attribute__((always_inline)) {
    r.k0 = r.k1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    r.hv = make_uint4(0u, 0u, 0u, 0u);
    if (i < count && i < base + REFINE_PER_WG) {
      const uint32_t* hk = p.heavy_key + (int64_t)c.x * HEAVY_KEY_STRIDE + sub * EPL;
      r.k0 = gather16(hk);
      if constexpr (EPL == 8) {
        r.k1 = gather16(hk + 4);
        r.hv = gather16(p.heavy_val + (int64_t)c.x * HEAVY_VAL_STRIDE + sub * 8);
      } else {
        const uint2 v2 = gather8(p.heavy_val + (int64_t)c.x * HEAVY_VAL_STRIDE + sub * 4);
        r.hv = make_uint4(v2.x, v2.y, 0u, 0u);
      }
    }
  };
  const uint32_t i_first = base + (threadIdx.x >> 3);
#if REFINE_PREFETCH >= 1
  uint2 c_cur = entry(i_first), c_nx = entry(i_first + 32);
#endif
#if REFINE_PREFETCH >= 2
  Rec r_cur;
  record(i_first, c_cur, r_cur);
#endif
  for (uint32_t i = i_first; i < base + REFINE_PER_WG; i += 32) {
    float corr = 0.f;
    int taken = 0;              // G8: operand products (integer units) of the listed same-bucket entries
    double back = 0.0;          // G8: real-valued products of those whose index values agree
#if REFINE_PREFETCH >= 2
    const uint2 c = c_cur;
    const Rec rc = r_cur;
    Rec r_nx;
    record(i + 32, c_nx, r_nx);
    const uint2 c_nx2 = entry(i + 64);
    c_cur = c_nx; c_nx = c_nx2; r_cur = r_nx;
#elif REFINE_PREFETCH == 1
    const uint2 c = c_cur;
    c_cur = c_nx; c_nx = entry(i + 64);
    Rec rc;
    record(i, c, rc);
#else
    const uint2 c = entry(i);
    Rec rc;
    record(i, c, rc);
#endif
    if (i < count) {
      const uint4 k0 = rc.k0, k1 = rc.k1;
      union { uint4 u; half8 h; } hvu;
      hvu.u = rc.hv;
      const half8 hv = hvu.h;
      const uint32_t keys[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const uint32_t key = keys[e];
        if (key != 0xFFFFFFFFu) {
          const uint32_t j = key >> 20;
          if constexpr (G8) {
            const uint2 wq = *(const uint2*)(qw + 2 * j);
            const uint32_t w = wq.x;
            const bool same_bucket = p.ungated || ((w >> 12) & 0x1u) == ((key >> 16) & 0x1u);
            const bool mismatch = !p.ungated && (w & 0xFFFu) != (key & 0xFFFu);
            const int lvl = (int)(((key >> 17) & 0x7u) << 4 | ((key >> 12) & 0xFu));      // the entry's int8 operand level (heavy_build_kernel)
            if (same_bucket && lvl > 0) {
              taken += (int)wq.y * lvl;
              if (!mismatch) {
                union { uint16_t u; _Float16 h; } qv; qv.u = (uint16_t)(w >> 16);
                const float d = p.abs_mode ? fabsf((float)hv[e]) : (float)hv[e];
                back += (double)(float)qv.h * (double)d;
              }
            }
          } else {
            const uint32_t w = qw[j];
            const bool same_bucket = ((w >> 12) & 0xFu) == ((key >> 16) & 0xFu);
            const bool mismatch = (w & 0xFFFu) != (key & 0xFFFu);
            union { uint16_t u; _Float16 h; } qv; qv.u = (uint16_t)(w >> 16);
            if (same_bucket && mismatch) corr += fabsf((float)qv.h * (float)hv[e]);
          }
        }
      }
    }
    if constexpr (G8) {
      taken += __shfl_xor(taken, 1, 64); taken += __shfl_xor(taken, 2, 64); taken += __shfl_xor(taken, 4, 64);
      back += __shfl_xor(back, 1, 64); back += __shfl_xor(back, 2, 64); back += __shfl_xor(back, 4, 64);
    } else {
      corr += __shfl_xor(corr, 1, 64);
      corr += __shfl_xor(corr, 2, 64);
      corr += __shfl_xor(corr, 4, 64);
    }
    if (sub == 0 && i < count) {
      float u2;
      if constexpr (G8) {
        // U counted `taken` units for the listed same-bucket entries; the entries whose index values agree contribute their real
        // product, the others nothing.  fp64: exact up to 2^-53 relative; the result is rounded UP to fp32.
        const double v = (double)__uint_as_float(c.y) - (double)taken * unit + back;
        u2 = (float)v;
        if ((double)u2 < v) u2 = nextafterf(u2, INFINITY);
      } else u2 = __uint_as_float(c.y) - corr;
      if (u2 >= t) {
        const uint32_t slot = atomicAdd(p.out_cnt + q, 1u);
        if (slot < p.out_cap) p.out[(int64_t)q * p.out_cap + slot] = make_uint2(c.x, __float_as_uint(u2));
      }
    }
  }
  if (!p.blk_off) return;
  }
}
// ---- dense-only int8 index: residual image and its refine level (RefineArgs::resid8)
// FOUR bits per value: nibble = 8 + rint((d - cs d8) * 14 / cs) in [1, 15] (what the int8 image lost, in 1/14 of the column's step; columns 2b and
// 2b + 1 in the low and high nibble of byte b).  A 768-column row is 384 bytes = three 128-byte lines: the level costs half the lines of an 8-bit
// residual and leaves 1/15 of the corpus term to the margin instead of 1/255 (+7 % survivors).
__global__ void __launch_bounds__(256) resid_build_kernel(const __half* __restrict__ vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls,
                                                          const float* __restrict__ col_scale, uint8_t* __restrict__ resid8, int ld) {
  const int64_t total = n_rows * (ld / 8);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = g / (ld / 8);
    const int c0 = (int)(g - row * (ld / 8)) * 16;                       // 16 columns = 8 bytes per thread
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = c0 + e;
      int r4 = 0;
      if (c < d_cls) {
        const float cs = col_scale[c];
        const float f = __half2float(vals_rm[row * k_rm + d_dlr + c]);
        const float rho = f - cs * (float)quant_i8(f, 1.f / cs);         // |rho| <= cs / 2 (a value beyond 127 steps does not occur: cs >= column maximum / 127)
        float t = rintf(rho * (14.f / cs));
        t = t < -7.f ? -7.f : (t > 7.f ? 7.f : t);                       // NaN / inf values: the margin of such an index is infinite anyway
        r4 = (int)t;
      }
      w[e >> 3] |= (uint32_t)(r4 + 8) << (4 * (e & 7));
    }
    *(uint2*)(resid8 + row * ld + c0 / 2) = make_uint2(w[0], w[1]);
  }
}
hipError_t launch_resid_build(const __half* vals_rm, int k_rm, int64_t n_rows, int d_dlr, int d_cls, const float* col_scale, uint8_t* resid8, int resid_ld, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int64_t blocks = (n_rows * (resid_ld / 8) + 255) / 256;
  hipLaunchKernelGGL(resid_build_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, vals_rm, k_rm, n_rows, d_dlr, d_cls, col_scale, resid8, resid_ld);
  return hipGetLastError();
}
// 16 lanes per candidate, CH columns of its residual row each (768 columns: 48 = three 8-byte loads; one load instruction of the candidate's
// 16 lanes = 128 contiguous bytes = one line).  The query factors a_c = q_c cs_c / 14 are quantised ONCE per workgroup to int8 (a_c = sa a8_c +
// da_c; what that loses is at most 7 sum_c |da_c|, summed exactly here and taken off the level's threshold) so that eight columns cost
// two v_dot4_i32_i8 and two mask operations instead of 32 vector instructions -- the first cut (fp32 factors, one shift + mask + convert +
// multiply-add per nibble) spent more time on its arithmetic than on its gathers.  Integer sums: nothing is rounded before the final conversion.
#ifndef DENSE_REFINE_WPE
#define DENSE_REFINE_WPE 8   // 64 registers, 8 waves per SIMD (the compiler picks 68: 7 waves): 10.4 -> 9.85 ms per config-2 step alone
#endif
#if DENSE_REFINE_WPE > 0
#define DENSE_REFINE_ATTR __attribute__((amdgpu_waves_per_eu(DENSE_REFINE_WPE)))
#else
#define DENSE_REFINE_ATTR
#endif
#ifndef DENSE_REFINE_LPC
#define DENSE_REFINE_LPC 16      // lanes per candidate: 16 (8-byte pieces, 4 candidates per wave per round) or 8 (16-byte pieces, 8 candidates per wave per round)
#endif
template <int CH>
__global__ void __launch_bounds__(256) DENSE_REFINE_ATTR dense_refine_kernel(RefineArgs p) {
  constexpr int LPC = DENSE_REFINE_LPC, PB = 128 / LPC, WPP = PB / 4;      // lanes per candidate; bytes / 32-bit words of a lane's piece of a 128-byte line
  static_assert(LPC == 16 || LPC == 8, "one load instruction of a candidate's lanes = one 128-byte line");
  __shared__ float a_s[1024];
  __shared__ uint32_t f_s[2 * 128];        // packed int8 factors: [0, 128) even columns of every group of eight, [128, 256) odd columns
  __shared__ float red[3 * 4];             // per wave: max |a|, then sum a8 and sum |da| (as floats: both are below 2^24 in magnitude / harmlessly rounded UP below)
  int q = blockIdx.y;
  uint32_t blk = blockIdx.x;
  const int sub = threadIdx.x & (LPC - 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NCOL = CH * 16;
  for (uint32_t fb = blockIdx.x;; fb += gridDim.x) {       // flat launches: grid stride over the block list (see rescore_kernel); else one pass
  if (p.blk_off && !flat_block(p.blk_off, p.n_queries, fb, q, blk)) return;
  uint32_t count = p.cnt[q];
  const uint32_t cap_q = p.ovf_cap ? p.cap + p.ovf_cap[q] : p.cap;      // two-tier lists (GemmArgs::ovf)
  if (count > cap_q) count = cap_q;
  const uint32_t base = blk * REFINE_PER_WG;
  if (base >= count) { if (p.blk_off) continue; return; }
  __syncthreads();                                           // the previous block's readers are done with the staged factors
  float mx = 0.f;
  for (int j = threadIdx.x; j < NCOL; j += 256) {
    const float a = j < p.d_cls ? p.q32[(int64_t)q * p.q32_ld + j] * p.col_scale[j] * (1.f / 14.f) : 0.f;
    a_s[j] = a;
    mx = fmaxf(mx, fabsf(a));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sa = mx > 0.f && mx < INFINITY ? mx * (1.f / 127.f) : 1.f, inv_sa = 1.f / sa;
  float s8 = 0.f, sd = 0.f;
  for (int g8 = threadIdx.x; g8 < NCOL / 8; g8 += 256) {     // one group of eight columns per thread: its two packed factor words
    uint32_t fe = 0u, fo = 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = a_s[g8 * 8 + e];
      float r = rintf(a * inv_sa);
      r = r < -127.f ? -127.f : (r > 127.f ? 127.f : r);     // (NaN: the index's margin is infinite anyway)
      s8 += r;
      sd += fabsf(a - sa * r);
      const uint32_t b = (uint32_t)(int)r & 0xffu;
      if (e & 1) fo |= b << (8 * (e >> 1)); else fe |= b << (8 * (e >> 1));
    }
    f_s[g8] = fe;
    f_s[128 + g8] = fo;
  }
  s8 = wave_sum(s8); sd = wave_sum(sd);
  if (lane == 0) { red[4 + wave] = s8; red[8 + wave] = sd; }
  __syncthreads();
  s8 = red[4] + red[5] + red[6] + red[7];
  sd = (red[8] + red[9] + red[10] + red[11]) * 1.0001f;
  // lane `sub` owns the PB-byte pieces sub, LPC + sub, ... of a row = columns 256 u + 2 PB sub ..: groups of eight 32 u + WPP sub + h, h < WPP
  constexpr int U = CH / 16;
  uint32_t fe[WPP * U], fo[WPP * U];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int h = 0; h < WPP; ++h) { fe[WPP * u + h] = f_s[32 * u + WPP * sub + h]; fo[WPP * u + h] = f_s[128 + 32 * u + WPP * sub + h]; }
  const int off = 8 * (int)s8;                               // the residuals are stored with an offset of 8 (s8 is an integer below 2^17: exact)
  const float t = p.thr[q] + p.thr_raise[q] - 7.f * sd;      // 7 sum |da|: what the int8 factors can be off by over a row's residuals
  // the loads of the NEXT 16 candidates go out before the current ones are summed (the kernel waits for round trips, not for bytes)
  struct Piece { uint32_t w[WPP]; };
  constexpr int CPR = 256 / LPC;               // candidates of the workgroup per round
  auto fetch = [&](uint32_t i, uint2& c, Piece (&v)[U]) __attribute__((always_inline)) {
    c = make_uint2(0u, 0u);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int h = 0; h < WPP; ++h) v[u].w[h] = 0x88888888u;
    if (i < count) {
      c = i < p.cap ? p.cand[(int64_t)q * p.cap + i] : p.ovf[(size_t)p.ovf_off[q] + (i - p.cap)];
      const uint8_t* r = p.resid8 + (int64_t)c.x * p.resid_ld + sub * PB;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (PB == 8) { const uint2 t = gather8(r + u * 128); v[u].w[0] = t.x; v[u].w[1] = t.y; }
        else { const uint4 t = gather16(r + u * 128); v[u].w[0] = t.x; v[u].w[1] = t.y; v[u].w[WPP - 2] = t.z; v[u].w[WPP - 1] = t.w; }
      }
    }
  };
  uint2 c;
  Piece v[U];
  fetch(base + (threadIdx.x / LPC), c, v);
  for (uint32_t i = base + (threadIdx.x / LPC); i < base + REFINE_PER_WG; i += CPR) {
    uint2 cn;
    Piece vn[U];
    fetch(i + CPR < base + REFINE_PER_WG ? i + CPR : 0xffffffffu, cn, vn);
    int isum = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int h = 0; h < WPP; ++h) {
        isum = __builtin_amdgcn_sdot4((int)fe[WPP * u + h], (int)(v[u].w[h] & 0x0f0f0f0fu), isum, false);
        isum = __builtin_amdgcn_sdot4((int)fo[WPP * u + h], (int)((v[u].w[h] >> 4) & 0x0f0f0f0fu), isum, false);
      }
    }
#pragma unroll
    for (int o = 1; o < LPC; o <<= 1) isum += __shfl_xor(isum, o, 64);
    if (sub == 0 && i < count) {
      float u2 = __uint_as_float(c.y) + sa * (float)(isum - off);
      u2 += fabsf(u2) * 4.8e-7f;                             // the product and the sum rounded up
      if (u2 >= t) {
        const uint32_t slot = atomicAdd(p.out_cnt + q, 1u);
        if (slot < p.out_cap) p.out[(int64_t)q * p.out_cap + slot] = make_uint2(c.x, __float_as_uint(u2));
      }
    }
    c = cn;
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = vn[u];
  }
  if (!p.blk_off) return;
  }
}
hipError_t launch_refine(const RefineArgs& a, hipStream_t s) {
  if (a.max_count == 0 || a.n_queries <= 0) return hipSuccess;
  if (a.resid8) {
    const dim3 grid = a.blk_off ? dim3(std::min<uint32_t>(a.flat_blocks, FLAT_GRID_MAX)) : dim3((a.max_count + REFINE_PER_WG - 1) / REFINE_PER_WG, (unsigned)a.n_queries);
    if (a.blk_off && !a.flat_blocks) return hipSuccess;
    switch (a.resid_ld / 8) {            // columns per lane = 2 x row bytes / 16 lanes
      case 16: hipLaunchKernelGGL(dense_refine_kernel<16>, grid, dim3(256), 0, s, a); break;
      case 32: hipLaunchKernelGGL(dense_refine_kernel<32>, grid, dim3(256), 0, s, a); break;
      case 48: hipLaunchKernelGGL(dense_refine_kernel<48>, grid, dim3(256), 0, s, a); break;
      case 64: hipLaunchKernelGGL(dense_refine_kernel<64>, grid, dim3(256), 0, s, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const bool g8 = a.g8_q8 != nullptr;
  const size_t lds = (size_t)a.d_dlr * (g8 ? 8 : 4);
  const dim3 grid = a.blk_off ? dim3(std::min<uint32_t>(a.flat_blocks, FLAT_GRID_MAX)) : dim3((a.max_count + REFINE_PER_WG - 1) / REFINE_PER_WG, (unsigned)a.n_queries);
  if (a.blk_off && !a.flat_blocks) return hipSuccess;
  if (g8) hipLaunchKernelGGL(refine_kernel<true>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(refine_kernel<false>, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ rescore
// Exact score of (query, row) pairs.  One wave per pair; lane l owns the 16-byte chunks l, l+64, ...
// of the row (read from the row-major fp16 copy of the corpus: one contiguous row per pair).
// Products of the stored fp16 values with the fp32 query are exact in fp64; the sum is accumulated
// in fp64 and rounded once to fp32, so the result does not depend on tiling, chunking or sharding.
// exact product of two fp16 values as fp32 (11 + 11 significand bits), one instruction, straight from packed registers
__device__ __forceinline__ float fmix_lo(uint32_t a, uint32_t b) { float r; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fmix_hi(uint32_t a, uint32_t b) { float r; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
// per fp16 pair: 0xFFFF where the index bytes agree (x = corpus bytes ^ query bytes; sel places two of them in the 16-bit lanes)
__device__ __forceinline__ uint32_t pair_mask(uint32_t x, uint32_t sel) {
  union { uint32_t u; ushort2v v; } t, one;
  t.u = __builtin_amdgcn_perm(0u, x, sel);
  one.u = 0x00010001u;
  t.v = __builtin_elementwise_min(t.v, one.v);
  t.v = t.v - one.v;
  return t.u;
}
// the same for 16-bit index values: x = two corpus ^ query index values, one per fp16 of the pair
__device__ __forceinline__ uint32_t pair_mask16(uint32_t x) {
  union { uint32_t u; ushort2v v; } t, one;
  t.u = x;
  one.u = 0x00010001u;
  t.v = __builtin_elementwise_min(t.v, one.v);
  t.v = t.v - one.v;
  return t.u;
}
// PATH 0: every path (72 registers with 60 bytes of scratch: the allocation is the maximum over the paths).  PATH 1 / 2: the instantiations of
// rescore_fast_kernel / rescore_narrow_kernel -- every query fp16-representable with no all-zero chunk, rows of more than / at most 128 chunks:
// the general loop's fast branch alone (52 registers) / the 16-lanes-per-pair path alone.
template <int PATH>
__device__ __forceinline__ void rescore_block(const RescoreArgs& p, const int q, const uint32_t blk) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t count;
  if (p.cand) { count = p.cnt[q]; if (count > p.cap) count = p.cap; }
  else if (p.rows32) count = p.count_all;
  else count = p.count_all;
  const uint32_t base = blk * RESCORE_CANDS_PER_WG;
  if (base >= count) return;
  const float* q32 = p.q32 + (int64_t)q * p.k_rm;
  const int16_t* qi = p.q_idx + (int64_t)q * p.d_dlr;
  const int nchunks = p.k_rm >> 3;
  const int dlr_chunks = p.d_dlr >> 3;
  const bool fast = p.q16 && *p.q_inexact == 0u && (!p.gate || p.d_dlr == 0 || p.c_idx_dtype == DHR_IDX_U8 || p.c_idx_dtype == DHR_IDX_I8);
  // Which of this lane's chunks (c = lane + 64 t, bit t) hold a non-zero query value: computed once per block, BEFORE the candidate
  // loop, so that the skip below does not put a query load in front of every corpus load.  The restricted batches of the two-stage
  // modes (gip_retrieval.py:130-136: only the columns with q > theta) keep 4-12 of 1 536 columns.
  // (only for batches in which SOME query has an all-zero chunk -- q_inexact[1], set by query_prep_kernel: an ordinary batch pays nothing)
  const bool some_zero = p.q_inexact && p.q_inexact[1] != 0u;
  uint32_t nz_mask = some_zero ? 0u : 0xFFFFFFFFu;
  for (int c = lane, t = 0; some_zero && c < nchunks; c += 64, ++t) {
    const float4 qa = *(const float4*)(q32 + c * 8);
    const float4 qb = *(const float4*)(q32 + c * 8 + 4);
    if (qa.x != 0.f || qa.y != 0.f || qa.z != 0.f || qa.w != 0.f || qb.x != 0.f || qb.y != 0.f || qb.z != 0.f || qb.w != 0.f) nz_mask |= 1u << t;
  }
  // Sparse queries (at most 16 non-zero chunks in the whole query: stage 1 of --theta 0.3 --rerank keeps 4-12 columns, a BM25 query 4-12
  // terms): a wave per pair would run 1-16 of its 64 lanes and still pay a full memory round trip per pair.  Instead 8 (up to 8 non-zero
  // chunks) or 16 lanes per pair, 8 / 4 pairs per wave: lane (pair slot ci, s) owns the s-th non-zero chunk.  Same products, fp64 sums
  // over at most 128 terms (the zero chunks the dense path adds are exact zeros).  Round 6: also over int16 slice indices (the whole-word
  // BM25 index of BASELINE config 1, whose 2 k exact rescorings per query took the general one-wave-per-pair loop: 8.4 of its 15.7 ms per
  // step) -- the 16-bit lanes of corpus ^ query index words are compared directly -- and up to 16 chunks (8 until then).
  // (its own kernel, rescore_sparse_kernel = PATH 3: the 16 chunk ids cost the general kernel, which sits at its register budget, another
  // 72 bytes of scratch per lane; the general kernel only counts the chunks and leaves these queries to it)
  if constexpr (PATH == 0) {
    if (p.split && p.q16 && *p.q_inexact == 0u && some_zero) {
      int m = 0;
      for (int t = 0; t * 64 < nchunks; ++t) m += __popcll(__ballot((nz_mask >> t) & 1u));
      if (m <= 16) return;
    }
  }
  if constexpr (PATH == 3) {
    if (!(p.q16 && *p.q_inexact == 0u && some_zero)) return;
    const bool idx16 = p.c_idx_dtype == DHR_IDX_I16;
    int nzc[16];
    int m = 0;
    for (int c0 = 0, t = 0; c0 < nchunks && m <= 16; c0 += 64, ++t) {
      uint64_t b = __ballot((nz_mask >> t) & 1u);
      while (b && m <= 16) { if (m < 16) nzc[m] = c0 + (__ffsll((unsigned long long)b) - 1); ++m; b &= b - 1; }
    }
    if (m > 16) return;
    {
      auto run = [&](auto lanes_c) __attribute__((always_inline)) {
        constexpr int L = decltype(lanes_c)::value, P = 64 / L;
        const int ci = lane / L, sl = lane % L;
        int myc = -1;
#pragma unroll
        for (int e = 0; e < L; ++e) if (e < m && sl == e) myc = nzc[e];
        uint4 qv = make_uint4(0u, 0u, 0u, 0u);
        uint4 qix = make_uint4(0u, 0u, 0u, 0u);          // the chunk's 8 query index values: 8 bytes (x, y) or 8 int16
        const bool gated_chunk = myc >= 0 && myc < dlr_chunks && p.gate;
        if (myc >= 0) qv = *(const uint4*)(p.q16 + (int64_t)q * p.k_rm + myc * 8);
        if (gated_chunk) {
          if (idx16) qix = *(const uint4*)(p.q_idx + (int64_t)q * p.d_dlr + myc * 8);
          else { const uint2 t8 = *(const uint2*)(p.q_idx8 + (int64_t)q * p.d_dlr + myc * 8); qix.x = t8.x; qix.y = t8.y; }
        }
        static_assert(RESCORE_CANDS_PER_WG % (4 * P) == 0, "4 waves x P pairs per round");
        for (uint32_t i0 = base + wave * P; i0 < base + RESCORE_CANDS_PER_WG && i0 < count; i0 += 4 * P) {
          const uint32_t i = i0 + ci;
          const bool live = i < count;
          uint32_t row = 0u;
          if (live) {
            if (p.cand) row = p.cand[(int64_t)q * p.cap + i].x;
            else if (p.rows32) row = p.rows32[(int64_t)q * p.ld_rows + i];
            else row = (uint32_t)(p.row0 + i);
          }
          const bool valid = live && (int64_t)row < p.n_rows;
          double acc = 0.0;
          if (valid && myc >= 0) {
            const uint4 dv = *(const uint4*)(p.vals_rm + (int64_t)row * p.k_rm + myc * 8);
            uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w};
            const uint32_t qq[4] = {qv.x, qv.y, qv.z, qv.w};
            if (gated_chunk && idx16) {
              const uint4 cix = *(const uint4*)((const int16_t*)p.c_idx + (int64_t)row * p.d_dlr + myc * 8);
              d[0] &= pair_mask16(cix.x ^ qix.x); d[1] &= pair_mask16(cix.y ^ qix.y);
              d[2] &= pair_mask16(cix.z ^ qix.z); d[3] &= pair_mask16(cix.w ^ qix.w);
            } else if (gated_chunk) {
              const uint2 cix = *(const uint2*)((const uint8_t*)p.c_idx + (int64_t)row * p.d_dlr + myc * 8);
              const uint32_t x0 = cix.x ^ qix.x, x1 = cix.y ^ qix.y;
              d[0] &= pair_mask(x0, 0x0c010c00u); d[1] &= pair_mask(x0, 0x0c030c02u);
              d[2] &= pair_mask(x1, 0x0c010c00u); d[3] &= pair_mask(x1, 0x0c030c02u);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc += (double)fmix_lo(d[e], qq[e]);
              acc += (double)fmix_hi(d[e], qq[e]);
            }
          }
#pragma unroll
          for (int o = 1; o < L; o <<= 1) acc += __shfl_xor(acc, o, 64);
          if (sl == 0 && live) {
            const float sc = valid ? (float)acc : -INFINITY;
            if (p.out_keys) p.out_keys[(int64_t)q * p.ld_keys + i] = valid ? make_key(sc, row) : 0ull;
            if (p.out_scores && q < p.n_queries) p.out_scores[(int64_t)q * p.ld_scores + i] = sc;
          }
        }
      };
      if (m <= 8) run(std::integral_constant<int, 8>{}); else run(std::integral_constant<int, 16>{});
      return;
    }
  }
  if constexpr (PATH != 3) {
  // Narrow rows (at most 128 chunks: dense-only indexes, 768 columns = 1.5 KB per row; the 768 + 128 BEIR layout): a wave per pair leaves half
  // of the lanes idle in its second pass and pays one 64-lane fp64 reduction per 1.5 KB -- measured 2.0 TB/s of row bytes against 5.4 TB/s on
  // 3.8 KB hybrid rows.  Instead 16 lanes per pair, four pairs per wave: lane s of a pair owns chunks s, s + 16, ... (ascending: six
  // independent 16-byte gathers in flight per lane), then the xor butterfly over the 16 lanes.  The summation order is fixed by the row
  // width alone, so scores stay reproducible between searches, shards and entry points.  (32 lanes per pair: 2.9-3.1 TB/s.)
  if constexpr (PATH == 0 || PATH == 2)
  if (fast && !some_zero && nchunks <= 128) {
#ifndef RESCORE_NARROW_L
#define RESCORE_NARROW_L 16      // lanes per pair on narrow rows (32: 2.9-3.1 TB/s, 16: 3.4; 8: A/B builds)
#endif
    constexpr int L = RESCORE_NARROW_L, P = 64 / L;
    const int g = lane / L, sl = lane % L;
    static_assert(RESCORE_CANDS_PER_WG % (4 * P) == 0, "4 waves x P pairs per round");
    for (uint32_t i0 = base + wave * P; i0 < base + RESCORE_CANDS_PER_WG && i0 < count; i0 += 4 * P) {
      const uint32_t i = i0 + g;
      const bool live = i < count;
      uint32_t row = 0u;
      if (live) {
        if (p.cand) row = p.cand[(int64_t)q * p.cap + i].x;
        else if (p.rows32) row = p.rows32[(int64_t)q * p.ld_rows + i];
        else row = (uint32_t)(p.row0 + i);
      }
      const bool valid = live && (int64_t)row < p.n_rows;
      double acc = 0.0;
      if (valid) {
        for (int c = sl; c < nchunks; c += L) {
          const uint4 dv = gather16(p.vals_rm + (int64_t)row * p.k_rm + c * 8);
          const uint4 qv = *(const uint4*)(p.q16 + (int64_t)q * p.k_rm + c * 8);
          uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w};
          const uint32_t qq[4] = {qv.x, qv.y, qv.z, qv.w};
          if (c < dlr_chunks && p.gate) {
            const uint2 ci = gather8((const uint8_t*)p.c_idx + (int64_t)row * p.d_dlr + c * 8);
            const uint2 qi8 = *(const uint2*)(p.q_idx8 + (int64_t)q * p.d_dlr + c * 8);
            const uint32_t x0 = ci.x ^ qi8.x, x1 = ci.y ^ qi8.y;
            d[0] &= pair_mask(x0, 0x0c010c00u); d[1] &= pair_mask(x0, 0x0c030c02u);
            d[2] &= pair_mask(x1, 0x0c010c00u); d[3] &= pair_mask(x1, 0x0c030c02u);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc += (double)fmix_lo(d[e], qq[e]);
            acc += (double)fmix_hi(d[e], qq[e]);
          }
        }
      }
#pragma unroll
      for (int o = L / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (sl == 0 && live) {
        const float sc = valid ? (float)acc : -INFINITY;
        if (p.out_keys) p.out_keys[(int64_t)q * p.ld_keys + i] = valid ? make_key(sc, row) : 0ull;
        if (p.out_scores && q < p.n_queries) p.out_scores[(int64_t)q * p.ld_scores + i] = sc;
      }
    }
    return;
  }
  // (Round 5: loading the row id of the wave's NEXT pair ahead of the current pair's gathers -- what gained 1.2 ms in refine_kernel -- made this
  // kernel slower, 15.4 -> 16.2 ms per config-3 step: it sits at its register budget (72, 60 bytes of scratch), and it is bound by bytes, not by the chain.)
  if constexpr (PATH != 2)
  for (uint32_t i = base + wave; i < base + RESCORE_CANDS_PER_WG && i < count; i += 4) {
    uint32_t row;
    if (p.cand) row = p.cand[(int64_t)q * p.cap + i].x;
    else if (p.rows32) row = p.rows32[(int64_t)q * p.ld_rows + i];
    else row = (uint32_t)(p.row0 + i);
    const bool valid = (int64_t)row < p.n_rows;
    double acc = 0.0;
    if (valid && fast) {
      // Queries exactly representable in fp16: d*q is exact in fp32 (v_fma_mix_f32 on the packed halves), converted and
      // added in fp64 -- bit-identical to the general path below at ~5 instead of ~8 vector instructions per element
      // (the kernel is VALU-bound on conversions and fp64 adds, not on the row gathers).
      for (int c = lane; c < nchunks; c += 64) {
        if (!((nz_mask >> (c >> 6)) & 1u)) continue;      // eight zero query values add exactly nothing: the corpus bytes are not fetched
        const uint4 dv = gather16(p.vals_rm + (int64_t)row * p.k_rm + c * 8);
        const uint4 qv = *(const uint4*)(p.q16 + (int64_t)q * p.k_rm + c * 8);
        uint32_t d[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint32_t qq[4] = {qv.x, qv.y, qv.z, qv.w};
        if (c < dlr_chunks && p.gate) {
          const uint2 ci = gather8((const uint8_t*)p.c_idx + (int64_t)row * p.d_dlr + c * 8);
          const uint2 qi8 = *(const uint2*)(p.q_idx8 + (int64_t)q * p.d_dlr + c * 8);
          const uint32_t x0 = ci.x ^ qi8.x, x1 = ci.y ^ qi8.y;
          d[0] &= pair_mask(x0, 0x0c010c00u); d[1] &= pair_mask(x0, 0x0c030c02u);
          d[2] &= pair_mask(x1, 0x0c010c00u); d[3] &= pair_mask(x1, 0x0c030c02u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc += (double)fmix_lo(d[e], qq[e]);
          acc += (double)fmix_hi(d[e], qq[e]);
        }
      }
      acc = wave_sum_f64(acc);
    } else if (PATH == 0 && valid) {
      for (int c = lane; c < nchunks; c += 64) {
        if (!((nz_mask >> (c >> 6)) & 1u)) continue;
        const half8 dv = *(const half8*)(p.vals_rm + (int64_t)row * p.k_rm + c * 8);
        const float4 qa = *(const float4*)(q32 + c * 8);
        const float4 qb = *(const float4*)(q32 + c * 8 + 4);
        const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        if (c < dlr_chunks && p.gate) {
          int ci[8];
          if (p.c_idx_dtype == DHR_IDX_I16) {
            const short8 v = *(const short8*)((const int16_t*)p.c_idx + (int64_t)row * p.d_dlr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) ci[e] = v[e];
          } else {
            const uint2 v = *(const uint2*)((const uint8_t*)p.c_idx + (int64_t)row * p.d_dlr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t w = (e < 4) ? v.x : v.y;
              const uint32_t byte = (w >> (8 * (e & 3))) & 0xFFu;
              ci[e] = (p.c_idx_dtype == DHR_IDX_I8) ? (int)(int8_t)byte : (int)byte;
            }
          }
          const short8 qiv = *(const short8*)(qi + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double d = (ci[e] == (int)qiv[e]) ? (double)(float)dv[e] : 0.0;
            acc = fma(d, (double)qv[e], acc);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fma((double)(float)dv[e], (double)qv[e], acc);
        }
      }
      acc = wave_sum_f64(acc);
    }
    if (lane == 0) {
      const float sc = valid ? (float)acc : -INFINITY;
      if (p.out_keys) p.out_keys[(int64_t)q * p.ld_keys + i] = valid ? make_key(sc, row) : 0ull;
      if (p.out_scores && q < p.n_queries) p.out_scores[(int64_t)q * p.ld_scores + i] = sc;
    }
  }
  }      // PATH != 3
}
// Flat launches walk the block list with a grid stride: the grid is the exact block count when the host knows it, and a fixed one when the
// controller runs without host read-backs (the list lengths then exist in device memory only).
#ifndef RESCORE_WPE
#define RESCORE_WPE 7          // waves per SIMD the register allocation aims at (7 = 72 registers, 8 = 64)
#endif
// Which kernel takes a batch is decided ON THE DEVICE (the flags of query_prep_kernel are never read by the host): the general kernel and the
// specialised one for the index's row width are both launched, one of them returns at once.
__device__ __forceinline__ bool rescore_fast_batch(const RescoreArgs& p) {
  return p.q16 && p.q_inexact[0] == 0u && p.q_inexact[1] == 0u &&
         (!p.gate || p.d_dlr == 0 || p.c_idx_dtype == DHR_IDX_U8 || p.c_idx_dtype == DHR_IDX_I8);
}
template <int PATH>
__device__ __forceinline__ void rescore_run(const RescoreArgs& p) {
  if (!p.blk_off) { rescore_block<PATH>(p, (int)blockIdx.y, blockIdx.x); return; }
  for (uint32_t b = blockIdx.x;; b += gridDim.x) {
    int q; uint32_t blk;
    if (!flat_block(p.blk_off, p.n_queries, b, q, blk)) return;
    rescore_block<PATH>(p, q, blk);
  }
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RESCORE_WPE))) rescore_kernel(RescoreArgs p) {
  if (p.split && rescore_fast_batch(p)) return;
  rescore_run<0>(p);
}
// The hybrid brute-force batch (round 5): the same sums in the same order from the fast branch alone, 52 registers, 8 waves per SIMD --
// 15.4 -> 13.9 ms of row gathers per config-3 step.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) rescore_fast_kernel(RescoreArgs p) {
  if (!rescore_fast_batch(p) || (p.k_rm >> 3) <= 128) return;
  rescore_run<1>(p);
}
// ... the queries with at most 16 non-zero chunks of a batch that has any (stage 1 of the theta modes, BM25 queries; any index dtype)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) rescore_sparse_kernel(RescoreArgs p) {
  if (!p.q16 || p.q_inexact[0] != 0u || p.q_inexact[1] == 0u) return;
  rescore_run<3>(p);
}
// ... and the narrow-row batch (dense-only 768, BM25, the 768 + 128 BEIR layout).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) rescore_narrow_kernel(RescoreArgs p) {
  if (!rescore_fast_batch(p) || (p.k_rm >> 3) > 128) return;
  rescore_run<2>(p);
}
hipError_t launch_rescore(const RescoreArgs& a_in, hipStream_t s) {
  if (a_in.max_count == 0 || a_in.n_queries <= 0) return hipSuccess;
  RescoreArgs a = a_in;
  // with an fp16 copy of the queries: the specialised kernel of the row width beside the general kernel (rescore_fast_batch decides on the device)
  a.split = (a.q16 && a.q_inexact) ? 1 : 0;
  const bool wide = (a.k_rm >> 3) > 128;
  dim3 grid;
  if (a.blk_off) {
    if (!a.flat_blocks) return hipSuccess;
    grid = dim3(std::min<uint32_t>(a.flat_blocks, FLAT_GRID_MAX));
  } else grid = dim3((a.max_count + RESCORE_CANDS_PER_WG - 1) / RESCORE_CANDS_PER_WG, (unsigned)a.n_queries);
  if (a.split && wide) hipLaunchKernelGGL(rescore_fast_kernel, grid, dim3(256), 0, s, a);
  if (a.split && !wide) hipLaunchKernelGGL(rescore_narrow_kernel, grid, dim3(256), 0, s, a);
  if (a.split) hipLaunchKernelGGL(rescore_sparse_kernel, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(rescore_kernel, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ select
// One workgroup per query.  LDS holds sort_n u64 keys: the running top-k (kp slots) plus up to
// sort_n-kp new keys that beat the current k-th key; a descending bitonic sort over the live prefix
// merges them.  Writes the new top-k, tau (exact k-th best so far) and thr = tau - margin.
__device__ __forceinline__ void bitonic_desc(uint64_t* keys, int n, int tid, int nthreads) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n >> 1); t += nthreads) {
        const int pos = 2 * t - (t & (stride - 1));
        const uint64_t a = keys[pos], b = keys[pos + stride];
        const bool desc = (pos & size) == 0;
        if ((a < b) == desc) { keys[pos] = b; keys[pos + stride] = a; }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int count_greater(const uint64_t* arr, int n, uint64_t key) {   // arr sorted descending
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (arr[mid] > key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Merge one query's rescored keys into its running top-k list (k <= 4096).  LDS: A = the running list (kps keys,
// sorted descending), B = the new keys that beat the current k-th (a chunk of the main pass brings ~k/M of them).
// Only B is sorted (bitonic, next power of two >= its fill); A and B are then merged by rank: every key's final
// position is its own index plus the number of keys of the other list that beat it (keys are unique: the row
// id is part of the key).  A full re-sort of kps + |B| keys per call cost 2.5x the barriers.
constexpr int SELECT_SM_THREADS = 256;
// EA = list entries per thread: 16 covers kp <= 4096; the top-1000 searches (kp = 1024) run the EA = 4 instantiation, whose 48 fewer registers
// double the workgroups a CU holds (round 5)
template <int EA>
__global__ void __launch_bounds__(SELECT_SM_THREADS) select_kernel(SelectArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* A = (uint64_t*)smem;
  int& fill = *(int*)(smem + (size_t)p.sort_n * 8);      // all LDS in the dynamic region (keeps it 16-B aligned)
  const int kps = p.kps;                      // <= p.kp: only this prefix of the running list is live
  uint64_t* B = A + kps;
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  uint32_t count = p.cnt ? p.cnt[q] : p.count_all;
  if (p.cnt && count > p.cap) count = p.cap;
  uint64_t* topk = p.topk_keys + (int64_t)q * p.kp;
  const uint64_t* in = p.in_keys + (int64_t)q * p.ld_keys;
  for (int j = tid; j < kps; j += SELECT_SM_THREADS) A[j] = topk[j];
  if (tid == 0) fill = 0;
  __syncthreads();
  uint32_t room = 64;                         // input keys per round: a power of two that fits behind A, at most 2048
  while (room * 2 <= (uint32_t)(p.sort_n - kps) && room < 2048) room <<= 1;
  const int k_keep = p.k_keep ? p.k_keep : p.k;
  for (uint32_t base = 0; base < count; base += room) {
    const uint64_t kth = A[k_keep - 1];
    __syncthreads();
    const uint32_t end = (base + room < count) ? base + room : count;
    for (uint32_t j = base + tid; j < end; j += SELECT_SM_THREADS) {
      const uint64_t key = in[j];
      if (key > kth) B[atomicAdd(&fill, 1)] = key;
    }
    __syncthreads();
    const int m = fill;
    if (m > 0) {
      int m2 = 2;
      while (m2 < m) m2 <<= 1;
      for (int j = m + tid; j < m2; j += SELECT_SM_THREADS) B[j] = 0ull;
      __syncthreads();
      bitonic_desc(B, m2, tid, SELECT_SM_THREADS);
      uint64_t ka[EA], kb[8];
      int da[EA], db[8];
#pragma unroll
      for (int e = 0; e < EA; ++e) {
        const int i = tid + e * SELECT_SM_THREADS;
        da[e] = kps;
        if (i < kps) { ka[e] = A[i]; da[e] = ka[e] ? i + count_greater(B, m, ka[e]) : kps; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = tid + e * SELECT_SM_THREADS;
        db[e] = kps;
        if (j < m) { kb[e] = B[j]; db[e] = j + count_greater(A, kps, kb[e]); }
      }
      // every live key has a distinct final position; positions no key lands on (the tail, when the list is
      // not full yet) must read as empty
      const int total = min(kps, m + count_greater(A, kps, 0ull));
      __syncthreads();
      for (int j = total + tid; j < kps; j += SELECT_SM_THREADS) A[j] = 0ull;
#pragma unroll
      for (int e = 0; e < EA; ++e)
        if (da[e] < kps) A[da[e]] = ka[e];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (db[e] < kps) A[db[e]] = kb[e];
    }
    __syncthreads();
    if (tid == 0) fill = 0;
    __syncthreads();
  }
  for (int j = tid; j < kps; j += SELECT_SM_THREADS) topk[j] = (j < k_keep) ? A[j] : 0ull;
  if (tid == 0) {
    const uint64_t kth = A[p.k - 1];
    float t = kth ? ordered_f32((uint32_t)(kth >> 32)) : -INFINITY;
    if (p.monotone) t = fmaxf(t, p.tau[q]);      // the rank that defines the threshold changes between the phases of a sampled run
    p.tau[q] = t;
    p.thr[q] = (q < p.n_queries) ? t - p.margin[q] : INFINITY;
  }
}

// Large k (4096 < k <= 16384, e.g. the documented --agip_topk 10000): the running list stays in memory and is merged IN PLACE.
// New keys that beat the current k-th come in sorted batches of SELECT_BIG_BATCH (LDS); every new key finds its final position by
// a binary search over the list in memory (read-only pass), then the list is swept from its tail to its head in tiles: a tile's
// entries are read, ranked against the batch (own index + number of new keys that beat them) and written at or behind their old
// place -- positions an earlier (= later-in-the-list) tile has already vacated.  16 KB of LDS and 256 threads: 8 workgroups per CU
// (rounds 2-4 held the whole list in LDS: 147 KB, one 1024-thread workgroup per CU, 1.0-2.3 ms per call; this form: see docs/experiments.md).
// Only the first k_keep slots of the kp-slot list are ever touched (the controller clears the list before a search).
constexpr int SELECT_BIG_BATCH = 2048;
constexpr int SELECT_BIG_THREADS = 256;
constexpr int SELECT_BIG_EPT = 4;                                    // list entries per thread and tile
__global__ void __launch_bounds__(SELECT_BIG_THREADS) select_big_kernel(SelectArgs p) {
  __shared__ uint64_t B[SELECT_BIG_BATCH];
  __shared__ int fill;
  const int q = blockIdx.x, tid = threadIdx.x;
  uint32_t count = p.cnt ? p.cnt[q] : p.count_all;
  if (p.cnt && count > p.cap) count = p.cap;
  uint64_t* topk = p.topk_keys + (int64_t)q * p.kp;
  const uint64_t* in = p.in_keys + (int64_t)q * p.ld_keys;
  const int k_keep = p.k_keep ? p.k_keep : p.k;
  constexpr int TILE = SELECT_BIG_THREADS * SELECT_BIG_EPT;
  for (uint32_t base = 0; base < count; base += SELECT_BIG_BATCH) {
    if (tid == 0) fill = 0;
    __syncthreads();                                                 // also orders the previous batch's writes before this read
    const uint64_t kth = topk[k_keep - 1];
    const uint32_t end = (base + SELECT_BIG_BATCH < count) ? base + SELECT_BIG_BATCH : count;
    for (uint32_t j = base + tid; j < end; j += SELECT_BIG_THREADS) {
      const uint64_t key = in[j];
      if (key > kth) B[atomicAdd(&fill, 1)] = key;
    }
    __syncthreads();
    const int m = fill;
    if (m == 0) continue;
    int m2 = 2;
    while (m2 < m) m2 <<= 1;
    for (int j = m + tid; j < m2; j += SELECT_BIG_THREADS) B[j] = 0ull;
    __syncthreads();
    bitonic_desc(B, m2, tid, SELECT_BIG_THREADS);
    uint64_t kb[SELECT_BIG_BATCH / SELECT_BIG_THREADS]; int db[SELECT_BIG_BATCH / SELECT_BIG_THREADS];
#pragma unroll
    for (int e = 0; e < SELECT_BIG_BATCH / SELECT_BIG_THREADS; ++e) {
      const int j = tid + e * SELECT_BIG_THREADS;
      kb[e] = (j < m) ? B[j] : 0ull;
      db[e] = (j < m) ? j + count_greater(topk, k_keep, kb[e]) : k_keep;
    }
    __syncthreads();                                                 // every rank against the old list is taken before the list moves
    for (int t0 = (k_keep - 1) / TILE * TILE; t0 >= 0; t0 -= TILE) {
      uint64_t ka[SELECT_BIG_EPT]; int da[SELECT_BIG_EPT];
#pragma unroll
      for (int e = 0; e < SELECT_BIG_EPT; ++e) {
        const int i = t0 + tid + e * SELECT_BIG_THREADS;
        da[e] = k_keep;
        if (i < k_keep) { ka[e] = topk[i]; da[e] = i + count_greater(B, m, ka[e]); }     // empty slots (0) move behind every key
      }
      __syncthreads();                                               // the tile is in registers: its slots may be overwritten
#pragma unroll
      for (int e = 0; e < SELECT_BIG_EPT; ++e)
        if (da[e] < k_keep) topk[da[e]] = ka[e];
    }
#pragma unroll
    for (int e = 0; e < SELECT_BIG_BATCH / SELECT_BIG_THREADS; ++e)
      if (db[e] < k_keep) topk[db[e]] = kb[e];
  }
  __syncthreads();
  if (tid == 0) {
    const uint64_t kth = topk[p.k - 1];
    float t = kth ? ordered_f32((uint32_t)(kth >> 32)) : -INFINITY;
    if (p.monotone) t = fmaxf(t, p.tau[q]);      // the rank that defines the threshold changes between the phases of a sampled run
    p.tau[q] = t;
    p.thr[q] = (q < p.n_queries) ? t - p.margin[q] : INFINITY;
  }
}

hipError_t launch_select(const SelectArgs& a, hipStream_t s) {
  if (a.kp > 16384) return launch_select_global(a, s);       // beyond the LDS: concatenate + segmented sort (select_global.hip)
  if (a.kp > 4096) {
    hipLaunchKernelGGL(select_big_kernel, dim3((unsigned)a.n_queries), dim3(SELECT_BIG_THREADS), 0, s, a);
    return hipGetLastError();
  }
  static int attr_bytes[64] = {}, attr_bytes4[64] = {};
  const int bytes = a.sort_n * 8 + 16;
  if (a.kp <= 4 * SELECT_SM_THREADS) {
    if (hipError_t e = ensure_lds_attr((const void*)select_kernel<4>, bytes, attr_bytes4); e != hipSuccess) return e;
    hipLaunchKernelGGL(select_kernel<4>, dim3((unsigned)a.n_queries), dim3(SELECT_SM_THREADS), bytes, s, a);
    return hipGetLastError();
  }
  if (hipError_t e = ensure_lds_attr((const void*)select_kernel<16>, bytes, attr_bytes); e != hipSuccess) return e;
  hipLaunchKernelGGL(select_kernel<16>, dim3((unsigned)a.n_queries), dim3(SELECT_SM_THREADS), bytes, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ product quantiser
// First stage of --PQIP: the reference delegates to faiss IndexPQ(d, M=64, nbits=8, inner product);
// faiss is not vendored, so this restates its published algorithm: per subspace Lloyd k-means (L2) on the sub-vectors,
// codes = nearest centroid, search score = sum over subspaces of <query sub-vector, centroid of the code> (ADC).
// Here the ADC scan is not a table walk: the codes are decoded once to fp16 vectors and searched with the dense bound
// GEMM (an exact inner product on the decoded vectors IS the ADC score), which is what the matrix cores are good at and
// 288 GB of HBM allow.  Kernels: assignment (+ optional accumulation for the k-means update), update, decode.
constexpr int PQ_K = 256;                      // centroids per subspace (nbits = 8)
constexpr int PQ_MAX_DSUB = 64;

// grid (point blocks of 256, M).  Centroids of the subspace in LDS; one thread per point.
template <int DSUB>   // compile-time sub-vector width (registers) or 0 = run-time width up to PQ_MAX_DSUB (scratch)
__global__ void __launch_bounds__(256) pq_assign_kernel(const __half* __restrict__ vals, int64_t ld, int64_t n, int64_t stride, int dsub_rt,
                                                        const float* __restrict__ cb /* [M][ksub][dsub] */, uint8_t* __restrict__ codes,
                                                        int64_t ld_codes, float* __restrict__ sums /* [M][ksub][dsub] or null */,
                                                        uint32_t* __restrict__ counts /* [M][ksub] or null */, float* __restrict__ err /* [M] or null */,
                                                        int ksub /* centroids per subspace = 2^nbits <= 256 */) {
  __shared__ float c[PQ_K * PQ_MAX_DSUB];
  __shared__ float cn[PQ_K];
  const int dsub = DSUB ? DSUB : dsub_rt;
  const int m = blockIdx.y;
  const float* cbm = cb + (int64_t)m * ksub * dsub;
  for (int i = threadIdx.x; i < ksub * dsub; i += 256) c[i] = cbm[i];
  __syncthreads();
  if ((int)threadIdx.x < ksub) {
    float s = 0.f;
    for (int j = 0; j < dsub; ++j) s += c[threadIdx.x * dsub + j] * c[threadIdx.x * dsub + j];
    cn[threadIdx.x] = s;
  }
  __syncthreads();
  const int64_t pi = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pi >= n) return;
  const __half* x = vals + pi * stride * ld + (int64_t)m * dsub;
  float xv[DSUB ? DSUB : PQ_MAX_DSUB];
#pragma unroll
  for (int j = 0; j < (DSUB ? DSUB : PQ_MAX_DSUB); ++j)
    if (j < dsub) xv[j] = __half2float(x[j]);
  // argmin_c |x - c|^2 = argmin_c (|c|^2 - 2 <x, c>); first minimum wins
  float best = INFINITY;
  int arg = 0;
  for (int k = 0; k < ksub; ++k) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < (DSUB ? DSUB : 1); ++j)
      if (DSUB) dot += xv[j] * c[k * dsub + j];
    if (!DSUB)
      for (int j = 0; j < dsub; ++j) dot += xv[j] * c[k * dsub + j];
    const float dist = cn[k] - 2.f * dot;
    if (dist < best) { best = dist; arg = k; }
  }
  if (codes) codes[pi * ld_codes + m] = (uint8_t)arg;
  if (sums) {
    for (int j = 0; j < dsub; ++j) atomicAdd(&sums[((int64_t)m * ksub + arg) * dsub + j], xv[j]);
    atomicAdd(&counts[m * ksub + arg], 1u);
    if (err) {
      float xx = 0.f;
      for (int j = 0; j < dsub; ++j) xx += xv[j] * xv[j];
      atomicAdd(&err[m], best + xx);
    }
  }
}
// centroid = mean of its points; an empty cluster takes a copy of the most populated cluster's centroid, nudged
// (faiss splits a large cluster the same way)
__global__ void __launch_bounds__(256) pq_update_kernel(float* __restrict__ cb, const float* __restrict__ sums, const uint32_t* __restrict__ counts,
                                                        int dsub, int ksub) {
  const int m = blockIdx.x, k = threadIdx.x;
  __shared__ uint32_t cnt[PQ_K];
  __shared__ int big;
  cnt[k] = k < ksub ? counts[m * ksub + k] : 0u;
  __syncthreads();
  if (k == 0) {
    int b = 0;
    for (int i = 1; i < ksub; ++i) if (cnt[i] > cnt[b]) b = i;
    big = b;
  }
  __syncthreads();
  if (k >= ksub) return;
  float* c = cb + ((int64_t)m * ksub + k) * dsub;
  if (cnt[k] > 0) {
    const float inv = 1.f / (float)cnt[k];
    for (int j = 0; j < dsub; ++j) c[j] = sums[((int64_t)m * ksub + k) * dsub + j] * inv;
  } else {
    const float inv = 1.f / (float)(cnt[big] ? cnt[big] : 1u);
    for (int j = 0; j < dsub; ++j) c[j] = sums[((int64_t)m * ksub + big) * dsub + j] * inv * (1.f + ((j + k) & 1 ? 1.f : -1.f) / 1024.f);
  }
}
// initial centroids: ksub evenly spaced training points
__global__ void __launch_bounds__(256) pq_init_kernel(const __half* __restrict__ vals, int64_t ld, int64_t n, int64_t stride, int dsub, int M,
                                                      float* __restrict__ cb, int ksub) {
  const int m = blockIdx.x, k = threadIdx.x;
  if (k >= ksub) return;
  const int64_t pi = (n >= ksub) ? (int64_t)k * (n / ksub) : (int64_t)(k % (n > 0 ? n : 1));
  const __half* x = vals + pi * stride * ld + (int64_t)m * dsub;
  for (int j = 0; j < dsub; ++j) cb[((int64_t)m * ksub + k) * dsub + j] = __half2float(x[j]);
}
// decoded vectors, fp16 [n][ld_out]: column m*dsub + j = centroid(code[m])[j]
__global__ void __launch_bounds__(256) pq_decode_kernel(const uint8_t* __restrict__ codes, int64_t ld_codes, int64_t n, int M, int dsub,
                                                        const float* __restrict__ cb, __half* __restrict__ out, int64_t ld_out, int ksub) {
  const int d = M * dsub;
  const int64_t total = n * d;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t row = g / d;
    const int col = (int)(g - row * d);
    const int m = col / dsub, j = col - m * dsub;
    out[row * ld_out + col] = __float2half(cb[((int64_t)m * ksub + codes[row * ld_codes + m]) * dsub + j]);
  }
}
hipError_t launch_pq_init(const __half* vals, int64_t ld, int64_t n, int64_t stride, int dsub, int M, float* cb, int ksub, hipStream_t s) {
  hipLaunchKernelGGL(pq_init_kernel, dim3(M), dim3(256), 0, s, vals, ld, n, stride, dsub, M, cb, ksub);
  return hipGetLastError();
}
hipError_t launch_pq_assign(const __half* vals, int64_t ld, int64_t n, int64_t stride, int dsub, int M, const float* cb, uint8_t* codes,
                            int64_t ld_codes, float* sums, uint32_t* counts, float* err, int ksub, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const dim3 grid((unsigned)((n + 255) / 256), M);
#define PQ_GO(D) hipLaunchKernelGGL(pq_assign_kernel<D>, grid, dim3(256), 0, s, vals, ld, n, stride, dsub, cb, codes, ld_codes, sums, counts, err, ksub)
  if (dsub == 12) PQ_GO(12); else if (dsub == 24) PQ_GO(24); else if (dsub == 14) PQ_GO(14); else if (dsub == 16) PQ_GO(16); else PQ_GO(0);
#undef PQ_GO
  return hipGetLastError();
}
hipError_t launch_pq_update(float* cb, const float* sums, const uint32_t* counts, int dsub, int M, int ksub, hipStream_t s) {
  hipLaunchKernelGGL(pq_update_kernel, dim3(M), dim3(256), 0, s, cb, sums, counts, dsub, ksub);
  return hipGetLastError();
}
hipError_t launch_pq_decode(const uint8_t* codes, int64_t ld_codes, int64_t n, int M, int dsub, const float* cb, __half* out, int64_t ld_out, int ksub,
                            hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t blocks = std::min<int64_t>((n * M * dsub + 255) / 256, 65536);
  hipLaunchKernelGGL(pq_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, codes, ld_codes, n, M, dsub, cb, out, ld_out, ksub);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ densify
// Fused densify (tevatron/DHR/utils.py:5-22 + the casts of tevatron/driver/encode.py:155-158): one pass over the
// [batch, vocab] lexical representations; thread (row, j) walks the n_groups vocabulary entries remove + g*dims + j
// (coalesced across j), keeps the maximum and the FIRST group attaining it, and writes the value (fp16 or fp32) and
// the group (uint8 / int16) straight into the caller's index-record arrays.  HBM-bound: every input byte is read once.
template <typename TIN>
__global__ void __launch_bounds__(256) densify_kernel(const TIN* __restrict__ lex, int64_t ld, int64_t batch, int remove, int dims,
                                                      int n_groups, void* __restrict__ out_val, int val_is_f32, int64_t ld_val,
                                                      void* __restrict__ out_idx, int idx_is_i16, int64_t ld_idx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (j >= dims || b >= batch) return;
  const TIN* src = lex + b * ld + remove + j;
  float best = (float)src[0];
  int arg = 0;
  int g = 1;
  for (; g + 8 <= n_groups; g += 8) {          // 8 independent loads in flight per thread
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (float)src[(int64_t)(g + u) * dims];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u] > best) { best = v[u]; arg = g + u; }
  }
  for (; g < n_groups; ++g) {
    const float v = (float)src[(int64_t)g * dims];
    if (v > best) { best = v; arg = g; }
  }
  if (val_is_f32) ((float*)out_val)[b * ld_val + j] = best;
  else ((__half*)out_val)[b * ld_val + j] = __float2half(best);
  if (idx_is_i16) ((int16_t*)out_idx)[b * ld_idx + j] = (int16_t)arg;
  else ((uint8_t*)out_idx)[b * ld_idx + j] = (uint8_t)arg;
}
hipError_t launch_densify(const void* lex, int in_is_f32, int64_t ld, int64_t batch, int remove, int dims, int n_groups, void* out_val,
                          int val_is_f32, int64_t ld_val, void* out_idx, int idx_is_i16, int64_t ld_idx, hipStream_t s) {
  if (batch <= 0) return hipSuccess;
  const dim3 grid((unsigned)((dims + 255) / 256), (unsigned)batch);
  if (in_is_f32)
    hipLaunchKernelGGL(densify_kernel<float>, grid, dim3(256), 0, s, (const float*)lex, ld, batch, remove, dims, n_groups, out_val, val_is_f32,
                       ld_val, out_idx, idx_is_i16, ld_idx);
  else
    hipLaunchKernelGGL(densify_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)lex, ld, batch, remove, dims, n_groups, out_val,
                       val_is_f32, ld_val, out_idx, idx_is_i16, ld_idx);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ emit / helpers
__global__ void emit_kernel(const uint64_t* __restrict__ topk_keys, int kp, int n_queries, int k, int64_t row_offset,
                            float* __restrict__ out_scores, int64_t* __restrict__ out_rows) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)n_queries * k) return;
  const int q = (int)(g / k), j = (int)(g - (int64_t)q * k);
  const uint64_t key = topk_keys[(int64_t)q * kp + j];
  if (key) {
    out_scores[g] = ordered_f32((uint32_t)(key >> 32));
    out_rows[g] = row_offset + (int64_t)(0xFFFFFFFFu - (uint32_t)key);
  } else {
    out_scores[g] = -INFINITY;
    out_rows[g] = -1;
  }
}
// Stage 2 of the two-stage modes: the rows of a sorted key list as a dense [n_queries][k] table of local rows
// (0xFFFFFFFF = empty slot) for the exact rescoring.
__global__ void keys_to_rows_kernel(const uint64_t* __restrict__ topk_keys, int kp, int n_queries, int k, uint32_t* __restrict__ rows) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)n_queries * k) return;
  const int q = (int)(g / k), j = (int)(g - (int64_t)q * k);
  const uint64_t key = topk_keys[(int64_t)q * kp + j];
  rows[g] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
}
// Threshold bootstrap (search_core): the m best rows of corpus tile 0 by BOUND score, per query.  The bound scores come from the GEMM's dump
// variant ([q][256] floats: an open filter would push all 65 536 pairs of a tile through the cold surplus path of the epilogue, one atomic
// each -- 0.46 ms for the 28 workgroups of a shard's first launch).  One workgroup per query: every thread ranks its row against the others
// (the keys are unique: the row is part of the key) and the m best write their rows in rank order; slots beyond the tile read 0xFFFFFFFF.
__global__ void __launch_bounds__(256) bound_topm_kernel(const float* __restrict__ bound, int n_rows, int n_queries, int m, uint32_t* __restrict__ rows) {
  __shared__ uint64_t key[256];
  const int q = blockIdx.x, t = threadIdx.x;
  uint64_t mine = 0ull;
  if (t < n_rows) mine = make_key(bound[(int64_t)q * 256 + t], (uint32_t)t);      // bound scores of rows 0 .. 255 as the GEMM's dump variant leaves them
  key[t] = mine;
  if (t < m) rows[(int64_t)q * m + t] = 0xFFFFFFFFu;
  __syncthreads();
  if (mine) {
    int rank = 0;
    for (int j = 0; j < 256; ++j) rank += key[j] > mine ? 1 : 0;
    if (rank < m) rows[(int64_t)q * m + rank] = 0xFFFFFFFFu - (uint32_t)mine;
  }
}
hipError_t launch_bound_topm(const float* bound, int n_rows, int n_queries, int m, uint32_t* rows, hipStream_t s) {
  if (n_queries <= 0) return hipSuccess;
  hipLaunchKernelGGL(bound_topm_kernel, dim3((unsigned)n_queries), dim3(256), 0, s, bound, n_rows, n_queries, m, rows);
  return hipGetLastError();
}
hipError_t launch_keys_to_rows(const uint64_t* topk_keys, int kp, int n_queries, int k, uint32_t* rows, hipStream_t s) {
  const int64_t total = (int64_t)n_queries * k;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(keys_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, topk_keys, kp, n_queries, k, rows);
  return hipGetLastError();
}
hipError_t launch_emit(const uint64_t* topk_keys, int kp, int n_queries, int k, int64_t row_offset, float* out_scores,
                       int64_t* out_rows, hipStream_t s) {
  const int64_t total = (int64_t)n_queries * k;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, topk_keys, kp, n_queries, k,
                     row_offset, out_scores, out_rows);
  return hipGetLastError();
}

__global__ void max_u32_kernel(const uint32_t* __restrict__ v, int n, uint32_t* out_max, unsigned long long* out_sum) {
  uint32_t m = 0;
  unsigned long long s = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t x = v[i];
    m = x > m ? x : m;
    s += x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t om = __shfl_xor(m, o, 64);
    m = om > m ? om : m;
    s += __shfl_xor(s, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out_max, m);
    atomicAdd(out_sum, s);
  }
}
hipError_t launch_max_u32(const uint32_t* v, int n, uint32_t* out_max, unsigned long long* out_sum, hipStream_t s) {
  hipLaunchKernelGGL(max_u32_kernel, dim3(32), dim3(256), 0, s, v, n, out_max, out_sum);
  return hipGetLastError();
}

__global__ void rows_to_local_kernel(const int64_t* __restrict__ rows, int64_t n, int64_t row_offset, int64_t n_rows,
                                     uint32_t* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int64_t r = rows[g] - row_offset;
  out[g] = (rows[g] >= 0 && r >= 0 && r < n_rows) ? (uint32_t)r : 0xFFFFFFFFu;
}
hipError_t launch_rows_to_local(const int64_t* rows, int64_t n, int64_t row_offset, int64_t n_rows, uint32_t* out,
                                hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(rows_to_local_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows, n, row_offset,
                     n_rows, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ sampled-threshold helpers
// A query's result is complete iff its candidate list did not overflow and its k-th best exact score
// reached the sampled threshold tau_hat (then every row with score >= tau_hat was collected).
__global__ void mark_overflow_kernel(const uint32_t* __restrict__ cnt, uint32_t cap, int n_queries, uint32_t* fail_flags) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n_queries && cnt[q] > cap) fail_flags[q] = 1u;
}
hipError_t launch_mark_overflow(const uint32_t* cnt, uint32_t cap, int n_queries, uint32_t* fail_flags, hipStream_t s) {
  hipLaunchKernelGGL(mark_overflow_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, cnt, cap, n_queries, fail_flags);
  return hipGetLastError();
}
__global__ void verify_kernel(const uint64_t* __restrict__ topk_keys, int kp, int k, const float* __restrict__ tau_hat,
                              int n_queries, uint32_t* fail_flags, uint32_t* n_fail) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  const uint64_t kth = topk_keys[(int64_t)q * kp + (k - 1)];
  const bool ok = fail_flags[q] == 0u && kth != 0ull && ordered_f32((uint32_t)(kth >> 32)) >= tau_hat[q];
  fail_flags[q] = ok ? 0u : 1u;
  if (!ok) atomicAdd(n_fail, 1u);
}
hipError_t launch_verify(const uint64_t* topk_keys, int kp, int k, const float* tau_hat, int n_queries, uint32_t* fail_flags,
                         uint32_t* n_fail, hipStream_t s) {
  hipLaunchKernelGGL(verify_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, topk_keys, kp, k, tau_hat, n_queries,
                     fail_flags, n_fail);
  return hipGetLastError();
}

__global__ void gather_queries_kernel(const float* __restrict__ q32, const int16_t* __restrict__ q_idx, int k_pad,
                                      int d_dlr, const int32_t* __restrict__ ids, int n, float* __restrict__ out32,
                                      int16_t* __restrict__ out_idx) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const int64_t src = ids[i];
  for (int j = threadIdx.x; j < k_pad; j += blockDim.x) out32[(int64_t)i * k_pad + j] = q32[src * k_pad + j];
  for (int j = threadIdx.x; j < d_dlr; j += blockDim.x) out_idx[(int64_t)i * d_dlr + j] = q_idx[src * d_dlr + j];
}
hipError_t launch_gather_queries(const float* q32, const int16_t* q_idx, int k_pad, int d_dlr, const int32_t* ids, int n,
                                 float* out32, int16_t* out_idx, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_queries_kernel, dim3(n), dim3(256), 0, s, q32, q_idx, k_pad, d_dlr, ids, n, out32, out_idx);
  return hipGetLastError();
}

__global__ void scatter_keys_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int kp,
                                    const int32_t* __restrict__ ids, int n) {
  const int i = blockIdx.x;
  if (i >= n) return;
  for (int j = threadIdx.x; j < kp; j += blockDim.x) dst[(int64_t)ids[i] * kp + j] = src[(int64_t)i * kp + j];
}
hipError_t launch_scatter_keys(const uint64_t* src, uint64_t* dst, int kp, const int32_t* ids, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_keys_kernel, dim3(n), dim3(256), 0, s, src, dst, kp, ids, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ staged-search helpers
__global__ void emit_scores_kernel(const uint64_t* __restrict__ topk_keys, int kp, int n_queries, int r, float* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)n_queries * r) return;
  const int q = (int)(g / r), j = (int)(g - (int64_t)q * r);
  const uint64_t key = topk_keys[(int64_t)q * kp + j];
  out[g] = key ? ordered_f32((uint32_t)(key >> 32)) : -INFINITY;
}
hipError_t launch_emit_scores(const uint64_t* topk_keys, int kp, int n_queries, int r, float* out, hipStream_t s) {
  const int64_t total = (int64_t)n_queries * r;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(emit_scores_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, topk_keys, kp, n_queries, r, out);
  return hipGetLastError();
}
__global__ void make_thr_kernel(const float* __restrict__ tau, const float* __restrict__ margin, int n_queries, int q_pad,
                                float* __restrict__ thr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < q_pad) thr[q] = q < n_queries ? tau[q] - margin[q] : INFINITY;
}
hipError_t launch_make_thr(const float* tau, const float* margin, int n_queries, int q_pad, float* thr, hipStream_t s) {
  hipLaunchKernelGGL(make_thr_kernel, dim3((q_pad + 255) / 256), dim3(256), 0, s, tau, margin, n_queries, q_pad, thr);
  return hipGetLastError();
}
// Main pass: raise the filter thresholds to the running exact ones (the k-th best exact score found so far is a
// valid threshold at any time; the sampled tau_hat is only the starting point).  The bound GEMM of the next
// chunk may be reading thr_hat meanwhile: either value of a word is a valid threshold.
__global__ void raise_thr_kernel(float* __restrict__ thr_hat, const float* __restrict__ thr_run, int n_queries) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n_queries) thr_hat[q] = fmaxf(thr_hat[q], thr_run[q]);
}
hipError_t launch_raise_thr(float* thr_hat, const float* thr_run, int n_queries, hipStream_t s) {
  hipLaunchKernelGGL(raise_thr_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, thr_hat, thr_run, n_queries);
  return hipGetLastError();
}
// Staged (sharded) search: the sampled run of this shard dropped sample rows below ITS OWN thresholds (maximum: tau_own).  The common
// threshold tau_ext normally lies far above them (a shard chases only its share of the union's rank); where it does not, rows of the
// sample tiles in [tau_ext, tau_own) may be missing from the shard's list, so the query is flagged and redone with local thresholds.
__global__ void flag_tau_above_kernel(const float* __restrict__ tau_own, const float* __restrict__ tau_ext, int n_queries, uint32_t* __restrict__ fail_flags) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n_queries && tau_own[q] > tau_ext[q]) fail_flags[q] = 1u;
}
hipError_t launch_flag_tau_above(const float* tau_own, const float* tau_ext, int n_queries, uint32_t* fail_flags, hipStream_t s) {
  hipLaunchKernelGGL(flag_tau_above_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, tau_own, tau_ext, n_queries, fail_flags);
  return hipGetLastError();
}
// Extrapolated threshold ("thresholds"): the rows seen so far are a scattered fraction f of the corpus, so the
// r-th best seen, r = k f + 6 sqrt(k f (1 - f)) + 4, lies below the final k-th best score except with negligible probability; a query
// for which it does not is caught by the verification against tau_hat (raised here too) and redone.
__global__ void raise_thr_rank_kernel(float* __restrict__ thr_hat, float* __restrict__ tau_hat, const uint64_t* __restrict__ topk_keys, int kp, int r,
                                      const float* __restrict__ margin, int n_queries) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_queries) return;
  const uint64_t key = topk_keys[(int64_t)q * kp + (r - 1)];
  if (key == 0ull) return;
  const float t = ordered_f32((uint32_t)(key >> 32));
  if (!(t > tau_hat[q])) return;
  tau_hat[q] = t;
  thr_hat[q] = fmaxf(thr_hat[q], t - margin[q]);
}
hipError_t launch_raise_thr_rank(float* thr_hat, float* tau_hat, const uint64_t* topk_keys, int kp, int r, const float* margin, int n_queries, hipStream_t s) {
  hipLaunchKernelGGL(raise_thr_rank_kernel, dim3((n_queries + 255) / 256), dim3(256), 0, s, thr_hat, tau_hat, topk_keys, kp, r, margin, n_queries);
  return hipGetLastError();
}
// Rows of the final list that reach tau (all valid rows when tau is null); -1 when the shard's lists overflowed.
__global__ void count_ge_kernel(const uint64_t* __restrict__ topk_keys, int kp, int k, const float* __restrict__ tau,
                                const uint32_t* __restrict__ fail_flags, int n_queries, int32_t* __restrict__ out) {
  const int q = blockIdx.x;
  if (q >= n_queries) return;
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const float t = tau ? tau[q] : -INFINITY;
  int c = 0;
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    const uint64_t key = topk_keys[(int64_t)q * kp + j];
    if (key && ordered_f32((uint32_t)(key >> 32)) >= t) ++c;
  }
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) out[q] = (fail_flags && fail_flags[q]) ? -1 : cnt;
}
hipError_t launch_count_ge(const uint64_t* topk_keys, int kp, int k, const float* tau, const uint32_t* fail_flags, int n_queries,
                           int32_t* out, hipStream_t s) {
  if (n_queries <= 0) return hipSuccess;
  hipLaunchKernelGGL(count_ge_kernel, dim3((unsigned)n_queries), dim3(256), 0, s, topk_keys, kp, k, tau, fail_flags, n_queries, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ merge_topk
// Per query: k_out best of n_in (score,row) pairs, order (score desc, row asc); row < 0 = padding.
// LDS: ordered score (u32) + position (u32) per entry; ties are resolved on the int64 rows in HBM.
__global__ void __launch_bounds__(1024) merge_topk_kernel(int n_in, int n_pad, const float* __restrict__ in_scores,
                                                          const int64_t* __restrict__ in_rows, int k_out,
                                                          float* __restrict__ out_scores, int64_t* __restrict__ out_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* sk = (uint32_t*)smem;
  uint32_t* pos = sk + n_pad;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float* s = in_scores + (int64_t)q * n_in;
  const int64_t* r = in_rows + (int64_t)q * n_in;
  for (int j = tid; j < n_pad; j += 1024) {
    const bool ok = j < n_in && r[j] >= 0;
    sk[j] = ok ? f32_ordered(s[j]) : 0u;
    pos[j] = ok ? (uint32_t)j : 0xFFFFFFFFu;
  }
  __syncthreads();
  auto before = [&](uint32_t ka, uint32_t pa, uint32_t kb, uint32_t pb) -> bool {   // a sorts before b
    if (pa == 0xFFFFFFFFu) return false;
    if (pb == 0xFFFFFFFFu) return true;
    if (ka != kb) return ka > kb;
    return r[pa] < r[pb];
  };
  for (int size = 2; size <= n_pad; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (n_pad >> 1); t += 1024) {
        const int p0 = 2 * t - (t & (stride - 1)), p1 = p0 + stride;
        const uint32_t ka = sk[p0], pa = pos[p0], kb = sk[p1], pb = pos[p1];
        const bool desc = (p0 & size) == 0;
        const bool swap = desc ? before(kb, pb, ka, pa) : before(ka, pa, kb, pb);
        if (swap) { sk[p0] = kb; pos[p0] = pb; sk[p1] = ka; pos[p1] = pa; }
      }
      __syncthreads();
    }
  for (int j = tid; j < k_out; j += 1024) {
    const uint32_t pj = (j < n_pad) ? pos[j] : 0xFFFFFFFFu;
    out_scores[(int64_t)q * k_out + j] = (pj != 0xFFFFFFFFu) ? s[pj] : -INFINITY;
    out_rows[(int64_t)q * k_out + j] = (pj != 0xFFFFFFFFu) ? r[pj] : -1;
  }
}
hipError_t launch_merge_topk(int n_queries, int n_in, const float* in_scores, const int64_t* in_rows, int k_out,
                             float* out_scores, int64_t* out_rows, hipStream_t s) {
  if (n_queries <= 0) return hipSuccess;
  int n_pad = 2;
  while (n_pad < n_in) n_pad <<= 1;
  const int bytes = n_pad * 8;
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  static int attr_bytes[64] = {};
  if (hipError_t e = ensure_lds_attr((const void*)merge_topk_kernel, bytes, attr_bytes); e != hipSuccess) return e;
  hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)n_queries), dim3(1024), bytes, s, n_in, n_pad, in_scores, in_rows,
                     k_out, out_scores, out_rows);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ merge_lists
// The same reduce for lists that arrive SORTED, in the layout an all-gather leaves them in: [n_lists, Q, L].
// No sort: every entry finds its output rank = its own position + the number of entries of every other list
// that come before it (searched in LDS).  Order: (score desc, row asc, list asc);
// padding (row < 0) ranks behind everything.  rows == nullptr: scores only, ties keep list order.
template <int NL>
__global__ void __launch_bounds__(1024) merge_lists_kernel(int n_lists, int L, int n_queries, const float* __restrict__ in_scores,
                                                          const int64_t* __restrict__ in_rows, int k_out,
                                                          float* __restrict__ out_scores, int64_t* __restrict__ out_rows,
                                                          int64_t stride_s, int64_t stride_r) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n_tot = n_lists * L;
  int64_t* rw = (int64_t*)smem;                           // [n_tot] rows (only with in_rows)
  int64_t* orow = rw + (in_rows ? n_tot : 0);             // [k_out] output rows, staged so that the HBM writes are coalesced
  uint32_t* sk = (uint32_t*)(orow + (in_rows ? k_out : 0));
  float* osc = (float*)(sk + n_tot);                      // [k_out] output scores
  const int q = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  for (int j = tid; j < k_out; j += nthr) {
    osc[j] = -INFINITY;
    if (in_rows) orow[j] = -1;
  }
  // eight lists' loads in flight per thread (one at a time they were serial HBM round trips; a flat index costs a division per entry and the
  // kernel is bound by instruction issue)
  for (int l0 = 0; l0 < n_lists; l0 += 8)
    for (int j = tid; j < L; j += nthr) {
      float sc[8];
      int64_t rr[8];
      const int64_t in_list = (int64_t)q * L + j;          // (list l's section starts at l * stride)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sc[u] = 0.f; rr[u] = 0;
        if (l0 + u < n_lists) {
          sc[u] = in_scores[(int64_t)(l0 + u) * stride_s + in_list];
          if (in_rows) rr[u] = in_rows[(int64_t)(l0 + u) * stride_r + in_list];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (l0 + u < n_lists) {
          const int e = (l0 + u) * L + j;
          uint32_t key = f32_ordered(sc[u]);
          if (in_rows) {
            if (rr[u] < 0) key = 0u;
            rw[e] = rr[u] < 0 ? INT64_MAX : rr[u];
          }
          sk[e] = key;
        }
    }
  __syncthreads();
  // Entries that cannot reach the output are cut first: with P = ceil(k_out / n_lists), every list holds P entries >= the SMALLEST of the
  // lists' P-th keys, so k_out entries reach that key and nothing below it is wanted.  At 8 lists of 448 for k = 1000 that leaves ~1 100 of
  // 3 584 entries, spread over all threads (until round 4 a thread owned a fixed run of one list: the lanes on the lists' tails searched
  // seven lists each only to find a rank beyond k_out: 0.85 -> 0.52 ms for 6 980 queries x 8 lists of 448).
  __shared__ int act[64];
  __shared__ uint32_t t_min;
  if (tid == 0) t_min = 0xFFFFFFFFu;
  __syncthreads();
  const int P = (k_out + n_lists - 1) / n_lists;
  if (tid < n_lists) atomicMin(&t_min, P <= L ? sk[tid * L + P - 1] : 0u);
  __syncthreads();
  if (tid < n_lists) {
    const uint32_t t = t_min;
    int lo = 0, hi = L;
    while (lo < hi) {                                      // first entry below t
      const int mid = (lo + hi) >> 1;
      if (sk[tid * L + mid] >= t) lo = mid + 1; else hi = mid;
    }
    act[tid] = lo;
  }
  __syncthreads();
  int n_act = 0;
  for (int m = 0; m < n_lists; ++m) n_act += act[m];
  // A thread owns a run of consecutive live entries (list-major).  The first entry of a list in its run finds its place in every other
  // list by binary search; the next ones only advance from there (the places are monotone along a sorted list).
  const int run = (n_act + nthr - 1) / nthr;
  const int f0 = tid * run, f1 = min(n_act, f0 + run);
  if (f0 < f1) {
    int l = 0, base = 0;
    while (f0 >= base + act[l]) { base += act[l]; ++l; }
    int j = f0 - base;
    bool fresh = true;
    int pos[NL];
    for (int f = f0; f < f1; ++f, ++j) {
      while (j >= act[l]) { ++l; j = 0; fresh = true; }
      const int e = l * L + j;
      const uint32_t key = sk[e];
      const int64_t row = in_rows ? rw[e] : 0;
      auto before = [&](int m, int i) -> bool {            // entry i of list m sorts before e: (score desc, row asc, list asc)
        const uint32_t km = sk[m * L + i];
        if (km != key) return km > key;
        if (!in_rows) return m < l;
        const int64_t rm = rw[m * L + i];
        return rm < row || (rm == row && m < l);
      };
      int rank = j;
#pragma unroll
      for (int m = 0; m < NL; ++m) {
        if (m < n_lists && m != l) {
          const int am = act[m];
          int p = fresh ? 0 : pos[m];
          if (fresh) {
            int hi = am;
            while (p < hi) {
              const int mid = (p + hi) >> 1;
              if (before(m, mid)) p = mid + 1; else hi = mid;
            }
          } else {
            while (p < am && before(m, p)) ++p;
          }
          pos[m] = p;
          rank += p;
        }
      }
      fresh = false;
      if (rank >= k_out) {                                 // ranks only grow along a list: the rest of this list in the run is not wanted
        f += act[l] - 1 - j;
        j = act[l] - 1;
        continue;
      }
      if (in_rows && row == INT64_MAX) continue;           // padding: the slot keeps (-inf, -1)
      osc[rank] = ordered_f32(key);
      if (in_rows) orow[rank] = row;
    }
  }
  __syncthreads();
  for (int j = tid; j < k_out; j += nthr) {
    out_scores[(int64_t)q * k_out + j] = osc[j];
    if (out_rows) out_rows[(int64_t)q * k_out + j] = orow[j];
  }
}
hipError_t launch_merge_lists(int n_queries, int n_lists, int list_len, const float* in_scores, const int64_t* in_rows, int k_out,
                              float* out_scores, int64_t* out_rows, hipStream_t s, int64_t stride_s, int64_t stride_r) {
  if (n_queries <= 0) return hipSuccess;
  if (stride_s <= 0) stride_s = (int64_t)n_queries * list_len;
  if (stride_r <= 0) stride_r = (int64_t)n_queries * list_len;
  const size_t bytes = ((size_t)n_lists * list_len + (size_t)k_out) * (in_rows ? 12 : 4);
  if (bytes > 160 * 1024 || n_lists > 64) return hipErrorInvalidValue;
  const int threads = 512;     // measured best of 256 / 512 / 1024 on every shard-reduce shape
  auto go = [&](auto kernel, size_t& attr_bytes) -> hipError_t {
    if (bytes > attr_bytes) {
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return e;
      attr_bytes = bytes;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_queries), dim3(threads), bytes, s, n_lists, list_len, n_queries, in_scores, in_rows,
                       k_out, out_scores, out_rows, stride_s, stride_r);
    return hipGetLastError();
  };
  static size_t a2 = 0, a4 = 0, a8 = 0, a16 = 0, a64 = 0;
  if (n_lists <= 2) return go(merge_lists_kernel<2>, a2);
  if (n_lists <= 4) return go(merge_lists_kernel<4>, a4);
  if (n_lists <= 8) return go(merge_lists_kernel<8>, a8);
  if (n_lists <= 16) return go(merge_lists_kernel<16>, a16);
  return go(merge_lists_kernel<64>, a64);
}

}  // namespace dhr
