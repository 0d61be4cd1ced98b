// Pieces shared by the two bound-GEMM kernels of a gated_i8 index: gemm_g8.hip (one workgroup per tile; the debug dump runs there) and
// gemm_g8p.hip (persistent workgroups: the LDS-DMA stream runs on across tiles).
#pragma once
#include "gemm_common.h"
#include <climits>

namespace dhr {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));

constexpr int G8_QOFF = 16384;                 // query part of a ring slot (the corpus part is 10 KiB gated / 16 KiB ungated)
constexpr int G8_SLOT = 32768;
constexpr int G8_META = 4 * G8_SLOT;           // behind the ring: 4 x 1 KiB = the tile's 256 row sums, and unit / threshold / shift of its 256 queries
constexpr int G8_RING_LDS = 4 * G8_SLOT + 4096 + 64;
constexpr int G8_NT = 512;


__device__ __forceinline__ void g8_smfmac(floatx16& c, const intx4& a, const intx8& b, uint32_t idx) {
  asm("v_smfmac_i32_32x32x64_i8 %0, %1, %2, %3" : "+v"(c) : "v"(a), "v"(b), "v"(idx));
}
__device__ __forceinline__ void g8_mfma(floatx16& c, const intx4& a, const intx4& b) {
  asm("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

struct G8Frag {
  intx4 a[4];                       // corpus fragments of the block's four 32-row blocks
  intx4 pw;                         // gated blocks: their position words (one 16-byte read)
  union { intx8 v; intx4 h[2]; } b[2];   // query fragments (ungated blocks use h[0])
};

// filter threshold of a query in accumulator units, rounded DOWN (a row is kept when sum >= thr)
__device__ __forceinline__ int g8_thr_units(float thr, float mul) {
  const float x = thr / mul;
  if (!(x < 2.1e9f)) return INT_MAX;          // +inf (padded query), NaN, mul == 0
  if (x < -2.1e9f) return INT_MIN;
  return (int)floorf(x - fabsf(x) * 1e-6f) - 1;
}
__device__ __forceinline__ float g8_score(int sum, float mul) {      // accumulator units -> score units, rounded up
  const float x = (float)sum * mul;
  return x + fabsf(x) * 2.4e-7f;
}


hipError_t launch_gemm_g8p(const GemmArgs& a, hipStream_t s);      // gemm_g8p.hip: persistent workgroups over the whole launch
bool gemm_g8p_ok(const GemmArgs& a);                                 // ... for the launches that kernel takes

}  // namespace dhr
