// Device-side pieces shared by the bound-GEMM translation units (kernels.hip, gemm_w4.hip).
#pragma once
#include "dhr_internal.h"

namespace dhr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int EPI_STACK = 32;          // private hit-stack slots per thread in the filter epilogue
constexpr int WX_META = 4 * SP_SLOT;              // behind the ring: 2 x 1 KiB = thresholds and (int8 image) units of the tile's 256 queries
constexpr int GEMM_RING_LDS = 4 * SP_SLOT + 2048 + 64;   // 4-slot stage ring of the 2:4 kernels (136 KiB) + the per-query constants

// Compressed query fragment -> smfmac B operand: each fp16 slice value v (bucket in the sign bit) becomes its two
// bucket columns (max(v,0), max(-v,0)), one v_pk_max_f16 per output register.  One asm block, so that the two
// wait states a matrix instruction needs after a VALU write of its operand (the compiler cannot see through
// inline asm) are inside it.
__device__ __forceinline__ void expand_bucket_columns(const uint32_t (&r)[4], uint32_t (&o)[8]) {
  asm("v_pk_max_f16 %0, %8, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %1, %8, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %2, %9, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %3, %9, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %4, %10, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %5, %10, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %6, %11, 0 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
      "v_pk_max_f16 %7, %11, 0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"
      "s_nop 1"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
}


hipError_t launch_gemm_wx(const GemmArgs& a, dim3 grid, int variant, hipStream_t s);   // variant 4 (4 waves) or 5 (8 waves)
hipError_t launch_gemm_g8(const GemmArgs& a, dim3 grid, hipStream_t s);                 // gated_i8 indexes (gemm_g8.hip)
#ifdef DHR_AB_VARIANTS
hipError_t launch_gemm_g8p(const GemmArgs& a, hipStream_t s);                           // ... with persistent workgroups (tools/ab/gemm_g8p.hip, A/B builds)
bool gemm_g8p_ok(const GemmArgs& a);
#endif

}  // namespace dhr
