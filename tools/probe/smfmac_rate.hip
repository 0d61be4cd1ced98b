// Issue-rate probe: v_smfmac_f32_32x32x32_f16 vs v_mfma_f32_32x32x16_f16 (register operands only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  half8 a; half16 b; 
  for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * (threadIdx.x + e));
  for (int e = 0; e < 16; ++e) b[e] = (_Float16)(0.02f * (threadIdx.x + e));
  half8 b8; for (int e = 0; e < 8; ++e) b8[e] = b[e];
  floatx16 c[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
  const int idx = 0x4444 | (0x4444 << 16);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b8, c[i], 0, 0, 0);
      else c[i] = __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c[i], idx, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_inst = 1024.0 * 4 * iters * 8;      // wave-instructions
    const double flops = n_inst * 2.0 * 32 * 32 * (mode == 0 ? 16 : 32);
    printf("%s: %.3f ms, %.1f TFLOP/s (logical), %.2f ns per wave-instruction-slot (1024 SIMDs)\n", mode == 0 ? "mfma 32x32x16 f16" : "smfmac 32x32x32 f16", ms, flops / ms / 1e9, ms * 1e6 / (n_inst / 1024));
  }
  return 0;
}
