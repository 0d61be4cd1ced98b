// Micro-benchmark behind DESIGN.md section 2: what would the north-star's "slice-index -> bucket, LDS-staged scatter" of the gated half
// cost?  That design accumulates every gate MATCH of every (query, row) pair into an LDS-resident score: on the benchmark's data a pair
// has ~70-100 matching slices, i.e. 6 980 x 8 841 823 x ~85 = 5.2e12 accumulates per batch.  This probe measures the chip's rate for
// exactly that primitive, in the three forms a kernel could use:
//   mode 0: ds_add_f32 (LDS atomic, no return) to data-dependent addresses (scores of a 256-row x 64-query block = 64 KiB in LDS)
//   mode 1: the same addresses, conflict-free by construction (lane l only ever touches bank l: the upper bound of the primitive)
//   mode 2: non-atomic read-modify-write (ds_read_b32, v_add, ds_write_b32) on lane-private addresses
//   mode 3: ds_add_u32 (integer LDS atomic: scores in fixed point), data-dependent addresses
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_scatter_probe lds_scatter_probe.hip   (without -munsafe-fp-atomics
// the float atomicAdd on LDS compiles to a compare-and-swap loop, 30x slower than the native ds_add_f32); run: ./lds_scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ addr, const float* __restrict__ val, int per_thread, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = (float*)smem;                       // 16384 floats = 64 KiB
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // every thread owns `per_thread` (address, value) pairs, re-used `iters` times (they sit in registers / L1: the probe times the LDS side)
  uint32_t a[16];
  float v[16];
  for (int j = 0; j < 16; ++j) {
    const uint32_t r = addr[(size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 16 + j] & 16383u;
    a[j] = (MODE == 0 || MODE == 3) ? r : ((r & ~63u) | (uint32_t)lane);      // modes 1, 2: lane l stays in bank l (and in its own words)
    v[j] = val[(threadIdx.x * 16 + j) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 2) { const float x = acc[a[j]]; acc[a[j]] = x + v[j]; }
      else if (MODE == 3) atomicAdd((uint32_t*)&acc[a[j]], __float_as_uint(v[j]) & 0xFFFFu);      // ds_add_u32
      else atomicAdd(&acc[a[j]], v[j]);             // ds_add_f32 (result unused)
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
  if (s == 123.456f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, const uint32_t* d_addr, const float* d_val, float* d_sink, int n_cu) {
  const int threads = 1024, iters = 2000, wgs = n_cu * 2 * 8;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 65536, 0, d_addr, d_val, 16, 10, d_sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 65536, 0, d_addr, d_val, 16, iters, d_sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)wgs * threads * 16.0 * iters;
  const double rate = ops / (ms * 1e-3);
  printf("%-58s %8.2f ms  %.3e accumulates/s  (%.1f per clock and CU at 2.4 GHz)  -> 5.2e12 accumulates = %.0f ms\n", name, ms, rate,
         rate / n_cu / 2.4e9, 5.2e12 / rate * 1e3);
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int n_cu = pr.multiProcessorCount;
  const size_t n = (size_t)n_cu * 16 * 1024 * 16;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 11); }
  float hv[4096];
  for (int i = 0; i < 4096; ++i) hv[i] = 0.001f * (float)(i % 97);
  uint32_t* d_addr; float *d_val, *d_sink;
  hipMalloc(&d_addr, n * 4); hipMalloc(&d_val, sizeof(hv)); hipMalloc(&d_sink, 16);
  hipMemcpy(d_addr, h, n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_val, hv, sizeof(hv), hipMemcpyHostToDevice);
  printf("%s, %d CUs; 2 workgroups of 1024 threads per CU, 64 KiB of LDS scores each\n", pr.name, n_cu);
  run<0>("ds_add_f32, data-dependent addresses (the scatter itself)", d_addr, d_val, d_sink, n_cu);
  run<1>("ds_add_f32, conflict-free (lane l in bank l)", d_addr, d_val, d_sink, n_cu);
  run<2>("ds_read + v_add + ds_write, conflict-free, non-atomic", d_addr, d_val, d_sink, n_cu);
  run<3>("ds_add_u32, data-dependent addresses (fixed-point scores)", d_addr, d_val, d_sink, n_cu);
  printf("for comparison: the whole search step (bound GEMM + refine + exact rescoring + select) takes ~137 ms\n");
  return 0;
}
