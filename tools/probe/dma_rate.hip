// Micro-benchmark: how fast can one CU pull L2-resident data (a) straight into LDS with global_load_lds,
// (b) into registers with global_load_dwordx4, (c) into registers and on into LDS with ds_write_b128?
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip ; run: ./dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// mode 0: LDS-DMA; 1: to registers; 2: registers -> ds_write_b128
template <int MODE>
__global__ void __launch_bounds__(1024) k(const char* src, size_t region_bytes, int regions, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // each WG streams one region (regions are shared by gridDim.x / regions WGs), wave w takes 8-KiB chunks w, w+nw, ...
  const char* base = src + (size_t)(blockIdx.x % regions) * region_bytes;
  const size_t chunks = region_bytes / 8192;
  char* lds = smem + wave * 8192;
  u4 acc = {0, 0, 0, 0};
  size_t c = wave;
  for (int it = 0; it < iters; ++it) {
    const char* g = base + c * 8192 + lane * 16;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g + j * 1024), LDS_PTR(lds + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      u4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const u4*)(g + j * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j]));
      if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *(u4*)(lds + j * 1024 + lane * 16) = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
      }
    }
    c += nw; if (c >= chunks) c -= chunks;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE == 2 || MODE == 0) acc[0] = *(uint32_t*)(smem + threadIdx.x * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE>
void run(const char* name, const char* src, size_t region, int regions, int waves, uint32_t* sink) {
  const int iters = 2000, blocks = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  size_t lds = (size_t)waves * 8192;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), lds, 0, src, region, regions, 200, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(waves * 64), lds, 0, src, region, regions, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * waves * iters * 8192.0;
  printf("%-28s region %7zu KiB x %3d regions, %2d waves/CU: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, region >> 10, regions, waves,
         bytes / ms / 1e6 / blocks, bytes / ms / 1e9);
}

int main() {
  char* src; uint32_t* sink;
  size_t total = 512u << 20;
  hipMalloc(&src, total); hipMemset(src, 1, total); hipMalloc(&sink, 64);
  for (int waves : {4, 8, 12, 16}) {
    for (auto cfg : std::vector<std::pair<size_t, int>>{{64u << 10, 1}, {1u << 20, 8}, {1u << 20, 32}, {2u << 20, 256}}) {
      if (waves * 8192 > 150 * 1024) continue;
      run<0>("LDS-DMA global_load_lds x4", src, cfg.first, cfg.second, waves, sink);
      run<1>("global_load_dwordx4 -> VGPR", src, cfg.first, cfg.second, waves, sink);
      run<2>("  ... -> ds_write_b128", src, cfg.first, cfg.second, waves, sink);
    }
  }
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
