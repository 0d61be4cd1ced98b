// Stand-alone driver of the bound GEMM for rocprofv3 --pmc passes (no torch in the process): builds a hybrid
// index from pseudo-random host arrays through the C ABI and times the GEMM over the whole shard.
// Build: hipcc --offload-arch=gfx950 -O2 -I../../include -o _bin/gemm_pmc gemm_pmc.cpp -L../../dhr_amd/csrc -ldhr_hip -Wl,-rpath,'$ORIGIN/../../../dhr_amd/csrc'
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "dhr_hip.h"
static uint32_t rng = 12345u;
static inline uint32_t next() { rng = rng * 1664525u + 1013904223u; return rng >> 8; }
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 500000;
  const int q = argc > 2 ? atoi(argv[2]) : 6980;
  const int d_dlr = argc > 3 ? atoi(argv[3]) : 768, d_cls = 768, k = d_dlr + d_cls;      // d_dlr = 0: dense-only index
  std::vector<__half> cv((size_t)n * k), qv((size_t)q * k);
  std::vector<uint8_t> ci((size_t)n * d_dlr + 1), qi((size_t)q * d_dlr + 1);
  auto fill = [&](std::vector<__half>& v, std::vector<uint8_t>& idx, int64_t rows) {
    for (int64_t r = 0; r < rows; ++r) {
      for (int j = 0; j < d_dlr; ++j) {
        const uint32_t u = next();
        const float x = (u % 13 == 0) ? 0.1f + (float)(u & 0xfff) * (2.9f / 4096.f) : (float)(u & 0xfff) * (0.02f / 4096.f);
        v[(size_t)r * k + j] = __float2half(x);
        idx[(size_t)r * d_dlr + j] = (uint8_t)((u >> 12) % 39);
      }
      for (int j = d_dlr; j < k; ++j) v[(size_t)r * k + j] = __float2half(((float)(next() & 0xffff) / 65536.f - 0.5f) * 0.35f);
    }
  };
  fill(cv, ci, n); fill(qv, qi, q);
  dhr_index_desc d{}; d.device = 0; d.mem_kind = DHR_MEM_HOST; d.n_rows = n; d.d_dlr = d_dlr; d.d_cls = d_cls;
  d.value = cv.data(); d.ld_value = k; d.index = d_dlr ? ci.data() : nullptr; d.index_dtype = d_dlr ? DHR_IDX_U8 : DHR_IDX_NONE; d.idx_buckets = 0; d.ld_index = d_dlr;
  dhr_index* ix = nullptr;
  if (dhr_index_create(&d, &ix)) { printf("create: %s\n", dhr_last_error()); return 1; }
  dhr_query_batch qb{}; qb.n_queries = q; qb.mem_kind = DHR_MEM_HOST; qb.value = qv.data(); qb.value_dtype = DHR_VAL_F16;
  qb.index_dtype = d_dlr ? DHR_IDX_U8 : DHR_IDX_NONE; qb.ld_value = k; qb.index = d_dlr ? qi.data() : nullptr; qb.ld_index = d_dlr;
  double ms = 0, fl = 0;
  if (dhr_debug_gemm_time(ix, &qb, 2, &ms, &fl, nullptr)) { printf("gemm: %s\n", dhr_last_error()); return 1; }
  double i8 = 0;
  dhr_index_get_info(ix, DHR_INFO_DENSE_I8, &i8);
  const int ts = d_dlr / 32, td = i8 != 0 ? 2 * ((d_cls + 127) / 128) : (d_cls + 31) / 32;     // sparse stages (34 KiB corpus + query), dense stages (16 + 16 KiB)
  const double tile_bytes = ts * 34816.0 + td * 32768.0;
  const double tiles = (double)((n + 255) / 256) * ((q + 255) / 256);
  printf("rows %lld queries %d dense_i8 %d: %.3f ms per launch, %.1f TFLOP/s issued; operand bytes through LDS per launch %.2f GB; unique operand bytes %.3f GB\n",
         (long long)n, q, (int)i8, ms, fl / ms / 1e9, tiles * tile_bytes / 1e9,
         ((double)((n + 255) / 256) * (ts * 18432.0 + td * 16384.0) + (double)((q + 255) / 256) * (ts + td) * 16384.0) / 1e9);
  dhr_index_destroy(ix);
  return 0;
}
