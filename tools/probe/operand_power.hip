// Round 5: does the ENCODING of the bound GEMM's int8 operands cost power?  The launch runs at the package power cap, so a
// matrix instruction is as fast as its operands are cheap to multiply: zeros run at the nominal rate, random bytes far below it
// (profiles/r03_smfmac8_probe.txt).  This probe times the two instructions of gemm_g8.hip on register operands only, 4 A x 2 B fragments per
// wave, 2 waves per SIMD, 512 workgroups, for operand images that carry the SAME information in different bit patterns:
//   dense  (v_mfma_i32_32x32x32_i8): int8 image of Gaussian columns as two's complement (today), with an offset that makes every value
//          non-negative (7 bits), at half the resolution, magnitudes only;
//   gated  (v_smfmac_i32_32x32x64_i8): corpus levels as today; the expanded query operand as today (level - 128, the other bucket
//          column -128) against plain 7-bit levels (other bucket column 0).
// Build: hipcc --offload-arch=gfx950 -O2 -o _bin/operand_power operand_power.hip       Run: _bin/operand_power [seconds per row]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
typedef int intx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(256) rate(const uint32_t* __restrict__ data, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* src = data + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 64;
  uint32_t w[64];
  for (int i = 0; i < 64; ++i) w[i] = src[i];
  intx4 a4[4]; intx8 b8[2]; intx4 b4[2]; int ix[4];
  for (int i = 0; i < 4; ++i) { for (int e = 0; e < 4; ++e) a4[i][e] = w[i * 8 + e]; ix[i] = w[56 + i]; }
  for (int j = 0; j < 2; ++j) { for (int e = 0; e < 8; ++e) b8[j][e] = w[32 + j * 8 + e]; for (int e = 0; e < 4; ++e) b4[j][e] = w[32 + j * 8 + e]; }
  const long long t0 = clock64();
  intx16 c[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (KIND == 1) c[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4[i], b4[j], c[i][j], 0, 0, 0);
        else c[i][j] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a4[i], b8[j], c[i][j], ix[i], 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += (float)c[i][j][e];
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static inline double unif() { return (double)(rnd() >> 11) / 9007199254740992.0; }
static inline double gauss() { return std::sqrt(-2.0 * std::log(unif() + 1e-300)) * std::cos(6.283185307179586 * unif()); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// byte generators ------------------------------------------------------------------------------------------------------------------
static uint8_t g_zero() { return 0; }
static uint8_t g_gauss24() { return (uint8_t)(int8_t)clampi((int)lrint(24.0 * gauss()), -127, 127); }          // today's dense image (max ~ 5.3 sigma = 127)
static uint8_t g_gauss24_off() { return (uint8_t)clampi(64 + (int)lrint(12.0 * gauss()), 0, 127); }          // 7 bits, non-negative (offset 64)
static uint8_t g_gauss12() { return (uint8_t)(int8_t)clampi((int)lrint(12.0 * gauss()), -63, 63); }           // 7 bits, two's complement
static uint8_t g_abs24() { return (uint8_t)clampi(std::abs((int)lrint(24.0 * gauss())), 0, 127); }           // magnitudes only (what the sign bits cost)
static uint8_t g_rand() { return (uint8_t)rnd(); }
static uint8_t g_corpus_level() { return unif() < 0.08 ? (uint8_t)(4 + rnd() % 124) : (uint8_t)1; }           // gated corpus: 8 % active entries, background = level 1
static int q_level8() { return unif() < 0.01 ? (int)(8 + rnd() % 248) : (int)(1 + rnd() % 2); }               // gated query, 8-bit levels
static int q_level7() { return unif() < 0.01 ? (int)(4 + rnd() % 124) : 1; }                                  // ... 7-bit levels

enum { B_PLAIN = 0, B_TODAY = 1, B_LEVEL7 = 2 };
static void fill(std::vector<uint32_t>& v, uint8_t (*ga)(), uint8_t (*gb)(), int b_mode) {
  uint8_t* p = (uint8_t*)v.data();
  for (size_t l = 0; l * 64 < v.size(); ++l) {
    uint8_t* w = p + l * 256;
    for (int i = 0; i < 128; ++i) w[i] = ga();                      // A fragments: u32 0..31
    if (b_mode == B_PLAIN) for (int i = 128; i < 192; ++i) w[i] = gb();
    else
      for (int i = 128; i < 192; i += 2) {                         // B fragments (expanded query): (bucket-0 column, bucket-1 column) per slice
        const bool b1 = (rnd() & 1) != 0;
        const uint8_t val = b_mode == B_TODAY ? (uint8_t)(int8_t)(q_level8() - 128) : (uint8_t)q_level7();
        const uint8_t none = b_mode == B_TODAY ? (uint8_t)0x80 : (uint8_t)0;
        w[i] = b1 ? none : val; w[i + 1] = b1 ? val : none;
      }
    for (int i = 192; i < 256; ++i) w[i] = 0;
    for (int i = 56; i < 60; ++i) {                                // position words: random valid 2:4 selections
      uint32_t x = 0;
      for (int g = 0; g < 8; ++g) { int p0 = (int)(rnd() % 3), p1 = p0 + 1 + (int)(rnd() % (3 - p0)); x |= (uint32_t)(p0 | (p1 << 2)) << (4 * g); }
      v[l * 64 + i] = x;
    }
  }
}
template <int KIND>
static void run_rate(const char* name, uint8_t (*ga)(), uint8_t (*gb)(), int b_mode, double seconds) {
  const int blocks = 512;
  const size_t n32 = (size_t)blocks * 256 * 64;
  std::vector<uint32_t> h(n32);
  fill(h, ga, gb, b_mode);
  uint32_t* d; float* out; long long* cyc;
  hipMalloc(&d, n32 * 4); hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
  hipMemcpy(d, h.data(), n32 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<KIND>), dim3(blocks), dim3(256), 0, 0, d, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = (int)(iters * (seconds * 1e3 / ms));
  }
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_inst = (double)blocks * 4 * iters * 8;
  printf("%-58s %6.2f ns per instruction and SIMD   clock %.3f GHz\n", name, ms * 1e6 / (n_inst / 1024), (double)c / ms / 1e6);
  fflush(stdout);
  hipFree(d); hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 1.5;
  run_rate<1>("warm-up", g_gauss24, g_gauss24, B_PLAIN, sec);
  for (int rep = 0; rep < 2; ++rep) {
    run_rate<1>("dense i8: zeros", g_zero, g_zero, B_PLAIN, sec);
    run_rate<1>("dense i8: Gaussian sigma 24, two's complement (today)", g_gauss24, g_gauss24, B_PLAIN, sec);
    run_rate<1>("dense i8: 7 bits, offset 64 (non-negative), both operands", g_gauss24_off, g_gauss24_off, B_PLAIN, sec);
    run_rate<1>("dense i8: 7 bits, offset 64, corpus operand only", g_gauss24_off, g_gauss24, B_PLAIN, sec);
    run_rate<1>("dense i8: 7 bits two's complement (sigma 12)", g_gauss12, g_gauss12, B_PLAIN, sec);
    run_rate<1>("dense i8: magnitudes only (sigma 24)", g_abs24, g_abs24, B_PLAIN, sec);
    run_rate<1>("dense i8: random bytes", g_rand, g_rand, B_PLAIN, sec);
    run_rate<2>("gated 2:4 i8: zeros", g_zero, g_zero, B_PLAIN, sec);
    run_rate<2>("gated 2:4 i8: corpus levels x query level - 128 (today)", g_corpus_level, nullptr, B_TODAY, sec);
    run_rate<2>("gated 2:4 i8: corpus levels x plain 7-bit query levels", g_corpus_level, nullptr, B_LEVEL7, sec);
    run_rate<2>("gated 2:4 i8: random bytes", g_rand, g_rand, B_PLAIN, sec);
  }
  return 0;
}
