// Probe of the 8-bit 2:4 matrix instructions of gfx950 (no ISA document in this image):
//   v_smfmac_i32_32x32x64_i8, v_smfmac_f32_32x32x64_fp8_fp8
// (1) register layouts, derived experimentally with the int8 form and then VALIDATED on random operands for both forms;
// (2) sustained rate under the package power cap on operand data shaped like the bound GEMM's gated half, next to the
//     instructions the kernel uses today (same box, same run).
// Build: hipcc --offload-arch=gfx950 -O2 -o _bin/smfmac8_probe smfmac8_probe.hip ; run: ./smfmac8_probe [seconds per rate row]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ layout
template <int FP8>
__global__ void one(const uint8_t* a, const uint8_t* b, const int* idx, float* out) {
  const int l = threadIdx.x;
  intx4 av; intx8 bv;
  for (int e = 0; e < 4; ++e) av[e] = ((const int*)a)[l * 4 + e];
  for (int e = 0; e < 8; ++e) bv[e] = ((const int*)b)[l * 8 + e];
  float r[16];
  if (FP8) {
    floatx16 c = {0};
    c = __builtin_amdgcn_smfmac_f32_32x32x64_fp8_fp8(av, bv, c, idx[l], 0, 0);
    for (int e = 0; e < 16; ++e) r[e] = c[e];
  } else {
    intx16 c = {0};
    c = __builtin_amdgcn_smfmac_i32_32x32x64_i8(av, bv, c, idx[l], 0, 0);
    for (int e = 0; e < 16; ++e) r[e] = (float)c[e];
  }
  for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = r[e];   // D: col = l&31, row = (e&3)+8*(e>>2)+4*(l>>5)
}
static uint8_t *g_da, *g_db; static int* g_di; static float* g_do;
static std::vector<float> run1(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, const std::vector<int>& idx, int fp8) {
  hipMemcpy(g_da, a.data(), 64 * 16, hipMemcpyHostToDevice);
  hipMemcpy(g_db, b.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(g_di, idx.data(), 64 * 4, hipMemcpyHostToDevice);
  if (fp8) hipLaunchKernelGGL(one<1>, dim3(1), dim3(64), 0, 0, g_da, g_db, g_di, g_do);
  else hipLaunchKernelGGL(one<0>, dim3(1), dim3(64), 0, 0, g_da, g_db, g_di, g_do);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), g_do, 4096, hipMemcpyDeviceToHost);
  return o;
}

// e4m3fn (OCP): codes 0..0x7E are the non-negative finite values in increasing order
static float e4m3_val(int c) { const int ex = (c >> 3) & 15, m = c & 7; return ex ? std::ldexp(1.f + m / 8.f, ex - 7) : std::ldexp(m / 8.f, -6); }
static uint8_t e4m3_up(float x) { for (int c = 0; c < 0x7F; ++c) if (e4m3_val(c) >= x) return (uint8_t)c; return 0x7E; }

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static inline double unif() { return (double)(rnd() >> 11) / 9007199254740992.0; }

static int mapH[2][16], mapE[2][16];      // A (lane half h, stored element E) at position 0 pairs with B (lane half, byte)

static void layout() {
  std::vector<uint8_t> a(64 * 16), b(64 * 32);
  std::vector<int> idx(64);
  printf("X1: A(L,E)=1, B=1, idx=0x44444444 -> D rows that are nonzero (row sum)\n");
  for (int L : {0, 1, 31, 32, 33, 63}) for (int E : {0, 1, 15}) {
    std::fill(a.begin(), a.end(), 0); std::fill(b.begin(), b.end(), 1); std::fill(idx.begin(), idx.end(), 0x44444444);
    a[L * 16 + E] = 1;
    auto o = run1(a, b, idx, 0);
    printf("  L=%2d E=%2d :", L, E);
    for (int i = 0; i < 32; ++i) { float s = 0; for (int j = 0; j < 32; ++j) s += o[i * 32 + j]; if (s != 0) printf(" row%d(%.0f)", i, s); }
    printf("\n");
  }
  printf("X2: A=1, B(L,e)=1, idx nibble p|p<<2 -> per p: (col, D[0][col])\n");
  for (int L : {0, 5, 32, 37}) for (int e = 0; e < 32; ++e) {
    printf("  L=%2d e=%2d :", L, e);
    for (int p = 0; p < 4; ++p) {
      std::fill(a.begin(), a.end(), 1); std::fill(b.begin(), b.end(), 0);
      int nib = p | (p << 2), w = 0; for (int n = 0; n < 8; ++n) w |= nib << (4 * n);
      std::fill(idx.begin(), idx.end(), w);
      b[L * 32 + e] = 1;
      auto o = run1(a, b, idx, 0);
      int col = -1; float v = 0; for (int j = 0; j < 32; ++j) if (o[j] != 0) { col = j; v = o[j]; }
      printf(" p%d:(%d,%.0f)", p, col, v);
    }
    printf("\n");
  }
  printf("X3: A(L,E)=1, B(l,e)=1+32*(l>>5)+e; D[L&31][0] with idx=0 (base), then with 2-bit field f=0..15 set to 3\n");
  for (int L : {0, 32}) for (int E = 0; E < 16; ++E) {
    printf("  L=%2d E=%2d :", L, E);
    for (int f = -1; f < 16; ++f) {
      std::fill(a.begin(), a.end(), 0);
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 32; ++e) b[l * 32 + e] = (uint8_t)(1 + 32 * (l >> 5) + e);
      std::fill(idx.begin(), idx.end(), f < 0 ? 0 : (3 << (2 * f)));
      a[L * 16 + E] = 1;
      auto o = run1(a, b, idx, 0);
      const int v = (int)o[(L & 31) * 32 + 0];
      if (f < 0) { mapH[L >> 5][E] = (v - 1) >> 5; mapE[L >> 5][E] = (v - 1) & 31; printf(" base=%d (half %d byte %d) |", v, (v - 1) >> 5, (v - 1) & 31); }
      else printf(" %d", v);
    }
    printf("\n");
  }
  // validation on random operands with the mapping X3 found: element E of lane half h at position p multiplies B (mapH, mapE + p)
  for (int fp8 = 0; fp8 < 2; ++fp8) {
    int bad = 0;
    for (int trial = 0; trial < 8; ++trial) {
      int pos[64][16];
      for (int l = 0; l < 64; ++l) {
        int w = 0;
        for (int g = 0; g < 8; ++g) {
          int p0 = (int)(rnd() % 4), p1 = (int)(rnd() % 4);
          if (p0 == p1) p1 = (p0 + 1) % 4;
          if (p0 > p1) std::swap(p0, p1);
          pos[l][2 * g] = p0; pos[l][2 * g + 1] = p1;
          w |= (p0 | (p1 << 2)) << (4 * g);
        }
        idx[l] = w;
        for (int E = 0; E < 16; ++E) a[l * 16 + E] = fp8 ? e4m3_up((float)(rnd() % 5)) : (uint8_t)(int8_t)((int)(rnd() % 15) - 7);
        for (int e = 0; e < 32; ++e) b[l * 32 + e] = fp8 ? e4m3_up((float)(rnd() % 5)) : (uint8_t)(int8_t)((int)(rnd() % 15) - 7);
      }
      auto val = [&](uint8_t c) -> double { return fp8 ? (double)e4m3_val(c) : (double)(int8_t)c; };
      auto o = run1(a, b, idx, fp8);
      for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double s = 0;
        for (int h = 0; h < 2; ++h) for (int E = 0; E < 16; ++E)
          s += val(a[(m + 32 * h) * 16 + E]) * val(b[(n + 32 * mapH[h][E]) * 32 + mapE[h][E] + pos[m + 32 * h][E]]);
        if (std::fabs(s - o[m * 32 + n]) > 1e-3) ++bad;
      }
    }
    printf("validation %s: %d mismatching outputs of %d\n", fp8 ? "fp8" : "i8", bad, 8 * 1024);
  }
}

// ------------------------------------------------------------------------------------------------ rate
// KIND 0: smfmac f16 32x32x32 (today's gated stages), 1: mfma i8 32x32x32 (today's ungated stages), 2: smfmac i8 32x32x64,
// 3: smfmac fp8 32x32x64, 4: dense fp8 32x32x64 (f8f6f4, unit scales).  4 A fragments x 2 B fragments per wave, B-major order
// (keep the B operand, vary A), 2 waves per SIMD.
template <int KIND>
__global__ void __launch_bounds__(256) rate(const uint32_t* __restrict__ data, float* out, int iters, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* src = data + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 64;
  uint32_t w[64];
  for (int i = 0; i < 64; ++i) w[i] = src[i];
  intx4 a4[4]; intx8 a8[4], b8[2]; intx4 b4[2]; int ix[4];
  for (int i = 0; i < 4; ++i) { for (int e = 0; e < 4; ++e) a4[i][e] = w[i * 8 + e]; for (int e = 0; e < 8; ++e) a8[i][e] = w[i * 8 + e]; ix[i] = w[56 + i]; }
  for (int j = 0; j < 2; ++j) { for (int e = 0; e < 8; ++e) b8[j][e] = w[32 + j * 8 + e]; for (int e = 0; e < 4; ++e) b4[j][e] = w[32 + j * 8 + e]; }
  const long long t0 = clock64();
  float s = 0.f;
  if constexpr (KIND == 1 || KIND == 2) {
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
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += (float)c[i][j][e];
  } else {
    floatx16 c[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
    union { half8 h; intx4 v; } ah[4]; union { half16 h; intx8 v; } bh[2];
    for (int i = 0; i < 4; ++i) ah[i].v = a4[i];
    for (int j = 0; j < 2; ++j) bh[j].v = b8[j];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (KIND == 0) c[i][j] = __builtin_amdgcn_smfmac_f32_32x32x32_f16(ah[i].h, bh[j].h, c[i][j], ix[i], 0, 0);
          if constexpr (KIND == 3) c[i][j] = __builtin_amdgcn_smfmac_f32_32x32x64_fp8_fp8(a4[i], b8[j], c[i][j], ix[i], 0, 0);
          if constexpr (KIND == 4) c[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], c[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += c[i][j][e];
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float dlr() { return (float)(unif() < 0.1 ? 0.1 + 2.9 * unif() : 0.02 * unif()); }
// pattern 0: DLR-like fp16 scaled by 200 (the fp16 gated image in accumulator units), 1: int8 image of Gaussian columns,
// 2: DLR-like int8, ceil(v / (3/127)), 3: DLR-like e4m3, rounded up, scaled by 128, 4: zeros, 5: random bytes
static void fill(std::vector<uint32_t>& v, int pattern) {
  uint8_t* p = (uint8_t*)v.data();
  const size_t nb = v.size() * 4;
  for (size_t i = 0; i < nb; i += 2) {
    uint16_t x = 0;
    switch (pattern) {
      case 0: x = f2h(dlr() * 200.f); break;
      case 1: { auto g = [&]() { double u = std::sqrt(-2.0 * std::log(unif() + 1e-300)) * std::cos(6.283185307179586 * unif()); int a = (int)lrint(32.0 * u); return (uint8_t)(a < -127 ? -127 : a > 127 ? 127 : a); }; x = (uint16_t)(g() | (g() << 8)); } break;
      case 2: { auto g = [&]() { return (uint8_t)std::min(127.0, std::ceil(dlr() / (3.0 / 127.0))); }; x = (uint16_t)(g() | (g() << 8)); } break;
      case 3: { auto g = [&]() { return e4m3_up(dlr() * 128.f); }; x = (uint16_t)(g() | (g() << 8)); } break;
      case 4: x = 0; break;
      case 5: x = (uint16_t)rnd(); break;
    }
    p[i] = (uint8_t)x; p[i + 1] = (uint8_t)(x >> 8);
  }
  // position words (u32 number 56..59 of every lane): random valid 2:4 selections
  for (size_t l = 0; l * 64 < v.size(); ++l)
    for (int i = 56; i < 60; ++i) {
      uint32_t w = 0;
      for (int g = 0; g < 8; ++g) { int p0 = (int)(rnd() % 3), p1 = p0 + 1 + (int)(rnd() % (3 - p0)); w |= (uint32_t)(p0 | (p1 << 2)) << (4 * g); }
      v[l * 64 + i] = w;
    }
}
template <int KIND>
static void run_rate(const char* name, int pattern, double macs_per_inst, double seconds) {
  const int blocks = 512;
  const size_t n32 = (size_t)blocks * 256 * 64;
  std::vector<uint32_t> h(n32);
  fill(h, pattern);
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
  printf("%-30s pattern %d: %8.1f ms  %7.1f logical TFLOP/s  %6.2f ns per instruction and SIMD  (s_memtime %.3f)\n", name, pattern, ms,
         n_inst * macs_per_inst * 2.0 / ms / 1e9, ms * 1e6 / (n_inst / 1024), (double)c / ms / 1e6);
  fflush(stdout);
  hipFree(d); hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 1.2;
  hipMalloc(&g_da, 64 * 16); hipMalloc(&g_db, 64 * 32); hipMalloc(&g_di, 64 * 4); hipMalloc(&g_do, 4096);
  layout();
  fflush(stdout);
  const double M = 32.0 * 32;
  run_rate<0>("smfmac f16 32x32x32", 0, M * 32, sec);      // warm-up row
  run_rate<0>("smfmac f16 32x32x32", 0, M * 32, sec);
  run_rate<1>("mfma i8 32x32x32", 1, M * 32, sec);
  run_rate<2>("smfmac i8 32x32x64", 2, M * 64, sec);
  run_rate<2>("smfmac i8 32x32x64", 5, M * 64, sec);
  run_rate<2>("smfmac i8 32x32x64", 4, M * 64, sec);
  run_rate<3>("smfmac fp8 32x32x64", 3, M * 64, sec);
  run_rate<3>("smfmac fp8 32x32x64", 4, M * 64, sec);
  run_rate<4>("mfma f8f6f4 fp8 32x32x64", 3, M * 64, sec);
  run_rate<0>("smfmac f16 32x32x32", 0, M * 32, sec);
  run_rate<3>("smfmac fp8 32x32x64", 3, M * 64, sec);
  return 0;
}
