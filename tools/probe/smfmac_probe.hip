// Probe of v_smfmac_f32_32x32x32_f16 register layouts on gfx950 (no ISA document in this image).
// hipcc --offload-arch=gfx950 -O2 -o smfmac_probe smfmac_probe.hip && ./smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half16 __attribute__((ext_vector_type(16)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int ABID>
__global__ void k(const _Float16* a, const _Float16* b, const int* idx, float* out) {
  const int l = threadIdx.x;
  half8 av; half16 bv;
  for (int e = 0; e < 8; ++e) av[e] = a[l * 8 + e];
  for (int e = 0; e < 16; ++e) bv[e] = b[l * 16 + e];
  floatx16 c = {0};
  c = __builtin_amdgcn_smfmac_f32_32x32x32_f16(av, bv, c, idx[l], 0, ABID);
  // D layout (dense 32x32 result): col = l&31, row = (e&3)+8*(e>>2)+4*(l>>5)
  for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[e];
}

static std::vector<float> run(const std::vector<_Float16>& a, const std::vector<_Float16>& b, const std::vector<int>& idx, int abid = 0) {
  _Float16 *da, *db; int* di; float* dout;
  hipMalloc(&da, 64 * 8 * 2); hipMalloc(&db, 64 * 16 * 2); hipMalloc(&di, 64 * 4); hipMalloc(&dout, 1024 * 4);
  hipMemcpy(da, a.data(), 64 * 8 * 2, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 64 * 16 * 2, hipMemcpyHostToDevice);
  hipMemcpy(di, idx.data(), 64 * 4, hipMemcpyHostToDevice);
  if (abid == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, da, db, di, dout);
  else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, da, db, di, dout);
  std::vector<float> o(1024);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(di); hipFree(dout);
  return o;
}

int main() {
  std::vector<_Float16> a(512), b(1024);
  std::vector<int> idx(64);
  // ---- exp 1: A single nonzero (lane L, elem E), B all ones, idx = 0x4444 (pos0=0,pos1=1): which D rows light up
  printf("exp1: A(L,E)=1, B=1 -> rows with nonzero D (value)\n");
  for (int L : {0, 1, 31, 32, 33, 63}) for (int E : {0, 1, 7}) {
    std::fill(a.begin(), a.end(), (_Float16)0.f); std::fill(b.begin(), b.end(), (_Float16)1.f);
    std::fill(idx.begin(), idx.end(), 0x44444444);
    a[L * 8 + E] = (_Float16)1.f;
    auto o = run(a, b, idx);
    printf("  L=%2d E=%d :", L, E);
    for (int i = 0; i < 32; ++i) { float s = 0; for (int j = 0; j < 32; ++j) s += o[i * 32 + j]; if (s != 0) printf(" row%d(sum %.0f)", i, s); }
    printf("\n");
  }
  // ---- exp 2: A all ones, B single nonzero (L,E); idx nibble = p|(p<<2): D column and value per p
  printf("exp2: A=1, B(L,E)=1, idx nibble p|p<<2 -> for p=0..3: (col, D[0][col])\n");
  for (int L : {0, 5, 32, 37}) for (int E = 0; E < 16; ++E) {
    printf("  L=%2d E=%2d :", L, E);
    for (int p = 0; p < 4; ++p) {
      std::fill(a.begin(), a.end(), (_Float16)1.f); std::fill(b.begin(), b.end(), (_Float16)0.f);
      int nib = p | (p << 2), w = 0; for (int n = 0; n < 8; ++n) w |= nib << (4 * n);
      std::fill(idx.begin(), idx.end(), w);
      b[L * 16 + E] = (_Float16)1.f;
      auto o = run(a, b, idx);
      int col = -1; float v = 0; for (int j = 0; j < 32; ++j) if (o[j] != 0) { col = j; v = o[j]; }
      printf(" p%d:(%d,%.0f)", p, col, v);
    }
    printf("\n");
  }
  // ---- exp 3: A single nonzero (L,E)=1; B[lane][e] = code lane_half*16+e (value = 1 + code); idx = 0 except one 2-bit field f set to 1,2,3
  printf("exp3: A(L,E)=1, B(l,e)=1+16*(l>>5)+e; D[row L&31][col 0] for idx=0 and for each 2-bit field f (value 3)\n");
  for (int L : {0, 32}) for (int E = 0; E < 8; ++E) {
    printf("  L=%2d E=%d :", L, E);
    for (int f = -1; f < 16; ++f) {
      std::fill(a.begin(), a.end(), (_Float16)0.f);
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) b[l * 16 + e] = (_Float16)(float)(1 + 16 * (l >> 5) + e);
      std::fill(idx.begin(), idx.end(), f < 0 ? 0 : (3 << (2 * f)));
      a[L * 8 + E] = (_Float16)1.f;
      auto o = run(a, b, idx);
      printf(" %s%.0f", f < 0 ? "base=" : "", o[(L & 31) * 32 + 0]);
    }
    printf("\n");
  }
  // ---- exp 4: same as exp3 base but ABID=1 with idx fields in the upper 16 bits
  printf("exp4: abid=1, idx = (3<<(2f)) for f=8..15\n");
  for (int E = 0; E < 8; ++E) {
    printf("  L= 0 E=%d :", E);
    for (int f = 8; f < 16; ++f) {
      std::fill(a.begin(), a.end(), (_Float16)0.f);
      for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) b[l * 16 + e] = (_Float16)(float)(1 + 16 * (l >> 5) + e);
      std::fill(idx.begin(), idx.end(), 3 << (2 * f));
      a[E] = (_Float16)1.f;
      auto o = run(a, b, idx, 1);
      printf(" %.0f", o[0]);
    }
    printf("\n");
  }
  return 0;
}
