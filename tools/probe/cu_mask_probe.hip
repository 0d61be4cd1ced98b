// Which CUs / XCDs does a CU-masked stream run on?  Each workgroup records (XCC_ID, SE_ID, CU_ID); the host prints,
// per mask, how many distinct CUs were used on each XCD.  Build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
__global__ void who(uint32_t* out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // spin a little so that workgroups spread over every enabled CU
  uint64_t t0 = clock64(); while (clock64() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int blocks = 4096;
  uint32_t* d; hipMalloc(&d, blocks * 8);
  hipLaunchKernelGGL(who, dim3(blocks), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<uint32_t> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::set<uint32_t> cus[8]; int wg[8] = {0};
  for (int i = 0; i < blocks; ++i) { const uint32_t x = h[2 * i] & 0xf; if (x < 8) { cus[x].insert(h[2 * i + 1] & 0xfff0); wg[x]++; } }   // HW_ID: wave/simd in the low 6 bits; cu [11:8], sh [12], se [15:13]
  printf("%-28s CUs per XCD:", name);
  int tot = 0;
  for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
  printf("  total %3d   WGs per XCD:", tot);
  for (int x = 0; x < 8; ++x) printf(" %4d", wg[x]);
  printf("\n");
  hipFree(d); hipStreamDestroy(s);
}
int main() {
  std::vector<uint32_t> full(8, 0xffffffffu);
  run("all 256 bits", full);
  { std::vector<uint32_t> m(8, 0); m[0] = m[1] = 0xffffffffu; run("bits 0-63", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; run("bits 0-31", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xff; run("bits 0-7", m); }
  { std::vector<uint32_t> m(8, 0x11111111u); run("every 4th bit", m); }
  { std::vector<uint32_t> m(8, 0); m[6] = m[7] = 0xffffffffu; run("bits 192-255", m); }
  { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = m[1] = 0; run("all but bits 0-63", m); }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
