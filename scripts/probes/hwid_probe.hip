// Which hardware wave slots do the waves of two co-resident 256-thread blocks get?  (GEMM phase-lock study)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
  extern __shared__ float smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
  long long t0 = clock64();
  float acc = 0;
  while (clock64() - t0 < spin) acc += smem[threadIdx.x];
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (acc == 12345.f);
  }
}
int main() {
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72000);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 72000, 0, d, 2000000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
  for (int b = 0; b < nb; ++b) {
    unsigned hw = h[b * 8], x = h[b * 8 + 1];
    unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if (b < 24 || (x == 0 && se == 0 && sh == 0 && cu < 2)) {
      printf("block %4d xcc %u se %u sh %u cu %2u :", b, x, se, sh, cu);
      for (int w = 0; w < 4; ++w) printf("  [simd %u slot %u]", (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15);
      printf("\n");
    }
  }
  return 0;
}
