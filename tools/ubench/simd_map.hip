// Which SIMD does wave w of a workgroup land on?  (HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8] on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = v;
}
int main() {
  unsigned* d; unsigned h[64];
  hipMalloc(&d, sizeof(h));
  for (int waves : {8, 9, 16}) {
    hipLaunchKernelGGL(k, dim3(2), dim3(64 * waves), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("%2d waves, block %d: simd of wave w:", waves, b);
      for (int w = 0; w < waves; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   cu %u\n", (h[b * 16] >> 8) & 15);
    }
  }
  return 0;
}
