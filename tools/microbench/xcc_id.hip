// Which XCD does workgroup b run on?  (s_getreg_b32 HW_REG_XCC_ID; the guide: blocks are dealt round-robin over the 8 XCDs)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
  const int n = 1280;
  unsigned *d, h[n];
  (void)hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  printf("raw register of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %x", h[i]); printf("\n");
  int hist[8][8] = {};
  for (int i = 0; i < n; ++i) hist[i % 8][h[i] & 7]++;
  for (int r = 0; r < 8; ++r) { printf("blockIdx %% 8 = %d: XCC", r); for (int c = 0; c < 8; ++c) printf(" %d:%d", c, hist[r][c]); printf("\n"); }
  return 0;
}
