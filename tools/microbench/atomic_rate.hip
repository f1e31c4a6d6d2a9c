// Micro-benchmark: sustained rate of float64 global atomics onto a SMALL set of hot addresses (gfx950, whole chip).
// Question behind it: per-batch flux tallies of a fused multi-batch launch.  A small domain (step cloud: 32 columns) has
// 64 hot words per batch; can every photon add to them straight in global memory (about 3.3e9 atomics/s), perhaps into
// one replica of the block per XCD or per workgroup, or must they be gathered in LDS first?
//   hipcc --offload-arch=gfx950 -O2 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// every wave: `iters` rounds; per round `lanes` of its 64 lanes add 1.0 to one of nAddr words of its replica
// replicaMode 0: one block; 1: one per XCD (XCC_ID); 2: blockIdx % nRep
__global__ void __launch_bounds__(256) hot_atomics(double *buf, int nAddr, int stride, int replicaMode, int nRep, int lanes, int iters,
                                                   int work, float *sink) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int rep = replicaMode == 0 ? 0 : (replicaMode == 1 ? (int)(xcc & 7u) : (int)(blockIdx.x % (unsigned)nRep));
  double *const mine = buf + (size_t)rep * stride;
  unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float f = (float)threadIdx.x;
  const bool active = (int)(threadIdx.x & 63) < lanes;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    for (int k = 0; k < work; ++k) f = f * 1.0000001f + 0.5f;   // stand-in for the tracing between two tallies
    if (active) unsafeAtomicAdd(mine + ((s >> 8) % (unsigned)nAddr), 1.0);
  }
  if (f == -1.0f) sink[0] = f;
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 7;
  const int stride = 1024;   // doubles between replicas (8 KB)
  double *buf; float *sink;
  CHK(hipMalloc(&buf, sizeof(double) * (size_t)stride * 4096));
  CHK(hipMalloc(&sink, 16));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  std::printf("%d CUs, %d workgroups of 4 waves\n", prop.multiProcessorCount, blocks);
  std::printf("%-28s %6s %6s %6s %12s %14s %s\n", "replicas", "nAddr", "lanes", "work", "ms", "atomics/s", "check");
  struct Case { const char *name; int mode, nRep; };
  const Case cases[] = {{"one block", 0, 1}, {"per XCD (8)", 1, 8}, {"blockIdx % 64", 2, 64}, {"blockIdx % 512", 2, 512}, {"per workgroup", 2, 4096}};
  for (const Case &c : cases)
    for (int nAddr : {64, 1024})
      for (int lanes : {3, 64})
        for (int work : {0, 300}) {
          if (work == 300 && lanes == 64) continue;
          const int iters = lanes == 64 ? 200 : 2000;
          CHK(hipMemset(buf, 0, sizeof(double) * (size_t)stride * 4096));
          hipLaunchKernelGGL(hot_atomics, dim3(blocks), dim3(256), 0, 0, buf, nAddr, stride, c.mode, c.nRep, lanes, 10, work, sink);   // warm-up
          CHK(hipMemset(buf, 0, sizeof(double) * (size_t)stride * 4096));
          CHK(hipEventRecord(a, 0));
          hipLaunchKernelGGL(hot_atomics, dim3(blocks), dim3(256), 0, 0, buf, nAddr, stride, c.mode, c.nRep, lanes, iters, work, sink);
          CHK(hipEventRecord(b, 0));
          CHK(hipEventSynchronize(b));
          float ms = 0;
          CHK(hipEventElapsedTime(&ms, a, b));
          std::vector<double> host((size_t)stride * 4096);
          CHK(hipMemcpy(host.data(), buf, sizeof(double) * host.size(), hipMemcpyDeviceToHost));
          double sum = 0;
          for (double v : host) sum += v;
          const double n = (double)blocks * 4 * lanes * iters;
          std::printf("%-28s %6d %6d %6d %12.3f %14.3e %s\n", c.name, nAddr, lanes, work, ms, n / ms * 1e3, sum == n ? "ok" : "LOST UPDATES");
          std::fflush(stdout);
        }
  return 0;
}
