// Micro-benchmark: per-wave cost (cycles) of the operations the photon kernel is made of, on gfx950.
// One wave per SIMD x 4 waves per SIMD variants; each test is a dependent-free batch of N ops per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <stdint.h>
#include "../../i3rc-monte-carlo-model_amd/csrc/philox.hpp"

#define REP 256
template <int OP>
__global__ void k(float *out, const float *in, long long *cycles) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 64] + 1.5f, c = in[threadIdx.x + 128] + 2.5f, acc = 0.f;
  uint32_t u = __float_as_uint(a) | 1u, v = __float_as_uint(b);
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < REP; ++i) {
    if (OP == 0) { acc += a * b; a += 1e-7f; }                                    // mul+add (2 valu)
    if (OP == 1) { acc += a / b; b += 1e-7f; }                                    // IEEE divide
    if (OP == 2) { acc += a * __builtin_amdgcn_rcpf(b); b += 1e-7f; }             // rcp multiply
    if (OP == 3) { acc += logf(fabsf(a) + 1e-3f); a += 1e-3f; }                   // accurate log
    if (OP == 4) { acc += __logf(fabsf(a) + 1e-3f); a += 1e-3f; }                 // native log
    if (OP == 5) { acc += cosf(a); a += 1e-3f; }                                  // accurate cos
    if (OP == 6) { acc += __cosf(a); a += 1e-3f; }                                // native cos
    if (OP == 7) { acc += sqrtf(fabsf(a)); a += 1e-3f; }                          // IEEE sqrt
    if (OP == 8) { acc += __builtin_amdgcn_sqrtf(fabsf(a)); a += 1e-3f; }         // native sqrt
    if (OP == 9) { i3rc::Philox4 o = i3rc::philox4x32_10(u, v, i, 0u, 10u, 7u); u ^= o.v[0]; acc += (float)o.v[1]; } // philox
    if (OP == 10) { acc += i3rc::u32_to_unit_float(u); u = u * 1664525u + 1013904223u; }  // u32 -> float via f64
    if (OP == 11) { uint64_t p = (uint64_t)u * v; u = (uint32_t)(p >> 32) ^ (uint32_t)p ^ i; acc += (float)(u & 7); }  // one mad_u64_u32 chain
    if (OP == 12) { acc += acosf(fminf(fabsf(a) * 0.01f, 1.0f)); a += 1e-3f; }    // accurate acos
    if (OP == 13) { acc += expf(-fabsf(a)); a += 1e-3f; }                         // accurate exp
    if (OP == 14) { acc += sinf(a); a += 1e-3f; }
    if (OP == 15) { u ^= u << 13; u ^= u >> 17; u ^= u << 5; acc += (float)(u & 7); } // xorshift32 (3 shifts, 3 xors)
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)u;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, int wavesPerBlock, float *out, float *in, long long *cyc) {
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * wavesPerBlock), 0, 0, out, in, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * wavesPerBlock), 0, 0, out, in, cyc);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
  // s_memtime ticks at a fixed 100 MHz-derived rate? report ticks per op per wave
  printf("%-28s waves/block %2d : %8.2f ticks per op (block wall), %8.2f per op per wave-slot\n", name, wavesPerBlock, s / 256 / REP,
         s / 256 / REP / ((wavesPerBlock + 3) / 4));
}
int main() {
  float *out, *in; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 4096); hipMalloc(&cyc, 256 * 8);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f + 0.001f * i;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int w : {4, 16}) {
    run<0>("mul+add (baseline 2 valu)", w, out, in, cyc);
    run<1>("IEEE divide", w, out, in, cyc);
    run<2>("rcp*mul", w, out, in, cyc);
    run<3>("logf accurate", w, out, in, cyc);
    run<4>("__logf", w, out, in, cyc);
    run<5>("cosf accurate", w, out, in, cyc);
    run<6>("__cosf", w, out, in, cyc);
    run<14>("sinf accurate", w, out, in, cyc);
    run<7>("sqrtf IEEE", w, out, in, cyc);
    run<8>("sqrt native", w, out, in, cyc);
    run<12>("acosf", w, out, in, cyc);
    run<13>("expf", w, out, in, cyc);
    run<9>("philox4x32-10 (4 draws)", w, out, in, cyc);
    run<11>("one mad_u64_u32 + 2 xor", w, out, in, cyc);
    run<10>("u32->float via f64", w, out, in, cyc);
    run<15>("xorshift32", w, out, in, cyc);
  }
  return 0;
}
