// Micro-benchmark: sustained ISSUE rate of single instruction kinds on the whole chip (gfx950), at k waves per SIMD.
// Settles the "2 or 4 cycles per wave64 vector instruction" question behind bench.py's roofline.issue.peak and prices the
// instruction kinds the photon kernel is made of (selects, compares, integer multiplies, f64 converts, packed f32).
// Each kernel runs LOOPS x 64 instructions of one kind per wave on 8 independent register chains (inline asm, so the
// compiler cannot fold or re-schedule them); time from HIP events, clock from s_memtime against the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>

#define LOOPS 4096
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

struct Clocks { unsigned long long shader, wall; };

#define KERNEL(NAME, DECL, BODY, SINK)                                                                  \
  __global__ void __launch_bounds__(256) NAME(float *out, const float *in, Clocks *clk) {                \
    DECL                                                                                                 \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();                     \
    for (int i = 0; i < LOOPS; ++i) { BODY }                                                             \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                     \
    SINK                                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk->shader = t1 - t0; clk->wall = w1 - w0; }             \
  }

#define FDECL float a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
              float b = in[threadIdx.x + 256], c = in[threadIdx.x + 512];
#define FSINK out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
#define UDECL uint32_t a0 = __float_as_uint(in[threadIdx.x]), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
              uint32_t b = __float_as_uint(in[threadIdx.x + 256]) | 1u, c = __float_as_uint(in[threadIdx.x + 512]);
#define USINK out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);

#define I_FMA(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "v"(b), "v"(c));
#define I_MUL(k) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_ADD(k) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_ADDU(k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_AND(k) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a##k) : "v"(b));
#define I_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : );
// the same select with the mask in a scalar register pair / into a register of its own / after a compare that has just
// written vcc (the form compilers emit)
#define I_CNDMASK_S(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(b) : );
#define I_CNDMASK_D(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a##k) : "v"(b), "v"(c) : );
#define I_CMP_CNDMASK(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
#define I_CMPS_CNDMASK(k) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(b) : "s20", "s21");
#define I_CMP_CNDMASK3(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
#define I_CNDMASK_ADD(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_SWAP(k) asm volatile("v_swap_b32 %0, %1" : "+v"(a##k), "+v"(b));
#define I_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a##k) : "v"(b));
#define I_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##k), "v"(b) : "vcc");
#define I_CMPS(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a##k), "v"(b) : "s20", "s21");
#define I_MULLO(k) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_MULHI(k) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_MUL24(k) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
#define I_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
#define I_LOG(k) asm volatile("v_log_f32 %0, %0" : "+v"(a##k));
#define I_CVTFU(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##k));
#define I_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define I_MIN(k) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_BFE(k) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a##k));
#define I_SALU(k) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define I_FMA_SALU(k) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(a##k) : "v"(b), "v"(c) : "s20", "scc");
#define I_READLANE(k) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a##k) : "s20");
#define I_DSREAD(k) asm volatile("ds_read_b32 %0, %1" : "=v"(a##k) : "v"(addr)); 

KERNEL(k_fma, FDECL, R64(I_FMA), FSINK)
KERNEL(k_mul, FDECL, R64(I_MUL), FSINK)
KERNEL(k_add, FDECL, R64(I_ADD), FSINK)
KERNEL(k_addu, UDECL, R64(I_ADDU), USINK)
KERNEL(k_and, UDECL, R64(I_AND), USINK)
KERNEL(k_lshladd, UDECL, R64(I_LSHLADD), USINK)
KERNEL(k_cndmask, UDECL asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");, R64(I_CNDMASK), USINK)
KERNEL(k_cndmask_s, UDECL asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");, R64(I_CNDMASK_S), USINK)
KERNEL(k_cndmask_d, UDECL asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");, R64(I_CNDMASK_D), USINK)
KERNEL(k_cmp_cndmask, UDECL, R64(I_CMP_CNDMASK), USINK)
KERNEL(k_cmps_cndmask, UDECL, R64(I_CMPS_CNDMASK), USINK)
KERNEL(k_cmp_cndmask3, UDECL, R64(I_CMP_CNDMASK3), USINK)
KERNEL(k_cndmask_add, UDECL asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");, R64(I_CNDMASK_ADD), USINK)
KERNEL(k_swap, UDECL, R64(I_SWAP), USINK out[0] = (float)b;)
KERNEL(k_mov, UDECL, R64(I_MOV), USINK)
KERNEL(k_cmp, FDECL, R64(I_CMP), FSINK)
KERNEL(k_cmps, FDECL, R64(I_CMPS), FSINK)
KERNEL(k_mullo, UDECL, R64(I_MULLO), USINK)
KERNEL(k_mulhi, UDECL, R64(I_MULHI), USINK)
KERNEL(k_mul24, UDECL, R64(I_MUL24), USINK)
KERNEL(k_rcp, FDECL, R64(I_RCP), FSINK)
KERNEL(k_sqrt, FDECL, R64(I_SQRT), FSINK)
KERNEL(k_log, FDECL, R64(I_LOG), FSINK)
KERNEL(k_cvtfu, UDECL, R64(I_CVTFU), USINK)
KERNEL(k_max3, FDECL, R64(I_MAX3), FSINK)
KERNEL(k_min, FDECL, R64(I_MIN), FSINK)
KERNEL(k_bfe, UDECL, R64(I_BFE), USINK)
KERNEL(k_salu, FDECL, R64(I_SALU), FSINK)
KERNEL(k_fma_salu, FDECL, R64(I_FMA_SALU), FSINK)
KERNEL(k_readlane, FDECL, R64(I_READLANE), FSINK)

// 64-bit register pairs: packed f32, f64, mad_u64_u32
#define PDECL double a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
              double b = in[threadIdx.x + 256], c = in[threadIdx.x + 512]; uint32_t ub = __float_as_uint(in[threadIdx.x + 256]) | 1u; float fb = in[threadIdx.x];
#define PSINK out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "v"(b), "v"(c));
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_PKADD(k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_MULF64(k) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(a##k) : "v"(b));
#define I_FMAF64(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a##k) : "v"(b), "v"(c));
#define I_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a##k) : "v"(ub) : "vcc");
#define I_CVTF64U(k) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a##k) : "v"(ub));
#define I_CVTF32F64(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fb) : "v"(a##k));
KERNEL(k_pkfma, PDECL, R64(I_PKFMA), PSINK)
KERNEL(k_pkmul, PDECL, R64(I_PKMUL), PSINK)
KERNEL(k_pkadd, PDECL, R64(I_PKADD), PSINK)
KERNEL(k_mulf64, PDECL, R64(I_MULF64), PSINK)
KERNEL(k_fmaf64, PDECL, R64(I_FMAF64), PSINK)
KERNEL(k_mad64, PDECL, R64(I_MAD64), PSINK)
KERNEL(k_cvtf64u, PDECL, R64(I_CVTF64U), PSINK)
KERNEL(k_cvtf32f64, PDECL, R64(I_CVTF32F64), PSINK out[0] = fb;)

__global__ void __launch_bounds__(256) k_dsread(float *out, const float *in, Clocks *clk) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  const unsigned addr = (unsigned)(threadIdx.x * 4);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < LOOPS; ++i) { R64(I_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)"); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk->shader = t1 - t0; clk->wall = w1 - w0; }
}

typedef void (*Kern)(float *, const float *, Clocks *);
static void run(const char *name, Kern k, int wavesPerSimd, int numCU, float *out, float *in, Clocks *clk, int wallKHz) {
  const int blocks = numCU * wavesPerSimd;   // 256 threads = 4 waves = one per SIMD; k blocks per CU = k waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, clk);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  Clocks h; hipMemcpy(&h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double instr = (double)blocks * 4 * LOOPS * 64;                 // wave-instructions
  const double rate = instr / (ms * 1e-3);
  const double ghz = (double)h.shader / ((double)h.wall / (wallKHz * 1e3)) / 1e9;   // shader clock seen by s_memtime
  const double cyc = (double)numCU * 4 * ghz * 1e9 / rate;              // cycles per wave-instruction per SIMD
  printf("%-18s %d waves/SIMD: %7.3f ms  %.3e wave-instr/s  %.2f cycles/instr/SIMD at %.2f GHz (s_memtime/wall_clock)\n", name, wavesPerSimd, ms, rate, cyc, ghz);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char **argv) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int wallKHz = 100000; hipDeviceGetAttribute(&wallKHz, hipDeviceAttributeWallClockRate, 0);
  printf("%s: %d CUs, clockRate %d kHz, wall clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate, wallKHz);
  float *out, *in; Clocks *clk;
  hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4); hipMalloc(&in, 4096); hipMalloc(&clk, sizeof(Clocks));
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f + 0.001f * i;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  struct { const char *n; Kern k; } tests[] = {
      {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_add_u32", k_addu}, {"v_and_b32", k_and},
      {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask (sgpr)", k_cndmask_s}, {"v_cndmask (new dst)", k_cndmask_d},
      {"v_swap_b32", k_swap}, {"v_mov_b32", k_mov}, {"v_cmp+v_cndmask", k_cmp_cndmask}, {"v_cmp+3 v_cndmask", k_cmp_cndmask3}, {"v_cndmask+v_add", k_cndmask_add}, {"v_cmp_s+nop+cndmsk", k_cmps_cndmask}, {"v_cmp (vcc)", k_cmp}, {"v_cmp (sgpr)", k_cmps},
      {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_bfe_u32", k_bfe}, {"v_mul_u32_u24", k_mul24},
      {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mad_u64_u32", k_mad64}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
      {"v_log_f32", k_log}, {"v_cvt_f32_u32", k_cvtfu}, {"v_cvt_f64_u32", k_cvtf64u}, {"v_mul_f64", k_mulf64}, {"v_fma_f64", k_fmaf64},
      {"v_cvt_f32_f64", k_cvtf32f64}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
      {"s_add_u32", k_salu}, {"v_fma + s_add", k_fma_salu}, {"v_readlane_b32", k_readlane}, {"ds_read_b32", k_dsread}};
  const int only = argc > 1 ? atoi(argv[1]) : 0;
  for (int w : {1, 2, 5, 8}) {
    if (only && w != only) continue;
    for (auto &t : tests) run(t.n, t.k, w, p.multiProcessorCount, out, in, clk, wallKHz);
  }
  return 0;
}
