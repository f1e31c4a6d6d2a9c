// Device-side photon tracer for gfx950 (CDNA4): the per-photon physics of
// Integrators/monteCarloRadiativeTransfer.f95 computeRT (:400-707) and the procedures it calls in the loop.
// float32 arithmetic in the reference's operator order (the build uses -ffp-contract=off), 1-based cell
// indices, so that the "step <= 0 => drop photon" escape, the 2*spacing() snaps and the periodic nudges
// (quirks Q2-Q5 of SURVEY.md 8a) behave exactly as they do on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/i3rc_hip.h"
#include "philox.hpp"

namespace i3rc {

constexpr float kTiny = FLT_MIN;
constexpr float kHuge = FLT_MAX;
constexpr float kPi   = 3.14159265358979312f;  // monteCarloRadiativeTransfer.f95:43 (rounded to real(4))

// Per-component phase-function tables (:100-105): inverse [nEntries][nInv], forward [nEntries][nFwd].
struct CompTables {
  const float *inv, *invCos, *fwd, *fwdOrig;   // invCos[k] = cos(inv[k]) (built by i3rc_hip_set_inverse_table)
  int nInv, nFwd;
};

// (member-wise copy: the source may live in the kernarg segment, address space 4, for which no implicit copy exists)
template <class CT>
__device__ __forceinline__ CompTables load_tables(const CT &c) {
  CompTables t;
  t.inv = c.inv; t.invCos = c.invCos; t.fwd = c.fwd; t.fwdOrig = c.fwdOrig; t.nInv = c.nInv; t.nFwd = c.nFwd;
  return t;
}

// Everything the kernel reads, by value in the kernarg segment (wave-uniform -> SGPRs).
struct DevProblem {
  int nx, ny, nz, ncomp;
  int xyRegular, zRegular;
  float x0, y0, z0, xMax, yMax, zMax, deltaX, deltaY, deltaZ;
  const float *xE, *yE, *zE;          // n+1 each (global; staged to LDS by every workgroup)
  const float *totalExt;              // [nz][ny][nx]
  // The same field in bricks of 32 cells = one 128-byte cache line (grids that do not fit in LDS): a ray's next cell
  // is then usually in the line it has just used, instead of nx or nx*ny floats away.  Brick edge lengths are powers
  // of two, 2^bsx x 2^bsy x 2^bsz cells; brick (X, Y, Z) starts at float 32 (X + nbx (Y + nby Z)).
  const float *extBrick;               // null: the grid is small enough to stay in L2 as it is (cheaper index)
  int bsx, bsy, bsz, nbx, nbxy;
  // ... and with it the CLEAR-AIR MAP (bricked fields only): for every footprint of 2^clearShift x 2^clearShift columns the
  // lowest and the highest layer that hold any extinction at all (one word: lo | hi << 16, 1-based; an empty footprint has
  // lo > hi), at most 4 KB, staged in LDS.  A cell outside [lo, hi] has extinction 0 -- the value the field holds -- without
  // a memory request: a third of the Landsat scene is clear air above the cloud tops, which every local-estimate ray
  // crosses on its way out, and those steps' loads were most of the field's traffic beyond L2.
  const uint32_t *clearMap;
  int clearShift, clearNx;
  // COLUMN RECORDS (round 4): a field in which every column's cells WITH extinction are one run of layers that share one value
  // -- the I3RC Landsat scene is such a field: per column a cloud of vertically uniform extinction between its base and its top --
  // is held as one 8-byte record per column {the value's bits, first layer | (run length - 1) << 16} (1-based; a clear column holds
  // the value 0).  128 x 128 columns are 128 KB where the field is 7.8 MB: the whole scene stays in every XCD's L2 (and much of it
  // in the vector L1), a cell's extinction is one load and three integer instructions, and what the field holds comes back bit
  // for bit.  null: the field has no such form (or i3rc_hip_select_grid_place asked for another place).
  const uint2 *colRec;
  // ... OVER A BASE PROFILE (round 5, GRID_COLBASE): a field that is such a column field PLUS a value per layer -- a cloud scene over a
  // horizontally uniform gas or aerosol: what a domain of several components adds up to -- totalExt(x, y, z) = base(z) + (z within the
  // column's run ? value(x, y) : 0), the float32 addition the host made when it summed the components (checked bit by bit where the
  // domain is handed over: i3rc_hip_column_records_base).  The records as above, base[nz] staged in LDS: the scene of 7.8 MB is 128 KB + 476 B.
  const float *colBase;
  const float *cumExt, *ssa;          // [ncomp][nz][ny][nx]
  const int32_t *pfIndex;             // [ncomp][nz][ny][nx]
  // TWO components (cloud + gas, cloud + aerosol): what a scattering reads of its cell in ONE 16-byte record -- the first component's
  // cumulative extinction, both single-scattering albedos, both table entries (16 bits each) -- instead of three words in three arrays
  // (three cache lines).  nullptr: no records (another number of components, an entry beyond 65535).
  const uint4 *cellRec;               // [nz][ny][nx] {cumExt_1, ssa_1, ssa_2, pfIndex_1 | pfIndex_2 << 16}
                                      // THREE components: two of them per cell, {cumExt_1, cumExt_2, ssa_1, ssa_2} {ssa_3, pfIndex_1 | pfIndex_2 << 16, pfIndex_3, -}
  const CompTables *comp;             // [ncomp] phase-function tables (device memory: indexed per lane)
  CompTables comp0;                   // tables of component 1 by value: the specialised (one-component) kernel reads them
                                      // from scalar registers, and its table loads are global_load, not flat_load
  // surface
  float albedo; int useBDRF; int nxs, nys;
  const float *xsE, *ysE, *brdf;
  // switches
  int useRayTracing, useRR, nDir, useHybrid, numOrdersOrig, useRRI, limitContrib;
  float zetaMin, maxContrib, maxExt;
  const float *dirCos;                // [nDir][3] (global; staged to LDS)
  // single-component shortcuts (specialised kernel): a value every cell shares is passed in the kernarg segment
  float uniformSsa;                   // >= 0: ssa of every cell of component 1; < 0: per-cell values differ
  int uniformPf;                      // >= 1: phaseFunctionIndex of every cell of component 1; 0: per-cell values differ
  // tallies (float64, packed; offsets in elements: i3rc_hip_create keeps the whole buffer below 2^31 elements)
  double *tally;
  int oUp, oDown, oAbs, oVol, oInt, oExc, oCnt;
  int ldsTallies;                     // 1: fluxUp / fluxDown privatised in LDS (ncol small)
  int ldsVolume;                      // 1: volumeAbsorption privatised in LDS (an absorbing domain of few cells)
  int ldsGrid;                        // 1: totalExt staged in LDS
  int ldsIntensity;                   // 1: intensityByComponent privatised in LDS ((ncomp+1)*nDir*ncol small)
  int rayQueueCap;                    // radiance runs: events a wave's ray queue holds (power of two; kRecWords floats each)
};

// XCD-aware photon order (fields beyond an XCD's L2): the photons of a launch sorted by the eighth of the domain (in y)
// they start in -- slabIds holds photon numbers relative to firstPhoton, slab after slab; see slab_count_kernel
struct SlabMeta { unsigned count[8], offset[8], fill[8], take[8]; };
struct RunArgs {
  uint32_t seed0, seed1;
  long long firstPhoton, nPhotons;
  unsigned long long *workCounter;    // device word, zeroed before launch
  const uint32_t *slabIds;            // null: photons are handed out in index order
  SlabMeta *slabMeta;
  int srcKind; float solarMu, solarPhi;
  float solarDx, solarDy, solarDz;    // makeDirectionCosines(solarMu, solarPhi) evaluated once on the host
  int chunk;                          // photon indices a wave reserves per visit of the work counter
  const float *sx, *sy, *sz, *smu, *sphi;   // explicit stream (device)
  // Fused multi-batch launch (PhiloxBatchStream kernels; i3rc_hip_run_batches): nBatches batches of a driver's loop in
  // ONE grid.  Batch b has the key (seed0, seed1 + b) and the photons firstPhoton .. firstPhoton + nPhotons - 1; the work
  // counter counts CHUNKS (chunk c = photons [j chunk, (j + 1) chunk) of batch c / chunksPerBatch, j = c mod chunksPerBatch:
  // a chunk never straddles two batches).  Tallies of batch b go to block b * replicas + r of the tally buffer (blocks of
  // blockStride float64 in the handle's tally layout), r = workgroup number mod replicas: a small domain's few hot
  // words take 4e8 float64 atomics/s in one block and 1.3e10/s in 64 of them (tools/microbench/atomic_rate.hip).
  unsigned nBatches, chunksPerBatch;
  int replicas;
  long long blockStride;
  // ... and its work counters to counterBlocks[(b * kCounterReplicas + workgroup mod kCounterReplicas) * I3RC_NUM_COUNTERS + k]:
  // a batch's counters are one cache line, and one line takes about 1e8 atomics/s -- with the counters in the tally block
  // itself (one replica for large domains) every batch of 1e6 Landsat photons paid 0.4 ms for them
  double *counterBlocks;
  const int *abortFlag;               // host-coherent word: once non-zero, waves take no further chunks (a discarded look-ahead)
  // replay
  const float *randoms; long long nRandoms; const long long *drawStart;
  int32_t *fate, *fateColumn; float *fateWeight; int32_t *fateOrder, *drawsUsed;
};

// LDS carve-up shared by all device functions.  The pointers carry the LDS address space in their type, so every
// access is a ds_* instruction even where the same value may come from LDS or from global memory (a generic pointer
// there ends in a flat_load of a selected address).
typedef __attribute__((address_space(3))) float lds_float;
// A workgroup's partial tally sums in LDS are FLOAT64 (round 5: ds_add_f64), as everything else a tally passes through: whichever
// way a batch is launched -- plain, fused, ahead of the calls, sharded -- its tallies differ only by the ORDER of float64 additions
// (1e-12 of a column's sum at most).  Until round 4 they were float32 and a plain launch differed from a fused one by up to 2e-5 of a
// column's absorption (some 1e4 float32 additions per column and workgroup), which the tests had to allow for -- and which would
// have hidden a small real difference.  -DI3RC_LDS_F32 is the old form (measurement knob, tools/README.md).
#ifdef I3RC_LDS_F32
typedef float tally_t;
#else
typedef double tally_t;
#endif
typedef __attribute__((address_space(3))) tally_t lds_tally;
constexpr int kTallyWords = (int)(sizeof(tally_t) / sizeof(float));
struct Lds {
  lds_float *xE, *yE, *zE;    // edges
  lds_tally *tUp, *tDown;     // privatised flux tallies (valid when ldsTallies)
  lds_tally *tVol;            // privatised volumeAbsorption (valid when ldsVolume)
  lds_float *ext;             // totalExt copy (valid when ldsGrid); bricked fields: the clear-air map (DevProblem::clearMap) as words; GRID_COLBASE: the base profile
  lds_float *dirCos;          // intensity directions
  lds_float *dirTab;          // ... and, per direction, what a ray of that direction derives from it (Ray::set_direction), 16 words: see photon_kernel
  lds_tally *tInt;            // privatised intensityByComponent (valid when ldsIntensity)
  lds_float *cosTab;          // the inverse table's cosines (one entry) where a kernel keeps them in LDS (photon_kernel, TBL)
  lds_float *queue;           // [waves][kRecWords][rayQueueCap]: every wave's ring of local-estimate events (see kernels.hpp)
};
__device__ __forceinline__ void lds_add(lds_tally *p, float v) { (void)__hip_atomic_fetch_add(p, (tally_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_add_f64 (ds_add_f32 with I3RC_LDS_F32)
// One record of a wave's ray queue: what the D local-estimate (shadow) rays of one scattering / reflection event need.
constexpr int kRecWords = 14;   // x y z | ix iy iz | weight | incoming direction (3) | info | photon id (2) | Philox block
// ... and one ready-made shadow ray of the wave's ready buffer (kReadyRays of them, one expand phase's worth)
constexpr int kReadyWords = 12;  // x y z | ix iy iz | component, direction, stage | weight | phase-function factor | free path | target | optical path so far
constexpr int kReadyRays = 64;
#ifndef I3RC_DIRECT_READY
#define I3RC_DIRECT_READY 128
#endif
constexpr int kDirectReady = I3RC_DIRECT_READY;   // ... of the one-direction radiance kernels, which have no event ring (photon_kernel, DIRECT): a power of two >= 128
constexpr int kCounterReplicas = 64;   // fused multi-batch launches: copies of a batch's counter block (RunArgs::counterBlocks)

// The carve-up of a workgroup's dynamic LDS, in WORDS from its start -- ONE function for both sides: photon_kernel sets its
// pointers (Lds) from it, the host sizes the launch's allocation from it (i3rc_hip.hip, lds_bytes).  (Round 4 kept two copies of
// this arithmetic and the advisor found them three floats apart: a 5 x 5 x 5 grid in LDS ended one word past its allocation.)
//   queues        radiance kernels with ray queues (INTENSITY && !Rng::kReplay): dirTab (16-byte aligned) and the waves' rings / ready stores
//   direct        ... of the one-direction form (photon_kernel, DIRECT)
//   grid          GridPlace of the instantiation;  intensity: its INTENSITY (a bricked field's clear-air map: flux kernels only)
//   waves         waves per workgroup;  tableWords: the inverse table's cosines behind everything else (TBL), else 0
struct LdsPlan { int xE, yE, zE, tallies, dirCos, dirTab, queue, tInt, ext, cosTab, end, tVol; };
template <class PR>
__host__ __device__ __attribute__((always_inline)) inline LdsPlan lds_plan(const PR &P, bool queues, bool direct, int grid, bool intensity, int waves, int tableWords) {
  LdsPlan o;
  int p = 0;
  o.xE = p; p += P.nx + 1;
  o.yE = p; p += P.ny + 1;
  o.zE = p; p += P.nz + 1;
  const int ncol = P.nx * P.ny;
  if (P.ldsTallies) p = (p + kTallyWords - 1) & ~(kTallyWords - 1);
  o.tallies = p;                               // fluxUp | fluxDown, ncol tally_t each (valid when ldsTallies)
  if (P.ldsTallies) p += 2 * ncol * kTallyWords;
  if (P.ldsVolume) p = (p + kTallyWords - 1) & ~(kTallyWords - 1);
  o.tVol = p;                                  // volumeAbsorption, ncol * nz tally_t (valid when ldsVolume: small absorbing domains)
  if (P.ldsVolume) p += ncol * P.nz * kTallyWords;
  o.dirCos = p; p += 3 * P.nDir;
  if (queues) p = (p + 3) & ~3;                // (the 128-bit reads of dirTab)
  o.dirTab = p;
  if (queues) p += 16 * P.nDir;
  o.queue = p;
  if (queues) p += waves * (kRecWords * P.rayQueueCap + kReadyWords * (direct ? kDirectReady : kReadyRays));
  if (P.ldsIntensity) p = (p + kTallyWords - 1) & ~(kTallyWords - 1);
  o.tInt = p;
  if (P.ldsIntensity) p += (P.ncomp + 1) * P.nDir * ncol * kTallyWords;
  o.ext = p;
  if (grid == 0 /* GRID_LDS */) p += ncol * P.nz;
  if (grid == 2 /* GRID_BRICKS */ && !intensity) p += P.clearNx * (((P.ny - 1) >> P.clearShift) + 1);   // (the clear-air map lives at Lds::ext)
  if (grid == 4 /* GRID_COLBASE */) p += P.nz;                                                           // (the base profile of the column records, likewise)
  o.cosTab = p; p += tableWords;
  o.end = p;
  return o;
}

// Fortran SPACING() for real(4)
__device__ __forceinline__ float spacingf(float x) {
  const uint32_t e = __float_as_uint(x) & 0x7f800000u;
  return e > (23u << 23) ? __uint_as_float(e - (23u << 23)) : kTiny;
}

// 2 * spacing(x), exactly: spacing() is a power of two (exponent field E - 23, never below tiny = field 1), so twice it is
// the field max(E, 24) - 22 -- three integer instructions instead of the five of 2.0f * spacingf(x)
__device__ __forceinline__ float two_spacingf(float x) {
  const uint32_t e = __float_as_uint(x) & 0x7f800000u;
  return __uint_as_float(max(e, 24u << 23) - (22u << 23));
}

// ---- correctly rounded float32 divide / sqrt from the hardware approximations --------------------------------
// The reference divides by the same direction cosine at every voxel step of a trace.  v_rcp_f32 (1 ulp) refined
// once gives r1 ~ 1/d; n/d is then q0 = n*r1 followed by two fused residual corrections -- the tail of the IEEE
// division expansion hipcc emits, minus its per-call reciprocal and range scaling: 5 FMA-class instructions instead
// of ~13, same correctly rounded quotient (bit-identity with `/` is checked on the device by
// tests/test_gpu_parity.py::test_exact_arithmetic_helpers and by the bit-exact tracer tests).  Valid for
// 1e-20 <= |d| and results far from overflow, which callers guarantee (else they use `/`).
__device__ __forceinline__ float refined_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float exact_div(float n, float d, float r1) {
  const float q0 = n * r1;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
  return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}
// v_sqrt_f32 is within 1 ulp: pick the neighbour with the right residual sign (the correction LLVM uses for
// correctly rounded f32 sqrt, without the denormal rescaling: arguments here are 0 or normal).
__device__ __forceinline__ float exact_sqrt(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sDn = __uint_as_float(__float_as_uint(s) - 1u), sUp = __uint_as_float(__float_as_uint(s) + 1u);
  const float eDn = __builtin_fmaf(-sDn, s, x), eUp = __builtin_fmaf(-sUp, s, x);
  s = eDn <= 0.0f ? sDn : s;
  s = eUp > 0.0f ? sUp : s;
  return x == 0.0f ? 0.0f : s;
}

// log() of a deviate in (0, 1] for the optical depth to travel (:480).
#ifdef I3RC_FAST_LOG
// hardware log2 (1 ulp) times ln 2: within 2 ulp of the correctly rounded value, a fifth of the instructions
__device__ __forceinline__ float sample_log(float u) { return __builtin_amdgcn_logf(u) * 0.693147180559945309f; }
#else
__device__ __forceinline__ float sample_log(float u) { return logf(u); }
#endif

// findIndex, Code/numericUtilities.f95:195-248 (1-based table; firstGuess <= 0: absent)
template <class Tab>
__device__ __forceinline__ int find_index(float value, Tab T, int n, int firstGuess) {
  int lower, upper;
  if (firstGuess > 0) {
    lower = firstGuess;
    int inc = 1;
    for (;;) {
      upper = min(lower + inc, n);
      if (lower == n || (T(lower) <= value && T(upper) > value)) break;
      if (T(lower) > value) { upper = lower; lower = max(upper - inc, 1); }
      else lower = upper;
      inc *= 2;
    }
  } else { lower = 0; upper = n; }
  for (;;) {
    if (lower == n || upper <= lower + 1) break;
    const int mid = (lower + upper) / 2;
    if (value >= T(mid)) lower = mid; else upper = mid;
  }
  return lower;
}

// makePeriodic :2063-2082
__device__ __forceinline__ float make_periodic(float a, float aMin, float aMax) {
  for (;;) {
    if (a <= aMax && a > aMin) break;
    float b;
    if (a > aMax) b = a - (aMax - aMin);
    else if (a == aMin) b = aMax;
    else b = a + (aMax - aMin);
    // no progress (NaN, infinity, or more than 2^24 widths away): the reference's loop never ends; see the oracle
    if (b == a || b != b) return aMax;
    a = b;
  }
  return a;
}

// makeDirectionCosines :2041-2059
__device__ __forceinline__ void make_dircos(float mu, float phi, float &dx, float &dy, float &dz) {
  const float sinTheta = exact_sqrt(1.0f - mu * mu);
  const float c = cosf(phi), s = sinf(phi);
  dx = sinTheta * c; dy = sinTheta * s; dz = mu;
}

// LDS byte address of an LDS pointer and back (LDS pointers are 32 bits wide)
__device__ __forceinline__ int lds_address(const lds_float *p) { return (int)(__UINTPTR_TYPE__)p; }
__device__ __forceinline__ float lds_read(int byteAddress) { return *(const lds_float *)(__UINTPTR_TYPE__)(unsigned)byteAddress; }

struct Ray {
  float x, y, z;
  float dx, dy, dz;
  float rx, ry, rz;   // refined reciprocals of the direction cosines (set_direction)
  int slow;           // 1: some |direction cosine| < 1e-20 -> plain IEEE division (and the 2*tiny test) per step
  int ix, iy, iz;
  float acc, target;
  // what a trace needs of its direction at every voxel step, worked out once per trace (set_direction):
  int ex, ey, ez;     // LDS byte address of edge 0 of the face the ray moves towards: face coordinate = [e + 4 index]
  int cx, cy, cz;     // cell increment (+1 / -1) along each axis
  float nudge;        // 2 cx as a float: the periodic wrap moves the position by nudge * spacing() (:1774-1788)
  __device__ __forceinline__ void set_direction(const Lds &L) {
    slow = fminf(fminf(fabsf(dx), fabsf(dy)), fabsf(dz)) < 1e-20f;
    rx = refined_rcp(dx); ry = refined_rcp(dy); rz = refined_rcp(dz);
    const bool px = dx >= 0.0f, py = dy >= 0.0f, pz = dz >= 0.0f;
    // edges are stored from index 0; cell i (1-based) lies between edge i - 1 and edge i
    ex = lds_address(L.xE) - (px ? 0 : 4); ey = lds_address(L.yE) - (py ? 0 : 4); ez = lds_address(L.zE) - (pz ? 0 : 4);
    cx = px ? 1 : -1; cy = py ? 1 : -1; cz = pz ? 1 : -1;
    nudge = px ? 2.0f : -2.0f;
  }
  // the same from a radiance direction's entry of Lds::dirTab (written once per workgroup by store_direction): four 128-bit LDS
  // reads instead of three reciprocals with their refinements and a dozen selects at every start of a local-estimate ray
  __device__ __forceinline__ void store_direction(lds_float *t) const {
    t[0] = dx; t[1] = dy; t[2] = dz; t[3] = nudge; t[4] = rx; t[5] = ry; t[6] = rz; t[7] = __int_as_float(slow);
    t[8] = __int_as_float(ex); t[9] = __int_as_float(ey); t[10] = __int_as_float(ez); t[11] = __int_as_float(cx);
    t[12] = __int_as_float(cy); t[13] = __int_as_float(cz); t[14] = 0.0f; t[15] = 0.0f;
  }
  __device__ __forceinline__ void load_direction(const lds_float *t) {
    typedef float vec4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const vec4 lds_vec4;
    const vec4 a = ((lds_vec4 *)t)[0], b = ((lds_vec4 *)t)[1], c = ((lds_vec4 *)t)[2], d = ((lds_vec4 *)t)[3];
    dx = a.x; dy = a.y; dz = a.z; nudge = a.w; rx = b.x; ry = b.y; rz = b.z; slow = __float_as_int(b.w);
    ex = __float_as_int(c.x); ey = __float_as_int(c.y); ez = __float_as_int(c.z); cx = __float_as_int(c.w);
    cy = __float_as_int(d.x); cz = __float_as_int(d.y);
  }
};

enum StepResult { STEP_CONTINUE = 0, STEP_DONE = 1, STEP_ERROR = 2, STEP_EXIT = 3 };   // DONE: target reached; EXIT: left through the top or the bottom

// (PR: DevProblem -- the kernel argument, held in scalar registers -- or the same struct read through a pointer into the
// kernarg segment, address space 4: see cold_args in kernels.hpp)
template <class PR>
__device__ __forceinline__ int cell_index(const PR &P, int ix, int iy, int iz) {
  // 24-bit multiplies (full rate): i3rc_hip_create guarantees nx*ny < 2^24 and nx*ny*nz < 2^30
  return (int)(__umul24((unsigned)(iz - 1), (unsigned)(P.nx * P.ny)) + __umul24((unsigned)(iy - 1), (unsigned)P.nx)) + (ix - 1);
}

// position of cell (ix, iy, iz) (1-based) in the bricked copy of the extinction field
template <class PR>
__device__ __forceinline__ int brick_index(const PR &P, int ix, int iy, int iz) {
  const unsigned x = (unsigned)(ix - 1), y = (unsigned)(iy - 1), z = (unsigned)(iz - 1);
  const unsigned brick = __umul24(z >> P.bsz, (unsigned)P.nbxy) + __umul24(y >> P.bsy, (unsigned)P.nbx) + (x >> P.bsx);
  const unsigned mx = (1u << P.bsx) - 1u, my = (1u << P.bsy) - 1u, mz = (1u << P.bsz) - 1u;
  const unsigned within = ((z & mz) << (P.bsx + P.bsy)) | ((y & my) << P.bsx) | (x & mx);
  return (int)((brick << 5) | within);
}
// extinction of the cell a ray is in: LDS copy when the grid fits; else global memory -- the plain field while it
// fits in an XCD's L2, the bricked copy beyond that (measured: Landsat 128x128x119, 7.8 MB, +29 % with bricks, fabric
// traffic 18 KB -> per photon; radar 138 KB and Landsat-36 2.4 MB are 8-11 % faster without the longer index)
// GRID says where, at COMPILE time: every kernel is instantiated once per place.  That takes the choice and its scalar
// registers out of the voxel-step loop -- and it is a matter of correctness, not only of speed: with a run-time flag
// tested inside the shadow-ray loops of the nested local estimate, the compiler (ROCm 7.2 LLVM, -O1 and -O3 alike)
// materialised the flag's negation as a LANE MASK under the exec mask of one loop (v_cndmask 0/1 + v_cmp_ne) and
// reused it in the next loop, whose active lanes were not all active there: those lanes took the LDS branch on a
// grid that lives in global memory and read zeros (found by the replay tests on the I3RC radar / Landsat fields;
// tests/test_build_isa.py keeps the pattern out of the kernels).
enum GridPlace { GRID_LDS = 0, GRID_GLOBAL = 1, GRID_BRICKS = 2, GRID_COLUMNS = 3, GRID_COLBASE = 4 };
// CLEARMAP: consult the clear-air map of a bricked field first (DevProblem::clearMap).  The flux kernels do -- Landsat-119 6.6 ->
// 7.2e8 photons/s, the scene tiled 2 x 2 5.5 -> 6.0e8 --; the radiance kernels do not: the look-up is an LDS read in front of
// every load, and with five waves per SIMD and no register to carry it a step ahead it cost them 11 %.
template <int GRID, bool CLEARMAP = false, class PR>
__device__ __forceinline__ float cell_extinction(const PR &P, const Lds &L, int ix, int iy, int iz) {
  if (GRID == GRID_LDS) return L.ext[cell_index(P, ix, iy, iz)];   // ds_read
  if (GRID == GRID_COLUMNS) {   // the column's record (DevProblem::colRec): its value within the run of layers, 0 outside
    // (ONE 8-byte buffer load: written as a plain load of a uint2, the compiler loaded the range word first, waited for it, and
    // fetched the value under a branch only for the lanes inside their run -- two memory latencies in a row at every voxel step,
    // Landsat-36 15 % slower than with the plain field.  The buffer descriptor is three scalar instructions from the pointer.)
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.colRec, 0, (P.nx * P.ny) << 3, 0x00020000);
    const u32x2 rec = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((__umul24((unsigned)(iy - 1), (unsigned)P.nx) + (unsigned)(ix - 1)) << 3), 0, 0);
    return ((unsigned)iz - (rec.y & 0xffffu)) <= (rec.y >> 16) ? __uint_as_float(rec.x) : 0.0f;
  }
  if (GRID == GRID_COLBASE) {   // ... over a base profile (DevProblem::colBase, in LDS at Lds::ext): the host's own float32 addition
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.colRec, 0, (P.nx * P.ny) << 3, 0x00020000);
    const u32x2 rec = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((__umul24((unsigned)(iy - 1), (unsigned)P.nx) + (unsigned)(ix - 1)) << 3), 0, 0);
    const float base = L.ext[iz - 1];
    return base + (((unsigned)iz - (rec.y & 0xffffu)) <= (rec.y >> 16) ? __uint_as_float(rec.x) : 0.0f);
  }
  if (GRID == GRID_BRICKS && CLEARMAP) {
    const uint32_t range = __float_as_uint(L.ext[__umul24((unsigned)(iy - 1) >> P.clearShift, (unsigned)P.clearNx) + ((unsigned)(ix - 1) >> P.clearShift)]);
    const unsigned z = (unsigned)iz;
    if (z < (range & 0xffffu) || z > (range >> 16)) return 0.0f;   // clear air: what the field holds there, without asking for it
    return P.extBrick[brick_index(P, ix, iy, iz)];
  }
  if (GRID == GRID_BRICKS) return P.extBrick[brick_index(P, ix, iy, iz)];
  return P.totalExt[cell_index(P, ix, iy, iz)];
}


// One iteration of accumulationLoop, accumulateExtinctionAlongPath :1690-1806.  hasTarget == false: trace to the
// boundary.  Written as straight-line predicated code (selects, no data-dependent branches except the two rare
// escapes): on a 64-lane wavefront the lanes take the reference's if/else arms in every combination at every step,
// so branches would execute both arms anyway and add exec-mask bookkeeping on top.  The arithmetic of each arm is
// exactly the reference's (checked bit for bit against the oracle by the tracer tests).
//
// LAZY ARRIVAL (round 4).  The step in which the optical path reaches its target (:1721-1731) ends the trace inside the cell:
// the reference advances the position by (target - path so far) / extinction -- a division by a value that changes from cell to
// cell: a reciprocal (quarter rate), its refinement and the five operations of exact_div, nine vector instructions that every
// lane of every step paid for although a trace arrives once.  trace_step_lazy leaves the position where it is and returns
// STEP_DONE with   r.acc = target - path so far,   r.target = -extinction   (negative: the mark of an arrival not yet finished;
// a reached cell has extinction > 0, and targets are never negative); finish_arrival does the division and the advance -- the
// same operations on the same values -- where the arrival is dealt with: the event phase for photons, the service phase for
// a local-estimate ray that goes on to its second leg (a ray that ends at its target needs no position at all).
// trace_step is the two together (nested local estimate, the tracer test hook).
// (BRANCHY: the general kernels keep the guarded division in a branch of its own -- the form below cost their radiance instantiations,
// which sit at the 168 registers of three waves per SIMD, four spilled vector registers)
// (SHORT: the three proximity tests behind exec masks of their own -- see below; the photon steps of the ring kernels, whose 96 registers
// the other form overran by two)
template <int GRID, bool CLEARMAP = false, bool BRANCHY = false, bool SHORT = false, class PR>
__device__ __forceinline__ StepResult trace_step_lazy(const PR &P, const Lds &L, Ray &r, bool hasTarget) {
  // the extinction of the current cell is requested first: its latency (LDS, or L2 / HBM for grids that do not fit
  // in LDS) is covered by the three face-distance divisions below
  const float ext = cell_extinction<GRID, CLEARMAP>(P, L, r.ix, r.iy, r.iz);
  const int cx = r.cx, cy = r.cy, cz = r.cz;
  const float ex = lds_read(r.ex + (r.ix << 2)), ey = lds_read(r.ey + (r.iy << 2)), ez = lds_read(r.ez + (r.iz << 2));
  float stx, sty, stz;
  if (BRANCHY) {
    if (__builtin_expect(r.slow, 0)) {   // a direction cosine of (almost) zero: the reference's guarded division
      stx = fabsf(r.dx) >= 2.0f * kTiny ? (ex - r.x) / r.dx : kHuge;
      sty = fabsf(r.dy) >= 2.0f * kTiny ? (ey - r.y) / r.dy : kHuge;
      stz = fabsf(r.dz) >= 2.0f * kTiny ? (ez - r.z) / r.dz : kHuge;
    } else {
      stx = exact_div(ex - r.x, r.dx, r.rx);
      sty = exact_div(ey - r.y, r.dy, r.ry);
      stz = exact_div(ez - r.z, r.dz, r.rz);
    }
  } else {
  // (every lane, every axis: on an axis whose cosine is below 1e-20 the quotient is rubbish and is replaced below)
  stx = exact_div(ex - r.x, r.dx, r.rx); sty = exact_div(ey - r.y, r.dy, r.ry); stz = exact_div(ez - r.z, r.dz, r.rz);
  // A direction cosine of (almost) zero: the reference's guarded division (:1697-1704: a face is never reached along an axis whose
  // |cosine| is below 2 tiny).  Such lanes are the rule, not the exception -- a sun at the zenith gives every photon dx = dy = 0
  // until its first scattering, a nadir radiance direction every ray of it -- so that in nine voxel-step phases of ten SOME lane
  // of the wave is one: a branch of its own for them (three guarded IEEE divisions behind three exec masks, then the fast path
  // for the others) was twenty vector and as many scalar instructions on top of every such phase.  Now: one uniform test, three
  // selects for the lanes concerned; only cosines between 2 tiny and 1e-20 (none in practice) still take the IEEE division.
  if (__ballot(r.slow != 0) != 0ull) {
    const float adx = fabsf(r.dx), ady = fabsf(r.dy), adz = fabsf(r.dz);
    const bool tx = adx < 1e-20f, ty = ady < 1e-20f, tz = adz < 1e-20f;
    stx = tx ? kHuge : stx; sty = ty ? kHuge : sty; stz = tz ? kHuge : stz;
    const bool mx = tx && adx >= 2.0f * kTiny, my = ty && ady >= 2.0f * kTiny, mz = tz && adz >= 2.0f * kTiny;
    if (__builtin_expect(__ballot(mx || my || mz) != 0ull, 0)) {
      if (mx) stx = (ex - r.x) / r.dx;
      if (my) sty = (ey - r.y) / r.dy;
      if (mz) stz = (ez - r.z) / r.dz;
    }
  }
  }
  float step = stx;
  step = sty < step ? sty : step;
  step = stz < step ? stz : step;
  // :1711-1714 `if(thisStep <= 0.)`, written so that a NaN step (a NaN direction or position) ends the trace as well:
  // the reference's comparison lets NaN through and its loop never ends
  if (__builtin_expect(!(step > 0.0f), 0)) { r.acc = -2.0f; return STEP_ERROR; }

  const float tauCell = step * ext;
  bool reach = false;
  if (hasTarget) reach = r.acc + tauCell > r.target;                  // :1721-1731
  const float adv = reach ? 0.0f : step;                               // (an arrival stays where it is: finish_arrival)
  const float rest = r.target - r.acc;
  r.target = reach ? -ext : r.target;
  r.acc = reach ? rest : r.acc + tauCell;

  const float ax = r.x + adv * r.dx, ay = r.y + adv * r.dy, az = r.z + adv * r.dz;
  const bool hx = !reach && stx <= step, hy = !reach && sty <= step, hz = !reach && stz <= step;   // face reached
  // :1744-1769 the face position itself when the face is reached, else the advanced position; the index moves on
  // when the face is reached or the position ends within 2 spacing() of it
  bool bx, by, bz;
  if (SHORT) {   // (the three tests behind exec masks of their own, as before round 4)
    bx = hx || (!reach && fabsf(ex - ax) <= two_spacingf(ax));
    by = hy || (!reach && fabsf(ey - ay) <= two_spacingf(ay));
    bz = hz || (!reach && fabsf(ez - az) <= two_spacingf(az));
  } else {
  // (bitwise, not short-circuit: every lane of the step makes the three comparisons -- nearly every lane needs two of them anyway --
  // instead of three exec-mask regions of four scalar instructions each: the scalar unit, one per compute unit, issues 0.6 ... 0.7
  // instructions per cycle in these kernels and is as busy as the vector units)
  const bool nx_ = fabsf(ex - ax) <= two_spacingf(ax), ny_ = fabsf(ey - ay) <= two_spacingf(ay), nz_ = fabsf(ez - az) <= two_spacingf(az);
  bx = hx | (!reach & nx_); by = hy | (!reach & ny_); bz = hz | (!reach & nz_);
  }
  r.x = hx ? ex : ax; r.y = hy ? ey : ay; r.z = hz ? ez : az;
  r.ix += bx ? cx : 0; r.iy += by ? cy : 0; r.iz += bz ? cz : 0;

  // periodic wrap :1774-1788 (y uses x's sign, as the reference does).  Few steps cross the domain's side walls: the
  // wrap is skipped by the whole wave when no lane needs it (a uniform branch on a ballot: two scalar instructions)
  // (the wave-uniform test as two compares whose lane masks are OR-ed in scalar registers: a ballot of the four conditions below
  // came out as four compares, a select of 0 / 1 and a fifth compare)
  if ((__builtin_amdgcn_uicmp((unsigned)(r.ix - 1), (unsigned)P.nx, 35 /* unsigned >= */) |
       __builtin_amdgcn_uicmp((unsigned)(r.iy - 1), (unsigned)P.ny, 35)) != 0ull) {
    const bool xLo = r.ix <= 0, xHi = r.ix >= P.nx + 1, yLo = r.iy <= 0, yHi = r.iy >= P.ny + 1;
    const float nudge = r.nudge;   // 2 cellIncrement(1) as a float: the nudges are +- 2 spacing()
    const float sxp = copysignf(two_spacingf(r.x), nudge), syp = copysignf(two_spacingf(r.y), nudge);
    r.x = xLo ? P.xMax + sxp : (xHi ? P.x0 + sxp : r.x);
    r.y = yLo ? P.yMax + syp : (yHi ? P.y0 + syp : r.y);
    r.ix = xLo ? P.nx : (xHi ? 1 : r.ix);
    r.iy = yLo ? P.ny : (yHi ? 1 : r.iy);
  }

  // :1793-1804: a ray that has left through the top or the bottom stands at zMax + 2 spacing(zMax) or at z0 -- also LAZY: the
  // trace is over, and the height is put there by whoever needs it (finish_exit: the event phase for photons; a local-estimate
  // ray's end only asks for its layer index)
  const bool top = r.iz > P.nz, bottom = r.iz < 1;
  return reach ? STEP_DONE : ((top || bottom) ? STEP_EXIT : STEP_CONTINUE);
}
// the height of a ray that has left the grid :1793-1804 (a ray inside keeps its own)
template <class PR>
__device__ __forceinline__ void finish_exit(const PR &P, Ray &r) {
  r.z = r.iz > P.nz ? P.zMax + two_spacingf(P.zMax) : (r.iz < 1 ? P.z0 : r.z);
}
// has the trace arrived at its target without its last advance (trace_step_lazy)?
__device__ __forceinline__ bool arrival_pending(const Ray &r) { return r.target < 0.0f; }
// the advance of the arriving step :1724-1731, :1736-1738 (the position ends inside the cell: no face, no index, no wrap)
__device__ __forceinline__ void finish_arrival(Ray &r) {
  const float ext = -r.target, rest = r.acc;
  float part = exact_div(rest, ext, refined_rcp(ext));
  if (__builtin_expect(ext < 1e-20f, 0)) part = rest / ext;
  r.x = r.x + part * r.dx; r.y = r.y + part * r.dy; r.z = r.z + part * r.dz;
}
// the reference's step: an arrival finished at once, the optical path at its target
template <int GRID, bool CLEARMAP = false, class PR>
__device__ __forceinline__ StepResult trace_step(const PR &P, const Lds &L, Ray &r, bool hasTarget) {
  const float target = r.target;
  const StepResult s = trace_step_lazy<GRID, CLEARMAP, false, false>(P, L, r, hasTarget);
  if (s == STEP_DONE) { finish_arrival(r); r.acc = target; r.target = target; }
  if (s == STEP_EXIT) finish_exit(P, r);
  return s;
}

// findXYIndicies :1353-1374, findZIndex :1376-1388
template <bool GENERAL = true, class PR>
__device__ __forceinline__ void find_xy(const PR &P, const Lds &L, float x, float y, int &ix, int &iy) {
  if (!GENERAL || P.xyRegular) {
    int i = min((int)((x - P.x0) / P.deltaX) + 1, P.nx);
    int j = min((int)((y - P.y0) / P.deltaY) + 1, P.ny);
    if (fabsf(L.xE[i] - x) < spacingf(x)) i = i + 1;
    if (fabsf(L.yE[j] - y) < spacingf(y)) j = j + 1;
    if (i == P.nx + 1) i = 1;
    if (j == P.ny + 1) j = 1;
    ix = i; iy = j;
  } else {
    const lds_float *xe = L.xE, *ye = L.yE;
    ix = find_index(x, [xe](int k) { return xe[k - 1]; }, P.nx + 1, ix);
    iy = find_index(y, [ye](int k) { return ye[k - 1]; }, P.ny + 1, iy);
    // x == xMax exactly (makePeriodic lets it through) gives nx + 1: the reference's regular branch wraps that to 1
    // (:1366-1367), its irregular branch does not and then indexes out of bounds; here both wrap
    if (ix == P.nx + 1) ix = 1;
    if (iy == P.ny + 1) iy = 1;
  }
}
template <bool GENERAL = true, class PR>
__device__ __forceinline__ void find_z(const PR &P, const Lds &L, float z, int &iz) {
  if (!GENERAL || P.zRegular) {
    int k = min((int)((z - P.z0) / P.deltaZ) + 1, P.nz);
    if (fabsf(L.zE[k] - z) < spacingf(z)) k = k + 1;
    iz = k;
  } else {
    const lds_float *ze = L.zE;
    iz = find_index(z, [ze](int k) { return ze[k - 1]; }, P.nz + 1, iz);
  }
}

// computeScatteringAngle :1390-1417 (quirk Q1: `left` is not rescaled by n) followed by cos(angle) (:687).
// The table holds cos(T(k)) (float64 cosine of the reference's float32 angle table, rounded once).  Because the
// interpolation weight `left` is < 1/n, cos((1-left) T(k) + left T(k+1)) and (1-left) cos T(k) + left cos T(k+1)
// differ by < left * dT^2 / 2 ~ 1e-11, far below one float32 ulp: the same cosine without a 57-op cosf per event.
// EXACT = false (production streams): (k - 1) / n by reciprocal -- `left` only weights the two neighbouring entries.
template <bool EXACT = true, class Tab = const float *>
__device__ __forceinline__ float scattering_cosine(float r, Tab cosTab, int n, float rcpN) {
  const int k = (int)(r * (float)n) + 1;
  if (k < n) {
    const float left = r - (EXACT ? exact_div((float)(k - 1), (float)n, rcpN) : (float)(k - 1) * rcpN);
    return (1.0f - left) * cosTab[k - 1] + left * cosTab[k];
  }
  return cosTab[n - 1];
}

// next_direct :2086-2113
// The production streams take the azimuth from the hardware sine / cosine (1e-7): there the square root and the two
// divisions use the hardware approximations as well (same order of error; the direction's norm is renewed by d at every
// scattering either way).  The replay stream keeps the correctly rounded operations of the reference's arithmetic.
template <class Rng>
__device__ __forceinline__ void next_direct(Rng &rng, float cosS, float &s0, float &s1, float &s2) {
  float d, ax, ay;
  rng.disc_point(ax, ay, d);   // a point of the unit disc (any radius): only its azimuth matters
  float b = Rng::kReplay ? exact_sqrt(exact_div(1.0f - cosS * cosS, d, refined_rcp(d)))
                         : __builtin_amdgcn_sqrtf((1.0f - cosS * cosS) * __builtin_amdgcn_rcpf(d));
  ax = ax * b;
  ay = ay * b;
  b = s0 * ax - s1 * ay;
  const float den = 1.0f + fabsf(s2);
  d = cosS - (Rng::kReplay ? exact_div(b, den, refined_rcp(den)) : b * __builtin_amdgcn_rcpf(den));
  s0 = s0 * d + ax;
  s1 = s1 * d - ay;
  s2 = s2 * cosS - copysignf(fabsf(b), s2 * b);
}

// lookUpPhaseFuncValsFromTable :1613-1652
__device__ __forceinline__ float lookup_phase(const float *tab, int n, float angle) {
  const float dTheta = kPi / (float)(n - 1);
  const int k = (int)(angle / dTheta) + 1;
  if (k < n) {
    const float w = 1.0f - (angle - (float)(k - 1) * dTheta) / dTheta;
    return w * tab[k - 1] + (1.0f - w) * tab[k];
  }
  return tab[n - 1];
}

// Hardware log / exp / reciprocal (v_log_f32, v_exp_f32, v_rcp_f32: 1-2 ulp) for the WEIGHTS of local-estimate rays
// (phase-function factor, transmission, roulette bounds): a Monte Carlo estimate's weights, not photon trajectories.
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// (measurement knob I3RC_FAST_ACOS, round 5: the local estimate's scattering angle (:1488-1490) from sqrt(1 - x) times a polynomial
// of degree 7 -- Abramowitz & Stegun 4.4.46, |error| <= 2e-8 rad on [0, 1], mirrored for x < 0 -- instead of libm's acosf: the angle
// only positions the look-up between two entries of the forward table, a ray's WEIGHT)
__device__ __forceinline__ float fast_acos(float x) {
  const float a = fabsf(x);
  float p = -0.0012624911f;
  p = __builtin_fmaf(p, a, 0.0066700901f); p = __builtin_fmaf(p, a, -0.0170881256f); p = __builtin_fmaf(p, a, 0.0308918810f);
  p = __builtin_fmaf(p, a, -0.0501743046f); p = __builtin_fmaf(p, a, 0.0889789874f); p = __builtin_fmaf(p, a, -0.2145988016f);
  p = __builtin_fmaf(p, a, 1.5707963050f);
  const float r = __builtin_amdgcn_sqrtf(fmaxf(1.0f - a, 0.0f)) * p;
  return x < 0.0f ? kPi - r : r;
}
// lookUpPhaseFuncValsFromTable :1613-1652 with the two divisions by the (uniform) table spacing as one reciprocal
__device__ __forceinline__ float lookup_phase_fast(const float *tab, int n, float angle) {
  const float rcpDTheta = (float)(n - 1) * (1.0f / kPi);
  const float pos = angle * rcpDTheta;
  const int k = (int)pos + 1;
  if (k < n) {
    const float f = pos - (float)(k - 1);
    return (1.0f - f) * tab[k - 1] + f * tab[k];
  }
  return tab[n - 1];
}

// computeSurfaceReflectance, Code/surfaceProperties.f95:121-148, with the Lambertian R (:154-162)
template <class PR>
__device__ __forceinline__ float surface_reflectance(const PR &P, float x, float y) {
  const float *xs = P.xsE, *ys = P.ysE;
  const int ix = find_index(make_periodic(x, xs[0], xs[P.nxs]), [xs](int k) { return xs[k - 1]; }, P.nxs + 1, 0);
  const int iy = find_index(make_periodic(y, ys[0], ys[P.nys]), [ys](int k) { return ys[k - 1]; }, P.nys + 1, 0);
  return P.brdf[(size_t)(iy - 1) * P.nxs + (ix - 1)];
}

}  // namespace i3rc
