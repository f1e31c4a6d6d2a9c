// gfx950 kernels of the I3RC photon tracer.
//
// Mapping (MI355X-first, not a port of the scalar Fortran loop nest):
//   * one photon per LANE, 64 photons in flight per wavefront, persistent waves: a lane that loses its photon
//     (exit, absorption, roulette, tracer drop) gets the next photon index from the wave's reservoir, which is
//     refilled from a device-wide counter with one returning atomic per <= 256 photons;
//   * the reference's three nested data-dependent loops (photon / order of scattering / voxel step,
//     computeRT :452-691 + accumulateExtinctionAlongPath :1690-1806) are flattened into a lane state machine with
//     ballot-gated phases: a VOXEL-STEP phase executed by the lanes that are tracing, an EVENT phase (exit tallies,
//     next photon, one Philox block for the whole wave, scatter / surface / new photon, new optical depth) that runs
//     when enough lanes wait for it, and for radiances a LIGHT phase: local-estimate (shadow) rays are traced by the
//     same voxel-step phase while the photon's own state is parked in LDS;
//   * cell edges (and, when they fit, the extinction grid) are staged in LDS with coalesced loads; grids beyond an
//     XCD's L2 are read from a copy in 32-cell bricks; flux and radiance tallies are privatised per workgroup in LDS
//     (ds_add_f32) and flushed once with float64 atomics; large domains tally straight to HBM with float64 atomics;
//   * per-photon Philox4x32-10 streams keyed by (seed, batch) make a photon's path independent of the launch
//     geometry, of every scheduling threshold and of the number of GPUs;
//   * work counters live in scalar registers (advanced by s_bcnt1 of ballots in uniform control flow).
#pragma once
#include <cstddef>
#include "tracer.hpp"

namespace i3rc {

enum LaneState { ST_TRACE = 0, ST_EVENT = 1, ST_DROPPED = 2, ST_NEW = 3, ST_DONE = 4,
                 ST_SHADOW = 5,   // tracing a local-estimate (shadow) ray towards a radiance direction
                 ST_LIGHT = 6 };  // a shadow ray ended, or the first one is due: needs the light phase

// Work counters (I3RC_CNT_*) are kept per WAVE, in scalar registers: they are only ever advanced in uniform control
// flow by the population count of a ballot, so they cost no vector registers and no vector instructions.  The
// nested local-estimate path (replay / max cross-section builds) counts its tracer work per lane.
struct WaveCounters {   // 32 bits are enough: a wave hands its counts over every time it refills its photon reservoir
  uint32_t photons = 0, dropped = 0, steps = 0, scat = 0, surf = 0, top = 0, roul = 0, shadow = 0, calls = 0;
};
struct NestedCounters { uint32_t shadow = 0, calls = 0; };

__device__ __forceinline__ unsigned count_lanes(bool p) { return (unsigned)__popcll(__ballot(p)); }
// number of lanes below this one whose bit is set in a wave-uniform mask
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ void add_global(double *p, float v) { unsafeAtomicAdd(p, (double)v); }

// Kernel arguments that are only needed now and then -- the reservoir refill, the counter hand-over, the epilogue --
// are read from the kernarg segment where they are used (scalar loads) instead of living in scalar registers through
// the whole photon loop: the flux kernel wanted 106 of the 102 there are, and every spilled one comes back as a
// v_readlane, i.e. a vector instruction, in the event phase.  (The empty asm keeps the loads from being hoisted.)
struct KernelArgs { DevProblem P; RunArgs A; };   // the kernarg segment of photon_kernel: (P, A, ...)
static_assert(offsetof(KernelArgs, A) == sizeof(DevProblem) && sizeof(DevProblem) % 8 == 0 && alignof(RunArgs) == 8,
              "the second kernel argument must follow the first without padding");
typedef const __attribute__((address_space(4))) KernelArgs *ColdArgs;
__device__ __forceinline__ ColdArgs cold_args() {
  ColdArgs k = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}


struct Tally {
  const DevProblem &P;
  const Lds &L;
  // (the tally buffer's base addresses stay in scalar registers: reading them from the kernarg segment at every
  // tally -- see cold_args -- was measured: -8 % where the tallies go to global memory, nothing gained elsewhere)
  __device__ __forceinline__ void down(int col, float w) const {
    if (P.ldsTallies) lds_add(&L.tDown[col], w); else add_global(P.tally + P.oDown + col, w);
  }
  // intensityByComponent(ix, iy, d, comp) (:574-579, :662-667)
  __device__ __forceinline__ void radiance(int comp, int d, int col, float v) const {
    const int i = (comp * P.nDir + d) * (P.nx * P.ny) + col;
    if (P.ldsIntensity) lds_add(&L.tInt[i], v); else add_global(P.tally + P.oInt + i, v);
  }
  // upward flux at the top (:513) or downward flux at the surface (:531): one atomic for either
  __device__ __forceinline__ void boundary(bool top, int col, float w) const {
    if (P.ldsTallies) lds_add((top ? L.tUp : L.tDown) + col, w);
    else add_global(P.tally + (top ? P.oUp : P.oDown) + col, w);
  }
  __device__ __forceinline__ void absorbed(int col, int cell, float w) const {
    if (P.ldsTallies) lds_add(&L.tAbs[col], w); else add_global(P.tally + P.oAbs + col, w);
    add_global(P.tally + P.oVol + cell, w);
  }
};

// computeIntensityContribution :1419-1611 for one event; adds straight into intensityByComponent.
template <int GRID, class Rng>
__device__ __forceinline__ void intensity_contribution(const DevProblem &P, const Lds &L, Rng &rng, NestedCounters &cnt,
                                                       float weight, float x, float y, float z, int ix, int iy, int iz,
                                                       float dx, float dy, float dz, int component, int order) {
  const int zIndexMax = P.nz + 1;
  const size_t ncol = (size_t)P.nx * P.ny;
  for (int d = 0; d < P.nDir; ++d) {
    const float ux = L.dirCos[3 * d], uy = L.dirCos[3 * d + 1], uz = L.dirCos[3 * d + 2];
    float normPF;
    if (component < 1) {
      normPF = 1.0f / kPi;
    } else {
      float proj = 0.0f;
      proj += dx * ux; proj += dy * uy; proj += dz * uz;
      if (fabsf(proj) > 1.0f) proj = copysignf(1.0f, proj);
      const float ang = acosf(proj);
      const size_t ncell = ncol * P.nz;
      const int pfi = max(P.pfIndex[(size_t)(component - 1) * ncell + cell_index(P, ix, iy, iz)], 1);
      const CompTables ct = P.comp[component - 1];
      const int n = ct.nFwd;
      const float *tab = ((P.useHybrid && order <= P.numOrdersOrig) ? ct.fwdOrig : ct.fwd) + (size_t)(pfi - 1) * n;
      normPF = lookup_phase(tab, n, ang) / ((4.0f * kPi) * fabsf(uz));
    }
    Ray r;
    r.x = x; r.y = y; r.z = z; r.ix = ix; r.iy = iy; r.iz = iz; r.dx = ux; r.dy = uy; r.dz = uz;
    r.set_direction(L);
    // ONE loop over the voxel steps of all the legs of this direction, the legs told apart by `stage` as in the light
    // phase of photon_kernel (0: plain local estimate, 1: small contribution, 2 / 3: the two legs of a large one).
    // Two loops one after the other (first leg, second leg) are what the compiler mishandled: see GridPlace.
    int stage = 0;
    float tauFree = 0.0f;
    r.acc = 0.0f; r.target = 0.0f;
    if (P.useRRI) {
      tauFree = -logf(fmaxf(kTiny, rng.next()));
      if (kPi * normPF <= P.zetaMin) { stage = 1; r.target = tauFree; }
      else { stage = 2; r.target = -logf(P.zetaMin / fmaxf(kTiny, kPi * normPF)); }   // tauMax
    }
    cnt.calls++;
    float con = 0.0f;
    for (;;) {
      cnt.shadow++;
      if (trace_step<GRID>(P, L, r, stage != 0) == STEP_CONTINUE) continue;
      const float tauB = r.acc;
      const bool outTop = r.iz >= zIndexMax;
      if (stage == 0) con = tauB >= 0.0f ? (weight * normPF) * expf(-tauB) : 0.0f;
      else if (stage == 1) {
        const float r2 = rng.next();
        con = (r2 <= kPi * normPF / P.zetaMin && outTop) ? weight * P.zetaMin / kPi : 0.0f;
      } else if (stage == 2) {
        if (outTop && tauB >= 0.0f) con = (weight * normPF) * expf(-tauB);
        else if (tauB >= 0.0f && r.iz >= 1) {   // second leg, up to the free path (not from below the grid: see the light phase)
          r.acc = 0.0f; r.target = tauFree; stage = 3;
          cnt.calls++;
          continue;
        }
      } else con = outTop ? weight * P.zetaMin / kPi : 0.0f;
      break;
    }
    if (P.limitContrib && con > P.maxContrib) {
      add_global(P.tally + P.oExc + component * P.nDir + d, con - P.maxContrib);
      con = P.maxContrib;
    }
    const Tally tl{P, L};
    tl.radiance(component, d, (r.iy - 1) * P.nx + (r.ix - 1), con);
  }
}

template <class Rng>
struct RngInit;
template <>
struct RngInit<PhiloxStream> {
  static __device__ __forceinline__ void init(PhiloxStream &g, const RunArgs &A) { g.init(A.seed0, A.seed1); }
  static __device__ __forceinline__ void start(PhiloxStream &g, const RunArgs &A, long long i) {
    g.start((uint64_t)(A.firstPhoton + i));
  }
};
template <>
struct RngInit<ReplayStream> {
  static __device__ __forceinline__ void init(ReplayStream &g, const RunArgs &A) { g.init(A.randoms, A.nRandoms); }
  static __device__ __forceinline__ void start(ReplayStream &g, const RunArgs &A, long long i) { g.start(A.drawStart[i]); }
};

// Wave-private reservoir of photon indices: one returning atomic per `chunk` photons instead of one per respawn
// round (the returning atomic costs microseconds; every wave would pay it in ~97 % of its event phases).  The two
// bounds are wave-uniform and held in scalar registers (readfirstlane tells the compiler so).
struct Reservoir {
  long long next, end;
  __device__ __forceinline__ void refill() {   // call in uniform control flow only
    const ColdArgs k = cold_args();
    unsigned long long *const counter = k->A.workCounter;
    const int chunk = k->A.chunk;
    const long long nPhotons = k->A.nPhotons;
    unsigned long long base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, (unsigned long long)chunk);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__shfl((unsigned)base, 0, 64));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)__shfl((unsigned)(base >> 32), 0, 64));
    const long long b = (long long)(((unsigned long long)hi << 32) | lo);
    next = b < nPhotons ? b : nPhotons;
    end = b + chunk < nPhotons ? b + chunk : nPhotons;
  }
};

#ifndef I3RC_MIN_WAVES
#define I3RC_MIN_WAVES 5
#endif
// GENERAL = false is the specialisation for the common problem class -- regular grid, ray tracing, one component,
// Lambertian albedo (no BRDF grid), Directional source, production RNG: the rare paths (grid searches, periodic
// re-wrapping loops, max-cross-section moves, BRDF lookups, component selection) are compiled out, which shrinks the
// loop's code and its scalar-register pressure.  GENERAL = true keeps every path behind run-time switches.
// (the general radiance kernel keeps the most state live: 4 waves per SIMD give it 128 vector registers and no spills)
template <class Rng, bool INTENSITY, bool GENERAL, int GRID>
__global__ void __launch_bounds__(256, (INTENSITY && GENERAL) ? 4 : I3RC_MIN_WAVES) photon_kernel(const DevProblem P, const RunArgs A, const int evThreshold, const int lightThreshold) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  {
    lds_float *p = (lds_float *)smem;
    L.xE = p; p += P.nx + 1;
    L.yE = p; p += P.ny + 1;
    L.zE = p; p += P.nz + 1;
    const int ncol = P.nx * P.ny;
    L.tUp = p; L.tDown = p + ncol; L.tAbs = p + 2 * ncol;
    if (P.ldsTallies) p += 3 * ncol;
    L.dirCos = p; p += 3 * P.nDir;
    L.park = p;
    if (INTENSITY && !Rng::kReplay) p += kParkWords * 256;
    L.tInt = p;
    if (P.ldsIntensity) p += (P.ncomp + 1) * P.nDir * ncol;
    L.ext = p;
  }
  for (int i = threadIdx.x; i < 3 * P.nDir; i += blockDim.x) L.dirCos[i] = P.dirCos[i];
  // coalesced staging of the edge vectors (and the extinction grid when it fits)
  for (int i = threadIdx.x; i <= P.nx; i += blockDim.x) L.xE[i] = P.xE[i];
  for (int i = threadIdx.x; i <= P.ny; i += blockDim.x) L.yE[i] = P.yE[i];
  for (int i = threadIdx.x; i <= P.nz; i += blockDim.x) L.zE[i] = P.zE[i];
  if (P.ldsTallies)
    for (int i = threadIdx.x; i < 3 * P.nx * P.ny; i += blockDim.x) L.tUp[i] = 0.0f;
  if (GRID == GRID_LDS) {
    const int ncell = P.nx * P.ny * P.nz;
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) L.ext[i] = P.totalExt[i];
  }
  if (P.ldsIntensity)
    for (int i = threadIdx.x; i < (P.ncomp + 1) * P.nDir * P.nx * P.ny; i += blockDim.x) L.tInt[i] = 0.0f;
  __syncthreads();

  constexpr bool REPLAY = Rng::kReplay;        // per-photon fates are recorded by i3rc_hip_run_replay only
  constexpr bool DEFER_BUILD = INTENSITY && !Rng::kReplay;   // radiances through the lane state machine (see DEFER below)
  constexpr bool NEED_PID = REPLAY || GENERAL; // explicit photon sources are indexed by photon number
  const Tally tally{P, L};
  const size_t ncell = (size_t)P.nx * P.ny * P.nz;
  const bool rayTracing = GENERAL ? (P.useRayTracing != 0) : true;
  const bool useBDRF = GENERAL ? (P.useBDRF != 0) : false;
  const bool multiComp = GENERAL ? (P.ncomp > 1) : false;
  const bool directional = GENERAL ? (A.srcKind == 0) : true;
  // Directional photons all start at z = z0 + (1 - spacing(1)) (zMax - z0): their start layer is wave-uniform
  const float zStart = P.z0 + (1.0f - spacingf(1.0f)) * (P.zMax - P.z0);
  int izStart = 1;
  find_z<true>(P, L, zStart, izStart);   // (once per wave: the specialised kernels take irregular layers too)
  const float rcpDeltaX = refined_rcp(P.deltaX), rcpDeltaY = refined_rcp(P.deltaY);
  const float surfaceZ = P.z0 + spacingf(P.z0);

  WaveCounters wc;
  NestedCounters nested;
  Rng rng;
  RngInit<Rng>::init(rng, A);
  Ray r;
  r.x = r.y = r.z = 0.0f; r.dx = r.dy = 0.0f; r.dz = -1.0f; r.ix = r.iy = r.iz = 1; r.acc = 0.0f; r.target = 0.0f;
  r.rx = r.ry = r.rz = 0.0f; r.slow = 1;
  r.ex = r.ey = r.ez = 0; r.cx = r.cy = r.cz = 1; r.nudge = 2.0f;
  float w = 0.0f;
  int order = 0;
  int st = ST_NEW;
  long long pid = -1;                 // photon number within the launch (NEED_PID builds)
  int fate = -1, fateCol = -1;        // REPLAY builds
  float fateW = 0.0f;
  Reservoir res;
  res.refill();
  // Radiance (local estimate) as part of the lane state machine instead of a loop nested in the event: after an
  // event the photon's own state is parked in LDS and the lane traces one shadow ray per radiance direction in the
  // common voxel-step phase; ray ends are handled in a light phase of their own (DEFER).  The replay build keeps
  // the reference's nested order (same deviates at the same places).
  constexpr bool DEFER = INTENSITY && !Rng::kReplay;
  float wI = 0.0f, normPF = 0.0f, tauFree = 0.0f;   // weight of the event, phase-function factor and free path of the current ray
  int dIdx = 0, stage = -1;                          // direction being traced; -1 none, 0 plain, 1 small-contribution RR, 2/3 two-leg RR
  bool pendingShadow = false;
  const bool defer = DEFER && rayTracing;            // max cross-section moves the photon inside the event: keep the nested order there

  // Thresholds: fixed when the caller asks for them (> 0), else adapted by every wave to its own photons at every
  // reservoir refill.  Event phase: the longer the photons' own traces (voxel steps per event), the more a
  // lane loses by waiting for others, so the threshold falls as 64 / sqrt(steps per event) (measured optima: 40 at
  // 2.5 steps per event, 32 at 3.5, 24 at 9, 16 at 14 ... 16, with or without shadow rays in the mix).  Light phase: likewise with
  // the length of the shadow rays, 70 / sqrt(steps per ray) within 16..32 (measured: 32 for the radar case's 2-step
  // nadir rays, 16 for the Landsat case's 19-step rays).  Thresholds only schedule work: no photon path depends on them.
  int evThr = evThreshold > 0 ? evThreshold : -evThreshold;
  int liThr = lightThreshold > 0 ? lightThreshold : -lightThreshold;
  const bool adaptEvent = evThreshold < 0, adaptLight = DEFER_BUILD && lightThreshold < 0;
  uint32_t raysStarted = 0;   // shadow rays since the last refill (wave-uniform)
  uint32_t refills = 0;       // visits of the work counter by this wave
  // re-fit the thresholds to what this wave has seen since its last hand-over (uniform control flow only)
  uint32_t raysSeen = 0;      // shadow rays since the last hand-over
  auto adapt_thresholds = [&]() {
    if (adaptEvent) {
      const float events = (float)(wc.scat + wc.photons + wc.surf), steps = (float)wc.steps;
      if (events > 0.0f && steps > 0.0f) {
        const int t = (int)(64.0f * __builtin_amdgcn_rsqf(steps * __builtin_amdgcn_rcpf(events)));
        evThr = __builtin_amdgcn_readfirstlane(t < 12 ? 12 : (t > 44 ? 44 : t));
      }
    }
    if (adaptLight) {
      raysSeen += raysStarted; raysStarted = 0u;
      if (raysSeen > 0u && wc.shadow > 0u) {
        const int t = (int)(70.0f * __builtin_amdgcn_rsqf((float)wc.shadow * __builtin_amdgcn_rcpf((float)raysSeen)));
        liThr = __builtin_amdgcn_readfirstlane(t < 16 ? 16 : (t > 32 ? 32 : t));
      }
    }
  };
  // hands the wave's work counters over to the tally buffer (uniform control flow only)
  auto flush_counters = [&]() {
    adapt_thresholds();
    raysSeen = 0u;
    if ((threadIdx.x & 63) == 0) {
      const uint32_t c[9] = {wc.photons, wc.dropped, wc.steps, wc.scat, wc.surf, wc.top, wc.roul, wc.shadow, wc.calls};
      const ColdArgs ka = cold_args();
      double *const counters = ka->P.tally + ka->P.oCnt;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (c[k] != 0u) unsafeAtomicAdd(counters + k, (double)c[k]);
    }
    wc = WaveCounters();
  };
  // a photon's record for i3rc_hip_run_replay; production builds keep no per-photon record
  auto close_photon = [&]() {
    if (REPLAY) {
      if (A.fate) {
        A.fate[pid] = fate; A.fateColumn[pid] = fateCol; A.fateWeight[pid] = fateW; A.fateOrder[pid] = order;
        A.drawsUsed[pid] = (int32_t)rng.draws_of_photon();
      }
    }
    rng.close();
  };

#ifdef I3RC_PROFILE_PHASES   // diagnostic build only (tools/phase_profile.sh): where do a wave's cycles go?
  unsigned long long profEv = 0, profSt = 0, profNEv = 0, profNSt = 0, profLanesEv = 0, profLanesSt = 0, profNew = 0;
  unsigned long long profSeg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long profMark = 0;
#define PROF_SEG(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); profSeg[k] += t_ - profMark; profMark = t_; } while (0)
#else
#define PROF_SEG(k) do {} while (0)
#endif
#ifdef I3RC_PROFILE_PHASES
#define PROF_T() __builtin_amdgcn_s_memtime()
#else
#define PROF_T() 0ull
#endif
  for (;;) {
    // ---------------------------------------------------------------- EVENT phase
    const bool wantEvent = st == ST_EVENT || st == ST_DROPPED || st == ST_NEW;
    const unsigned long long evMask = __ballot(wantEvent);
    const unsigned long long trMask = __ballot(st == ST_TRACE || st == ST_SHADOW);
    unsigned long long liMask = 0ull;
    if (DEFER) {
      liMask = __ballot(st == ST_LIGHT);
      // ------------------------------------------------------------ LIGHT phase (shadow-ray ends and starts)
      if (liMask != 0ull && (__popcll(liMask) >= liThr || trMask == 0ull)) {
#ifdef I3RC_PROFILE_PHASES
        const unsigned long long profL0 = __builtin_amdgcn_s_memtime();
        profSeg[6] += (unsigned long long)__popcll(liMask);   // lanes served by light phases
        profNew++;                                            // light phases
#endif
        if (st == ST_LIGHT) {
          lds_float *park = L.park + threadIdx.x;
          if (stage >= 0) {                                              // the ray that just ended (:1517-1596)
            const float tauB = r.acc;
            const bool outTop = r.iz >= P.nz + 1;
            float con = 0.0f;
            if (stage == 0) con = tauB >= 0.0f ? (wI * normPF) * expf(-tauB) : 0.0f;
            else if (stage == 1) {
              const float r2 = rng.next();
              con = (r2 <= kPi * normPF / P.zetaMin && outTop) ? wI * P.zetaMin / kPi : 0.0f;
            } else if (stage == 2) {
              if (outTop && tauB >= 0.0f) con = (wI * normPF) * expf(-tauB);
              else if (tauB >= 0.0f && r.iz >= 1) {                      // second leg, up to the free path (:1576-1587)
                r.acc = 0.0f; r.target = tauFree; stage = 3; st = ST_SHADOW;
              }
              // (a first leg that left through the BOTTOM -- a downward radiance direction -- gets no second leg: the
              // reference starts one from outside the grid, reads zPosition(0) / totalExt(:, :, 0) out of bounds and
              // discards the outcome, since only an exit through the top counts; the contribution is 0 either way)
            } else con = outTop ? wI * P.zetaMin / kPi : 0.0f;
            if (st == ST_LIGHT) {
              const int comp = __float_as_int(park[13 * 256]) & 0xff;
              if (P.limitContrib && con > P.maxContrib) {                // :1598-1609
                add_global(P.tally + P.oExc + comp * P.nDir + dIdx, con - P.maxContrib);
                con = P.maxContrib;
              }
              tally.radiance(comp, dIdx, (r.iy - 1) * P.nx + (r.ix - 1), con);
              dIdx++; stage = -1;
            }
          }
          if (st == ST_LIGHT) {
            r.x = park[0]; r.y = park[256]; r.z = park[2 * 256];
            r.ix = __float_as_int(park[3 * 256]); r.iy = __float_as_int(park[4 * 256]); r.iz = __float_as_int(park[5 * 256]);
            if (dIdx < P.nDir) {                                         // next radiance direction (:1473-1510)
              const float ux = L.dirCos[3 * dIdx], uy = L.dirCos[3 * dIdx + 1], uz = L.dirCos[3 * dIdx + 2];
              const int info = __float_as_int(park[13 * 256]);
              const int comp = info & 0xff;
              if (comp < 1) normPF = 1.0f / kPi;
              else {
                float proj = 0.0f;
                proj += park[10 * 256] * ux; proj += park[11 * 256] * uy; proj += park[12 * 256] * uz;
                if (fabsf(proj) > 1.0f) proj = copysignf(1.0f, proj);
                const float ang = acosf(proj);
                const int pfi = __float_as_int(park[14 * 256]);
                const CompTables ct = GENERAL ? P.comp[comp - 1] : P.comp0;
                const float *tab = ((info & 0x100) ? ct.fwdOrig : ct.fwd) + (size_t)(pfi - 1) * ct.nFwd;
                normPF = lookup_phase(tab, ct.nFwd, ang) / ((4.0f * kPi) * fabsf(uz));
              }
              r.dx = ux; r.dy = uy; r.dz = uz;
              r.set_direction(L);
              r.acc = 0.0f; r.target = 0.0f;
              if (!P.useRRI) stage = 0;
              else {
                tauFree = -logf(fmaxf(kTiny, rng.next()));
                if (kPi * normPF <= P.zetaMin) { stage = 1; r.target = tauFree; }
                else { stage = 2; r.target = -logf(P.zetaMin / fmaxf(kTiny, kPi * normPF)); }
              }
              st = ST_SHADOW;
            } else if (w <= kTiny) {
              st = ST_NEW;                                               // killed by roulette at this event
            } else {                                                     // back to the photon's own path
              r.dx = park[6 * 256]; r.dy = park[7 * 256]; r.dz = park[8 * 256];
              r.target = park[9 * 256]; r.acc = 0.0f;
              r.set_direction(L);
              st = ST_TRACE;
            }
          }
        }
        // every lane that entered as ST_LIGHT and leaves as ST_SHADOW has started exactly one tracer call
        const unsigned started = (unsigned)__popcll(__ballot(st == ST_SHADOW) & liMask);
        wc.calls += started;
        raysStarted += started;
#ifdef I3RC_PROFILE_PHASES
        profSeg[7] += __builtin_amdgcn_s_memtime() - profL0;    // cycles in light phases
#endif
      }
    }
    if (evMask == 0ull && trMask == 0ull && liMask == 0ull) break;
    const unsigned long long profT0 = PROF_T();
    if (evMask != 0ull && (__popcll(evMask) >= evThr || (trMask == 0ull && liMask == 0ull))) {
#ifdef I3RC_PROFILE_PHASES
      profNEv++; profLanesEv += __popcll(evMask);
      profMark = __builtin_amdgcn_s_memtime();
#endif
      // ---- part A: endings that need no random number -- tracer drop, exit through the top, arrival at a black
      //      surface -- are tallied first so that the lanes can be given their next photon before the wave
      //      generates its random block (part C), which then serves old and new photons in one go.
      const bool blackSurface = !REPLAY && !useBDRF && !(P.albedo > kTiny) && !INTENSITY;
      const bool isEv = wantEvent && st == ST_EVENT;
      const bool dropped = wantEvent && st == ST_DROPPED;                 // :488-489
      const bool atTop = isEv && r.z >= P.zMax;                           // :499-514
      const bool atSurface = isEv && !atTop && r.z <= surfaceZ;           // :515-531
      const bool atBlack = atSurface && blackSurface;                     // ... and :560-562
      wc.dropped += count_lanes(dropped);
      wc.top += count_lanes(atTop);
      wc.surf += count_lanes(atSurface);
      if (dropped || atTop || atBlack) {
        // one merged branch for the three endings: a single tally atomic and a single bookkeeping block
        if (!dropped) {
          if (!rayTracing) {   // max cross-section: step back to the boundary (:504-511, :521-528)
            const float zB = atTop ? P.zMax : P.z0;
            r.x = make_periodic(r.x - r.dx * fabsf((r.z - zB) / r.dz), P.x0, P.xMax);
            r.y = make_periodic(r.y - r.dy * fabsf((r.z - zB) / r.dz), P.y0, P.yMax);
            find_xy<GENERAL>(P, L, r.x, r.y, r.ix, r.iy);
          }
          const int c2 = (r.iy - 1) * P.nx + (r.ix - 1);
          tally.boundary(atTop, c2, w);
          if (REPLAY) { fateCol = c2; fateW = w; }
        }
        if (REPLAY) {
          order += atBlack ? 1 : 0;
          fate = dropped ? 3 : (atTop ? 0 : 1);
        }
        close_photon();
        st = ST_NEW;
      }
      PROF_SEG(0);
      // ---- part B (uniform): hand out photon indices from the wave's reservoir
      const bool isNew = wantEvent && st == ST_NEW;
      const unsigned long long newMask = __ballot(isNew);
      if (newMask != 0ull) {
        int need = __popcll(newMask);
        int rank = lanes_below(newMask);
        long long mine = -1;
        long long avail = res.end - res.next;
        if (avail < (long long)need && res.end < A.nPhotons) {   // drain the reservoir, then refill it (chunk >= 64 covers the rest)
          if (isNew && rank < (int)avail) mine = res.next + rank;
          wc.photons += (unsigned)avail;                        // numPhotonsProcessed :459
          need -= (int)avail;
          rank -= (int)avail;
          // the counters go to the tally buffer (and the thresholds are re-fitted) at every fourth refill: the nine
          // atomics of a hand-over all go to the same nine addresses, from every wave of the chip
          if ((++refills & 3u) == 0u || wc.steps > 0x40000000u || wc.shadow > 0x40000000u) flush_counters();
          else adapt_thresholds();
          res.refill();
          avail = res.end - res.next;
        }
        const int taken = (int)(avail < (long long)need ? avail : (long long)need);   // < need only when the batch is exhausted
        if (isNew && mine < 0 && rank >= 0 && rank < taken) mine = res.next + rank;
        res.next += taken;
        wc.photons += (unsigned)taken;
        if (isNew) {
          if (mine < 0) st = ST_DONE;
          else {
            RngInit<Rng>::start(rng, A, mine);
            if (NEED_PID) pid = mine;
          }
        }
      }
      PROF_SEG(1);
      // ---- part C: one random block per lane for this event, then the event itself
      bool didScatter = false, didRoulette = false, startedTrace = false;   // per-lane flags -> wave counters below
      if (wantEvent && st != ST_DONE) {
        rng.begin_event(DEFER_BUILD && P.useRRI != 0);   // the local estimate's roulette draws after the event
        PROF_SEG(2);
        if (st == ST_NEW) {                                               // :453-470
          float px, py, pz;
          if (directional) {   // newPhotonStream_Directional, Code/monteCarloIllumination.f95:91-99
            px = rng.first(); py = rng.second();
            pz = 1.0f - spacingf(1.0f);
            r.dx = A.solarDx; r.dy = A.solarDy; r.dz = A.solarDz;
          } else {
            px = A.sx[pid]; py = A.sy[pid]; pz = A.sz[pid];
            make_dircos(A.smu[pid], A.sphi[pid], r.dx, r.dy, r.dz);
          }
          order = 0;
          if (REPLAY) { fate = -1; fateCol = -1; fateW = 0.0f; }
          w = 1.0f;
          r.x = P.x0 + px * (P.xMax - P.x0);
          r.y = P.y0 + py * (P.yMax - P.y0);
          r.z = P.z0 + pz * (P.zMax - P.z0);
          r.ix = 1; r.iy = 1; r.iz = 1;
          if (!GENERAL) {   // findXYIndicies :1359-1369 with the divisions by the (uniform) cell sizes done by reciprocal
            int i = min((int)exact_div(r.x - P.x0, P.deltaX, rcpDeltaX) + 1, P.nx);
            int j = min((int)exact_div(r.y - P.y0, P.deltaY, rcpDeltaY) + 1, P.ny);
            if (fabsf(L.xE[i] - r.x) < spacingf(r.x)) i = i + 1;
            if (fabsf(L.yE[j] - r.y) < spacingf(r.y)) j = j + 1;
            r.ix = i == P.nx + 1 ? 1 : i;
            r.iy = j == P.ny + 1 ? 1 : j;
            r.iz = izStart;
          } else {
            find_xy<GENERAL>(P, L, r.x, r.y, r.ix, r.iy);
            find_z<GENERAL>(P, L, r.z, r.iz);
          }
          st = ST_TRACE;
        }
        PROF_SEG(3);
        if (st == ST_EVENT) {
          if (r.z <= surfaceZ) {                                          // :515-580
            order++;
            if (!rayTracing) {
              r.x = make_periodic(r.x - r.dx * fabsf((r.z - P.z0) / r.dz), P.x0, P.xMax);
              r.y = make_periodic(r.y - r.dy * fabsf((r.z - P.z0) / r.dz), P.y0, P.yMax);
              find_xy<GENERAL>(P, L, r.x, r.y, r.ix, r.iy);
            }
            r.iz = 1;
            r.z = surfaceZ;
            const int c2 = (r.iy - 1) * P.nx + (r.ix - 1);
            tally.down(c2, w);
            if (REPLAY) { fateCol = c2; fateW = w; }
            float mu = exact_sqrt(rng.first());
            while (!(fabsf(mu) > 2.0f * kTiny)) mu = exact_sqrt(rng.next());   // :546-549
            const float phi = (2.0f * kPi) * rng.second();
            if (useBDRF) w = w * surface_reflectance(P, r.x, r.y);
            else w = w * P.albedo;
            if (w <= kTiny) { if (REPLAY) fate = 1; st = ST_NEW; }
            else {
              make_dircos(mu, phi, r.dx, r.dy, r.dz);
              if (defer) {
                pendingShadow = true; wI = w;
                L.park[13 * 256 + threadIdx.x] = __int_as_float(0);     // component 0: the surface
              } else if (INTENSITY)
                intensity_contribution<GRID>(P, L, rng, nested, w, r.x, r.y, r.z, r.ix, r.iy, r.iz, r.dx, r.dy, r.dz, 0, order);
              st = ST_TRACE;
            }
          } else {                                                        // :581-689
            bool scatterThis = true;
            int cell = 0;
            if (!rayTracing) {
              // max cross-section never updates the cell indices after a move (:494-496 has no index search): they are
              // those of the photon's start or of its last surface hit.  A start index of nz + 1 (a photon that starts
              // within spacing() of the domain top: thin elevated domains) is outside the grid; the reference reads
              // totalExt out of bounds there.  Here: no extinction outside the grid, hence never a scattering.
              const bool inGrid = (unsigned)(r.iz - 1) < (unsigned)P.nz;
              cell = inGrid ? cell_index(P, r.ix, r.iy, r.iz) : 0;
              const float extHere = inGrid ? P.totalExt[cell] : 0.0f;
              scatterThis = rng.next() < extHere / P.maxExt;
            }
            if (scatterThis) {
              order++;
              didScatter = true;
              // :606-632 (quirk Q2 kept): a scattering in a cell without extinction steps back over the cell face.  The
              // tracer only stops inside a cell whose extinction is positive (acc + step * 0 > target cannot hold), so
              // with ray tracing the case cannot arise and the look-up is left to the max-cross-section build.
              if (!rayTracing && cell_extinction<GRID>(P, L, r.ix, r.iy, r.iz) <= 0.0f) {
                if (r.x - L.xE[r.ix - 1] <= 0.0f && r.dx > 0.0f) {
                  r.x = r.x - spacingf(r.x);
                  r.ix = r.ix - 1;
                  if (r.ix <= 0) { r.ix = P.nx; r.x = L.xE[r.ix - 1]; r.x = r.x - 2.0f * spacingf(r.x); }
                }
                if (r.y - L.yE[r.iy - 1] <= 0.0f && r.dy > 0.0f) {
                  r.y = r.y - spacingf(r.y);
                  r.iy = r.iy - 1;
                  if (r.iy <= 0) { r.iy = P.ny; r.y = L.xE[r.iy - 1]; r.y = r.x - 2.0f * spacingf(r.y); }
                }
                if (r.z - L.zE[r.iz - 1] <= 0.0f && r.dz > 0.0f) { r.z = r.z - spacingf(r.z); r.iz = r.iz - 1; }
              }
              // the cell's properties are read only where the domain does not share one value (see DevProblem)
              const bool needCell = GENERAL || !(P.uniformSsa >= 0.0f) || P.uniformSsa < 1.0f || P.uniformPf < 1;
              if (needCell) cell = cell_index(P, r.ix, r.iy, r.iz);
              int comp = 1;                                               // :637-638
              if (multiComp || REPLAY) {
                const float rc = rng.next();
                if (multiComp) {
                  const float *cum = P.cumExt + cell;
                  comp = find_index(rc, [cum, ncell](int k) { return k == 1 ? 0.0f : cum[(size_t)(k - 2) * ncell]; },
                                    P.ncomp + 1, 0);
                }
              }
              // single-scattering albedo and phase-function entry of the cell; a value shared by the whole (one-component)
              // domain comes from the kernel arguments instead of two dependent memory reads
              float ssa;
              if (!GENERAL && P.uniformSsa >= 0.0f) ssa = P.uniformSsa;
              else ssa = P.ssa[(size_t)(comp - 1) * ncell + cell];
              if (ssa < 1.0f) {                                           // :642-649
                tally.absorbed((r.iy - 1) * P.nx + (r.ix - 1), cell, w * (1.0f - ssa));
                w = w * ssa;
              }
              int pfi;
              if (!GENERAL && P.uniformPf >= 1) pfi = P.uniformPf;
              else pfi = max(P.pfIndex[(size_t)(comp - 1) * ncell + cell], 1);   // (index 0 marks clear cells: never a table offset of -1)
              if (defer) {                                                // :654-668, traced after this event
                pendingShadow = true; wI = w;
                lds_float *park = L.park + threadIdx.x;
                park[10 * 256] = r.dx; park[11 * 256] = r.dy; park[12 * 256] = r.dz;   // incoming direction
                const int useOrig = (P.useHybrid && order <= P.numOrdersOrig) ? 0x100 : 0;
                park[13 * 256] = __int_as_float(comp | useOrig);
                park[14 * 256] = __int_as_float(pfi);
              } else if (INTENSITY)
                intensity_contribution<GRID>(P, L, rng, nested, w, r.x, r.y, r.z, r.ix, r.iy, r.iz, r.dx, r.dy, r.dz, comp, order);
              if (P.useRR && w < 0.5f) {                                  // :673-680
                didRoulette = true;
                if (rng.spare() >= w / 1.0f) w = 0.0f; else w = 1.0f;
              }
              if (w <= kTiny) { if (REPLAY) fate = 2; st = ST_NEW; }
              else {
                const CompTables ct = GENERAL ? P.comp[comp - 1] : P.comp0;
                const float cosS = scattering_cosine(rng.first(), ct.invCos + (size_t)(pfi - 1) * ct.nInv, ct.nInv,
                                                     refined_rcp((float)ct.nInv));
                next_direct(rng, cosS, r.dx, r.dy, r.dz);                 // :684-687
                st = ST_TRACE;
              }
            } else {
              st = ST_TRACE;
            }
          }
        }
        PROF_SEG(4);
        if (st == ST_TRACE) {                                             // :480
          const float tau = -sample_log(fmaxf(kTiny, rng.path()));
          r.acc = 0.0f; r.target = tau;
          if (rayTracing) { startedTrace = true; r.set_direction(L); }
          else {                                                          // :494-496 max cross-section move
            r.x = make_periodic(r.x + r.dx * tau / P.maxExt, P.x0, P.xMax);
            r.y = make_periodic(r.y + r.dy * tau / P.maxExt, P.y0, P.yMax);
            r.z = r.z + r.dz * tau / P.maxExt;
            st = ST_EVENT;
          }
        }
        PROF_SEG(5);
        if (defer && pendingShadow) {   // park the photon (alive or killed by roulette) and trace its shadow rays first
          lds_float *park = L.park + threadIdx.x;
          park[0] = r.x; park[256] = r.y; park[2 * 256] = r.z;
          park[3 * 256] = __int_as_float(r.ix); park[4 * 256] = __int_as_float(r.iy); park[5 * 256] = __int_as_float(r.iz);
          park[6 * 256] = r.dx; park[7 * 256] = r.dy; park[8 * 256] = r.dz;
          park[9 * 256] = r.target;
          pendingShadow = false; dIdx = 0; stage = -1;
          st = ST_LIGHT;
        }
        // a photon that died in part C (roulette, absorbing surface) is closed here and respawns at the next event phase
        if (st == ST_NEW) close_photon();
      }
      wc.scat += count_lanes(didScatter);
      wc.roul += count_lanes(didRoulette);
      wc.calls += count_lanes(startedTrace);
    }
    const unsigned long long profT1 = PROF_T();
#ifdef I3RC_PROFILE_PHASES
    profEv += profT1 - profT0;
    profNSt++; profLanesSt += __popcll(__ballot(st == ST_TRACE || st == ST_SHADOW));
#endif
    // ---------------------------------------------------------------- VOXEL-STEP phase
    const bool own = st == ST_TRACE, shadowRay = DEFER && st == ST_SHADOW;
    wc.steps += count_lanes(own);
    if (DEFER) wc.shadow += count_lanes(shadowRay);
    if (own || shadowRay) {
      const StepResult s = trace_step<GRID>(P, L, r, own || stage != 0);
      if (s == STEP_DONE) st = own ? ST_EVENT : ST_LIGHT;
      else if (s == STEP_ERROR) st = own ? ST_DROPPED : ST_LIGHT;   // a failed shadow ray contributes nothing (:1531-1535)
    }
#ifdef I3RC_PROFILE_PHASES
    profSt += PROF_T() - profT1;
#endif
  }
#ifdef I3RC_PROFILE_PHASES
  if ((threadIdx.x & 63) == 0) {
    unsafeAtomicAdd(P.tally + P.oCnt + 10, (double)profEv);
    unsafeAtomicAdd(P.tally + P.oCnt + 11, (double)profSt);
    unsafeAtomicAdd(P.tally + P.oCnt + 12, (double)profNEv);
    unsafeAtomicAdd(P.tally + P.oCnt + 13, (double)profNSt);
    unsafeAtomicAdd(P.tally + P.oCnt + 14, (double)profLanesEv);
    unsafeAtomicAdd(P.tally + P.oCnt + 15, (double)profLanesSt);
    // segment shares are packed into the volume-absorption tally of cells 0..5 (diagnostic build only, omega = 1 runs)
    for (int k = 0; k < 8; ++k) unsafeAtomicAdd(P.tally + P.oVol + k, (double)profSeg[k]);
    unsafeAtomicAdd(P.tally + P.oVol + 8, (double)profNew);
  }
#endif

  // ------------------------------------------------------------------ epilogue: flush tallies + counters
  __syncthreads();
  {
    const ColdArgs ka = cold_args();   // (offsets and sizes straight from the kernarg segment: see cold_args)
    double *const out = ka->P.tally;
    if (ka->P.ldsTallies) {
      const int ncol = ka->P.nx * ka->P.ny;
      const int oUp = ka->P.oUp, oDown = ka->P.oDown, oAbs = ka->P.oAbs;
      for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
        const float u = L.tUp[i], d = L.tDown[i], a = L.tAbs[i];
        if (u != 0.0f) add_global(out + oUp + i, u);
        if (d != 0.0f) add_global(out + oDown + i, d);
        if (a != 0.0f) add_global(out + oAbs + i, a);
      }
    }
    if (ka->P.ldsIntensity) {
      const int nInt = (ka->P.ncomp + 1) * ka->P.nDir * ka->P.nx * ka->P.ny;
      const int oInt = ka->P.oInt;
      for (int i = threadIdx.x; i < nInt; i += blockDim.x) {
        const float v = L.tInt[i];
        if (v != 0.0f) add_global(out + oInt + i, v);
      }
    }
    // nested local-estimate work and the deviate count are per lane; everything else is already per wave
    const double nestedShadow = wave_sum((double)nested.shadow), nestedCalls = wave_sum((double)nested.calls);
    const double draws = wave_sum((double)rng.total());
    flush_counters();
    if ((threadIdx.x & 63) == 0) {
      double *const counters = out + ka->P.oCnt;
      if (nestedShadow != 0.0) unsafeAtomicAdd(counters + I3RC_CNT_SHADOW_STEPS, nestedShadow);
      if (nestedCalls != 0.0) unsafeAtomicAdd(counters + I3RC_CNT_TRACER_CALLS, nestedCalls);
      if (draws != 0.0) unsafeAtomicAdd(counters + I3RC_CNT_RNG_DRAWS, draws);
    }
  }
}

// Test hook: independent tracer calls, one ray per thread.
__global__ void __launch_bounds__(256) trace_rays_kernel(const DevProblem P, long long n, const float *dir, float *pos,
                                                         int32_t *idx, const float *target, float *tau, int32_t *steps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  L.xE = (lds_float *)smem; L.yE = L.xE + P.nx + 1; L.zE = L.yE + P.ny + 1;
  L.tUp = L.tDown = L.tAbs = L.ext = L.dirCos = nullptr;
  for (int i = threadIdx.x; i <= P.nx; i += blockDim.x) L.xE[i] = P.xE[i];
  for (int i = threadIdx.x; i <= P.ny; i += blockDim.x) L.yE[i] = P.yE[i];
  for (int i = threadIdx.x; i <= P.nz; i += blockDim.x) L.zE[i] = P.zE[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ray r;
  r.x = pos[3 * i]; r.y = pos[3 * i + 1]; r.z = pos[3 * i + 2];
  r.dx = dir[3 * i]; r.dy = dir[3 * i + 1]; r.dz = dir[3 * i + 2];
  r.ix = idx[3 * i]; r.iy = idx[3 * i + 1]; r.iz = idx[3 * i + 2];
  r.set_direction(L);
  r.acc = 0.0f;
  const bool hasTarget = target[i] >= 0.0f;
  r.target = target[i];
  int ns = 0;
  StepResult s;
  do { ns++; s = trace_step<GRID_BRICKS>(P, L, r, hasTarget); } while (s == STEP_CONTINUE && ns < (1 << 24));
  pos[3 * i] = r.x; pos[3 * i + 1] = r.y; pos[3 * i + 2] = r.z;
  idx[3 * i] = r.ix; idx[3 * i + 1] = r.iy; idx[3 * i + 2] = r.iz;
  tau[i] = r.acc;
  steps[i] = ns;
}

// Test hook: exact_div / exact_sqrt against the IEEE operations they replace.
__global__ void arith_check_kernel(long long n, const float *num, const float *den, unsigned long long *mismatch) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = num[i], b = den[i];
  const float q = exact_div(a, b, refined_rcp(b)), qRef = a / b;
  if (__float_as_uint(q) != __float_as_uint(qRef)) atomicAdd(&mismatch[0], 1ull);
  const float x = fabsf(a);
  if (__float_as_uint(exact_sqrt(x)) != __float_as_uint(sqrtf(x))) atomicAdd(&mismatch[1], 1ull);
}

// Test hook: raw Philox blocks as the photon streams see them.
__global__ void philox_kernel(uint32_t seed0, uint32_t seed1, long long firstPhoton, long long n, int blocksPerPhoton,
                              uint32_t *out, float *outf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PhiloxStream g;
  g.init(seed0, seed1);
  g.start((uint64_t)(firstPhoton + i));
  for (int b = 0; b < blocksPerPhoton; ++b) {
    const Philox4 o = philox4x32_10(g.id_lo, g.id_hi, (uint32_t)b, 0u, seed0, seed1);
    g.begin_event();   // block b of the stream, as the photon kernel draws it
    const float roles[4] = {g.first(), g.second(), g.path(), g.spare()};
    for (int k = 0; k < 4; ++k) {
      out[(i * blocksPerPhoton + b) * 4 + k] = o.v[k];
      outf[(i * blocksPerPhoton + b) * 4 + k] = roles[k];
    }
  }
}

}  // namespace i3rc
