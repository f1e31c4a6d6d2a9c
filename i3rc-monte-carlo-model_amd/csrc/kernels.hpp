// gfx950 kernels of the I3RC photon tracer.
//
// Mapping (MI355X-first, not a port of the scalar Fortran loop nest):
//   * one photon per LANE, 64 photons in flight per wavefront, persistent waves: a lane that loses its photon
//     (exit, absorption, roulette, tracer drop) gets the next photon index from the wave's reservoir, which is
//     refilled from a device-wide counter with one returning atomic per <= 256 photons;
//   * the reference's three nested data-dependent loops (photon / order of scattering / voxel step,
//     computeRT :452-691 + accumulateExtinctionAlongPath :1690-1806) are flattened into a lane state machine with
//     ballot-gated phases: a VOXEL-STEP phase executed by the lanes that are tracing and an EVENT phase (exit tallies,
//     next photon, one Philox block for the whole wave, scatter / surface / new photon, new optical depth) that runs
//     when enough lanes wait for it;
//   * radiances: an event pushes ONE record into its wave's ray queue (a ring in LDS) and the photon goes on; when the
//     ring holds a wavefront's worth of local-estimate (shadow) rays the wave changes to RAY MODE: all lanes turn
//     (event, direction) pairs into ready rays (a ray whose roulette is already lost is dropped there), lanes take a
//     ready ray as soon as theirs has ended -- shadow rays run with nearly full wavefronts, independent of the photons
//     that caused them; with ONE radiance direction (an event is one ray, and the roulette ends most rays where they are made) there
//     is no ring: the event phase makes its event's ray ready itself and only survivors go to LDS, into a ready store of two
//     wavefronts (template parameter DIRECT);
//   * the batches of a driver's loop share ONE grid (PhiloxBatchStream: a lane carries its photon's batch in its Philox key, a
//     local-estimate ray in its info word; per-batch tally blocks in global memory), so that one batch's tail -- the few photons
//     with a thousand scatterings -- is filled by the next batch's photons;
//   * cell edges (and, when they fit, the extinction grid) are staged in LDS with coalesced loads; a field whose columns
//     each hold one run of one value (the I3RC Landsat scene) is read as one 8-byte record per column; other grids beyond an
//     XCD's L2 are read from a copy in 32-cell bricks; flux and radiance tallies are privatised per workgroup in LDS
//     (ds_add_f64) and flushed once with float64 atomics; large domains tally straight to HBM with float64 atomics;
//   * per-photon Philox4x32-10 streams keyed by (seed, batch) make a photon's path independent of the launch
//     geometry, of every scheduling threshold and of the number of GPUs;
//   * work counters live in scalar registers (advanced by s_bcnt1 of ballots in uniform control flow).
#pragma once
#include <cstddef>
#include "tracer.hpp"

namespace i3rc {

enum LaneState { ST_TRACE = 0, ST_EVENT = 1, ST_DROPPED = 2, ST_NEW = 3, ST_DONE = 4,
                 ST_EXIT = 5 };   // an event known to end the photon (left through the top / onto a black surface): see the turnover quorum

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
__device__ __forceinline__ void add_global(double *p, double v) { unsafeAtomicAdd(p, v); }

// Kernel arguments that are only needed now and then -- the reservoir refill, the counter hand-over, the epilogue --
// are read from the kernarg segment where they are used (scalar loads) instead of living in scalar registers through
// the whole photon loop: the flux kernel wanted 106 of the 102 there are, and every spilled one comes back as a
// v_readlane, i.e. a vector instruction, in the event phase.  (The empty asm keeps the loads from being hoisted.)
struct KernelArgs { DevProblem P; RunArgs A; };   // the kernarg segment of photon_kernel: (P, A, ...)
static_assert(offsetof(KernelArgs, A) == sizeof(DevProblem) && sizeof(DevProblem) % 8 == 0 && alignof(RunArgs) == 8,
              "the second kernel argument must follow the first without padding");
typedef const __attribute__((address_space(4))) KernelArgs *ColdArgs;
typedef __attribute__((address_space(4))) DevProblem ColdProblem;   // (as a template argument: Tally<ColdProblem>)
__device__ __forceinline__ ColdArgs cold_args() {
  ColdArgs k = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}


// BATCHED (fused multi-batch launch): every lane tallies into the block of ITS photon's batch (blk, worked out once per event
// phase), straight in global memory -- the hot words of a small domain are spread over several replicas of the block instead
// of being gathered in LDS, whose partial sums would have to be kept per batch.
template <class PR, bool BATCHED = false>
struct Tally {
  const PR &P;
  const Lds &L;
  double *blk;   // BATCHED: the lane's tally block (else the launch's one buffer, P.tally)
  // (the tally buffer's base addresses stay in scalar registers: reading them from the kernarg segment at every
  // tally -- see cold_args -- was measured: -8 % where the tallies go to global memory, nothing gained elsewhere)
  __device__ __forceinline__ double *base() const { return BATCHED ? blk : P.tally; }
  __device__ __forceinline__ void down(int col, float w) const {
    if (!BATCHED && P.ldsTallies) lds_add(&L.tDown[col], w); else add_global(base() + P.oDown + col, w);
  }
  // intensityByComponent(ix, iy, d, comp) (:574-579, :662-667)
  __device__ __forceinline__ void radiance(int comp, int d, int col, float v) const {
    const int i = (comp * P.nDir + d) * (P.nx * P.ny) + col;
    if (!BATCHED && P.ldsIntensity) lds_add(&L.tInt[i], v); else add_global(base() + P.oInt + i, v);
  }
  // upward flux at the top (:513) or downward flux at the surface (:531): one atomic for either
  __device__ __forceinline__ void boundary(bool top, int col, float w) const {
    if (!BATCHED && P.ldsTallies) lds_add((top ? L.tUp : L.tDown) + col, w);
    else add_global(base() + (top ? P.oUp : P.oDown) + col, w);
  }
  // Absorption (:644-647: fluxAbsorbed(ix, iy) and volumeAbsorption(ix, iy, iz) take the same increment): ONE tally, the cell's.  The
  // column's sum is the sum of its cells' and is formed from them after the launch (absorbed_columns_kernel, i3rc_hip.hip): the same
  // float64 additions in another order.  Scattered float64 atomics execute at the memory side, some 2e10 a second for the whole chip
  // (profiles/r05_atomic_rate.txt): two per scattering made the I3RC cases' absorbing versions (omega = 0.99) two to three times slower
  // than the conservative ones on the Landsat fields -- and nineteen times on the step cloud, whose 512 cells every wave of the chip
  // added to; a domain of few cells gathers them per workgroup in LDS (ldsVolume).
  __device__ __forceinline__ void absorbed(int cell, float w) const {
    if (!BATCHED && P.ldsVolume) lds_add(&L.tVol[cell], w); else add_global(base() + P.oVol + cell, w);
  }
  __device__ __forceinline__ bool volume_in_lds() const { return !BATCHED && P.ldsVolume; }
  __device__ __forceinline__ void absorbed_sum(int cell, double sum) const { add_global(base() + P.oVol + cell, sum); }   // (a lane's run in one cell: photon_kernel)
};

// computeIntensityContribution :1419-1611 for one event; adds straight into intensityByComponent.
template <int GRID, class Rng, class PR>
__device__ __forceinline__ void intensity_contribution(const PR &P, const Lds &L, Rng &rng, NestedCounters &cnt,
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
      const CompTables ct = load_tables(P.comp[component - 1]);
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
      if (trace_step_lazy<GRID, false, true>(P, L, r, stage != 0) == STEP_CONTINUE) continue;   // (lazy: only a second leg needs the arrival's position, see the service phase of photon_kernel)
      const float tauB = r.acc;
      const bool outTop = r.iz >= zIndexMax;
      if (stage == 0) con = tauB >= 0.0f ? (weight * normPF) * expf(-tauB) : 0.0f;
      else if (stage == 1) {
        const float r2 = rng.next();
        con = (r2 <= kPi * normPF / P.zetaMin && outTop) ? weight * P.zetaMin / kPi : 0.0f;
      } else if (stage == 2) {
        if (outTop && tauB >= 0.0f) con = (weight * normPF) * expf(-tauB);
        else if (tauB >= 0.0f && r.iz >= 1) {   // second leg, up to the free path (not from below the grid: see the light phase)
          finish_arrival(r);                     // (the first leg has arrived at tauMax inside the grid: nothing else gets here)
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
    const Tally<PR> tl{P, L, nullptr};
    tl.radiance(component, d, (r.iy - 1) * P.nx + (r.ix - 1), con);
  }
}

template <class Rng>
struct RngInit;
template <>
struct RngInit<PhiloxStream> {
  static __device__ __forceinline__ void init(PhiloxStream &g, const RunArgs &A) { g.init(A.seed0, A.seed1); }
  template <class AR>
  static __device__ __forceinline__ void start(PhiloxStream &g, const AR &A, long long i) {
    g.start((uint64_t)(A.firstPhoton + i));
  }
};
template <>
struct RngInit<PhiloxBatchStream> {
  static __device__ __forceinline__ void init(PhiloxBatchStream &g, const RunArgs &A) { g.init(A.seed0, A.seed1); }
};
template <>
struct RngInit<ReplayStream> {
  static __device__ __forceinline__ void init(ReplayStream &g, const RunArgs &A) { g.init(A.randoms, A.nRandoms); }
  template <class AR>
  static __device__ __forceinline__ void start(ReplayStream &g, const AR &A, long long i) { g.start(A.drawStart[i]); }
};

// Wave-private reservoir of photon indices: one returning atomic per `chunk` photons instead of one per respawn
// round (the returning atomic costs microseconds; every wave would pay it in ~97 % of its event phases).  The two
// bounds are wave-uniform and held in scalar registers (readfirstlane tells the compiler so).
struct Reservoir {
  long long next, end;   // (XCD-aware order: positions in the sorted list; end < 0: every tile has run dry)
  // (Shorter chunks towards the end of a launch -- what is left shared out over all waves, down to one photon per lane -- were
  // tried in round 3: nothing at 1e8 photons, -1 % at 1.25e7, -6 % at 1e6.  A launch ends with its longest photon histories,
  // not with the last chunks.)
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
  // Fused multi-batch launch: the work counter hands out chunk numbers; a chunk lies within one batch (RunArgs).  `batch` is
  // the batch of the photons in hand (wave-uniform); end < 0 once the launch's chunks have run out or the host has called
  // the launch off (a look-ahead that is not wanted any more ends within one chunk per wave).
  unsigned batch = 0u;
  __device__ __forceinline__ void refill_batched() {   // call in uniform control flow only
    const ColdArgs k = cold_args();
    unsigned long long *const counter = k->A.workCounter;
    const unsigned chunk = (unsigned)k->A.chunk, perBatch = k->A.chunksPerBatch;
    const unsigned long long total = (unsigned long long)k->A.nBatches * perBatch;
    const long long nPhotons = k->A.nPhotons;
    unsigned long long c = total;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_load(k->A.abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
      c = atomicAdd(counter, 1ull);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__shfl((unsigned)c, 0, 64));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)__shfl((unsigned)(c >> 32), 0, 64));
    c = ((unsigned long long)hi << 32) | lo;
    if (c >= total) { next = 0; end = -1; return; }
    const unsigned b = (unsigned)(c / perBatch), j = (unsigned)(c - (unsigned long long)b * perBatch);
    batch = b;
    next = (long long)j * chunk;
    end = next + chunk < nPhotons ? next + chunk : nPhotons;
  }
  // XCD-aware order: positions in RunArgs::slabIds instead of photon numbers.  A wave takes from the slab of its own XCD
  // (whose L2 then holds that eighth of the field) and goes round the other slabs when that one has run dry.  How many
  // slabs of its round have run dry the wave keeps in LDS (read and written at refills only: no register for it in
  // kernels that have none to spare).
  __device__ __forceinline__ void refill_slabs(int *tried) {   // call in uniform control flow only
    const ColdArgs k = cold_args();
    SlabMeta *const m = k->A.slabMeta;
    const unsigned chunk = (unsigned)k->A.chunk;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    next = end = -1;   // (nothing left: avail = 0 and end < 0)
    int slabTry = __builtin_amdgcn_readfirstlane(*tried);
    for (; slabTry < 8; ++slabTry) {
      const int s = (int)((xcc + (unsigned)slabTry) & 7u);
      const unsigned cnt = m->count[s], off = m->offset[s];
      unsigned base = 0;
      if ((threadIdx.x & 63) == 0) base = atomicAdd(&m->take[s], chunk);
      base = __builtin_amdgcn_readfirstlane((unsigned)__shfl(base, 0, 64));
      if (base < cnt) {
        next = (long long)off + base;
        end = (long long)off + (base + chunk < cnt ? base + chunk : cnt);
        break;
      }
    }
    if ((threadIdx.x & 63) == 0) *tried = slabTry;
  }
};

// The two passes that sort a launch's photons by start slab: a photon's start position is the first two deviates of its own
// stream (block 0, words 0 and 1: photon_kernel, part C), so its slab -- one of eight tiles of the domain -- is known before it
// is traced.  Pass 1 counts per workgroup (each workgroup owns a contiguous range of the launch's photons), the scan turns
// the counts into every workgroup's first position in every slab's list, pass 2 writes the photon numbers there: no global
// atomics, the same list for the same launch every time.  Both passes recompute the photon's first Philox block (about 1 ms
// for 1e8 photons; a first version with one global counter per slab took 490 ms: 1.5 million waves on eight words).
constexpr int kSlabSortBlocks = 2048;
__device__ __forceinline__ int start_slab(uint32_t seed0, uint32_t seed1, unsigned long long photon, int tx, int ty) {
  // tx x ty = 8 tiles of the domain (the host picks the squarest tiling: the smaller a tile's perimeter, the fewer photons
  // leave it): measured on the 128 x 128 Landsat field 2 x 4 6.50e8, 4 x 2 6.40e8, 1 x 8 6.35e8 photons/s
  const Philox4 o = philox4x32_10((uint32_t)photon, (uint32_t)(photon >> 32), 0u, 0u, seed0, seed1);
  const int sx = (int)(u32_to_unit_float(o.v[0]) * (float)tx), sy = (int)(u32_to_unit_float(o.v[1]) * (float)ty);
  return (sy >= ty ? ty - 1 : sy) * tx + (sx >= tx ? tx - 1 : sx);
}
__global__ void __launch_bounds__(256) slab_count_kernel(uint32_t seed0, uint32_t seed1, long long firstPhoton, long long n, long long span,
                                                         int tx, int ty, unsigned *blockCounts) {
  __shared__ unsigned cnt[8];
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0u;
  __syncthreads();
  const long long lo = (long long)blockIdx.x * span, hi = lo + span < n ? lo + span : n;
  unsigned mine[8] = {};   // (per wave: only lane 0's copy is used)
  for (long long base = lo; base < hi; base += 256) {   // uniform trip count within the block
    const long long i = base + threadIdx.x;
    const int s = i < hi ? start_slab(seed0, seed1, (unsigned long long)(firstPhoton + i), tx, ty) : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine[k] += (unsigned)__popcll(__ballot(s == k));
  }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&cnt[k], mine[k]);
  __syncthreads();
  if (threadIdx.x < 8) blockCounts[blockIdx.x * 8 + threadIdx.x] = cnt[threadIdx.x];
}
// one thread per slab: the slab's size, then (after all sizes are known) every workgroup's first position in its list
__global__ void slab_scan_kernel(int nBlocks, const unsigned *blockCounts, SlabMeta *m, unsigned *blockBase) {
  __shared__ unsigned total[8];
  const int s = (int)threadIdx.x;
  unsigned c = 0;
  for (int b = 0; b < nBlocks; ++b) c += blockCounts[b * 8 + s];
  total[s] = c;
  __syncthreads();
  unsigned off = 0;
  for (int k = 0; k < s; ++k) off += total[k];
  m->count[s] = c; m->offset[s] = off; m->take[s] = 0u; m->fill[s] = 0u;
  for (int b = 0; b < nBlocks; ++b) { blockBase[b * 8 + s] = off; off += blockCounts[b * 8 + s]; }
}
__global__ void __launch_bounds__(256) slab_fill_kernel(uint32_t seed0, uint32_t seed1, long long firstPhoton, long long n, long long span,
                                                        int tx, int ty, const unsigned *blockBase, uint32_t *ids) {
  __shared__ unsigned cursor[8];
  if (threadIdx.x < 8) cursor[threadIdx.x] = blockBase[blockIdx.x * 8 + threadIdx.x];
  __syncthreads();
  const long long lo = (long long)blockIdx.x * span, hi = lo + span < n ? lo + span : n;
  for (long long base = lo; base < hi; base += 256) {
    const long long i = base + threadIdx.x;
    const int s = i < hi ? start_slab(seed0, seed1, (unsigned long long)(firstPhoton + i), tx, ty) : -1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long mask = __ballot(s == k);
      if (mask == 0ull) continue;
      const int leader = __builtin_ctzll(mask);
      unsigned at = 0;
      if ((int)(threadIdx.x & 63) == leader) at = atomicAdd(&cursor[k], (unsigned)__popcll(mask));
      at = (unsigned)__shfl(at, leader, 64);
      if (s == k) ids[at + (unsigned)lanes_below(mask)] = (uint32_t)i;
    }
  }
}

#ifndef I3RC_RADIANCE_WAVES
#define I3RC_RADIANCE_WAVES 5   /* (the Landsat + 7 directions case gains 9 % over 4; 6 would spill) */
#endif
#ifndef I3RC_MIN_WAVES
#define I3RC_MIN_WAVES 5
#endif
// The specialised flux kernels need 54 vector registers: told to plan for eight waves per SIMD (instead of five) the
// compiler schedules them differently -- step cloud 3.13 -> 3.27e9 photons/s on the same box, 32 layers 2.56 -> 2.67e9,
// radar 640 flux 1.24 -> 1.29e9, Landsat-36 -0.5 %; the bricked kernels (at most five workgroups per CU anyway) lose 1 %.
#ifndef I3RC_FLUX_WAVES
#define I3RC_FLUX_WAVES 8
#endif
// (the fused multi-batch kernels carry five more vector registers per lane -- the batch and the per-lane counts -- and want 66:
// planned for seven waves per SIMD they keep them all; for eight, two go to scratch and nothing is gained: 2.97 against
// 2.83 ... 3.06e9 photons/s on the step cloud, 1.22 against 1.08e9 on the radar field, within the noise elsewhere)
#ifndef I3RC_FUSED_WAVES
#define I3RC_FUSED_WAVES 7
#endif
// (the radiance kernels for several components carry the stream's cursor and the component on top of the one-component kernels' 96
// registers: planned for five waves per SIMD they keep 4 ... 14 of them in scratch; measured against four waves: DESIGN.md section 8)
#ifndef I3RC_MULTI_WAVES
#define I3RC_MULTI_WAVES 4
#endif
// GENERAL = false is the specialisation for the common problem class -- regular grid, ray tracing, one component,
// Lambertian albedo (no BRDF grid), Directional source, production RNG: the rare paths (grid searches, periodic
// re-wrapping loops, max-cross-section moves, BRDF lookups, component selection) are compiled out, which shrinks the
// loop's code and its scalar-register pressure.  GENERAL = true keeps every path behind run-time switches.
// (radiance kernels keep two rays per lane live -- the photon's and a shadow ray's: 4 waves per SIMD give them 128 vector registers)
// TBL (round 3, specialised flux kernels): workgroups of 1024 threads, two per compute unit, that also keep the 40 KB of the
// inverse table's cosines (one entry) in LDS: the two dependent table reads of a scattering then come from LDS instead of
// L2 or beyond (+1.6 % on the step cloud, +22 % on Landsat-36; the launch chooses it: i3rc_hip.hip).
// DIRECT (round 4, radiance kernels, ONE radiance direction): no event ring.  With one direction an event is one ray, and with the
// roulette most rays end where they are made: the event phase itself turns its event into a ready ray (the EXPAND arithmetic, at
// the event phase's lane count -- what EXPAND had with a ring that an event phase half fills), only the survivors go to LDS, and
// the LDS the ring took pays for a ready store of two wavefronts: rays are traced when a wavefront of SURVIVORS has gathered.
// MULTI (round 5; with GENERAL = false, radiance kernels): the common class WIDENED by what production domains bring -- several components,
// an irregular x / y grid (only a photon's start looks its cell up: the tracer reads edges, whatever their spacing), a gridded surface
// (only a reflection looks its reflectance up) -- while max cross-section, the other photon sources and the replay stream stay with the
// general kernels.  For domains of SEVERAL components -- cloud + aerosol + gas is what
// Tools/PhysicalPropertiesToDomain.f95 makes --: the component of a scattering by a compare chain over the cell's cumulative
// extinctions (:637-638: the findIndex of (/0, cumulativeExt/), largest i with table(i) <= deviate), single-scattering albedo and
// phase-function entry read per cell and component (:642, :684-686), tables per component.  Everything else -- regular grid, ray
// tracing, no BRDF grid, Directional source -- as compiled out as in the one-component kernels; the deviates are drawn as the general
// kernels draw them (the component's from the stream's cursor), so that a MULTI launch traces the general kernels' photons.
template <class Rng, bool INTENSITY, bool GENERAL, int GRID, bool TBL = false, bool DIRECT = false, bool MULTI = false>
__global__ void __launch_bounds__(TBL ? 1024 : 256, INTENSITY ? (GENERAL ? 3 : (MULTI ? I3RC_MULTI_WAVES : I3RC_RADIANCE_WAVES)) : ((GENERAL || GRID == GRID_BRICKS) ? I3RC_MIN_WAVES : ((Rng::kBatched && !TBL) ? (MULTI ? I3RC_FUSED_WAVES - 2 : I3RC_FUSED_WAVES) : I3RC_FLUX_WAVES))) photon_kernel(const DevProblem P, const RunArgs A, const int evThreshold, const int lightThreshold) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  {
    // (one carve-up for the kernel and for the host's allocation: lds_plan, tracer.hpp)
    const LdsPlan lp = lds_plan(P, INTENSITY && !Rng::kReplay, DIRECT, GRID, INTENSITY, TBL ? 16 : 4, 0);
    lds_float *const base = (lds_float *)smem;
    const int ncol = P.nx * P.ny;
    L.xE = base + lp.xE; L.yE = base + lp.yE; L.zE = base + lp.zE;
    L.tUp = (lds_tally *)(base + lp.tallies); L.tDown = L.tUp + ncol; L.tVol = (lds_tally *)(base + lp.tVol);
    L.dirCos = base + lp.dirCos; L.dirTab = base + lp.dirTab; L.queue = base + lp.queue;
    L.tInt = (lds_tally *)(base + lp.tInt); L.ext = base + lp.ext; L.cosTab = base + lp.cosTab;
  }
  for (int i = threadIdx.x; i < 3 * P.nDir; i += blockDim.x) L.dirCos[i] = P.dirCos[i];
  // coalesced staging of the edge vectors (and the extinction grid when it fits)
  for (int i = threadIdx.x; i <= P.nx; i += blockDim.x) L.xE[i] = P.xE[i];
  for (int i = threadIdx.x; i <= P.ny; i += blockDim.x) L.yE[i] = P.yE[i];
  for (int i = threadIdx.x; i <= P.nz; i += blockDim.x) L.zE[i] = P.zE[i];
  if (P.ldsTallies)
    for (int i = threadIdx.x; i < 2 * P.nx * P.ny; i += blockDim.x) L.tUp[i] = (tally_t)0;
  if (P.ldsVolume)
    for (int i = threadIdx.x; i < P.nx * P.ny * P.nz; i += blockDim.x) L.tVol[i] = (tally_t)0;
  if (GRID == GRID_LDS) {
    const int ncell = P.nx * P.ny * P.nz;
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) L.ext[i] = P.totalExt[i];
  }
  if (GRID == GRID_COLBASE)                 // the base profile of column records over a per-layer value (DevProblem::colBase)
    for (int i = threadIdx.x; i < P.nz; i += blockDim.x) L.ext[i] = P.colBase[i];
  if (GRID == GRID_BRICKS && !INTENSITY) {   // the clear-air map of a bricked field (DevProblem::clearMap): flux kernels, see cell_extinction
    const int nWords = P.clearNx * (((P.ny - 1) >> P.clearShift) + 1);
    for (int i = threadIdx.x; i < nWords; i += blockDim.x) L.ext[i] = __uint_as_float(P.clearMap[i]);
  }
  if (P.ldsIntensity)
    for (int i = threadIdx.x; i < (P.ncomp + 1) * P.nDir * P.nx * P.ny; i += blockDim.x) L.tInt[i] = (tally_t)0;
  if (TBL) {
    const float *src = P.comp0.invCos + (size_t)(P.uniformPf >= 1 ? P.uniformPf - 1 : 0) * P.comp0.nInv;   // (else the table has one entry)
    for (int i = threadIdx.x; i < P.comp0.nInv; i += blockDim.x) L.cosTab[i] = src[i];
  }
  __syncthreads();
  if (INTENSITY && !Rng::kReplay) {   // what a ray of each radiance direction derives from it, once per workgroup (Ray::load_direction)
    for (int d = threadIdx.x; d < P.nDir; d += blockDim.x) {
      Ray t;
      t.dx = L.dirCos[3 * d]; t.dy = L.dirCos[3 * d + 1]; t.dz = L.dirCos[3 * d + 2];
      t.set_direction(L);
      t.store_direction(L.dirTab + 16 * d);
    }
    __syncthreads();
  }

  constexpr bool REPLAY = Rng::kReplay;        // per-photon fates are recorded by i3rc_hip_run_replay only
  constexpr bool BATCHED = Rng::kBatched;      // fused multi-batch launch: every lane knows its photon's batch (rng.batch)
  static_assert(!BATCHED || !GENERAL, "fused multi-batch launches: specialised kernels");
  static_assert(!MULTI || (!GENERAL && !TBL && !Rng::kReplay && (INTENSITY || BATCHED)), "MULTI: the widened class -- radiance kernels, and the fused flux kernels (plain flux launches of the class run the general flux kernel)");
  // Work counters of a fused launch.  Flux kernels: exact per batch, gathered per lane (below).  Radiance kernels have no
  // vector register to spare for that: their counters stay per WAVE and are handed to the batch whose photons the wave was
  // given last -- photons and dropped photons (what the normalisation needs) are exact per batch, the others over the group.
#ifdef I3RC_FUSED_WAVE_COUNTS   /* (measurement knob: the fused flux kernels count per wavefront as the radiance kernels do) */
  constexpr bool LANE_COUNTS = false;
#else
  constexpr bool LANE_COUNTS = BATCHED && !INTENSITY;
#endif
  constexpr bool NEED_PID = REPLAY || GENERAL; // explicit photon sources are indexed by photon number
  const size_t ncell = (size_t)P.nx * P.ny * P.nz;
  const bool rayTracing = GENERAL ? (P.useRayTracing != 0) : true;
  const bool useBDRF = (GENERAL || MULTI) ? (P.useBDRF != 0) : false;
  const bool multiComp = (GENERAL || MULTI) ? (P.ncomp > 1) : false;   // (one component: no deviate is drawn for the choice, :637 -- the widened class too)
  const bool directional = GENERAL ? (A.srcKind == 0) : true;
  // Directional photons all start at z = z0 + (1 - spacing(1)) (zMax - z0): their start layer is wave-uniform
  const float zStart = P.z0 + (1.0f - spacingf(1.0f)) * (P.zMax - P.z0);
  int izStart = 1;
  find_z<true>(P, L, zStart, izStart);   // (once per wave: the specialised kernels take irregular layers too)
  const float rcpDeltaX = refined_rcp(P.deltaX), rcpDeltaY = refined_rcp(P.deltaY);
  const float surfaceZ = P.z0 + spacingf(P.z0);
  const bool blackSurface = !Rng::kReplay && !useBDRF && !(P.albedo > kTiny) && !INTENSITY;   // arrival at the surface ends the photon

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
  // Absorption where the tally goes to global memory: what a photon's CONSECUTIVE scatterings in one cell add to it leaves as one
  // atomic -- when the next scattering lies in another cell, or the photon ends (so that a fused launch's lane never holds a sum of
  // another batch's block).  In clouds whose cells are optically thick (an LES field: 55 m cells, free paths of 10 m) a photon is
  // scattered several times in a row in the cell it is in; the memory side takes some 2e10 scattered atomics a second, which one per
  // scattering reaches at 1.1e9 photons/s (Tally::absorbed).  The sum is float64: the same additions in another grouping.
#ifndef I3RC_MERGE_ABSORPTION
#define I3RC_MERGE_ABSORPTION 1
#endif
  // (flux kernels: the radiance kernels, the general ones at three waves per SIMD and the fused table-in-LDS kernels at eight have no
  // three registers to spare -- with them they spilled 6 ... 20 bytes a lane)
  constexpr bool MERGE = I3RC_MERGE_ABSORPTION != 0 && !INTENSITY && !(Rng::kBatched && TBL);
  int pendCell = -1;
  double pendSum = 0.0;
  long long pid = -1;                 // photon number within the launch (NEED_PID builds)
  int fate = -1, fateCol = -1;        // REPLAY builds
  float fateW = 0.0f;
  // XCD-aware photon order: kernels on fields beyond an XCD's L2 (bricks), when the host has sorted the launch's photons
  constexpr bool SLABS = GRID == GRID_BRICKS && !Rng::kReplay;
  __shared__ int slabsTried[16];           // per wave (see Reservoir::refill_slabs; 16 waves in the 1024-thread instantiations)
  Reservoir res;
  if constexpr (BATCHED) res.refill_batched();
  else if (SLABS && A.slabIds != nullptr) {
    if ((threadIdx.x & 63) == 0) slabsTried[threadIdx.x >> 6] = 0;
    res.refill_slabs(&slabsTried[threadIdx.x >> 6]);
  } else res.refill();

  // ---- BATCHED: work counters per batch --------------------------------------------------------------------------------
  // The wave's scalar counters (wc) cannot tell the batches of its lanes apart: they only steer the thresholds here.  What
  // a batch's counter block is owed is gathered per LANE -- voxel steps in a register of their own, scatterings | roulette
  // plays and surface arrivals | exits through the top as two pairs of 16-bit counts, the deviates in the stream's own
  // count -- and handed over (a handful of atomics) when the lane's next photon belongs to another batch, when a count is
  // about to leave its 16 bits, and at the end.  Tracer calls need no count: every photon ends in exactly one of four ways
  // (top, surface, roulette, tracer error) and every event that does not end it starts a trace, so a batch's calls are its
  // scatterings + surface arrivals + exits through the top + dropped photons.  Photons are counted where they are handed
  // out (wave-uniform: resTaken), dropped ones at once (rare).
  uint32_t accSteps = 0u, accA = 0u, accB = 0u;   // steps; scatterings | roulette << 16; surface arrivals | exits top << 16
  uint32_t resTaken = 0u;                         // photons of the reservoir's batch handed out since the last hand-over
  const unsigned rep = BATCHED ? blockIdx.x % (unsigned)A.replicas : 0u;
  auto lane_block = [&](uint32_t b) -> double * {   // tally block of batch b for this workgroup (RunArgs)
    const ColdArgs k = cold_args();
    return k->P.tally + (size_t)(b * (unsigned)k->A.replicas + rep) * (size_t)(unsigned)k->A.blockStride;
  };
  auto counter_block = [&](uint32_t b) -> double * {   // ... and its counter block
    return cold_args()->A.counterBlocks + (size_t)(b * (unsigned)kCounterReplicas + (blockIdx.x & (unsigned)(kCounterReplicas - 1))) * I3RC_NUM_COUNTERS;
  };
  // (uniform control flow: the lanes with `want` hand their counts over.  The counts of all lanes that share a batch --
  // nearly always all of them -- are summed across the wave first and one lane adds them to the batch's counter block:
  // sixty-four lanes adding to the same seven words one after the other cost a launch 6 ms at its end and made chunks of
  // 256 photons half as fast as chunks of 1024.)
  auto flush_lanes = [&](bool want) {
    if constexpr (BATCHED) {
      unsigned long long todo = __ballot(want);
      while (todo != 0ull) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)rng.batch, (int)__builtin_ctzll(todo));
        const bool sel = want && rng.batch == b;
        // (one sum at a time, each ending in a scalar register: six at once cost the kernel six spilled vector registers)
        auto wave_total = [](uint32_t x) -> uint32_t {
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off, 64);
          return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
        };
        const uint32_t steps = wave_total(sel ? accSteps : 0u);
        const uint32_t scat = wave_total(sel ? (accA & 0xffffu) : 0u), roul = wave_total(sel ? (accA >> 16) : 0u);
        const uint32_t surf = wave_total(sel ? (accB & 0xffffu) : 0u), top = wave_total(sel ? (accB >> 16) : 0u);
        const uint32_t draws = wave_total(sel ? rng.take_used() : 0u);
        if ((threadIdx.x & 63) == 0) {
          double *const c = counter_block(b);
          if (steps) unsafeAtomicAdd(c + I3RC_CNT_CELL_STEPS, (double)steps);
          if (scat) unsafeAtomicAdd(c + I3RC_CNT_SCATTERINGS, (double)scat);
          if (roul) unsafeAtomicAdd(c + I3RC_CNT_ROULETTE, (double)roul);
          if (surf) unsafeAtomicAdd(c + I3RC_CNT_SURFACE_HITS, (double)surf);
          if (top) unsafeAtomicAdd(c + I3RC_CNT_EXITS_TOP, (double)top);
          if (scat + surf + top) unsafeAtomicAdd(c + I3RC_CNT_TRACER_CALLS, (double)(scat + surf + top));
          if (draws) unsafeAtomicAdd(c + I3RC_CNT_RNG_DRAWS, (double)draws);
        }
        if (sel) accSteps = accA = accB = 0u;
        todo &= ~__ballot(sel);
      }
    }
  };
  auto hand_over_taken = [&](uint32_t b) {          // uniform control flow
    if (resTaken != 0u && (threadIdx.x & 63) == 0)
      unsafeAtomicAdd(counter_block(b) + I3RC_CNT_PHOTONS, (double)resTaken);
    resTaken = 0u;
  };

  // ---- Radiance (local estimate) through a per-wave RAY QUEUE -------------------------------------------------------
  // A scattering or reflection event does not trace its D local-estimate (shadow) rays itself and the photon does not
  // wait for them: the event is pushed as ONE record (position, cell, weight, incoming direction, table entry, and the
  // photon's Philox coordinates) into the wave's ring in LDS, and the photon goes on.  When the wave holds enough rays
  // to fill the wavefront (or the ring has no room for the next event phase, or the photons are finished) it changes to
  // RAY MODE, which has three phases:
  //   EXPAND   all 64 lanes turn one (event, direction) pair each into a ready-made ray -- phase-function factor (acos,
  //            table look-up), the ray's own Philox block, free path, roulette stage and target -- and put it into the
  //            wave's buffer of ready rays: the expensive part of a ray's start runs with full wavefronts;
  //   SERVICE  lanes whose ray has ended tally it and take the next ready ray (a dozen LDS reads), run when `liThr`
  //            lanes wait;
  //   STEP     one voxel step of every lane's ray -- nearly all lanes, since a lane whose ray ends is served soon.
  // Photons and shadow rays have their own registers (r, sr) and their own loops: the wave runs the photon loop until a
  // change of mode is due, then the ray loop, and back.
  // When nothing is left to hand out and fewer than kLowWater rays are still under way the wave goes back to its photons;
  // the unfinished rays go back to the ready buffer as they are and are resumed at the next change.  Every ray draws its two
  // deviates from a Philox block of its own, counter (photon, block of the event, direction + 1 in the fourth word; the
  // photon's own stream has 0 there): radiances do not depend on the schedule either.  Ray starts and ends use the
  // hardware log / exp / reciprocal (2 ulp): these are weights of a Monte Carlo estimate, not trajectories.
  // The replay build and max cross-section keep the reference's nested order (intensity_contribution).
  constexpr bool DEFER = INTENSITY && !Rng::kReplay;
  // (max cross-section moves the photon inside the event: nested order there)
#ifdef I3RC_NESTED_BUILD
  // measurement build (tools/variant_bench.py build nested="-DI3RC_NESTED_BUILD"): the GENERAL kernels keep the reference's nested order
  // -- every ray traced where its event happens, the roulette played after the trace, libm in the weights -- with the production random
  // streams: a second implementation of :1419-1611 on the device to hold the ray queue against (profiles/r04_parity_xl.txt).  As a
  // run-time switch it cost the general radiance kernels nine spilled vector registers.
  const bool defer = DEFER && rayTracing && !GENERAL;
#else
  const bool defer = DEFER && rayTracing;
#endif
  enum { R_EMPTY = 0, R_TRACE = 1, R_ENDED = 2 };
#ifndef I3RC_STEP_AHEAD
#define I3RC_STEP_AHEAD 2
#endif
#ifndef I3RC_LOW_WATER
#define I3RC_LOW_WATER 64   /* = the ready buffer: with nothing left to expand, a wave leaves its rays unless a whole wavefront of them is in hand */
#endif
#ifndef I3RC_THIRD_STEP
#define I3RC_THIRD_STEP 4   /* a third ray step per pass when the service phase is this much further away: Landsat + 7 directions +2.4 % */
#endif
#ifndef I3RC_TURN_MIN
#define I3RC_TURN_MIN 4
#endif
#ifndef I3RC_TURN_FORCE
#define I3RC_TURN_FORCE 12
#endif
#ifndef I3RC_PHOTON_STEP_AHEAD
#define I3RC_PHOTON_STEP_AHEAD 64   /* off: measured -1.6 % (step cloud) ... +2.8 % (Landsat-36), -3 % on the radar field */
#endif
  // measured (Landsat + 7 directions, 2e7 photons, before the lazy roulette): low water 16 / 32 / 48 / 56 / 64 -> 3.1 / 4.2 / 4.6 / 4.7 /
  // 4.7e7 photons/s; two steps per pass +8 %.  With the lazy roulette (most rays end in EXPAND): low water 40 / 48 / 56 / 64 ->
  // 8.2 / 9.1 / 9.2 / 9.4e7 (radar-64 + nadir 5.8 / 5.9 / 6.1 / 6.2e8); expand batch 16 / 32 / 48 / 64 -> radar-64 5.8 / 6.1 /
  // 6.3 / 6.4e8; both at 64: +5.5 % (radar-64), +3 % (Landsat + 7 directions), +4 % (radar 640 + nadir)
  constexpr int kTurnMin = Rng::kReplay ? 1 : I3RC_TURN_MIN, kTurnForce = Rng::kReplay ? 1 : I3RC_TURN_FORCE;
#ifndef I3RC_EXPAND_BATCH
#define I3RC_EXPAND_BATCH 64   /* = the ready buffer: expand when it is empty, a whole wavefront at a time */
#endif
  constexpr int kExpandBatch = I3RC_EXPAND_BATCH;   // an expand phase runs when the ready buffer has room for this many rays
  // DIRECT: rays are traced once this many survivors are ready (or when the next event phase's rays would not fit into the store
  // of 128: wantSlots), and the wave goes back to its photons when nothing is left to hand out and fewer than
  // kDirectLeave rays are still under way (those few go back to the store: with rays of one or two voxel steps, a wave that
  // left with a wavefront's worth under way -- the ring mode's rule -- would write back and take up again most of its rays)
#ifndef I3RC_DIRECT_ENTER
#define I3RC_DIRECT_ENTER 96   /* (radar-64 + nadir, 5e7 photons: 64 -> 59.1 ms, 80 -> 58.8, 96 -> 58.2, 112 -> 58.3; leave level 8 ... 48: within 1 %) */
#endif
#ifndef I3RC_DIRECT_LEAVE
#define I3RC_DIRECT_LEAVE 24
#endif
  constexpr int kDirectEnter = I3RC_DIRECT_ENTER, kDirectLeave = I3RC_DIRECT_LEAVE;
  // (an event phase that would not fit -- possible when kDirectEnter is set above kDirectReady - 64 -- sends the wave to its rays first: wantSlots)
  static_assert(!DIRECT || (kDirectReady >= kDirectEnter && (kDirectReady & (kDirectReady - 1)) == 0), "the ready store holds the entry level; a power of two");
  constexpr int kLowWater = I3RC_LOW_WATER, kStepAhead = I3RC_STEP_AHEAD, kPhotonStepAhead = I3RC_PHOTON_STEP_AHEAD;
  bool wantSlots = false, photonsLeft = true;         // wave-uniform
  unsigned qTail = 0u, qHeadEv = 0u, qHeadSub = 0u;   // events pushed / events expanded completely / rays expanded of event qHeadEv
  unsigned rdHead = 0u, rdTail = 0u;                  // ready rays taken / made
  constexpr int kReady = DIRECT ? kDirectReady : kReadyRays;     // slots of the wave's ready store
  lds_float *const qBase = L.queue + (threadIdx.x >> 6) * (kRecWords * P.rayQueueCap + kReadyWords * kReady);
  lds_float *const rdBase = qBase + kRecWords * P.rayQueueCap;   // ready rays: word w of slot s at rdBase[w * kReady + s]
  const unsigned qMask = (unsigned)P.rayQueueCap - 1u;
  const unsigned qMagic = ((1u << 20) + (unsigned)P.nDir - 1u) / (unsigned)(P.nDir > 0 ? P.nDir : 1);   // t / nDir = (t * qMagic) >> 20 for t < nDir + 64 <= 319 (nDir <= 255)
  bool pendingShadow = false;
  float wI = 0.0f, inDx = 0.0f, inDy = 0.0f, inDz = 0.0f;   // of the event being pushed
  int evInfo = 0;

  // Thresholds: fixed when the caller asks for them (> 0), else adapted by every wave to its own photons at every
  // reservoir refill.  Event phase: the longer the photons' own traces (voxel steps per event), the more a
  // lane loses by waiting for others, so the threshold falls with the steps per event: 44 - 2 steps per event for the flux
  // kernels, 44 - 1.2 steps per event for the radiance kernels, within 12 ... 44 (adapt_thresholds; until round 4 58 / sqrt(steps per event)).  Ray mode's service phase: likewise with the length of
  // the shadow rays, 70 / sqrt(steps per ray) within 16..32.  Thresholds only schedule work: no photon path depends on them.
  int evThr = evThreshold > 0 ? evThreshold : -evThreshold;
  int liThr = lightThreshold > 0 ? lightThreshold : -lightThreshold;
  const bool adaptEvent = evThreshold < 0, adaptLight = DEFER && lightThreshold < 0;
  uint32_t raysStarted = 0;   // tracer calls for shadow rays since the last refill (wave-uniform)
  uint32_t raysSkipped = 0;   // local-estimate rays known to contribute nothing before any tracing (lost roulette): whole kernel, per wave
  uint32_t refills = 0;       // visits of the work counter by this wave
  // re-fit the thresholds to what this wave has seen since its last hand-over (uniform control flow only)
  uint32_t raysSeen = 0;      // shadow rays since the last hand-over
  auto adapt_thresholds = [&]() {
    if (adaptEvent) {
      const float events = (float)(wc.scat + wc.photons + wc.surf), steps = (float)wc.steps;
      if (events > 0.0f && steps > 0.0f) {
        // (round 4, the step phase a third cheaper than it was: 44 - slope * steps per event fits the measured optima -- step cloud 36 ... 40
        // at 2.5 steps per event, Landsat-36 28 at 9.5, Landsat-119 18 ... 20 at 15 -- where 58 / sqrt(steps per event) sat below them on
        // the long traces; radiance kernels, whose photons share the wave's time with their rays, want the flatter slope)
#ifndef I3RC_EVTHR_SLOPE_FLUX
#define I3RC_EVTHR_SLOPE_FLUX 2.0f
#endif
#ifndef I3RC_EVTHR_SLOPE_RADIANCE
#define I3RC_EVTHR_SLOPE_RADIANCE 1.2f
#endif
        const int t = (int)(44.0f - (INTENSITY ? I3RC_EVTHR_SLOPE_RADIANCE : I3RC_EVTHR_SLOPE_FLUX) * (steps * __builtin_amdgcn_rcpf(events)));
        evThr = __builtin_amdgcn_readfirstlane(t < 12 ? 12 : (t > 44 ? 44 : t));
      }
    }
    if (adaptLight) {
      raysSeen += raysStarted; raysStarted = 0u;
      if (raysSeen > 0u && wc.shadow > 0u) {
#ifndef I3RC_LITHR_COEF
#define I3RC_LITHR_COEF 70.0f
#endif
#ifndef I3RC_LITHR_MIN
#define I3RC_LITHR_MIN 16
#endif
        const int t = (int)(I3RC_LITHR_COEF * __builtin_amdgcn_rsqf((float)wc.shadow * __builtin_amdgcn_rcpf((float)raysSeen)));
        liThr = __builtin_amdgcn_readfirstlane(t < I3RC_LITHR_MIN ? I3RC_LITHR_MIN : (t > 32 ? 32 : t));
      }
    }
  };
  // hands the wave's work counters over to the tally buffer (uniform control flow only)
  auto flush_counters = [&]() {
    adapt_thresholds();
    raysSeen = 0u;
    if constexpr (BATCHED && !LANE_COUNTS) {   // per wave, to the batch in hand (LANE_COUNTS above)
      uint32_t draws = rng.take_used();
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) draws += (uint32_t)__shfl_xor((int)draws, off, 64);
      if ((threadIdx.x & 63) == 0) {
        double *const c = counter_block(res.batch);
        const uint32_t v[8] = {wc.steps, wc.scat, wc.surf, wc.top, wc.roul, wc.shadow, wc.scat + wc.surf + wc.top + wc.calls, draws};
        const int at[8] = {I3RC_CNT_CELL_STEPS, I3RC_CNT_SCATTERINGS, I3RC_CNT_SURFACE_HITS, I3RC_CNT_EXITS_TOP, I3RC_CNT_ROULETTE,
                           I3RC_CNT_SHADOW_STEPS, I3RC_CNT_TRACER_CALLS, I3RC_CNT_RNG_DRAWS};
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (v[k] != 0u) unsafeAtomicAdd(c + at[k], (double)v[k]);
        if (raysSkipped != 0u) unsafeAtomicAdd(c + I3RC_CNT_RAYS_SKIPPED, (double)raysSkipped);
      }
      raysSkipped = 0u;
    }
    if (!BATCHED && (threadIdx.x & 63) == 0) {   // (BATCHED flux kernels: per lane and batch, see flush_lanes)
      const uint32_t c[9] = {wc.photons, wc.dropped, wc.steps, wc.scat, wc.surf, wc.top, wc.roul, wc.shadow, wc.calls};
      const ColdArgs ka = cold_args();
      double *const counters = ka->P.tally + ka->P.oCnt;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (c[k] != 0u) unsafeAtomicAdd(counters + k, (double)c[k]);
    }
    wc = WaveCounters();
  };
  // One (event, direction) pair -> a ready-made local-estimate ray (:1473-1559): the phase-function factor (acos, table
  // look-up), the ray's own Philox block (counter: photon, block of its event, direction + 1), free path, roulette stage and
  // target.  false: the ray is known to contribute nothing and is not traced (below).  Called by the EXPAND phase of ray mode
  // (the event from the wave's ring) and, in DIRECT kernels, by the event phase itself (the event from registers).
  // The ray's info word: component | direction << 8 | stage << 16 (0: plain local estimate, 1: small contribution, 2 / 3: the two
  // legs of a large one) | roulette won << 18 | batch << 19 (fused launches: the batch of the ray's photon, relative to the launch's first).
  auto make_ray = [&](const auto &Px, const auto &Ax, int info, float inX, float inY, float inZ, uint32_t photonLo, uint32_t photonHi,
                      uint32_t eventBlock, uint32_t batch, int dIdx, float &word6, float &normOut, float &tauFreeOut, float &targetOut) -> bool {
    const int comp = info & 0xff;
    const float uz = L.dirCos[3 * dIdx + 2];
    float norm;
    if (comp < 1) norm = 1.0f / kPi;
    else {
      float proj = 0.0f;
      proj += inX * L.dirCos[3 * dIdx]; proj += inY * L.dirCos[3 * dIdx + 1]; proj += inZ * uz;
      if (fabsf(proj) > 1.0f) proj = copysignf(1.0f, proj);
#ifdef I3RC_FAST_ACOS
      const float ang = fast_acos(proj);
#else
      const float ang = acosf(proj);
#endif
      const int pfi = (int)((unsigned)info >> 16);   // (table entries up to 65535: the record's upper half is unsigned)
      const CompTables ct = (GENERAL || MULTI) ? load_tables(Px.comp[comp - 1]) : load_tables(Px.comp0);
      const float *tab = ((info & 0x100) ? ct.fwdOrig : ct.fwd) + (size_t)(pfi - 1) * ct.nFwd;
      norm = fast_div(lookup_phase_fast(tab, ct.nFwd, ang), (4.0f * kPi) * fabsf(uz));
    }
    int stg = 0;
    bool won = false;
    float tauFree = 0.0f, target = 0.0f;
    if (Px.useRRI) {
      const Philox4 q4 = philox4x32_10(photonLo, photonHi, eventBlock, (uint32_t)dIdx + 1u, Ax.seed0, BATCHED ? Ax.seed1 + batch : Ax.seed1);
      rng.count_draws(2u);
      tauFree = -fast_log(fmaxf(kTiny, u32_to_unit_float(q4.v[0])));
      // small contribution: it counts -- with the weight of the roulette's bound -- with probability pi normPF / zetaMin
      // (:1551-1559); the deviate is compared here, the outcome travels as bit 18 of the ray's info word
      const float r2 = u32_to_unit_float(q4.v[1]);
      if (kPi * norm <= Px.zetaMin) { stg = 1; won = r2 * Px.zetaMin <= kPi * norm; target = tauFree; }
      else { stg = 2; target = -fast_log(fast_div(Px.zetaMin, fmaxf(kTiny, kPi * norm))); }   // tauMax
    }
    word6 = __int_as_float(comp | (dIdx << 8) | (stg << 16) | (won ? 1 << 18 : 0) | (int)(batch << 19));
    normOut = norm; tauFreeOut = tauFree; targetOut = target;
    // A small contribution that has lost its roulette (:1554: the deviate is independent of the path) is 0 whatever
    // the trace would find: such a ray is not traced at all -- the same estimator, evaluated lazily.  (The
    // reference traces first and draws afterwards; the replay build keeps that order.)
    return stg != 1 || won;   // (a small contribution that lost its roulette is dropped; stages 0 and 2 are always traced)
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
  // diagnostic build only (-DI3RC_PROFILE_PHASES, tools/phase_profile.py): where do a wave's cycles go?  Per phase kind:
  // cycles (s_memtime), number of phases, lanes served.  The results leave through the volume-absorption tally.
  enum { PH_EVENT = 0, PH_STEP = 1, PH_RAYSTEP = 2, PH_EXPAND = 3, PH_SERVICE = 4, PH_KINDS = 5 };
#ifdef I3RC_PROFILE_PHASES
  unsigned long long profCycles[PH_KINDS] = {}, profCount[PH_KINDS] = {}, profLanes[PH_KINDS] = {};
  unsigned long long profMark = 0;
  const unsigned long long profStart = __builtin_amdgcn_s_memtime();
#define PROF_BEGIN() (profMark = __builtin_amdgcn_s_memtime())
#define PROF_END(kind, lanes) do { profCycles[kind] += __builtin_amdgcn_s_memtime() - profMark; profCount[kind]++; profLanes[kind] += (unsigned long long)(lanes); } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_END(kind, lanes) do {} while (0)
#endif
  for (;;) {
    if (DEFER) {
      // ============================================================================================== RAY MODE?
      // enough local-estimate rays to fill the wavefront (or no room for the next event phase, or no photons left)?
      const int work = DIRECT ? (int)(rdTail - rdHead) : (int)((qTail - qHeadEv) * (unsigned)P.nDir - qHeadSub) + (int)(rdTail - rdHead);
      photonsLeft = __ballot(st != ST_DONE) != 0ull;
      if (work > 0 && (work >= (DIRECT ? kDirectEnter : 64) || !photonsLeft || wantSlots)) {
        wantSlots = false;
        // this lane's shadow ray: registers of the ray loop only (rays still under way when the wave leaves go back to the
        // ready buffer), so that the photon loop's register allocation knows nothing of them and the other way round
        Ray sr;                                             // (every field is set when a lane takes a ray, before any use: no initial
        int rst = R_EMPTY, sInfo;                           //  values, which would cost a register copy each at every pass of the photon loop)
        float sW, sNorm, sTauFree;                          // state; component | direction << 8 | stage << 16 (bit 24: see EXPAND); weight, phase-function factor, free path
        // (One back edge only: EXPAND runs in a loop of its own and SERVICE goes on into the step phase.  With three ways
        // back to the loop's head -- after an expand, after a service, after the steps -- the compiler kept the rays' 27
        // registers in one set at the head and in another at the latch and copied them to and fro, some forty vector
        // moves in every pass: "Exchanging two register sets")
        for (;;) {
          if constexpr (!DIRECT) for (;;) {
            const int ringRays0 = (int)((qTail - qHeadEv) * (unsigned)P.nDir - qHeadSub);
            const int ready0 = (int)(rdTail - rdHead);
            if (!(ringRays0 > 0 && ready0 <= kReady - kExpandBatch)) break;
            // ------------------------------------------------------------ EXPAND phase: (event, direction) -> ready ray
            const int room = kReady - ready0;
            const int n = ringRays0 < room ? ringRays0 : room;
            const int lane = (int)(threadIdx.x & 63);
            PROF_BEGIN();
            const ColdArgs kx = cold_args();   // (the problem through the kernarg segment, as in the event phase)
            const auto &Px = kx->P;
            const auto &Ax = kx->A;
            bool keep = false;
            float evWord6 = 0.0f, evNorm = 0.0f, evTauFree = 0.0f, evTarget = 0.0f;
            const lds_float *evRec = qBase;
            if (lane < n) {                                                  // next radiance direction (:1473-1510)
              const unsigned t = qHeadSub + (unsigned)lane;
              const unsigned eOff = (t * qMagic) >> 20;
              const int dIdx = (int)(t - eOff * (unsigned)Px.nDir);
              const lds_float *rec = qBase + ((qHeadEv + eOff) & qMask);
              const int cap = Px.rayQueueCap;
              // (fused launches: photon numbers are below 2^32 and the record's twelfth word carries the photon's batch instead)
              const uint32_t word12 = __float_as_uint(rec[12 * cap]);
              keep = make_ray(Px, Ax, __float_as_int(rec[10 * cap]), rec[7 * cap], rec[8 * cap], rec[9 * cap], __float_as_uint(rec[11 * cap]),
                              BATCHED ? 0u : word12, __float_as_uint(rec[13 * cap]), BATCHED ? word12 : 0u, dIdx, evWord6, evNorm, evTauFree, evTarget);
              evRec = rec;
            }
            const unsigned long long keepMask = __ballot(keep);
            if (keep) {
              const int cap = Px.rayQueueCap;
              lds_float *out = rdBase + ((rdTail + (unsigned)lanes_below(keepMask)) & (unsigned)(kReady - 1));
              out[0] = evRec[0]; out[kReady] = evRec[cap]; out[2 * kReady] = evRec[2 * cap];
              out[3 * kReady] = evRec[3 * cap]; out[4 * kReady] = evRec[4 * cap]; out[5 * kReady] = evRec[5 * cap];
              out[6 * kReady] = evWord6;
              out[7 * kReady] = evRec[6 * cap];
              out[8 * kReady] = evNorm; out[9 * kReady] = evTauFree; out[10 * kReady] = evTarget; out[11 * kReady] = 0.0f;
            }
            {   // ring bookkeeping (wave-uniform)
              const unsigned t = qHeadSub + (unsigned)n;
              const unsigned e = (t * qMagic) >> 20;
              qHeadEv += e; qHeadSub = t - e * (unsigned)Px.nDir;
              const unsigned kept = (unsigned)__popcll(keepMask);
              rdTail += kept;
              raysSkipped += (unsigned)n - kept;
            }
            PROF_END(PH_EXPAND, n);
          }
          const int ringRays = DIRECT ? 0 : (int)((qTail - qHeadEv) * (unsigned)P.nDir - qHeadSub);   // rays waiting in the ring, not yet expanded
          const int ready = (int)(rdTail - rdHead);                                        // ready-made rays
          const unsigned long long actMask = __ballot(rst == R_TRACE), endMask = __ballot(rst == R_ENDED);
          const int nAct = (int)__popcll(actMask), nIdle = 64 - nAct;
          // with photons still to run, the wave leaves its rays once there is nothing left to hand out and few are under way
          // (no more rays can be made ready at this point: the ring is empty, or the ready buffer is full)
          // (DIRECT: see kDirectLeave -- nothing left to hand out and only a few rays still under way)
          const bool leaving = DIRECT ? (photonsLeft && ready == 0 && nAct < kDirectLeave)
                                      : (ringRays == 0 && photonsLeft && ready + nAct < kLowWater);   // too few for a wavefront: gather more first
          const bool canServe = endMask != 0ull || (!leaving && ready > 0 && nIdle > 0);
          const bool serve = canServe && (nIdle >= liThr || nAct == 0 || leaving);
          if (serve) {
            // ------------------------------------------------------------ SERVICE phase (shadow-ray ends and starts)
            PROF_BEGIN();
            const ColdArgs kx = cold_args();
            const auto &Px = kx->P;
            bool secondLeg = false;
            if (rst == R_ENDED) {                                           // the ray that just ended (:1517-1596)
              double *rayBlk = nullptr;                                     // (fused launches: the tally block of the ray's batch)
              if constexpr (BATCHED) rayBlk = lane_block((uint32_t)sInfo >> 19);
              const Tally<ColdProblem, BATCHED> tally{Px, L, rayBlk};
              const float tauB = sr.acc;
              const bool outTop = sr.iz >= Px.nz + 1;
              const int comp = sInfo & 0xff, dIdx = (sInfo >> 8) & 0xff, stage = (sInfo >> 16) & 3;
              const float direct = tauB >= 0.0f ? (sW * sNorm) * fast_exp(-tauB) : 0.0f;   // plain local estimate
              const float capped = outTop ? sW * Px.zetaMin * (1.0f / kPi) : 0.0f;          // roulette survivor
              float con = 0.0f;
              if (stage == 0) con = direct;
              else if (stage == 1) con = (sInfo & (1 << 18)) ? capped : 0.0f;
              else if (stage == 2) {
                if (outTop) con = direct;
                else if (tauB >= 0.0f && sr.iz >= 1) {                      // second leg, up to the free path (:1576-1587)
                  // (the first leg has ARRIVED at tauMax inside the grid -- the only way to get here: an exit through the bottom has
                  // iz < 1, a failed trace tauB = -2 -- and its last advance is still to be made: trace_step_lazy.  Every other ray
                  // that ends at its target contributes nothing, wherever it stands: stages 1 and 3 count only through the top.)
                  finish_arrival(sr);
                  sr.acc = 0.0f; sr.target = sTauFree; sInfo |= 1 << 16; rst = R_TRACE; secondLeg = true;   // (stage 2 -> 3)
                }
                // (a first leg that left through the BOTTOM -- a downward radiance direction -- gets no second leg: the
                // reference starts one from outside the grid, reads zPosition(0) / totalExt(:, :, 0) out of bounds and
                // discards the outcome, since only an exit through the top counts; the contribution is 0 either way)
              } else con = capped;
              if (rst == R_ENDED) {
                if (Px.limitContrib && con > Px.maxContrib) {                // :1598-1609
                  add_global(tally.base() + Px.oExc + comp * Px.nDir + dIdx, con - Px.maxContrib);
                  con = Px.maxContrib;
                }
                // (most rays of the roulette end without a contribution: no atomic for adding nothing -- on the radar
              // field the radiance atomics were 1.4 KB of HBM writes per photon and half of the waves' cycles were waits)
              if (con != 0.0f) tally.radiance(comp, dIdx, (sr.iy - 1) * Px.nx + (sr.ix - 1), con);
                rst = R_EMPTY;
              }
            }
            // hand ready rays to the free lanes, in order
            const unsigned long long freeMask = __ballot(rst == R_EMPTY);
            const int nFree = (int)__popcll(freeMask);
            const int take = leaving ? 0 : (nFree < ready ? nFree : ready);   // (a wave about to leave ends its rays but takes no new ones)
            const int rank = lanes_below(freeMask);
            if (rst == R_EMPTY && rank < take) {
              const lds_float *in = rdBase + ((rdHead + (unsigned)rank) & (unsigned)(kReady - 1));
              sr.x = in[0]; sr.y = in[kReady]; sr.z = in[2 * kReady];
              sr.ix = __float_as_int(in[3 * kReady]); sr.iy = __float_as_int(in[4 * kReady]); sr.iz = __float_as_int(in[5 * kReady]);
              sInfo = __float_as_int(in[6 * kReady]);
              sW = in[7 * kReady]; sNorm = in[8 * kReady]; sTauFree = in[9 * kReady];
              sr.target = in[10 * kReady]; sr.acc = in[11 * kReady];
#ifdef I3RC_NO_DIRTAB   /* (measurement knob: the direction's derived values worked out at every ray start, as before round 4) */
              const int dIdx = (sInfo >> 8) & 0xff;
              sr.dx = L.dirCos[3 * dIdx]; sr.dy = L.dirCos[3 * dIdx + 1]; sr.dz = L.dirCos[3 * dIdx + 2];
              sr.set_direction(L);
#else
              sr.load_direction(L.dirTab + 16 * ((sInfo >> 8) & 0xff));
#endif
              rst = R_TRACE;
            }
            rdHead += (unsigned)take;
            const unsigned started = (unsigned)take + count_lanes(secondLeg);   // every start is one tracer call
            wc.calls += started;
            raysStarted += started;
            PROF_END(PH_SERVICE, __popcll(endMask) + take);
          }
          // rays still under way go back to the ready buffer as they are.  (A wave that leaves has taken no new rays in the
          // service phase just before, but rays that ended there may have begun their second leg: with those the rays under
          // way may no longer fit beside the ready ones -- then the wave stays, as it did when the next pass re-counted.)
          const unsigned long long backMask = __ballot(rst == R_TRACE);   // (after the service phase, if there was one)
          const int nBack = (int)__popcll(backMask);
          const int readyNow = (int)(rdTail - rdHead);
          if ((leaving && (DIRECT || ready + nBack < kLowWater)) || (nBack == 0 && readyNow == 0 && ringRays == 0)) {
            if (rst == R_TRACE) {
              lds_float *out = rdBase + ((rdTail + (unsigned)lanes_below(backMask)) & (unsigned)(kReady - 1));
              out[0] = sr.x; out[kReady] = sr.y; out[2 * kReady] = sr.z;
              out[3 * kReady] = __int_as_float(sr.ix); out[4 * kReady] = __int_as_float(sr.iy); out[5 * kReady] = __int_as_float(sr.iz);
              out[6 * kReady] = __int_as_float(sInfo);
              out[7 * kReady] = sW; out[8 * kReady] = sNorm; out[9 * kReady] = sTauFree;
              out[10 * kReady] = sr.target; out[11 * kReady] = sr.acc;
            }
            rdTail += (unsigned)nBack;
            wc.calls -= (unsigned)nBack;      // (they are counted again when they are taken up: one tracer call each, whatever the schedule)
            raysStarted -= (unsigned)nBack;
            break;
          }
          // -------------------------------------------------------------- VOXEL-STEP phase (shadow rays)
          // (a second step in the same pass while the next service phase is some lanes away: the phase logic above --
          // ballots, counts, the decisions -- is paid once for both; written out twice, see the photons' step phase)
          auto ray_step = [&]() {
            const bool tracing = rst == R_TRACE;
            const unsigned nTracing = count_lanes(tracing);
            wc.shadow += nTracing;
            PROF_BEGIN();
            if (tracing) {   // a failed shadow ray contributes nothing (:1531-1535: its optical path is -2)
              // (an arrival: see the service phase.  The ray's own stage says whether it has a target: the wave-uniform P.useRRI says the
              // same, and as a run-time flag in this loop it came out as a lane mask made under another loop's exec mask -- GridPlace's trap,
              // caught by tests/test_build_isa.py)
              if (trace_step_lazy<GRID, false, GENERAL, !DIRECT>(P, L, sr, ((sInfo >> 16) & 3) != 0) != STEP_CONTINUE) rst = R_ENDED;   // (ring kernels: the short form, see trace_step_lazy)
            }
            PROF_END(PH_RAYSTEP, nTracing);
          };
          const int idleNow = 64 - nBack;   // (a service phase has handed rays out)
          // (lanes still idle after a service phase and rays left to make ready or to hand out: no step with a wave part
          // full, the next pass expands / serves again first -- as a `continue` after the service phase did, without its back edge)
          const bool fillFirst = serve && idleNow >= liThr && (ringRays > 0 || readyNow > 0);
          if (!fillFirst) {
            ray_step();
            if (liThr - idleNow > kStepAhead) ray_step();
#if I3RC_THIRD_STEP < 64
            if (liThr - idleNow > kStepAhead + I3RC_THIRD_STEP) ray_step();
#endif
          }
        }
        // what the photons' rays derive from their directions is worked out again here, so that those eleven registers per
        // lane are free during the ray loop (the radiance kernels then fit five waves per SIMD)
        r.set_direction(L);
      }
    }
    // ================================================================================================ PHOTON MODE
    // ---------------------------------------------------------------- EVENT phase
    // Two kinds of lanes wait for it: photons with an interaction due (scattering, reflection) and TURNOVER lanes, whose
    // photon has ended (left through the top, reached a black surface, dropped by the tracer) or that have none yet.
    // Closing a photon and starting the next one is a third of the event phase's instructions and in most event phases
    // two or three lanes need it: turnover lanes therefore sit out until kTurnMin of them have gathered (or kTurnForce
    // of them call for an event phase of their own, or nothing else is left to do).
    const bool wantScat = st == ST_EVENT;
    const bool wantTurn = st == ST_EXIT || st == ST_DROPPED || st == ST_NEW;
    const unsigned long long scMask = __ballot(wantScat), tuMask = __ballot(wantTurn);
    const unsigned long long trMask = __ballot(st == ST_TRACE);
    if (scMask == 0ull && tuMask == 0ull && trMask == 0ull) break;   // (no photons left; any ray work has been done in ray mode above)
    const int nSc = (int)__popcll(scMask), nTu = (int)__popcll(tuMask);
    bool runEvent = nSc + nTu > 0 && (nSc >= evThr || nTu >= kTurnForce || trMask == 0ull);
    const bool doTurn = nTu >= kTurnMin || trMask == 0ull || nSc == 0;
    const bool wantEvent = wantScat || (doTurn && wantTurn);
    const unsigned long long evMask = __ballot(wantEvent);            // the lanes this event phase serves
    if (DEFER && runEvent && (DIRECT ? kReady - (int)(rdTail - rdHead) : (int)(P.rayQueueCap - (int)(qTail - qHeadEv))) < nSc) {
      // every lane of the event phase may push one record: without room for all of them the rays are served first
      wantSlots = true;
      runEvent = false;
      if (trMask == 0ull) continue;
    }
    if (runEvent) {
      // The event phase reads the problem through a pointer into the kernarg segment (scalar loads where they are
      // needed) instead of keeping some sixty more values in scalar registers through the whole kernel: the kernel
      // wanted well over twice the scalar registers there are, and every spilled one comes back as a v_readlane,
      // a vector instruction, inside the voxel-step loops.  The step phases use the kernel argument itself.
      const ColdArgs ke = cold_args();
      const auto &Pe = ke->P;
      const auto &Ae = ke->A;
      if constexpr (LANE_COUNTS) {   // a 16-bit count about to overflow (one event adds at most one to each): hand over now
        // (steps: a hand-over sums the lanes' counts across the wave in 32 bits -- 64 lanes of less than 2^25 each)
        const bool full = ((accA | accB) & 0x80008000u) != 0u || accSteps >= (1u << 25);
        if (__ballot(full) != 0ull) flush_lanes(full);
      }
      double *laneBlk = nullptr;
      if constexpr (BATCHED) laneBlk = lane_block(rng.batch);
      const Tally<ColdProblem, BATCHED> tally{Pe, L, laneBlk};
      PROF_BEGIN();
      // a photon that has arrived at its optical depth makes the last advance of its trace here (trace_step_lazy / finish_arrival):
      // once per event instead of a division at every voxel step
      if (wantEvent && st == ST_EVENT && arrival_pending(r)) finish_arrival(r);
      // ... and one that has left the grid gets its height (finish_exit).  (The general kernels do that where the step ends: here it
      // cost them nine spilled vector registers -- and with max cross-section the layer index says nothing about the height.)
      if (!GENERAL && wantEvent && (st == ST_EVENT || st == ST_EXIT)) finish_exit(Pe, r);
      // ---- part A: endings that need no random number -- tracer drop, exit through the top, arrival at a black
      //      surface -- are tallied first so that the lanes can be given their next photon before the wave
      //      generates its random block (part C), which then serves old and new photons in one go.
      const bool isEv = wantEvent && (st == ST_EVENT || st == ST_EXIT);
      const bool dropped = wantEvent && st == ST_DROPPED;                 // :488-489
      const bool atTop = isEv && r.z >= Pe.zMax;                           // :499-514
      const bool atSurface = isEv && !atTop && r.z <= surfaceZ;           // :515-531
      const bool atBlack = atSurface && blackSurface;                     // ... and :560-562
      wc.dropped += count_lanes(dropped);
      wc.top += count_lanes(atTop);
      wc.surf += count_lanes(atSurface);
      if constexpr (BATCHED) {
        if constexpr (LANE_COUNTS) accB += (atSurface ? 1u : 0u) + (atTop ? 0x10000u : 0u);
        if (dropped) {
          double *const c = counter_block(rng.batch);
          unsafeAtomicAdd(c + I3RC_CNT_DROPPED, 1.0);
          unsafeAtomicAdd(c + I3RC_CNT_TRACER_CALLS, 1.0);
        }
      }
      if (dropped || atTop || atBlack) {
        // one merged branch for the three endings: a single tally atomic and a single bookkeeping block
        if (!dropped) {
          if (!rayTracing) {   // max cross-section: step back to the boundary (:504-511, :521-528)
            const float zB = atTop ? Pe.zMax : Pe.z0;
            r.x = make_periodic(r.x - r.dx * fabsf((r.z - zB) / r.dz), Pe.x0, Pe.xMax);
            r.y = make_periodic(r.y - r.dy * fabsf((r.z - zB) / r.dz), Pe.y0, Pe.yMax);
            find_xy<GENERAL>(Pe, L, r.x, r.y, r.ix, r.iy);
          }
          const int c2 = (r.iy - 1) * Pe.nx + (r.ix - 1);
          tally.boundary(atTop, c2, w);
          if (REPLAY) { fateCol = c2; fateW = w; }
        }
        if (REPLAY) {
          order += atBlack ? 1 : 0;
          fate = dropped ? 3 : (atTop ? 0 : 1);
        }
        if (MERGE && pendCell >= 0) { tally.absorbed_sum(pendCell, pendSum); pendCell = -1; }
        close_photon();
        st = ST_NEW;
      }
      // ---- part B (uniform): hand out photon indices from the wave's reservoir
      const bool isNew = wantEvent && st == ST_NEW;
      const unsigned long long newMask = __ballot(isNew);
      if constexpr (BATCHED) {
        if (newMask != 0ull) {
          int need = __popcll(newMask);
          int rank = lanes_below(newMask);
          long long mine = -1;
          unsigned mineBatch = 0u;
          for (;;) {   // (uniform; a batch's last chunk may be short of `need`: then the next chunk serves the rest)
            const long long left = res.end - res.next;
            const int avail = left > 0 ? (left < 64 ? (int)left : 64) : 0;
            const int take = avail < need ? avail : need;
            if (isNew && mine < 0 && rank >= 0 && rank < take) { mine = res.next + rank; mineBatch = res.batch; }
            res.next += take; resTaken += (unsigned)take; wc.photons += (unsigned)take;
            need -= take; rank -= take;
            if (need == 0 || res.end < 0) break;
            if ((++refills & 3u) == 0u || wc.steps > 0x40000000u) flush_counters();   // (re-fits the thresholds; the counts go per lane)
            else adapt_thresholds();
            const unsigned oldBatch = res.batch;
            res.refill_batched();
            if (res.end < 0 || res.batch != oldBatch) hand_over_taken(oldBatch);
          }
          // a lane whose next photon belongs to another batch hands over what it has counted for the batch it leaves
          if constexpr (LANE_COUNTS) {
            const bool leaves = isNew && mine >= 0 && mineBatch != rng.batch && (accSteps | accA | accB | rng.used) != 0u;
            if (__ballot(leaves) != 0ull) flush_lanes(leaves);
          }
          if (isNew) {
            if (mine < 0) st = ST_DONE;   // (only when the launch has no chunks left)
            else rng.start((uint64_t)(Ae.firstPhoton + mine), mineBatch);
          }
        }
      } else
      if (newMask != 0ull) {
        int need = __popcll(newMask);
        int rank = lanes_below(newMask);
        long long mine = -1;
        long long avail = res.end - res.next;
        const bool slabs = SLABS && Ae.slabIds != nullptr;
        if (avail < (long long)need && (slabs ? res.end >= 0 : res.end < Ae.nPhotons)) {   // drain the reservoir, then refill it (chunk >= 64 covers the rest)
          if (isNew && rank < (int)avail) mine = res.next + rank;
          wc.photons += (unsigned)avail;                        // numPhotonsProcessed :459
          need -= (int)avail;
          rank -= (int)avail;
          // the counters go to the tally buffer (and the thresholds are re-fitted) at every fourth refill: the nine
          // atomics of a hand-over all go to the same nine addresses, from every wave of the chip
          if ((++refills & 3u) == 0u || wc.steps > 0x40000000u || wc.shadow > 0x40000000u) flush_counters();
          else adapt_thresholds();
          if (slabs) res.refill_slabs(&slabsTried[threadIdx.x >> 6]); else res.refill();
          avail = res.end - res.next;
        }
        const int taken = (int)(avail < (long long)need ? avail : (long long)need);   // < need only when the batch is exhausted
        if (isNew && mine < 0 && rank >= 0 && rank < taken) mine = res.next + rank;
        res.next += taken;
        wc.photons += (unsigned)taken;
        if (isNew) {
          if (mine < 0) st = ST_DONE;
          else {
            if (slabs) mine = (long long)Ae.slabIds[mine];   // position in the sorted list -> photon number
            if constexpr (!BATCHED) RngInit<Rng>::start(rng, Ae, mine);
            if (NEED_PID) pid = mine;
          }
        }
      }
      // ---- part C: one random block per lane for this event, then the event itself
      bool didScatter = false, didRoulette = false, startedTrace = false;   // per-lane flags -> wave counters below
      if (wantEvent && st != ST_DONE) {
        rng.begin_event();
        if (st == ST_NEW) {                                               // :453-470
          float px, py, pz;
          if (directional) {   // newPhotonStream_Directional, Code/monteCarloIllumination.f95:91-99
            px = rng.first(); py = rng.second();
#ifdef I3RC_EXPERIMENT_SLAB   // measurement only (tools/locality_experiment.py): all photons start in one eighth of the domain
            py *= 0.125f;
#endif
            pz = 1.0f - spacingf(1.0f);
            r.dx = Ae.solarDx; r.dy = Ae.solarDy; r.dz = Ae.solarDz;
          } else {
            px = Ae.sx[pid]; py = Ae.sy[pid]; pz = Ae.sz[pid];
            make_dircos(Ae.smu[pid], Ae.sphi[pid], r.dx, r.dy, r.dz);
          }
          order = 0;
          if (REPLAY) { fate = -1; fateCol = -1; fateW = 0.0f; }
          w = 1.0f;
          r.x = Pe.x0 + px * (Pe.xMax - Pe.x0);
          r.y = Pe.y0 + py * (Pe.yMax - Pe.y0);
          r.z = Pe.z0 + pz * (Pe.zMax - Pe.z0);
          r.ix = 1; r.iy = 1; r.iz = 1;
          if (!GENERAL && !(MULTI && !Pe.xyRegular)) {   // findXYIndicies :1359-1369 with the divisions by the (uniform) cell sizes done by reciprocal
            int i = min((int)exact_div(r.x - Pe.x0, Pe.deltaX, rcpDeltaX) + 1, Pe.nx);
            int j = min((int)exact_div(r.y - Pe.y0, Pe.deltaY, rcpDeltaY) + 1, Pe.ny);
            if (fabsf(L.xE[i] - r.x) < spacingf(r.x)) i = i + 1;
            if (fabsf(L.yE[j] - r.y) < spacingf(r.y)) j = j + 1;
            r.ix = i == Pe.nx + 1 ? 1 : i;
            r.iy = j == Pe.ny + 1 ? 1 : j;
            r.iz = izStart;
          } else {
            find_xy<true>(Pe, L, r.x, r.y, r.ix, r.iy);   // (the general kernels, and the several-components ones on an irregular x / y grid)
            find_z<true>(Pe, L, r.z, r.iz);
          }
          st = ST_TRACE;
        }
        if (st == ST_EVENT) {
          if (r.z <= surfaceZ) {                                          // :515-580
            order++;
            if (!rayTracing) {
              r.x = make_periodic(r.x - r.dx * fabsf((r.z - Pe.z0) / r.dz), Pe.x0, Pe.xMax);
              r.y = make_periodic(r.y - r.dy * fabsf((r.z - Pe.z0) / r.dz), Pe.y0, Pe.yMax);
              find_xy<GENERAL>(Pe, L, r.x, r.y, r.ix, r.iy);
            }
            r.iz = 1;
            r.z = surfaceZ;
            const int c2 = (r.iy - 1) * Pe.nx + (r.ix - 1);
            tally.down(c2, w);
            if (REPLAY) { fateCol = c2; fateW = w; }
            float mu = exact_sqrt(rng.first());
            while (!(fabsf(mu) > 2.0f * kTiny)) mu = exact_sqrt((GENERAL || MULTI) ? rng.next() : rng.fresh());   // :546-549
            const float turn = rng.second();                               // phi = 2 pi turn (:550)
            if (useBDRF) w = w * surface_reflectance(Pe, r.x, r.y);
            else w = w * Pe.albedo;
            if (w <= kTiny) { if (REPLAY) fate = 1; st = ST_NEW; }
            else {
              if (REPLAY) make_dircos(mu, (2.0f * kPi) * turn, r.dx, r.dy, r.dz);
              else {
                // production streams: the hardware sine / cosine take their argument in revolutions -- two instructions
                // instead of the ~180 of libm's sinf + cosf (argument reduction for any float), as for the scattering azimuth
                const float sinTheta = exact_sqrt(1.0f - mu * mu);
                r.dx = sinTheta * __builtin_amdgcn_cosf(turn); r.dy = sinTheta * __builtin_amdgcn_sinf(turn); r.dz = mu;
              }
              if (defer) { pendingShadow = true; wI = w; evInfo = 0; }    // component 0: the surface (:567-580)
              else if (INTENSITY)
                intensity_contribution<GRID>(Pe, L, rng, nested, w, r.x, r.y, r.z, r.ix, r.iy, r.iz, r.dx, r.dy, r.dz, 0, order);
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
              const bool inGrid = (unsigned)(r.iz - 1) < (unsigned)Pe.nz;
              cell = inGrid ? cell_index(Pe, r.ix, r.iy, r.iz) : 0;
              const float extHere = inGrid ? Pe.totalExt[cell] : 0.0f;
              scatterThis = rng.next() < extHere / Pe.maxExt;
            }
            if (scatterThis) {
              order++;
              didScatter = true;
              // :606-632 (quirk Q2 kept): a scattering in a cell without extinction steps back over the cell face.  The
              // tracer only stops inside a cell whose extinction is positive (acc + step * 0 > target cannot hold), so
              // with ray tracing the case cannot arise and the look-up is left to the max-cross-section build.
              if (!rayTracing && cell_extinction<GRID>(Pe, L, r.ix, r.iy, r.iz) <= 0.0f) {
                if (r.x - L.xE[r.ix - 1] <= 0.0f && r.dx > 0.0f) {
                  r.x = r.x - spacingf(r.x);
                  r.ix = r.ix - 1;
                  if (r.ix <= 0) { r.ix = Pe.nx; r.x = L.xE[r.ix - 1]; r.x = r.x - 2.0f * spacingf(r.x); }
                }
                if (r.y - L.yE[r.iy - 1] <= 0.0f && r.dy > 0.0f) {
                  r.y = r.y - spacingf(r.y);
                  r.iy = r.iy - 1;
                  if (r.iy <= 0) { r.iy = Pe.ny; r.y = L.xE[r.iy - 1]; r.y = r.x - 2.0f * spacingf(r.y); }
                }
                if (r.z - L.zE[r.iz - 1] <= 0.0f && r.dz > 0.0f) { r.z = r.z - spacingf(r.z); r.iz = r.iz - 1; }
              }
              // the cell's properties are read only where the domain does not share one value (see DevProblem)
              const bool needCell = GENERAL || MULTI || !(Pe.uniformSsa >= 0.0f) || Pe.uniformSsa < 1.0f || Pe.uniformPf < 1;
              if (needCell) cell = cell_index(Pe, r.ix, r.iy, r.iz);
              int comp = 1;                                               // :637-638
#ifndef I3RC_CELL_RECORD_READS
#define I3RC_CELL_RECORD_READS 1
#endif
              // TWO components (cloud + gas, cloud + aerosol: the usual production domain): what the scattering needs of its cell comes as
              // ONE 16-byte record (DevProblem::cellRec) -- one cache line, one trip to L2 -- and the component's pair is picked when the
              // deviate has been compared, instead of one read to choose the component and then two more, from two more arrays, that wait
              // for it.  (Asking the three arrays for both components' words at once -- five reads, five lines -- was measured: 9 - 15 %
              // SLOWER on the flux workloads; it is the lines that cost.  findIndex on (/0, c1, c2/) without a first guess answers 1 or 2,
              // never 3: see below.)
              bool fromRecord = false;
              float ssaRec = 0.0f;
              int pfiRec = 0;
              if (I3RC_CELL_RECORD_READS && (GENERAL || MULTI) && !REPLAY && multiComp && Pe.cellRec != nullptr) {
                const float rc = rng.next();
                if (Pe.ncomp == 2) {
                  const uint4 rec = Pe.cellRec[cell];
                  const float c0 = __uint_as_float(rec.x), s0 = __uint_as_float(rec.y), s1 = __uint_as_float(rec.z);
                  const int p0 = (int)(rec.w & 0xffffu), p1 = (int)(rec.w >> 16);
                  const bool second = rc >= c0;
                  comp = second ? 2 : 1; ssaRec = second ? s1 : s0; pfiRec = second ? p1 : p0;
                } else {   // three components (droplets + aerosol + gas): 32 bytes of the same line
                  const uint4 a = Pe.cellRec[2 * (size_t)cell], b = Pe.cellRec[2 * (size_t)cell + 1];
                  const bool ge0 = rc >= __uint_as_float(a.x), ge1 = rc >= __uint_as_float(a.y);
                  comp = 1 + (ge0 ? 1 : 0) + (ge1 ? 1 : 0);
                  ssaRec = comp == 1 ? __uint_as_float(a.z) : (comp == 2 ? __uint_as_float(a.w) : __uint_as_float(b.x));
                  pfiRec = comp == 1 ? (int)(b.y & 0xffffu) : (comp == 2 ? (int)(b.y >> 16) : (int)b.z);
                }
                fromRecord = true;
              } else
              if (multiComp || REPLAY) {
                const float rc = rng.next();
                if (multiComp) {
                  const float *cum = Pe.cumExt + cell;
                  if constexpr (MULTI) {
                    // findIndex without a first guess answers the largest i <= size(table) - 1 with table(i) <= value (its bisection starts
                    // with upperBound = size(table) and never returns it: Code/numericUtilities.f95:234-247, pinned by
                    // tests/golden/ref_numerics.npz): with table = (/0, cumulativeExt/) that is 1 + the number of the cell's first
                    // ncomp - 1 cumulative extinctions at or below the deviate -- the last one is never looked at.  A compare chain: every
                    // lane the same ncomp - 1 independent reads, where the bisection's reads depend on one another.
                    const int nc = Pe.ncomp;
                    for (int k = 0; k < nc - 1; ++k) comp += rc >= cum[(size_t)k * ncell] ? 1 : 0;
                  } else
                  comp = find_index(rc, [cum, ncell](int k) { return k == 1 ? 0.0f : cum[(size_t)(k - 2) * ncell]; },
                                    Pe.ncomp + 1, 0);
                }
              }
              // single-scattering albedo and phase-function entry of the cell; a value shared by the whole (one-component)
              // domain comes from the kernel arguments instead of two dependent memory reads
              float ssa;
              if ((GENERAL || MULTI) && fromRecord) ssa = ssaRec;
              else if (!GENERAL && !MULTI && Pe.uniformSsa >= 0.0f) ssa = Pe.uniformSsa;
              else if (I3RC_CELL_RECORD_READS && !GENERAL && !MULTI && !REPLAY && Pe.cellRec != nullptr) {
                // ONE component whose cells share neither albedo nor table entry (a Mie cloud: every cell its effective radius): the two as one
                // 8-byte record (the same pointer: i3rc_hip_create makes the record that fits the domain, and only where NEITHER is shared, so
                // that the kernels of the BASELINE workloads -- both shared -- never ask)
                const uint2 rec = ((const uint2 *)Pe.cellRec)[cell];
                ssa = __uint_as_float(rec.x); pfiRec = (int)rec.y; fromRecord = true;
              }
              else ssa = Pe.ssa[(size_t)(comp - 1) * ncell + cell];
              if (ssa < 1.0f) {                                           // :642-649
                const float inc = w * (1.0f - ssa);
                if (!MERGE || tally.volume_in_lds()) tally.absorbed(cell, inc);
                else {
                  if (cell != pendCell) {
                    if (pendCell >= 0) tally.absorbed_sum(pendCell, pendSum);
                    pendCell = cell; pendSum = 0.0;
                  }
                  pendSum += (double)inc;
                }
                w = w * ssa;
              }
              int pfi;
              if ((GENERAL || MULTI) && fromRecord) pfi = max(pfiRec, 1);
              else if (!GENERAL && !MULTI && Pe.uniformPf >= 1) pfi = Pe.uniformPf;
              else if (!GENERAL && !MULTI && fromRecord) pfi = max(pfiRec, 1);
              else pfi = max(Pe.pfIndex[(size_t)(comp - 1) * ncell + cell], 1);   // (index 0 marks clear cells: never a table offset of -1)
              if (defer) {                                                // :654-668: pushed after this event, traced in ray mode
                pendingShadow = true; wI = w;
                inDx = r.dx; inDy = r.dy; inDz = r.dz;                     // incoming direction
                evInfo = comp | ((Pe.useHybrid && order <= Pe.numOrdersOrig) ? 0x100 : 0) | (int)((unsigned)pfi << 16);
              } else if (INTENSITY)
                intensity_contribution<GRID>(Pe, L, rng, nested, w, r.x, r.y, r.z, r.ix, r.iy, r.iz, r.dx, r.dy, r.dz, comp, order);
              if (Pe.useRR && w < 0.5f) {                                  // :673-680
                didRoulette = true;
                if (rng.spare() >= w / 1.0f) w = 0.0f; else w = 1.0f;
              }
              if (w <= kTiny) { if (REPLAY) fate = 2; st = ST_NEW; }
              else {
                const CompTables ct = (GENERAL || MULTI) ? load_tables(Pe.comp[comp - 1]) : load_tables(Pe.comp0);
                float cosS;
                if constexpr (TBL) cosS = scattering_cosine<REPLAY>(rng.first(), (const lds_float *)L.cosTab, ct.nInv, refined_rcp((float)ct.nInv));
                else cosS = scattering_cosine<REPLAY>(rng.first(), ct.invCos + (size_t)(pfi - 1) * ct.nInv, ct.nInv, refined_rcp((float)ct.nInv));
                next_direct(rng, cosS, r.dx, r.dy, r.dz);                 // :684-687
                st = ST_TRACE;
              }
            } else {
              st = ST_TRACE;
            }
          }
        }
        if (st == ST_TRACE) {                                             // :480
          // (production streams: hardware log2, within 2 ulp -- an optical depth, not a trajectory's bit pattern; the replay
          // build follows the reference's deviates with libm's logf)
          const float tau = REPLAY ? -sample_log(fmaxf(kTiny, rng.path())) : -fast_log(fmaxf(kTiny, rng.path()));
          r.acc = 0.0f; r.target = tau;
          if (rayTracing) { startedTrace = true; r.set_direction(L); }
          else {                                                          // :494-496 max cross-section move
            r.x = make_periodic(r.x + r.dx * tau / Pe.maxExt, Pe.x0, Pe.xMax);
            r.y = make_periodic(r.y + r.dy * tau / Pe.maxExt, Pe.y0, Pe.yMax);
            r.z = r.z + r.dz * tau / Pe.maxExt;
            st = ST_EVENT;
          }
        }
        // a photon that died in part C (roulette, absorbing surface) is closed here and respawns at the next event phase
        if (st == ST_NEW) {
          if (MERGE && pendCell >= 0) { tally.absorbed_sum(pendCell, pendSum); pendCell = -1; }
          close_photon();
        }
      }
      if constexpr (DIRECT) {   // one radiance direction: the event becomes a ready ray here and now (make_ray), survivors only go to LDS
        if (defer) {
          bool keep = false;
          float word6 = 0.0f, norm = 0.0f, tauFree = 0.0f, target = 0.0f;
          if (pendingShadow)
            keep = make_ray(Pe, Ae, evInfo, inDx, inDy, inDz, rng.photon_lo(), rng.photon_hi(), rng.event_block(), rng.lane_batch(), 0, word6, norm, tauFree, target);
          const unsigned long long madeMask = __ballot(pendingShadow), keepMask = __ballot(keep);
          if (keep) {
            lds_float *out = rdBase + ((rdTail + (unsigned)lanes_below(keepMask)) & (unsigned)(kReady - 1));
            out[0] = r.x; out[kReady] = r.y; out[2 * kReady] = r.z;
            out[3 * kReady] = __int_as_float(r.ix); out[4 * kReady] = __int_as_float(r.iy); out[5 * kReady] = __int_as_float(r.iz);
            out[6 * kReady] = word6;
            out[7 * kReady] = wI;
            out[8 * kReady] = norm; out[9 * kReady] = tauFree; out[10 * kReady] = target; out[11 * kReady] = 0.0f;
          }
          pendingShadow = false;
          const unsigned kept = (unsigned)__popcll(keepMask);
          rdTail += kept;
          raysSkipped += (unsigned)__popcll(madeMask) - kept;
        }
      } else
      if (defer) {   // push the events of this phase into the wave's ring: one record serves all D rays of an event
        const unsigned long long pushMask = __ballot(pendingShadow);
        if (pendingShadow) {
          lds_float *rec = qBase + ((qTail + (unsigned)lanes_below(pushMask)) & qMask);
          const int cap = Pe.rayQueueCap;
          rec[0] = r.x; rec[cap] = r.y; rec[2 * cap] = r.z;
          rec[3 * cap] = __int_as_float(r.ix); rec[4 * cap] = __int_as_float(r.iy); rec[5 * cap] = __int_as_float(r.iz);
          rec[6 * cap] = wI;
          rec[7 * cap] = inDx; rec[8 * cap] = inDy; rec[9 * cap] = inDz;
          rec[10 * cap] = __int_as_float(evInfo);
          rec[11 * cap] = __uint_as_float(rng.photon_lo()); rec[12 * cap] = __uint_as_float(BATCHED ? rng.lane_batch() : rng.photon_hi());
          rec[13 * cap] = __uint_as_float(rng.event_block());
          pendingShadow = false;
        }
        qTail += (unsigned)__popcll(pushMask);
      }
      if constexpr (LANE_COUNTS) accA += (didScatter ? 1u : 0u) + (didRoulette ? 0x10000u : 0u);
      wc.scat += count_lanes(didScatter);
      wc.roul += count_lanes(didRoulette);
      if constexpr (!BATCHED) wc.calls += count_lanes(startedTrace);   // (BATCHED: a batch's photon traces = its scatterings + surface arrivals + exits + drops)
      PROF_END(PH_EVENT, __popcll(evMask));
    }
    // -------------------------------------------------------------- VOXEL-STEP phase (photons)
    // (a second step in the same pass while the next event phase is some lanes away: the phase logic is paid once for
    // both.  Written out twice: a loop around the step made the whole photon loop's code worse, -17 % on the step cloud)
    auto photon_step = [&]() {
      const bool tracing = st == ST_TRACE;
      const unsigned nTracing = count_lanes(tracing);
      wc.steps += nTracing;
      PROF_BEGIN();
      if (tracing) {
        if constexpr (LANE_COUNTS) accSteps++;
        const StepResult s = trace_step_lazy<GRID, !INTENSITY, GENERAL>(P, L, r, true);   // (an arrival is finished by the event phase)
        if (GENERAL && s == STEP_EXIT) finish_exit(P, r);
        // (an exit through the top, or onto a black surface, ends the photon: such lanes wait for the turnover quorum)
        if (s != STEP_CONTINUE)
          st = s == STEP_DONE ? ST_EVENT : (s == STEP_ERROR ? ST_DROPPED : ((r.iz >= 1 || blackSurface) ? ST_EXIT : ST_EVENT));
      }
      PROF_END(PH_STEP, nTracing);
    };
    photon_step();
#if I3RC_PHOTON_STEP_AHEAD < 64
    if (!runEvent && evThr - nSc > kPhotonStepAhead) photon_step();
#endif
  }
#ifdef I3RC_PROFILE_PHASES
  if ((threadIdx.x & 63) == 0) {   // (diagnostic build, omega = 1 runs: the volume-absorption tally is free)
    unsafeAtomicAdd(P.tally + P.oVol + 3 * PH_KINDS, (double)(__builtin_amdgcn_s_memtime() - profStart));   // the wave's whole loop
    for (int k = 0; k < PH_KINDS; ++k) {
      unsafeAtomicAdd(P.tally + P.oVol + 3 * k, (double)profCycles[k]);
      unsafeAtomicAdd(P.tally + P.oVol + 3 * k + 1, (double)profCount[k]);
      unsafeAtomicAdd(P.tally + P.oVol + 3 * k + 2, (double)profLanes[k]);
    }
  }
#endif

  // ------------------------------------------------------------------ epilogue: flush tallies + counters
  if constexpr (BATCHED) {   // what the lanes still hold for their last batches; the tallies themselves are in global memory already
    if constexpr (LANE_COUNTS) flush_lanes(true); else flush_counters();
    hand_over_taken(res.batch);
    return;
  }
  __syncthreads();
  {
    const ColdArgs ka = cold_args();   // (offsets and sizes straight from the kernarg segment: see cold_args)
    double *const out = ka->P.tally;
    if (ka->P.ldsTallies) {
      const int ncol = ka->P.nx * ka->P.ny;
      const int oUp = ka->P.oUp, oDown = ka->P.oDown;
      for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
        const tally_t u = L.tUp[i], d = L.tDown[i];
        if (u != (tally_t)0) add_global(out + oUp + i, u);
        if (d != (tally_t)0) add_global(out + oDown + i, d);
      }
    }
    if (ka->P.ldsVolume) {
      const int nVol = ka->P.nx * ka->P.ny * ka->P.nz, oVol = ka->P.oVol;
      for (int i = threadIdx.x; i < nVol; i += blockDim.x) {
        const tally_t v = L.tVol[i];
        if (v != (tally_t)0) add_global(out + oVol + i, v);
      }
    }
    if (ka->P.ldsIntensity) {
      const int nInt = (ka->P.ncomp + 1) * ka->P.nDir * ka->P.nx * ka->P.ny;
      const int oInt = ka->P.oInt;
      for (int i = threadIdx.x; i < nInt; i += blockDim.x) {
        const tally_t v = L.tInt[i];
        if (v != (tally_t)0) add_global(out + oInt + i, v);
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
      if (raysSkipped != 0u) unsafeAtomicAdd(counters + I3RC_CNT_RAYS_SKIPPED, (double)raysSkipped);
    }
  }
}

// Test hook: independent tracer calls, one ray per thread.
// (CLEARMAP = false: domains of more than 65534 layers, whose clear-air map -- 16 bits per bound -- is not used: i3rc_hip.hip)
// (GRID: the bricked copy -- what the hook reads unless i3rc_hip_select_grid_place says otherwise --, the plain field, or the column records)
template <int GRID, bool CLEARMAP>
__global__ void __launch_bounds__(256) trace_rays_kernel(const DevProblem P, long long n, const float *dir, float *pos,
                                                         int32_t *idx, const float *target, float *tau, int32_t *steps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  L.xE = (lds_float *)smem; L.yE = L.xE + P.nx + 1; L.zE = L.yE + P.ny + 1;
  L.tUp = L.tDown = L.tVol = nullptr; L.dirCos = nullptr;
  L.ext = L.zE + P.nz + 1;   // the clear-air map of the bricked field
  if (GRID == GRID_BRICKS && CLEARMAP)
    for (int i = threadIdx.x; i < P.clearNx * (((P.ny - 1) >> P.clearShift) + 1); i += blockDim.x) L.ext[i] = __uint_as_float(P.clearMap[i]);
  if (GRID == GRID_COLBASE)
    for (int i = threadIdx.x; i < P.nz; i += blockDim.x) L.ext[i] = P.colBase[i];
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
  do { ns++; s = trace_step<GRID, CLEARMAP>(P, L, r, hasTarget); } while (s == STEP_CONTINUE && ns < (1 << 24));
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

// Test hooks: findIndex and computeSurfaceReflectance as the photon kernels evaluate them (tracer.hpp find_index, surface_reflectance),
// one thread per value -- held against the reference's own routines (tests/golden/ref_numerics.npz) by tests/test_gpu_ref_numerics.py.
__global__ void find_index_kernel(int n, const float *table, long long m, const float *values, const int32_t *firstGuess, int32_t *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[i] = find_index(values[i], [table](int k) { return table[k - 1]; }, n, firstGuess ? firstGuess[i] : 0);
}
struct SurfaceOnly { const float *xsE, *ysE, *brdf; int nxs, nys; };   // (what surface_reflectance reads of a DevProblem)
__global__ void surface_reflectance_kernel(const SurfaceOnly S, long long m, const float *x, const float *y, float *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[i] = surface_reflectance(S, x[i], y[i]);
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
