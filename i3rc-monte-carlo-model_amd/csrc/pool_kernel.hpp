// Photon-pool kernel (EXPERIMENT, selected with I3RC_KERNEL_POOL only): the flux-only tracer of the common problem
// class (regular grid, ray tracing, one component, Lambertian surface, Directional source, production RNG) with
// better-filled wavefronts.  Measured on MI355X (step cloud 32x1x16, 5e7 photons): 1.98e9 photons/s against 2.61e9 of
// photon_kernel -- lane occupancy rises as predicted (0.55 -> 0.65 of all vector thread-cycles) but the LDS round trip
// of the photon state adds ~70 vector instructions per phase (address selects, register initialisation for idle
// lanes, state-byte bookkeeping), more than the occupancy returns.  Kept because it cross-checks the state machine:
// three kernels must give every photon the same fate (tests/test_gpu_parity.py).
//
// photon_kernel (kernels.hpp) ties one photon to one lane: a lane whose photon waits for the other phase idles, and
// the wave runs its voxel-step and event phases at ~2/3 lane occupancy.  Here every lane owns kPerLane = 2 photons,
// whose state lives in LDS (structure of arrays, one private column per lane and photon: conflict-free, no
// synchronisation).  Each iteration the wave ballots which lanes could serve an event and which a voxel step, runs the
// phase more lanes can serve, and every such lane loads the photon of that kind into registers, advances it by ONE
// step or ONE event, and stores it back.  A lane idles only when both of its photons wait for the other phase:
// occupancy rises from 0.62 / 0.81 (event / step phase) to 0.78 / 0.92 for ~25 LDS and ~10 vector instructions per
// phase.  (A wave-wide pool with compaction fills every phase completely but its bookkeeping -- state bytes, ranks,
// slot list, scattered LDS addresses -- cost more vector instructions than the last 15 % of occupancy return: measured.)
// The phase bodies are the ones of photon_kernel<PhiloxStream, false, false> (same arithmetic, same per-photon Philox
// streams: a photon's path, and therefore every tally and work counter, is identical -- tests compare them).
#pragma once
#include "kernels.hpp"

namespace i3rc {

constexpr int kPerLane = 2;             // photons per lane
constexpr int kPool = 64 * kPerLane;    // photons per wave
enum PoolField { F_X = 0, F_Y, F_Z, F_DX, F_DY, F_DZ, F_RX, F_RY, F_RZ, F_IX, F_IY, F_IZ, F_ACC, F_TARGET, F_W, F_ID, F_BLOCK,
                 kPoolFields };
constexpr int kPoolWordsPerWave = kPoolFields * kPool;
constexpr int kPoolBytesPerBlock = 4 * kPoolWordsPerWave * 4;   // 4 waves

template <int GRID>
__global__ void __launch_bounds__(256, 4) photon_pool_kernel(const DevProblem P, const RunArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  lds_float *poolBase;
  {
    lds_float *p = (lds_float *)smem;
    L.xE = p; p += P.nx + 1;
    L.yE = p; p += P.ny + 1;
    L.zE = p; p += P.nz + 1;
    const int ncol = P.nx * P.ny;
    L.tUp = p; L.tDown = p + ncol; L.tAbs = p + 2 * ncol;
    if (P.ldsTallies) p += 3 * ncol;
    L.dirCos = p; L.park = p; L.tInt = p;
    L.ext = p;
    if (P.ldsGrid) p += ncol * P.nz;
    poolBase = p;
  }
  for (int i = threadIdx.x; i <= P.nx; i += blockDim.x) L.xE[i] = P.xE[i];
  for (int i = threadIdx.x; i <= P.ny; i += blockDim.x) L.yE[i] = P.yE[i];
  for (int i = threadIdx.x; i <= P.nz; i += blockDim.x) L.zE[i] = P.zE[i];
  if (P.ldsTallies)
    for (int i = threadIdx.x; i < 3 * P.nx * P.ny; i += blockDim.x) L.tUp[i] = 0.0f;
  if (P.ldsGrid) {
    const int ncell = P.nx * P.ny * P.nz;
    for (int i = threadIdx.x; i < ncell; i += blockDim.x) L.ext[i] = P.totalExt[i];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lds_float *pool = poolBase + wave * kPoolWordsPerWave + lane;   // [kPoolFields][kPerLane][64]: this lane's columns
  int kindA = ST_NEW, kindB = ST_NEW;                               // LaneState of the lane's two photons
  __syncthreads();

  const Tally tally{P, L};
  const float zStart = P.z0 + (1.0f - spacingf(1.0f)) * (P.zMax - P.z0);
  int izStart = 1;
  find_z<false>(P, L, zStart, izStart);
  const float rcpDeltaX = refined_rcp(P.deltaX), rcpDeltaY = refined_rcp(P.deltaY);
  const float surfaceZ = P.z0 + spacingf(P.z0);
  const bool blackSurface = !(P.albedo > kTiny);

  WaveCounters wc;
  PhiloxStream rng;
  rng.init(A.seed0, A.seed1);
  Reservoir res;
  res.refill();
  auto flush_counters = [&]() {
    if (lane == 0) {
      const uint32_t c[9] = {wc.photons, wc.dropped, wc.steps, wc.scat, wc.surf, wc.top, wc.roul, wc.shadow, wc.calls};
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (c[k] != 0u) unsafeAtomicAdd(P.tally + P.oCnt + k, (double)c[k]);
    }
    wc = WaveCounters();
  };

  for (;;) {
    // ---- which phase can more lanes serve?
    const bool eA = kindA == ST_EVENT || kindA == ST_DROPPED || kindA == ST_NEW;
    const bool eB = kindB == ST_EVENT || kindB == ST_DROPPED || kindB == ST_NEW;
    const bool tA = kindA == ST_TRACE, tB = kindB == ST_TRACE;
    const int nE = (int)count_lanes(eA || eB), nT = (int)count_lanes(tA || tB);
    if (nE + nT == 0) break;
    const bool doEvent = nE >= nT;
    const bool active = doEvent ? (eA || eB) : (tA || tB);
    const bool useB = doEvent ? !eA : !tA;             // the first of the lane's photons that is of the phase's kind
    lds_float *ph = pool + (useB ? 64 : 0);            // field f of that photon: ph[f * kPool]

    if (!doEvent) {
      // ---------------------------------------------------------------- VOXEL-STEP phase (one step of each photon)
      wc.steps += (unsigned)nT;
      if (active) {
        Ray r;
        r.x = ph[F_X * kPool]; r.y = ph[F_Y * kPool]; r.z = ph[F_Z * kPool];
        r.dx = ph[F_DX * kPool]; r.dy = ph[F_DY * kPool]; r.dz = ph[F_DZ * kPool];
        r.rx = ph[F_RX * kPool]; r.ry = ph[F_RY * kPool]; r.rz = ph[F_RZ * kPool];
        r.ix = __float_as_int(ph[F_IX * kPool]); r.iy = __float_as_int(ph[F_IY * kPool]); r.iz = __float_as_int(ph[F_IZ * kPool]);
        r.acc = ph[F_ACC * kPool]; r.target = ph[F_TARGET * kPool];
        // a direction cosine below 1e-20 (or zero: reciprocal inf / NaN) takes the tracer's guarded IEEE divisions
        r.slow = !(fabsf(r.rx) <= 1e20f) || !(fabsf(r.ry) <= 1e20f) || !(fabsf(r.rz) <= 1e20f);
        {   // the per-trace constants of trace_step (this experiment keeps them out of the pool)
          const bool px = r.dx >= 0.0f, py = r.dy >= 0.0f, pz = r.dz >= 0.0f;
          r.ex = lds_address(L.xE) - (px ? 0 : 4); r.ey = lds_address(L.yE) - (py ? 0 : 4); r.ez = lds_address(L.zE) - (pz ? 0 : 4);
          r.cx = px ? 1 : -1; r.cy = py ? 1 : -1; r.cz = pz ? 1 : -1;
          r.nudge = px ? 2.0f : -2.0f;
        }
        const StepResult s = trace_step<GRID>(P, L, r, true);
        ph[F_X * kPool] = r.x; ph[F_Y * kPool] = r.y; ph[F_Z * kPool] = r.z;
        ph[F_IX * kPool] = __int_as_float(r.ix); ph[F_IY * kPool] = __int_as_float(r.iy); ph[F_IZ * kPool] = __int_as_float(r.iz);
        ph[F_ACC * kPool] = r.acc;
        if (s != STEP_CONTINUE) {
          const int k = s == STEP_DONE ? ST_EVENT : ST_DROPPED;
          if (useB) kindB = k; else kindA = k;
        }
      }
      continue;
    }

    // ------------------------------------------------------------------ EVENT phase (one event of each photon)
    int st = active ? (useB ? kindB : kindA) : (int)ST_DONE;
    float z = 0.0f, dx = 0.0f, dy = 0.0f, dz = -1.0f, w = 0.0f;
    int ix = 1, iy = 1, iz = 1;
    float x = 0.0f, y = 0.0f;
    bool movedXY = false;   // x, y and their cell indices were (re)set by this event
    uint32_t index = 0u;    // photon number within the launch (F_ID)
    if (active) {
      z = ph[F_Z * kPool];
      dx = ph[F_DX * kPool]; dy = ph[F_DY * kPool]; dz = ph[F_DZ * kPool];
      ix = __float_as_int(ph[F_IX * kPool]); iy = __float_as_int(ph[F_IY * kPool]); iz = __float_as_int(ph[F_IZ * kPool]);
      w = ph[F_W * kPool];
      index = __float_as_uint(ph[F_ID * kPool]);
      const unsigned long long photon = (unsigned long long)A.firstPhoton + index;
      rng.id_lo = (uint32_t)photon; rng.id_hi = (uint32_t)(photon >> 32);
      rng.block = __float_as_uint(ph[F_BLOCK * kPool]);
      rng.have = 0;   // next() (retries only) starts on a block of this photon, never on another photon's leftovers
    }
    // ---- part A: endings that need no random number (tracer drop, exit through the top, arrival at a black surface)
    const bool isEv = st == ST_EVENT;
    const bool dropped = st == ST_DROPPED;                               // :488-489
    const bool atTop = isEv && z >= P.zMax;                              // :499-514
    const bool atSurface = isEv && !atTop && z <= surfaceZ;              // :515-531
    const bool atBlack = atSurface && blackSurface;                      // ... and :560-562
    wc.dropped += count_lanes(dropped);
    wc.top += count_lanes(atTop);
    wc.surf += count_lanes(atSurface);
    if (dropped || atTop || atBlack) {
      if (!dropped) tally.boundary(atTop, (iy - 1) * P.nx + (ix - 1), w);
      st = ST_NEW;
    }
    // ---- part B (uniform): hand out photon indices from the wave's reservoir
    const bool isNew = st == ST_NEW;
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
        flush_counters();
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
          const unsigned long long photon = (unsigned long long)(A.firstPhoton + mine);
          rng.id_lo = (uint32_t)photon; rng.id_hi = (uint32_t)(photon >> 32);
          rng.block = 0u;
          index = (uint32_t)mine;   // the host keeps launches of this kernel below 2^32 photons
        }
      }
    }
    // ---- part C: one random block per lane for this event, then the event itself
    bool didScatter = false, didRoulette = false, startedTrace = false;
    float target = 0.0f;
    if (active && st != ST_DONE) {
      rng.begin_event();
      if (st == ST_NEW) {                                               // :453-470, newPhotonStream_Directional :91-99
        const float px = rng.first(), py = rng.second();
        dx = A.solarDx; dy = A.solarDy; dz = A.solarDz;
        w = 1.0f;
        x = P.x0 + px * (P.xMax - P.x0);
        y = P.y0 + py * (P.yMax - P.y0);
        z = zStart;
        int i = min((int)exact_div(x - P.x0, P.deltaX, rcpDeltaX) + 1, P.nx);   // findXYIndicies :1359-1369
        int j = min((int)exact_div(y - P.y0, P.deltaY, rcpDeltaY) + 1, P.ny);
        if (fabsf(L.xE[i] - x) < spacingf(x)) i = i + 1;
        if (fabsf(L.yE[j] - y) < spacingf(y)) j = j + 1;
        ix = i == P.nx + 1 ? 1 : i;
        iy = j == P.ny + 1 ? 1 : j;
        iz = izStart;
        movedXY = true;
        st = ST_TRACE;
      } else if (st == ST_EVENT) {
        if (z <= surfaceZ) {                                            // :515-580 (reflecting Lambertian surface)
          iz = 1;
          z = surfaceZ;
          tally.down((iy - 1) * P.nx + (ix - 1), w);
          float mu = exact_sqrt(rng.first());
          while (!(fabsf(mu) > 2.0f * kTiny)) mu = exact_sqrt(rng.next());
          const float phi = (2.0f * kPi) * rng.second();
          w = w * P.albedo;
          if (w <= kTiny) st = ST_NEW;
          else { make_dircos(mu, phi, dx, dy, dz); st = ST_TRACE; }
        } else {                                                        // :581-689
          didScatter = true;
          int cell = cell_index(P, ix, iy, iz);
          const float extHere = cell_extinction<GRID>(P, L, ix, iy, iz);
          if (extHere <= 0.0f) {                                        // :606-632 (quirk Q2 kept)
            x = ph[F_X * kPool]; y = ph[F_Y * kPool];
            movedXY = true;
            if (x - L.xE[ix - 1] <= 0.0f && dx > 0.0f) {
              x = x - spacingf(x);
              ix = ix - 1;
              if (ix <= 0) { ix = P.nx; x = L.xE[ix - 1]; x = x - 2.0f * spacingf(x); }
            }
            if (y - L.yE[iy - 1] <= 0.0f && dy > 0.0f) {
              y = y - spacingf(y);
              iy = iy - 1;
              if (iy <= 0) { iy = P.ny; y = L.xE[iy - 1]; y = x - 2.0f * spacingf(y); }
            }
            if (z - L.zE[iz - 1] <= 0.0f && dz > 0.0f) { z = z - spacingf(z); iz = iz - 1; }
            cell = cell_index(P, ix, iy, iz);
          }
          float ssa;
          if (P.uniformSsa >= 0.0f) ssa = P.uniformSsa; else ssa = P.ssa[cell];
          if (ssa < 1.0f) {                                             // :642-649
            tally.absorbed((iy - 1) * P.nx + (ix - 1), cell, w * (1.0f - ssa));
            w = w * ssa;
          }
          if (P.useRR && w < 0.5f) {                                    // :673-680
            didRoulette = true;
            if (rng.spare() >= w / 1.0f) w = 0.0f; else w = 1.0f;
          }
          if (w <= kTiny) st = ST_NEW;
          else {
            int pfi;
            if (P.uniformPf >= 1) pfi = P.uniformPf; else pfi = max(P.pfIndex[cell], 1);
            const CompTables ct = P.comp0;
            const float cosS = scattering_cosine(rng.first(), ct.invCos + (size_t)(pfi - 1) * ct.nInv, ct.nInv,
                                                 refined_rcp((float)ct.nInv));
            next_direct(rng, cosS, dx, dy, dz);                         // :684-687
            st = ST_TRACE;
          }
        }
      }
      if (st == ST_TRACE) {                                             // :480
        target = -sample_log(fmaxf(kTiny, rng.path()));
        startedTrace = true;
        ph[F_DX * kPool] = dx; ph[F_DY * kPool] = dy; ph[F_DZ * kPool] = dz;
        ph[F_RX * kPool] = refined_rcp(dx); ph[F_RY * kPool] = refined_rcp(dy); ph[F_RZ * kPool] = refined_rcp(dz);
        ph[F_ACC * kPool] = 0.0f; ph[F_TARGET * kPool] = target;
        ph[F_W * kPool] = w;
        ph[F_Z * kPool] = z; ph[F_IZ * kPool] = __int_as_float(iz);
        if (movedXY) {
          ph[F_X * kPool] = x; ph[F_Y * kPool] = y;
          ph[F_IX * kPool] = __int_as_float(ix); ph[F_IY * kPool] = __int_as_float(iy);
        }
      }
      ph[F_ID * kPool] = __uint_as_float(index);
      ph[F_BLOCK * kPool] = __uint_as_float(rng.block);
    }
    if (active) { if (useB) kindB = st; else kindA = st; }
    wc.scat += count_lanes(didScatter);
    wc.roul += count_lanes(didRoulette);
    wc.calls += count_lanes(startedTrace);
  }

  // ------------------------------------------------------------------ epilogue: flush tallies + counters
  __syncthreads();
  if (P.ldsTallies) {
    const int ncol = P.nx * P.ny;
    for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
      const float u = L.tUp[i], d = L.tDown[i], a = L.tAbs[i];
      if (u != 0.0f) add_global(P.tally + P.oUp + i, u);
      if (d != 0.0f) add_global(P.tally + P.oDown + i, d);
      if (a != 0.0f) add_global(P.tally + P.oAbs + i, a);
    }
  }
  const double draws = wave_sum((double)rng.total());
  flush_counters();
  if (lane == 0 && draws != 0.0) unsafeAtomicAdd(P.tally + P.oCnt + I3RC_CNT_RNG_DRAWS, draws);
}

}  // namespace i3rc
