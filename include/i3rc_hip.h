/*
 * include/i3rc_hip.h -- C ABI of the MI355X (gfx950) photon-tracing integrator.
 *
 * Drop-in boundary for the hot path of the I3RC community Monte Carlo model:
 *   Integrators/monteCarloRadiativeTransfer.f95  computeRadiativeTransfer (:262-398) -> computeRT (:400-707)
 * The reference has no FFI; its boundary is the Fortran-95 module `monteCarloRadiativeTransfer`.  The
 * Fortran shell in i3rc-monte-carlo-model_amd/fortran/ keeps that module API and calls these entry points
 * through iso_c_binding; Python tests and bench.py bind them with ctypes.
 *
 * Conventions
 *   - return 0 on success, non-zero on failure; i3rc_hip_last_error() gives the text the shell hands to
 *     setStateToFailure (Code/ErrorMessages.f95:159-233).
 *   - host arrays are borrowed for the duration of the call only; device state is owned by the handle.
 *   - grids are x-fastest: cell (ix,iy,iz) 1-based  ->  [(iz-1)*ny + (iy-1)]*nx + (ix-1); per-component
 *     arrays add a slowest component axis (Fortran (nx,ny,nz,ncomp) column-major, passed as is).
 *   - tallies come back RAW (un-normalised sums over photons) in float64; the caller applies the
 *     normalisation of computeRadiativeTransfer :353-395 (the shell does; so does i3rc_hip_normalise).
 *   - no torch types, no C++ types: plain pointers and sizes.
 */
#ifndef I3RC_HIP_H
#define I3RC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct i3rc_hip_integrator i3rc_hip_integrator;

/* The reference limits neither (Code/opticalProperties.f95:133-230, monteCarloRadiativeTransfer.f95:1026-1045; its driver reads
 * at most 20 directions, Example-Drivers/monteCarloDriver.f95:63 maxNumRad): here a ray record carries either in 8 bits. */
#define I3RC_MAX_COMPONENTS 255
#define I3RC_MAX_DIRECTIONS 255

/* Algorithm switches and scalars: the private components of type(integrator)
 * (monteCarloRadiativeTransfer.f95:50-142) that specifyParameters (:830-1069) sets. */
typedef struct i3rc_params {
  float   surfaceAlbedo;                   /* :70  used when useSurfaceBDRF == 0 */
  int32_t useSurfaceBDRF;                  /* :83  BRDF grid given by i3rc_hip_set_surface */
  int32_t useRayTracing;                   /* :63  1 = photon tracing (0 = max cross-section) */
  int32_t useRussianRoulette;              /* :65  RussianRouletteW fixed at 1 (:66) */
  int32_t useHybridPhaseFunsForIntenCalcs; /* :118 */
  int32_t numOrdersOrigPhaseFunIntenCalcs; /* :120 */
  int32_t useRussianRouletteForIntensity;  /* :123 */
  float   zetaMin;                         /* :124 */
  int32_t limitIntensityContributions;     /* :127 */
  float   maxIntensityContribution;        /* :128 */
} i3rc_params;

/* Photon source for one batch: what new_PhotonStream (Code/monteCarloIllumination.f95:46-50) encodes. */
typedef struct i3rc_source {
  int32_t kind;          /* 0 = Directional generated on device (newPhotonStream_Directional :62-104);
                            1 = explicit stream: host arrays x,y,z (0..1), mu, phi (radians), n each */
  float   solarMu;       /* kind 0: as passed to new_PhotonStream (sign ignored: mu = -abs(solarMu)) */
  float   solarAzimuth;  /* kind 0: degrees */
  const float *x, *y, *z, *mu, *phi; /* kind 1 */
} i3rc_source;

/* Counter block appended to the tallies (float64 so the whole buffer reduces with one all-reduce). */
enum {
  I3RC_CNT_PHOTONS = 0,   /* photons started (numPhotonsProcessed, :459) */
  I3RC_CNT_DROPPED,       /* tracer-error drops (nBad, :488) */
  I3RC_CNT_CELL_STEPS,    /* accumulateExtinctionAlongPath iterations, photon paths */
  I3RC_CNT_SCATTERINGS,   /* scattering events (:591) */
  I3RC_CNT_SURFACE_HITS,  /* fluxDown tallies (:531) */
  I3RC_CNT_EXITS_TOP,     /* fluxUp tallies (:513) */
  I3RC_CNT_ROULETTE,      /* roulette plays (:673) */
  I3RC_CNT_SHADOW_STEPS,  /* tracer iterations spent in local-estimate rays (:1517-1595) */
  I3RC_CNT_TRACER_CALLS,  /* calls of the tracer, all kinds */
  I3RC_CNT_RNG_DRAWS,     /* uniform deviates consumed */
  I3RC_CNT_RAYS_SKIPPED,  /* local-estimate rays not traced at all: small contributions that lost their roulette (:1554) -- the
                             deviate is independent of the path, so the contribution is 0 whatever a trace would find */
  I3RC_NUM_COUNTERS = 16
};

/* Layout (in float64 elements) of the packed tally buffer of a handle. */
typedef struct i3rc_tally_layout {
  int64_t fluxUp, fluxDown, fluxAbsorbed; /* offsets; nx*ny each                          (:135-136) */
  int64_t volumeAbsorption;               /* nx*ny*nz                                      (:137)     */
  /* (fluxAbsorbed of a raw block is the sum of its column's volumeAbsorption words, formed on the device after every launch: the
   *  reference adds one increment to both, :644-647, the kernels tally the cell) */
  int64_t intensityByComponent;           /* (ncomp+1)*nDir*nx*ny, component 0 = surface   (:139-140) */
  int64_t intensityExcess;                /* (ncomp+1)*nDir                                (:130)     */
  int64_t counters;                       /* I3RC_NUM_COUNTERS                                        */
  int64_t total;                          /* number of float64 elements                               */
} i3rc_tally_layout;

/* ---- lifetime -------------------------------------------------------------------------------------- */

/* new_Integrator (:162-254): deep-copies the optical-property grids to `device`.
 * totalExt[nz][ny][nx]; cumExt/ssa/pfIndex [ncomp][nz][ny][nx] as produced by
 * getOpticalPropertiesByComponent (Code/opticalProperties.f95:429-539) incl. the 1+spacing(1.) nudge (:233). */
int i3rc_hip_create(i3rc_hip_integrator **h, int device, int nx, int ny, int nz, int ncomp,
                    const float *xEdges, const float *yEdges, const float *zEdges,
                    const float *totalExt, const float *cumExt, const float *ssa, const int32_t *pfIndex);

/* finalize_Integrator (:1258-1349) */
int i3rc_hip_destroy(i3rc_hip_integrator *h);

const char *i3rc_hip_last_error(const i3rc_hip_integrator *h); /* h may be NULL: error of the last failed create */

/* ---- setup ----------------------------------------------------------------------------------------- */

/* inversePhaseFunctions(comp)%values(nSteps, nEntries) (:104-105, built by tabulateInversePhaseFunctions
 * :1809-1861); comp is 1-based; t is [nEntries][nSteps]. */
int i3rc_hip_set_inverse_table(i3rc_hip_integrator *h, int comp, int nSteps, int nEntries, const float *t);

/* tabulatedPhaseFunctions / tabulatedOrigPhaseFunctions (:100-103, :1863-1923); orig may equal hybrid. */
int i3rc_hip_set_forward_tables(i3rc_hip_integrator *h, int comp, int nSteps, int nEntries,
                                const float *hybrid, const float *orig);

/* specifyParameters (:830-1069), scalar switches */
int i3rc_hip_set_params(i3rc_hip_integrator *h, const i3rc_params *p);

/* specifyParameters(surfaceBDRF=) (:954-956): Lambertian albedo grid brdf[nys][nxs] on its own edges
 * (Code/surfaceProperties.f95:34-38; uniform surface = 1x1 with edges (0, huge) :98-117). */
int i3rc_hip_set_surface(i3rc_hip_integrator *h, int nxs, int nys, const float *xsEdges, const float *ysEdges,
                         const float *brdf);

/* specifyParameters(intensityMus=, intensityPhis=) (:1026-1045): dirCos[nDir][3] already converted with
 * makeDirectionCosines (:2041-2059).  nDir = 0 switches computeIntensity off. */
int i3rc_hip_set_directions(i3rc_hip_integrator *h, int nDir, const float *dirCos);

/* ---- tallies --------------------------------------------------------------------------------------- */

int i3rc_hip_get_tally_layout(const i3rc_hip_integrator *h, i3rc_tally_layout *layout);

/* Optional: accumulate into caller-owned DEVICE memory (e.g. a torch tensor handed to RCCL) of at least
 * layout.total float64 elements.  NULL restores the handle's own buffer. */
int i3rc_hip_bind_tally_buffer(i3rc_hip_integrator *h, void *devicePtr, size_t bytes);

/* Run on the caller's HIP stream (hipStream_t as void*), so that launches are ordered with the caller's own work on
 * it (zeroing a bound tally buffer, an RCCL all-reduce of it).  NULL is a stream like any other: HIP's null stream,
 * which is what torch.cuda.current_stream() is unless the caller changed it.  i3rc_hip_use_own_stream goes back to
 * the private non-blocking stream every handle starts with. */
int i3rc_hip_set_stream(i3rc_hip_integrator *h, void *hipStream);
int i3rc_hip_use_own_stream(i3rc_hip_integrator *h);

/* computeRadiativeTransfer :296-309: zero all tallies and counters (asynchronous on the stream). */
int i3rc_hip_zero_tallies(i3rc_hip_integrator *h);

/* ---- the hot path ---------------------------------------------------------------------------------- */

/* computeRT (:400-707) for one batch of nPhotons, ASYNCHRONOUS on the handle's stream; tallies accumulate.
 * RNG: per-photon Philox4x32-10 stream keyed by (seed0, seed1) -- the driver's seed=(/iseed, batch/)
 * (Example-Drivers/monteCarloDriver.f95:277) -- with the photon's index (firstPhoton + i) as counter, so a
 * photon's trajectory does not depend on how batches are split across launches or GPUs.
 * An explicit source (kind 1) is checked as the reference's photon-stream constructors check theirs
 * (Code/monteCarloIllumination.f95:78-83, :124, :204, :369-371): relative positions in [0, 1], |mu| in (tiny, 1],
 * finite azimuth; anything else fails with the reference's message.
 * Where the reference indexes out of bounds or never returns (a photon starting at zIndex nz + 1, max cross-section
 * on a domain without extinction, makePeriodic on a position 2^24 widths away, a NaN step) the kernels follow the
 * rules written down in DESIGN.md section 3; they never read outside their arrays and every wave ends.
 * Environment: I3RC_POISON=1 fills fresh device allocations with 0xFF bytes (debugging aid). */
int i3rc_hip_launch_batch(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t firstPhoton,
                          int64_t nPhotons, const i3rc_source *src);

/* The batch loop of a driver as ONE call (Example-Drivers/monteCarloDriver.f95:283-326: per batch a random-number
 * sequence seeded (/iseed, batch/), a photon stream of numPhotonsPerBatch photons, computeRadiativeTransfer,
 * reportResults).  Batch k = 0 .. nBatches-1 is traced with the key (seed0, seed1 + k) into tallies of its own;
 * its raw tallies (layout of i3rc_hip_get_tally_layout, float64; normalise them with i3rc_hip_normalise) are written
 * to hostTallies + k * layout.total.  Every batch equals i3rc_hip_zero_tallies + i3rc_hip_launch_batch(seed0,
 * seed1 + k, 0, nPhotons) + i3rc_hip_fetch_tallies, photon for photon: the same integer work counters, tallies equal to
 * the order of the float64 additions.
 * A batch ends with a long tail -- the few photons of a million that scatter a thousand times keep a handful of
 * wavefronts busy for a millisecond --, so batches of the reference drivers' size (1e5 ... 1e6 photons) must overlap:
 *   - FUSED (problems of the common class: regular x / y grid, ray tracing, one component, no BRDF grid -- with or, since
 *     round 4, without radiance directions; two batches or more of at most 2e7 photons): a group of consecutive batches is
 *     ONE kernel launch (photon_kernel<PhiloxBatchStream, ...>).  Every lane carries the batch of its photon in its Philox
 *     key, a local-estimate ray the batch of its photon in its info word; tallies go to per-batch blocks in global memory
 *     (spread over replicas where a domain has few columns): one batch's tail is filled by the next batch's photons
 *     inside the launch.  Groups hold about 2.5e8 photons (bounded by 1 GiB of tally blocks), up to three groups are in
 *     flight.  Work counters: flux problems gather them per lane and hand them over per batch -- every counter of a batch is
 *     what a launch of its own gives.  Radiance kernels have no register for that: photons and dropped photons (what the
 *     normalisation needs) are exact per batch, their other counters (steps, scatterings, ray starts ...) are counted per
 *     wavefront and handed to the batch the wavefront was given last -- exact over the batches of a group, not per batch.
 *     The TALLIES are per batch in every case.  Timings: profiles/r05_fused_timing.txt (step cloud, radar and Landsat fields,
 *     with and without radiances, against one launch per batch).
 *   - otherwise up to inFlight batches (1..8; 0 = 6) are on the device at a time, each a launch on a HIP stream of its
 *     own with its own tally buffer (GPU_MAX_HW_QUEUES=8 in the environment gives these another 10-15 %).
 * i3rc_hip_set_batch_fusion chooses between the two.  Directional sources only (an explicit stream differs from batch
 * to batch).  Synchronous: returns when all batches are in hostTallies; the handle's own tally buffer and stream are
 * not touched. */
int i3rc_hip_run_batches(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons,
                         const i3rc_source *src, int inFlight, double *hostTallies);

/* Layout (in float64 elements) of a block of batch moments (i3rc_hip_run_batches_moments): the quantities reportResults (:711-826)
 * hands out, in its shapes. */
typedef struct i3rc_moments_layout {
  int64_t fluxUp, fluxDown, fluxAbsorbed;                /* nx*ny each ([ny][nx])                       */
  int64_t volumeAbsorption;                              /* nx*ny*nz ([nz][ny][nx])                     */
  int64_t intensity;                                     /* nDir*nx*ny ([nDir][ny][nx])                 */
  int64_t absorbedProfile;                               /* nz         (:780)                           */
  int64_t meanFluxUp, meanFluxDown, meanFluxAbsorbed;    /* 1 each     (:739-742)                       */
  int64_t meanIntensity;                                 /* nDir                                        */
  int64_t total;
} i3rc_moments_layout;
int i3rc_hip_get_moments_layout(const i3rc_hip_integrator *h, i3rc_moments_layout *layout);

/* A driver's batch loop with its STATISTICS gathered on the device.  The reference's drivers keep, of every batch, only the
 * first two moments of what reportResults returns (Example-Drivers/monteCarloDriver.f95:300-321: stats(1) += x, stats(2) += x**2
 * for the domain means, the pixel fluxes, the absorption profile and volume, the radiances; reduced over processes at :333-352,
 * turned into mean and standard error at :358-378).  This call traces batches k = 0 .. nBatches-1 exactly as i3rc_hip_run_batches
 * does (keys (seed0, seed1 + k), fused groups where the problem allows), normalises every batch's block ON THE DEVICE as
 * i3rc_hip_normalise does (:327-395, float64, rounded to the reference's real(4)) and adds x and x*x per element -- and per
 * domain mean / profile layer / mean radiance -- to two blocks of the layout above, in float64: sum[e] = sum over the batches
 * of x_k[e], sumSquares[e] likewise of x_k[e]^2.  Only those two blocks come back (and, optionally, the work counters summed
 * over the batches, I3RC_NUM_COUNTERS values): the per-batch tally blocks -- 5 MB per batch for a 128 x 128 x 36 domain --
 * are never copied to the host.  Directional sources only.  Synchronous. */
int i3rc_hip_run_batches_moments(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons,
                                 const i3rc_source *src, double *sum, double *sumSquares, double *counters);

/* computeRadiativeTransfer (:262-398) for ONE batch of a driver's loop -- i3rc_hip_zero_tallies + i3rc_hip_launch_batch(seed0,
 * seed1, 0, nPhotons) + i3rc_hip_fetch_tallies into hostTallies, in a tally buffer and on a stream of the library's own --
 * that LOOKS AHEAD: the reference's drivers call computeRadiativeTransfer once per batch with
 * seed = (/iseed, batch/) (monteCarloDriver.f95:277, :287), and a call cannot return before the batch's last photon has
 * (the tail of a launch: see i3rc_hip_run_batches).  When a call is the same batch as the previous call with the next
 * seed word, the library takes that for such a loop and traces the following batches (seed1 + 1, seed1 + 2, ...) behind
 * this one: on problems whose batches can share a grid in FUSED GROUPS of 8, 16, 32 ... 256 batches (up to three groups
 * under way), else up to lookAhead (0..7; 0 = never look ahead) single batches; the next call then finds its batch under
 * way or done.  A call that is not the expected batch (other seed, photon count, sun), and every change of the problem
 * (tables, parameters, surface, directions, tuning), calls the work launched ahead off -- fused groups poll an abort word
 * and end within one chunk per wavefront --, waits for it and forgets it: results never depend on the guess, a wrong
 * guess (and the end of the loop) costs the device a few milliseconds.  So does i3rc_hip_destroy.  Launches made ahead are
 * not recorded in i3rc_hip_kernel_ms_history.  A loop that is only guessed at keeps each of its three slots within 96 MiB of
 * pinned memory (groups of a Landsat-sized field then hold 18 batches); a group that cannot be launched for want of memory is
 * not tried again at every call -- such a loop goes on with single batches launched ahead.  The unchanged reference driver
 * end to end: profiles/r05_driver_timing.txt.  Directional sources only.  Synchronous. */
int i3rc_hip_compute_batch(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t nPhotons,
                           const i3rc_source *src, int lookAhead, double *hostTallies);

/* Announces a driver's batch loop to the look-ahead of i3rc_hip_compute_batch: the caller is going to ask for the batches
 * with the seed words seed1, seed1 + 1 ... seed1 + nBatches - 1 (nPhotons photons each, this sun), one
 * i3rc_hip_compute_batch call per batch, in this order.  On problems whose batches can share a grid (see
 * i3rc_hip_run_batches) the library starts tracing them at once, in fused groups of 32, 64, 128, 256 ... batches, three groups
 * under way at a time and none beyond the announced loop (their slots are made for the loop's largest group at once); *accepted is 1.  The calls then find their batches done or under
 * way while the caller works on the ones it has: the shell's computeRadiativeTransferBatches / selectBatchResults stream
 * a driver's loop this way.  On other problems -- and when the first group cannot be launched (memory: the reason stays in
 * i3rc_hip_last_error) -- nothing happens, the call returns 0 and *accepted is 0: i3rc_hip_run_batches (or the calls' own
 * look-ahead) serves those.  A call for any other batch, and every change of the problem, calls the announced work
 * off, as for any look-ahead.  Asynchronous. */
int i3rc_hip_expect_batches(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int nBatches, int64_t nPhotons,
                            const i3rc_source *src, int *accepted);

/* Test hook: same kernel, but every uniform deviate is read from `randoms` (float32, e.g. the reference's
 * MT19937 stream): photon i consumes randoms[drawStart[i]], randoms[drawStart[i]+1], ... in the reference's
 * draw order.  Optional per-photon outputs (host arrays of n, may be NULL):
 * fate (0 exit top, 1 died at surface, 2 roulette kill, 3 dropped), last flux column, its weight, order,
 * number of deviates consumed.  Synchronous. */
int i3rc_hip_run_replay(i3rc_hip_integrator *h, int64_t nPhotons, const i3rc_source *src,
                        const float *randoms, int64_t nRandoms, const int64_t *drawStart,
                        int32_t *fate, int32_t *fateColumn, float *fateWeight, int32_t *fateOrder,
                        int32_t *drawsUsed);

/* Test hook: nRays independent calls of accumulateExtinctionAlongPath (:1654-1807) on the device.
 * pos/dir [nRays][3], idx [nRays][3] (1-based) in/out; target[nRays] (<0: trace to the boundary);
 * tau[nRays] out (-2 = tracer error), steps[nRays] out.  Synchronous. */
int i3rc_hip_trace_rays(i3rc_hip_integrator *h, int64_t nRays, const float *dir, float *pos, int32_t *idx,
                        const float *target, float *tau, int32_t *steps);

/* Wait for the stream; copy the packed tally buffer (layout.total float64) to the host. */
int i3rc_hip_synchronize(i3rc_hip_integrator *h);
int i3rc_hip_fetch_tallies(i3rc_hip_integrator *h, double *hostTallies);

/* computeRadiativeTransfer :327-395 on a host copy of the packed tallies: redistribute intensityExcess,
 * divide by photons per column (regular grid: N/(nx*ny), else column-area weighted), volumeAbsorption also
 * by layer depth.  Output float32 arrays (any may be NULL) in reportResults' shapes (:711-826):
 * fluxUp/fluxDown/fluxAbsorbed [ny][nx], volumeAbsorption [nz][ny][nx], intensity [nDir][ny][nx],
 * intensityByComponent [ncomp+1][nDir][ny][nx]. */
int i3rc_hip_normalise(const i3rc_hip_integrator *h, const double *hostTallies,
                       float *fluxUp, float *fluxDown, float *fluxAbsorbed, float *volumeAbsorption,
                       float *intensity, float *intensityByComponent);

/* Timing of the most recent i3rc_hip_launch_batch, measured with HIP events on the launch stream.
 * Synchronises.  Returns milliseconds in *ms. */
int i3rc_hip_last_kernel_ms(i3rc_hip_integrator *h, float *ms);
/* Durations of the last n (<= 64) launches, oldest first; events are recorded around each launch without
 * synchronising, so a timed region of many launches can be read back afterwards. */
int i3rc_hip_kernel_ms_history(i3rc_hip_integrator *h, int n, float *ms);

/* Number of kernel launches timed so far (a batch longer than the launch limit is several launches). */
int64_t i3rc_hip_timed_launch_count(const i3rc_hip_integrator *h);

/* Name of the kernel instantiation the most recent launch ran, as rocprofv3 lists it without the namespace
 * (e.g. "photon_kernel<PhiloxStream, false, false, GRID_LDS>"); "" before the first launch. */
const char *i3rc_hip_last_kernel_name(const i3rc_hip_integrator *h);

/* Experiment knobs (not part of the reference API): lanes that must be waiting before a wavefront runs its
 * event phase (1..64; 0 = default: every wave adapts it to its photons' voxel steps per event, 44 - 2 steps per event
 * -- radiance kernels 44 - 1.2 steps per event -- within 12..44) and workgroups per CU (0 = occupancy query). */
int i3rc_hip_set_tuning(i3rc_hip_integrator *h, int evThreshold, int blocksPerCU);
/* ... and lanes whose local-estimate (shadow) ray has ended before the wavefront runs its light phase (1..64;
 * 0 = default: adapted to the length of the rays, 70 / sqrt(steps per ray) within 16..32; radiance runs only). */
int i3rc_hip_set_light_threshold(i3rc_hip_integrator *h, int lanes);

/* A Directional batch longer than this many photons is cut into several kernel launches over consecutive photon
 * ranges (same result: every photon has its own random stream).  0 = default, 2^22 photons per compute unit (about
 * 1e9 on an MI355X): the work counters a wave hands over are 32-bit (until round 4 also: workgroups kept partial sums in float32,
 * which stops counting at 2^24; they are float64 now). */
int i3rc_hip_set_launch_limit(i3rc_hip_integrator *h, int64_t photons);

/* Fusion of a loop's batches (i3rc_hip_run_batches, the look-ahead of i3rc_hip_compute_batch): -1 = automatic (default:
 * problems the specialised flux kernels run, two batches or more of at most 2e7 photons), 0 = never (every batch a launch
 * of its own, several in flight), 1 = whenever the problem allows.  Environment: I3RC_FUSED=0 switches fusion off for the
 * process, I3RC_FUSED_CHUNK (photons a wavefront takes from one batch at a time, default 512) and
 * I3RC_FUSED_GROUP_PHOTONS (photons per fused launch, default 2.5e8) are tuning knobs. */
int i3rc_hip_set_batch_fusion(i3rc_hip_integrator *h, int mode);

/* Test / tuning knob.  A launch normally (AUTO) runs the one-photon-per-lane kernel specialised for the common
 * problem class (regular x / y grid, ray tracing, one component, no BRDF grid, Directional source) when the problem is
 * in it, else the general kernel.  All kernels trace the same photon paths from the same per-photon random streams,
 * so tests run one against the other.
 *   GENERAL: always the general kernel;  LANE: same choice as AUTO;  RING: as AUTO, but a radiance problem with ONE
 *   direction goes through the event ring like any other (AUTO gives such problems the kernels without a ring, in which
 *   the event phase itself makes the event's ray ready and a ready store of two wavefronts gathers the survivors). */
enum { I3RC_KERNEL_AUTO = 0, I3RC_KERNEL_GENERAL = 1, I3RC_KERNEL_LANE = 2, I3RC_KERNEL_RING = 3 };
int i3rc_hip_select_kernel(i3rc_hip_integrator *h, int variant);
/* = i3rc_hip_select_kernel(h, on ? I3RC_KERNEL_GENERAL : I3RC_KERNEL_AUTO) */
int i3rc_hip_force_general_kernel(i3rc_hip_integrator *h, int on);

/* Test / tuning knob: where the kernels read the extinction field (getOpticalPropertiesByComponent's totalExt, :172) from.
 *   AUTO (default): LDS when the field fits beside the rest; else COLUMN RECORDS when the field has them -- every column's cells
 *   with extinction are one run of layers holding one value (the I3RC Landsat scene: per column a cloud of vertically uniform
 *   extinction), kept as 8 bytes per column: first layer, run length, value; else the plain field while it fits in an XCD's L2
 *   (4 MB); else a copy in bricks of 32 cells.  LINEAR / BRICKS / COLUMNS force one of them (COLUMNS fails when the field has no
 *   such form), also for i3rc_hip_trace_rays.  Every place returns what the field holds, bit for bit: photon paths do not depend
 *   on it (tests run one against the other).  Environment: I3RC_COLUMNS=0 takes the column records out of AUTO. */
enum { I3RC_GRID_AUTO = 0, I3RC_GRID_LINEAR = 1, I3RC_GRID_BRICKS = 2, I3RC_GRID_COLUMNS = 3 };
int i3rc_hip_select_grid_place(i3rc_hip_integrator *h, int place);
/* 1 when the field has column records, else 0 */
int i3rc_hip_has_column_records(const i3rc_hip_integrator *h);
/* The test i3rc_hip_create applies, as host code of its own (no device needed): does the field totalExt [nz][ny][nx] have the form --
 * in every column the cells whose extinction is not +0 are one run of layers holding one value, bit for bit?  Returns 1 / 0 and,
 * when records is not NULL, writes [ny * nx][2] words: the value's bits; first layer (1-based) | (run length - 1) << 16. */
int i3rc_hip_column_records(int nx, int ny, int nz, const float *totalExt, uint32_t *records);
/* ... and the same OVER A BASE PROFILE: totalExt(x, y, z) = base(z) + (z within the column's run ? value(x, y) : 0) in float32 arithmetic --
 * a cloud scene of one run of one value per column plus a horizontally uniform component (gas, aerosol): what several components add up to
 * (getOpticalPropertiesByComponent, Code/opticalProperties.f95:523-537).  i3rc_hip_create keeps such a field as the records and base[nz]
 * when the domain has several components.  Host code, no device needed; returns 1 / 0; records as above, base[nz] (must not be NULL). */
int i3rc_hip_column_records_base(int nx, int ny, int nz, const float *totalExt, uint32_t *records, float *base);

/* Test / tuning knob: 0 = a plain launch adds every tally straight to the float64 buffer in global memory instead of gathering a
 * workgroup's partial sums (float64 as well) in LDS first: the same float64 additions in one more order.  Default 1.
 * Environment: I3RC_LDS_TALLIES=0 for the whole process. */
int i3rc_hip_set_lds_tallies(i3rc_hip_integrator *h, int on);

/* Test hook, host code only (no device needed): the carve-up of a workgroup's dynamic LDS that photon_kernel makes -- the very
 * function the kernel sets its pointers from and the launch sizes its allocation from (csrc/tracer.hpp, lds_plan).
 *   q[0..16]  = nx, ny, nz, ncomp, nDir, ldsTallies, ldsIntensity, rayQueueCap, clearNx, clearShift,
 *               queues (radiance kernel with ray queues), direct (its one-direction form), grid place (0 LDS, 1 global, 2 bricks,
 *               3 column records), intensity (radiance kernel), waves per workgroup, words of the inverse table kept in LDS,
 *               ldsVolume (volume-absorption tallies gathered in LDS: absorbing domains of few cells)
 *   out[0..11] = word offsets of: x edges, y edges, z edges, flux tallies (up, down), directions, per-direction ray constants, ray
 *               queues, radiance tallies, extinction grid / clear-air map, inverse table; the end (= words a launch allocates); and
 *               the volume-absorption tallies (which lie between the flux tallies and the directions). */
int i3rc_hip_lds_plan(const int32_t *q, int32_t *out);

/* Test hook: the raw Philox4x32-10 blocks (out[n][blocksPerPhoton][4]) of photons firstPhoton..+n-1 and the
 * float32 deviates the photon streams derive from them (outf, same shape). */
int i3rc_hip_philox_blocks(i3rc_hip_integrator *h, uint32_t seed0, uint32_t seed1, int64_t firstPhoton, int64_t n,
                           int blocksPerPhoton, uint32_t *out, float *outf);

/* Test hook: over n pairs (num[i], den[i]) counts where the kernels' reciprocal-based correctly rounded division
 * differs from IEEE num/den, and where their corrected hardware sqrt differs from sqrtf(|num|). */
int i3rc_hip_arith_check(i3rc_hip_integrator *h, int64_t n, const float *num, const float *den, int64_t *divMismatch,
                         int64_t *sqrtMismatch);

/* Test hooks: findIndex (Code/numericUtilities.f95:195-248) and computeSurfaceReflectance (Code/surfaceProperties.f95:121-162) as the
 * photon kernels evaluate them (csrc/tracer.hpp find_index, surface_reflectance), one thread per value: out[i] = findIndex(values[i],
 * table(1:n), firstGuess[i]) (firstGuess NULL, or an entry <= 0: the argument is absent); out[i] = the reflectance at (x[i], y[i]) of
 * the surface set by i3rc_hip_set_surface.  tests/test_gpu_ref_numerics.py holds both against the reference's own routines. */
int i3rc_hip_find_index(i3rc_hip_integrator *h, int n, const float *table, int64_t m, const float *values, const int32_t *firstGuess,
                        int32_t *out);
int i3rc_hip_surface_reflectance(i3rc_hip_integrator *h, int64_t m, const float *x, const float *y, float *out);

/* Library / device probe that needs no GPU work: returns the number of HIP devices (or -1). */
int i3rc_hip_device_count(void);
const char *i3rc_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
