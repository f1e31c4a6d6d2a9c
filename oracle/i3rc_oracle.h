/*
 * oracle/i3rc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, IEEE float32, scalar, single thread) of the photon-tracing hot path of the
 * I3RC community Monte Carlo model.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (HIP) path never calls it.
 *
 * Parity status: PARITY UNPINNED by a reference binary or by reference-held vectors.  The reference needs
 * netCDF-Fortran, which this image lacks, and the build rules forbid stand-in libraries, so no oracle/_ref binary
 * exists; the reference ships no tests, golden vectors or expected outputs.  What the restatement is pinned against:
 * (i) closed forms of the very loop -- Beer-Lambert transmission per column and layer at omega = 0, an empty domain
 * over a Lambertian surface, first-order scattering of a slab (tests/test_closed_form.py, independent of this code) --
 * and MULTIPLE scattering: fluxes and radiances of homogeneous slabs (optical depth 0.1 / 1 / 10, omega 1 / 0.9, surface albedo
 * 0 / 0.5, planeParallel.nml among them) against an adding-doubling solver in float64 that shares nothing with this code
 * (tests/plane_parallel_solver.py, tests/test_plane_parallel.py);
 * (ii) the product's independent Fortran implementation of the six photon-stream constructors, bit for bit
 * (tests/test_fortran_shell.py); (iii) the canonical MT19937 known answers and the reference-run numbers recorded at
 * survey time in SURVEY.md 6 / 8c and BASELINE.md 2 (RNG KATs for seed=(/10,1/), inverse and forward table spot values,
 * planeParallel.nml result 0.16420 / 0.83580 / 0.00363, step-cloud fluxes and per-photon work counters:
 * tests/test_oracle_pins.py).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic is float32 with no FMA contraction (compile with -ffp-contract=off), matching the
 * reference as built for baseline x86-64.  Cell indices are 1-based as in the Fortran.
 */
#ifndef I3RC_ORACLE_H
#define I3RC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Code/RandomNumbersForMC.f95 ------------------------------------------------------------------ */
typedef struct {
  int32_t state[624];
  int32_t cur;
  int64_t draws; /* instrumentation: number of ints drawn */
} orc_mt;

void    orc_mt_seed_scalar(orc_mt *t, int32_t seed);                 /* :169-185 */
void    orc_mt_seed_vector(orc_mt *t, const int32_t *seed, int n);   /* :187-239 */
int32_t orc_mt_int(orc_mt *t);                                       /* :243-258 */
double  orc_mt_double(orc_mt *t);                                    /* :275-290 */
float   orc_mt_real(orc_mt *t);                                      /* :292-299 */

/* ---- Code/numericUtilities.f95 --------------------------------------------------------------------- */
float orc_spacing(float x);                                          /* Fortran SPACING() for real(4) */
int   orc_find_index(float value, const float *table, int n, int firstGuess); /* :195-248, 1-based; firstGuess<=0: absent */
void  orc_lobatto(int n, float *mus, float *weights);                /* :15-102 */
void  orc_gauss_legendre(int n, float *mus, float *weights);         /* :104-173 */
void  orc_legendre_polynomials(int maxL, const float *mus, int m, float *P);   /* :175-193, P[j * (maxL + 1) + l] */

/* ---- Code/scatteringPhaseFunctions.f95, Code/inversePhaseFunctions.f95 ----------------------------- */
/* Legendre phase function: coefficients l=1..nCoef (P0=1 implicit), values at angles (radians). :486-496 */
void orc_phase_values_legendre(const float *coef, int nCoef, const float *angles, int nAngles, float *values);
/* Tabulated phase function, interpolated linearly in cos(angle). :497-524 */
void orc_phase_values_tabulated(const float *tabAngles, const float *tabValues, int nTab,
                                const float *angles, int nAngles, float *values);
/* normalizePhaseFunction :1329-1345 (what new_PhaseFunction / new_PhaseFunctionTable do to tabulated values) */
void orc_normalize_tabulated(const float *tabAngles, const float *tabValues, int nTab, float *normalized);
/* Inverse (CDF -> angle) tables. inversePhaseFunctions.f95:68-176 */
void orc_inverse_table_legendre(const float *coef, int nCoef, int nSteps, float *table);
void orc_inverse_table_tabulated(const float *tabAngles, const float *tabValues, int nTab, int nSteps, float *table);
/* Forward tables equally spaced in angle 0..pi. monteCarloRadiativeTransfer.f95:1896-1901 */
void orc_forward_table_legendre(const float *coef, int nCoef, int nSteps, float *table);
void orc_forward_table_tabulated(const float *tabAngles, const float *tabValues, int nTab, int nSteps, float *table);
/* Hybrid (Gaussian forward peak) tables. monteCarloRadiativeTransfer.f95:1925-2039.  values is [nEntries][nSteps] */
void orc_hybrid_phase_functions(int nSteps, int nEntries, const float *values, float widthDegrees, float *newValues);

/* ---- Integrators/monteCarloRadiativeTransfer.f95 ---------------------------------------------------- */
typedef struct {
  /* grid (borrowed pointers) */
  int nx, ny, nz, ncomp;
  const float *xEdges, *yEdges, *zEdges;        /* n+1 each */
  const float *totalExt;                        /* [nz][ny][nx] */
  const float *cumExt, *ssa;                    /* [ncomp][nz][ny][nx] */
  const int32_t *pfIndex;                       /* [ncomp][nz][ny][nx], 1-based entries */
  /* tables per component */
  const float *const *inverseTables;            /* [ncomp] -> [nEntries][nInvSteps] */
  const int   *nInvSteps;                       /* [ncomp] */
  const float *const *forwardTables;            /* hybrid (or original if hybrid off) [ncomp] -> [nEntries][nFwdSteps] */
  const float *const *forwardOrigTables;
  const int   *nFwdSteps;
  /* surface */
  float surfaceAlbedo;
  int   useSurfaceBDRF;
  int   nxs, nys;                               /* BRDF grid cells */
  const float *xsEdges, *ysEdges;               /* nxs+1, nys+1 */
  const float *brdf;                            /* [nys][nxs] Lambertian albedo */
  /* algorithm switches  (:63-66, :118-129) */
  int   useRayTracing, useRussianRoulette;
  int   nDir;
  const float *dirCos;                          /* [nDir][3] */
  int   useHybrid, numOrdersOrig;
  int   useRRForIntensity;
  float zetaMin;
  int   limitContrib;
  float maxContrib;
} orc_problem;

typedef struct {
  /* raw (un-normalised) tallies, float32 accumulated exactly as the reference does */
  float *fluxUp, *fluxDown, *fluxAbsorbed;      /* [ny][nx] */
  float *volumeAbsorption;                      /* [nz][ny][nx] */
  float *intensity;                             /* [nDir][ny][nx] */
  float *intensityByComp;                       /* [ncomp+1][nDir][ny][nx] */
  float *intensityExcess;                       /* [ncomp+1][nDir] */
  /* instrumentation */
  int64_t nPhotons, nBad, tracerCalls, cellSteps, scatterings, surfaceHits, roulettePlays, exitsTop;
  /* optional per-photon record for replay tests (may be NULL) */
  int64_t *drawStart;     /* [n+1]  RNG ints consumed before photon i started (relative to batch start) */
  int32_t *fate;          /* [n]    0 exit top, 1 absorbed at surface/weight<=tiny, 2 roulette kill, 3 dropped (tracer error) */
  int32_t *fateColumn;    /* [n]    (iy-1)*nx + (ix-1) of the last flux tally, or -1 */
  float   *fateWeight;    /* [n]    weight at the last flux tally */
  int32_t *fateOrder;     /* [n]    scattering order at the end */
} orc_tallies;

/* Trace one batch: photons given as 5 arrays (x,y,z in [0,1], mu, phi) exactly like photonStream
 * (Code/monteCarloIllumination.f95:34-41).  Follows computeRT :400-707.  Returns number of photons processed. */
int64_t orc_compute_rt(const orc_problem *p, orc_mt *rng, int64_t n,
                       const float *xs, const float *ys, const float *zs, const float *mus, const float *phis,
                       orc_tallies *t);

/* The normalisation of computeRadiativeTransfer :327-395 applied in place to raw tallies. */
void orc_normalise(const orc_problem *p, int64_t numPhotonsProcessed, orc_tallies *t);
/* computeSurfaceReflectance, Code/surfaceProperties.f95:121-162, for m points (brdf [nys][nxs], x fastest) */
void orc_surface_reflectance(int nxs, int nys, const float *xsEdges, const float *ysEdges, const float *brdf, int m,
                             const float *x, const float *y, float *out);

/* Single tracer call, exported for bit-exact kernel tests. :1654-1807.  hasTarget=0 -> trace to boundary. */
float orc_trace(const orc_problem *p, const float dir[3], float pos[3], int idx[3], int hasTarget, float target,
                int64_t *cellSteps);

/* Directional photon stream, newPhotonStream_Directional (Code/monteCarloIllumination.f95:62-104). */
void orc_photons_directional(orc_mt *rng, float solarMu, float solarAzimuthDeg, int64_t n,
                             float *xs, float *ys, float *zs, float *mus, float *phis);

/* the other sources of Code/monteCarloIllumination.f95 (illumination.c): RandomAzimuth :106, Flux :148, Spotlight :187,
 * Internal_Flux :228, Internal_Intensity :333.  deltaX / deltaY < 0 stand for an absent optional argument. */
void orc_photons_random_azimuth(orc_mt *rng, float solarMu, int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis);
void orc_photons_flux(orc_mt *rng, int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis);
void orc_photons_spotlight(float solarMu, float solarAzimuthDeg, float solarX, float solarY, int64_t n,
                           float *xs, float *ys, float *zs, float *mus, float *phis);
void orc_photons_internal_flux(orc_mt *rng, float detX, float detY, float detZ, int pointsUp, float deltaX, float deltaY,
                               int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis);
void orc_photons_internal_intensity(orc_mt *rng, float detX, float detY, float detZ, float detMu, float detPhiDeg,
                                    float deltaX, float deltaY, int64_t n, float *xs, float *ys, float *zs, float *mus,
                                    float *phis);

/* Helpers shared with tests */
void orc_regular_flags(const orc_problem *p, int *xyRegular, int *zRegular);     /* :193-211 */

#ifdef __cplusplus
}
#endif
#endif
