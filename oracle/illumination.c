/* CPU oracle -- TEST INFRASTRUCTURE, not product code (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it).  Plain-C float32 restatement of the photon sources of
 * /root/reference/Code/monteCarloIllumination.f95 other than the Directional one (which is in integrator.c):
 * RandomAzimuth :106-146, Flux :148-185, Spotlight :187-226, Internal_Flux :228-331, Internal_Intensity :333-424.
 * Deviates are taken from the reference's MT19937 (numerics.c) in the reference's order; every quirk is kept:
 *   - Internal_Intensity stores detectorPhi as it is given, in DEGREES (:392), while every other source converts to radians;
 *   - the finite detector is not centred: position + delta * (1 - r / 2) (:307, :313, :407, :413);
 *   - a replaced mu (= 0 exactly) of a downward detector comes back POSITIVE (:287 negates before :295-303 replace).
 * Parity unpinned by a reference binary (the reference cannot be built here): pinned against the
 * product's independent Fortran implementation of the same constructors bit for bit (tests/test_fortran_shell.py) and
 * by distribution checks (tests/test_oracle_pins.py). */
#include <math.h>
#include <float.h>
#include "i3rc_oracle.h"

/* acos(-1.) in real(4), as the reference writes pi in this module */
static float pi_f(void) { return acosf(-1.0f); }

/* :106-146 */
void orc_photons_random_azimuth(orc_mt *rng, float solarMu, int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis) {
  for (int64_t i = 0; i < n; i++) {
    xs[i] = orc_mt_real(rng);
    ys[i] = orc_mt_real(rng);
    phis[i] = orc_mt_real(rng) * 2.0f * pi_f();     /* :138 left to right: (r * 2.) * acos(-1.) */
  }
  const float z = 1.0f - orc_spacing(1.0f), mu = -fabsf(solarMu);
  for (int64_t i = 0; i < n; i++) { zs[i] = z; mus[i] = mu; }
}

/* :148-185 */
void orc_photons_flux(orc_mt *rng, int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis) {
  for (int64_t i = 0; i < n; i++) {
    xs[i] = orc_mt_real(rng);
    ys[i] = orc_mt_real(rng);
    mus[i] = -sqrtf(orc_mt_real(rng));
    phis[i] = orc_mt_real(rng) * 2.0f * pi_f();
  }
  const float z = 1.0f - orc_spacing(1.0f);
  for (int64_t i = 0; i < n; i++) zs[i] = z;
}

/* :187-226 (no deviates) */
void orc_photons_spotlight(float solarMu, float solarAzimuthDeg, float solarX, float solarY, int64_t n,
                           float *xs, float *ys, float *zs, float *mus, float *phis) {
  const float z = 1.0f - orc_spacing(1.0f), mu = -fabsf(solarMu), phi = solarAzimuthDeg * pi_f() / 180.0f;
  for (int64_t i = 0; i < n; i++) { xs[i] = solarX; ys[i] = solarY; zs[i] = z; mus[i] = mu; phis[i] = phi; }
}

/* :304-315, :404-415 */
static void finite_detector(orc_mt *rng, float *pos, float delta, int64_t n) {
  for (int64_t i = 0; i < n; i++) pos[i] = pos[i] + delta * (1.0f - 0.5f * orc_mt_real(rng));
}

/* :228-331; deltaX / deltaY < 0: argument absent */
void orc_photons_internal_flux(orc_mt *rng, float detX, float detY, float detZ, int pointsUp, float deltaX, float deltaY,
                               int64_t n, float *xs, float *ys, float *zs, float *mus, float *phis) {
  float z = detZ;
  if (pointsUp) z = fmaxf(z, 2.0f * FLT_MIN); else z = fminf(z, 1.0f - orc_spacing(1.0f));   /* :277-281 */
  for (int64_t i = 0; i < n; i++) { xs[i] = detX; ys[i] = detY; zs[i] = z; }
  for (int64_t i = 0; i < n; i++) {                                                           /* :282-286 */
    mus[i] = sqrtf(orc_mt_real(rng));
    phis[i] = orc_mt_real(rng) * 2.0f * pi_f();
  }
  if (!pointsUp) for (int64_t i = 0; i < n; i++) mus[i] = -mus[i];                            /* :287 */
  /* :294-303: rounds of replacement in index order, until no |mu| < 2 tiny is left (entry test <=, loop test <) */
  int64_t toReplace = 0;
  for (int64_t i = 0; i < n; i++) toReplace += fabsf(mus[i]) <= 2.0f * FLT_MIN;
  while (toReplace > 0) {
    for (int64_t i = 0; i < n; i++) if (fabsf(mus[i]) < 2.0f * FLT_MIN) mus[i] = sqrtf(orc_mt_real(rng));
    toReplace = 0;
    for (int64_t i = 0; i < n; i++) toReplace += fabsf(mus[i]) < 2.0f * FLT_MIN;
  }
  if (deltaX >= 0.0f) finite_detector(rng, xs, deltaX, n);
  if (deltaY >= 0.0f) finite_detector(rng, ys, deltaY, n);
}

/* :333-424; deltaX / deltaY < 0: argument absent */
void orc_photons_internal_intensity(orc_mt *rng, float detX, float detY, float detZ, float detMu, float detPhiDeg,
                                    float deltaX, float deltaY, int64_t n, float *xs, float *ys, float *zs, float *mus,
                                    float *phis) {
  float z = detZ;
  if (detMu > FLT_MIN) z = fmaxf(z, 2.0f * FLT_MIN); else z = fminf(z, 1.0f - orc_spacing(1.0f));   /* :396-400 */
  for (int64_t i = 0; i < n; i++) { xs[i] = detX; ys[i] = detY; zs[i] = z; mus[i] = detMu; phis[i] = detPhiDeg; }
  if (deltaX >= 0.0f) finite_detector(rng, xs, deltaX, n);
  if (deltaY >= 0.0f) finite_detector(rng, ys, deltaY, n);
}
