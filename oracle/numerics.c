/*
 * oracle/numerics.c -- TEST INFRASTRUCTURE (see i3rc_oracle.h).  CPU restatement of the reference's
 * RNG, table search, Lobatto nodes, phase-function evaluation and inverse/forward/hybrid table builders.
 * All arithmetic is float32, evaluated in the reference's operator order, no FMA contraction.
 */
#include "i3rc_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================================================
 * MT19937, Code/RandomNumbersForMC.f95
 * =================================================================================================== */
#define MT_N 624
#define MT_M 397

/* initialize_scalar :169-185 */
void orc_mt_seed_scalar(orc_mt *t, int32_t seed) {
  uint32_t *s = (uint32_t *)t->state;
  s[0] = (uint32_t)seed;
  for (int i = 1; i < MT_N; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
  t->cur = MT_N;
  t->draws = 0;
}

/* initialize_vector :187-239 (init_by_array with the reference's own index bookkeeping) */
void orc_mt_seed_vector(orc_mt *t, const int32_t *seed, int n) {
  uint32_t *s = (uint32_t *)t->state;
  orc_mt_seed_scalar(t, 19650218);
  int nWraps = 0;
  int nFirst = MT_N > n ? MT_N : n;
  for (int k = 1; k <= nFirst; k++) {
    int i = (k + nWraps) % MT_N;
    int j = (k - 1) % n;
    if (i == 0) {
      s[0] = s[MT_N - 1];
      s[1] = (s[1] ^ ((s[0] ^ (s[0] >> 30)) * 1664525u)) + (uint32_t)seed[j] + (uint32_t)j;
      nWraps++;
    } else {
      s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1664525u)) + (uint32_t)seed[j] + (uint32_t)j;
    }
  }
  for (int i = nFirst % MT_N + nWraps + 1; i <= MT_N - 1; i++)
    s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
  s[0] = s[MT_N - 1];
  for (int i = 1; i <= nFirst % MT_N + nWraps; i++)
    s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
  s[0] = 0x80000000u;
  t->cur = MT_N;
  t->draws = 0;
}

/* twist :125-133 */
static inline uint32_t mt_twist(uint32_t u, uint32_t v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

/* nextState :135-153 */
static void mt_next_state(orc_mt *t) {
  uint32_t *s = (uint32_t *)t->state;
  int k;
  for (k = 0; k < MT_N - MT_M; k++) s[k] = s[k + MT_M] ^ mt_twist(s[k], s[k + 1]);
  for (; k < MT_N - 1; k++) s[k] = s[k + MT_M - MT_N] ^ mt_twist(s[k], s[k + 1]);
  s[MT_N - 1] = s[MT_M - 1] ^ mt_twist(s[MT_N - 1], s[0]);
  t->cur = 0;
}

/* getRandomInt :243-258, temper :155-166 */
int32_t orc_mt_int(orc_mt *t) {
  if (t->cur >= MT_N) mt_next_state(t);
  uint32_t y = (uint32_t)t->state[t->cur++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  t->draws++;
  return (int32_t)y;
}

/* getRandomDouble :275-290 : unsigned value / (2^32 - 1) */
double orc_mt_double(orc_mt *t) {
  int32_t v = orc_mt_int(t);
  double d = v < 0 ? (double)v + 4294967296.0 : (double)v;
  return d / 4294967295.0;
}

/* getRandomReal :292-299 */
float orc_mt_real(orc_mt *t) { return (float)orc_mt_double(t); }

/* ===================================================================================================
 * Code/numericUtilities.f95
 * =================================================================================================== */

/* Fortran SPACING(x) for real(4): 2**(exponent(x)-24), TINY for x==0 or when that underflows. */
float orc_spacing(float x) {
  if (x == 0.0f) return FLT_MIN;
  int e;
  (void)frexpf(fabsf(x), &e);
  float r = ldexpf(1.0f, e - 24);
  return r < FLT_MIN ? FLT_MIN : r;
}

/* findIndex :195-248.  table is 1-based in the reference: T(i) == table[i-1]. */
int orc_find_index(float value, const float *table, int n, int firstGuess) {
#define T(i) table[(i)-1]
  int lower, upper;
  if (firstGuess > 0) {
    lower = firstGuess;
    int inc = 1;
    for (;;) {
      upper = lower + inc < n ? lower + inc : n;
      if (lower == n || (T(lower) <= value && T(upper) > value)) break;
      if (T(lower) > value) {
        upper = lower;
        lower = upper - inc > 1 ? upper - inc : 1;
      } else {
        lower = upper;
      }
      inc *= 2;
    }
  } else {
    lower = 0;
    upper = n;
  }
  for (;;) {
    if (lower == n || upper <= lower + 1) break;
    int mid = (lower + upper) / 2;
    if (value >= T(mid)) lower = mid; else upper = mid;
  }
  return lower;
#undef T
}

/* computeLegendrePolynomials :175-193 for one mu: P[0..maxL] */
static void legendre_p(int maxL, float mu, float *P) {
  P[0] = 1.0f;
  if (maxL >= 1) P[1] = mu;
  for (int l = 1; l <= maxL - 1; l++)
    P[l + 1] = (((float)(2 * l + 1) * mu) * P[l] - (float)l * P[l - 1]) / (float)(l + 1);
}

/* computeLobattoTerms :15-102 */
void orc_lobatto(int n, float *mus, float *weights) {
  const float relAcc = 3.0f;
  const int maxIter = 25;
  float pi = acosf(-1.0f);
  int mid = (n + 1) / 2;
  int m = mid - 1; /* number of interior roots searched */
  float *trial = calloc(m > 0 ? m : 1, sizeof(float)), *last = calloc(m > 0 ? m : 1, sizeof(float));
  float *P = calloc((size_t)(m > 0 ? m : 1) * n, sizeof(float)); /* P[j*n + l], l = 0..n-1 */
  float c1 = (n % 2 == 1) ? 1.0f : 0.5f;
  float denom = ((float)n - 1.0f) + 0.5f;
  for (int j = 0; j < m; j++) trial[j] = sinf(pi * ((float)(j + 1) - c1) / denom);

#define NEWTON_UPDATE(j)                                                                                  \
  do {                                                                                                    \
    float mu = trial[j];                                                                                  \
    const float *Pj = P + (size_t)(j)*n;                                                                  \
    float d1 = ((float)(n - 1) * (mu * Pj[n - 1] - Pj[n - 2])) / (mu * mu - 1.0f);                       \
    float d2 = ((2.0f * mu) * d1 - ((float)(n * (n - 1)) * Pj[n - 1])) / (1.0f - mu * mu);              \
    last[j] = mu;                                                                                         \
    trial[j] = mu - d1 / d2;                                                                              \
  } while (0)

  for (int j = 0; j < m; j++) legendre_p(n - 1, trial[j], P + (size_t)j * n);
  for (int j = 0; j < m; j++) NEWTON_UPDATE(j);

  int it = 0;
  for (;;) {
    int done = 1;
    for (int j = 0; j < m; j++)
      if (!(fabsf(trial[j] - last[j]) <= relAcc * orc_spacing(trial[j]))) done = 0;
    if (done) break;
    for (int j = 0; j < m; j++) legendre_p(n - 1, trial[j], P + (size_t)j * n);
    for (int j = 0; j < m; j++)
      if (fabsf(trial[j] - last[j]) > relAcc * orc_spacing(trial[j])) NEWTON_UPDATE(j);
    it++;
    if (it > maxIter) break;
  }
#undef NEWTON_UPDATE

  /* assemble (1-based in the reference): mus(1) = -1; mus(mid:2:-1) = -trial(:) */
  mus[0] = -1.0f;
  weights[0] = 2.0f / (float)(n * (n - 1));
  for (int j = 0; j < m; j++) {
    const float *Pj = P + (size_t)j * n;
    mus[mid - 1 - j] = -trial[j];
    weights[mid - 1 - j] = 2.0f / ((float)(n * (n - 1)) * (Pj[n - 1] * Pj[n - 1]));
  }
  if (n % 2 == 0) {
    for (int k = 0; k < mid; k++) { /* mus(mid+1:n) = -mus(mid:1:-1) */
      mus[mid + k] = -mus[mid - 1 - k];
      weights[mid + k] = weights[mid - 1 - k];
    }
  } else {
    /* mus(mid:n) = -mus(mid:1:-1): right-hand side evaluated before assignment */
    float *tm = malloc(sizeof(float) * mid), *tw = malloc(sizeof(float) * mid);
    for (int k = 0; k < mid; k++) { tm[k] = -mus[mid - 1 - k]; tw[k] = weights[mid - 1 - k]; }
    for (int k = 0; k < mid; k++) { mus[mid - 1 + k] = tm[k]; weights[mid - 1 + k] = tw[k]; }
    free(tm); free(tw);
  }
  free(trial); free(last); free(P);
}

/* computeGaussLegendreTerms :104-173 (no caller in the reference; restated because the module exports it and
 * oracle/_ref pins it) */
void orc_gauss_legendre(int n, float *mus, float *weights) {
  const float relAcc = 2.0f;
  const int maxIter = 25;
  float pi = acosf(-1.0f);
  int mid = (n + 1) / 2;
  float *trial = calloc(mid, sizeof(float)), *last = calloc(mid, sizeof(float)), *deriv = calloc(mid, sizeof(float));
  float *P = calloc((size_t)mid * (n + 1), sizeof(float)); /* P[j*(n+1) + l], l = 0..n */
  for (int j = 0; j < mid; j++) trial[j] = cosf(pi * ((float)(j + 1) - 0.25f) / ((float)n + 0.5f));

#define NEWTON_UPDATE(j)                                                                   \
  do {                                                                                     \
    float mu = trial[j];                                                                   \
    const float *Pj = P + (size_t)(j) * (n + 1);                                           \
    deriv[j] = ((float)n * (mu * Pj[n] - Pj[n - 1])) / (mu * mu - 1.0f);                   \
    last[j] = mu;                                                                          \
    trial[j] = mu - Pj[n] / deriv[j];                                                      \
  } while (0)

  for (int j = 0; j < mid; j++) legendre_p(n, trial[j], P + (size_t)j * (n + 1));
  for (int j = 0; j < mid; j++) NEWTON_UPDATE(j);
  int it = 0;
  for (;;) {
    int done = 1;
    for (int j = 0; j < mid; j++)
      if (!(fabsf(trial[j] - last[j]) <= relAcc * orc_spacing(trial[j]))) done = 0;
    if (done) break;
    for (int j = 0; j < mid; j++) legendre_p(n, trial[j], P + (size_t)j * (n + 1));
    for (int j = 0; j < mid; j++)
      if (fabsf(trial[j] - last[j]) > relAcc * orc_spacing(trial[j])) NEWTON_UPDATE(j);
    it++;
    if (it > maxIter) break;
  }
#undef NEWTON_UPDATE
  for (int j = 0; j < mid; j++) {   /* mus(:mid) = -trial; weights(:mid) = 2 / ((1 - trial**2) * deriv**2) */
    mus[j] = -trial[j];
    weights[j] = 2.0f / ((1.0f - trial[j] * trial[j]) * (deriv[j] * deriv[j]));
  }
  if (n % 2 == 0) {
    for (int k = 0; k < mid; k++) { mus[mid + k] = -mus[mid - 1 - k]; weights[mid + k] = weights[mid - 1 - k]; }
  } else {   /* mus(mid:n) = -mus(mid:1:-1): the right-hand side is evaluated before the assignment */
    float *tm = malloc(sizeof(float) * mid), *tw = malloc(sizeof(float) * mid);
    for (int k = 0; k < mid; k++) { tm[k] = -mus[mid - 1 - k]; tw[k] = weights[mid - 1 - k]; }
    for (int k = 0; k < mid; k++) { mus[mid - 1 + k] = tm[k]; weights[mid - 1 + k] = tw[k]; }
    free(tm); free(tw);
  }
  free(trial); free(last); free(deriv); free(P);
}

/* computeLegendrePolynomials :175-193 for m values of mu: P[j * (maxL + 1) + l] */
void orc_legendre_polynomials(int maxL, const float *mus, int m, float *P) {
  for (int j = 0; j < m; j++) legendre_p(maxL, mus[j], P + (size_t)j * (maxL + 1));
}

/* ===================================================================================================
 * Code/scatteringPhaseFunctions.f95 : getPhaseFunctionValues
 * =================================================================================================== */

/* getPhaseFunctionValues_one, Legendre branch :481-496:
 *   value = matmul( (/1, c(:)/) * (/(2l+1)/), P(0:maxL, :) ), accumulated in ascending l. */
void orc_phase_values_legendre(const float *coef, int nCoef, const float *angles, int nAngles, float *values) {
  if (nCoef == 0) { for (int i = 0; i < nAngles; i++) values[i] = 0.5f; return; }
  float *P = malloc(sizeof(float) * (nCoef + 1));
  float *w = malloc(sizeof(float) * (nCoef + 1));
  w[0] = 1.0f;
  for (int l = 1; l <= nCoef; l++) w[l] = coef[l - 1] * (float)(2 * l + 1);
  for (int i = 0; i < nAngles; i++) {
    legendre_p(nCoef, cosf(angles[i]), P);
    float s = 0.0f;
    for (int l = 0; l <= nCoef; l++) s += w[l] * P[l];
    values[i] = s;
  }
  free(P); free(w);
}

/* getPhaseFunctionValues_one, tabulated branch :497-524 */
void orc_phase_values_tabulated(const float *tabAngles, const float *tabValues, int nTab,
                                const float *angles, int nAngles, float *values) {
  for (int i = 0; i < nAngles; i++) {
    int idx = orc_find_index(angles[i], tabAngles, nTab, 0);
    int idxp = idx + 1;
    float dMu;
    if (idx < nTab) dMu = cosf(tabAngles[idxp - 1]) - cosf(tabAngles[idx - 1]);
    else { dMu = FLT_MAX; idxp = idx; }
    float wgt = 1.0f - (cosf(angles[i]) - cosf(tabAngles[idx - 1])) / dMu;
    values[i] = wgt * tabValues[idx - 1] + (1.0f - wgt) * tabValues[idxp - 1];
  }
}

/* normalizePhaseFunction :1329-1345: scale so that the trapezoid integral over cos(angle) equals 2
 * (applied by the constructors of tabulated phase functions :145, :300-303) */
void orc_normalize_tabulated(const float *tabAngles, const float *tabValues, int nTab, float *normalized) {
  float dot = 0.0f;
  for (int i = 0; i + 1 < nTab; i++)
    dot += (cosf(tabAngles[i + 1]) - cosf(tabAngles[i])) * (0.5f * (tabValues[i + 1] + tabValues[i]));
  for (int i = 0; i < nTab; i++) normalized[i] = (-tabValues[i] * 2.0f) / dot;
}

/* ===================================================================================================
 * Code/inversePhaseFunctions.f95 : computeInversePhaseFunction :68-176
 * =================================================================================================== */
static void inverse_from_cdf_inputs(int nAngles, const float *mus, const float *values, int nSteps, float *table) {
  float *cdf = malloc(sizeof(float) * nAngles);
  int *ind = malloc(sizeof(int) * nSteps);
  cdf[0] = 0.0f;
  for (int i = 1; i < nAngles; i++) /* :122-124 */
    cdf[i] = cdf[i - 1] + ((mus[i] - mus[i - 1]) * 0.5f) * (values[i] + values[i - 1]);
  float total = cdf[nAngles - 1];
  for (int i = 0; i < nAngles; i++) cdf[i] = cdf[i] / total; /* :128 */

  ind[0] = orc_find_index(0.0f, cdf, nAngles, 0); /* :132 */
  for (int i = 2; i <= nSteps; i++) {
    float p = (float)(i - 1) / (float)(nSteps - 1);
    ind[i - 1] = orc_find_index(p, cdf, nAngles, ind[i - 2]);
  }
#define CDF(k) cdf[(k)-1]
#define MU(k) mus[(k)-1]
#define V(k) values[(k)-1]
  for (int i = 1; i <= nSteps - 1; i++) { /* :138-169 */
    float p = (float)(i - 1) / (float)(nSteps - 1);
    int k = ind[i - 1];
    if (CDF(k + 1) - CDF(k) <= orc_spacing(CDF(k))) {
      table[i - 1] = acosf(MU(k));
    } else if (fabsf(V(k) - V(k + 1)) <= orc_spacing(V(k))) {
      table[i - 1] = acosf(MU(k) + (MU(k + 1) - MU(k)) * (p - CDF(k)) / (CDF(k + 1) - CDF(k)));
    } else {
      float rad = ((CDF(k + 1) - p) * (V(k) * V(k)) + (p - CDF(k)) * (V(k + 1) * V(k + 1))) / (CDF(k + 1) - CDF(k));
      table[i - 1] = acosf(MU(k) + (MU(k + 1) - MU(k)) / (V(k) - V(k + 1)) * (V(k) - sqrtf(rad)));
    }
  }
#undef CDF
#undef MU
#undef V
  table[nSteps - 1] = 0.0f; /* :170 */
  free(cdf); free(ind);
}

/* Legendre branch :101-115: values on max(nMoments,2) Lobatto nodes */
void orc_inverse_table_legendre(const float *coef, int nCoef, int nSteps, float *table) {
  int nAngles = nCoef > 2 ? nCoef : 2;
  float *mus = malloc(sizeof(float) * nAngles), *wts = malloc(sizeof(float) * nAngles);
  float *ang = malloc(sizeof(float) * nAngles), *val = malloc(sizeof(float) * nAngles), *tmp = malloc(sizeof(float) * nAngles);
  orc_lobatto(nAngles, mus, wts);
  for (int i = 0; i < nAngles; i++) ang[i] = acosf(mus[nAngles - 1 - i]);
  orc_phase_values_legendre(coef, nCoef, ang, nAngles, tmp);
  for (int i = 0; i < nAngles; i++) val[i] = tmp[nAngles - 1 - i];
  inverse_from_cdf_inputs(nAngles, mus, val, nSteps, table);
  free(mus); free(wts); free(ang); free(val); free(tmp);
}

/* Angle-value branch :90-100: native angles */
void orc_inverse_table_tabulated(const float *tabAngles, const float *tabValues, int nTab, int nSteps, float *table) {
  float *mus = malloc(sizeof(float) * nTab), *val = malloc(sizeof(float) * nTab), *tmp = malloc(sizeof(float) * nTab);
  orc_phase_values_tabulated(tabAngles, tabValues, nTab, tabAngles, nTab, tmp);
  for (int i = 0; i < nTab; i++) { mus[i] = cosf(tabAngles[nTab - 1 - i]); val[i] = tmp[nTab - 1 - i]; }
  inverse_from_cdf_inputs(nTab, mus, val, nSteps, table);
  free(mus); free(val); free(tmp);
}

/* ===================================================================================================
 * Forward tables: monteCarloRadiativeTransfer.f95:1896-1901 -> getPhaseFunctionValues_table :531-648
 * =================================================================================================== */
static const float kPiIntegrator = 3.14159265358979312f; /* monteCarloRadiativeTransfer.f95:43 */

static void forward_angles(int nSteps, float *angles) {
  for (int j = 0; j < nSteps; j++) angles[j] = ((float)j / (float)(nSteps - 1)) * kPiIntegrator; /* :1900 */
}

/* Legendre entries of a table :575-580,:621-626: P scaled by (2l+1) first, then sum c_l * P'_l */
void orc_forward_table_legendre(const float *coef, int nCoef, int nSteps, float *table) {
  float *angles = malloc(sizeof(float) * nSteps);
  forward_angles(nSteps, angles);
  if (nCoef == 0) { for (int i = 0; i < nSteps; i++) table[i] = 0.5f; free(angles); return; }
  float *P = malloc(sizeof(float) * (nCoef + 1));
  for (int i = 0; i < nSteps; i++) {
    legendre_p(nCoef, cosf(angles[i]), P);
    float s = 0.0f;
    for (int l = 0; l <= nCoef; l++) {
      float c = l == 0 ? 1.0f : coef[l - 1];
      s += c * ((float)(2 * l + 1) * P[l]);
    }
    table[i] = s;
  }
  free(P); free(angles);
}

/* One-angle-set tabulated entries :581-609,:627-632 */
void orc_forward_table_tabulated(const float *tabAngles, const float *tabValues, int nTab, int nSteps, float *table) {
  float *angles = malloc(sizeof(float) * nSteps);
  forward_angles(nSteps, angles);
  int prev = 0;
  for (int i = 0; i < nSteps; i++) {
    int idx = orc_find_index(angles[i], tabAngles, nTab, prev);
    prev = idx;
    int idxp = idx + 1;
    float dMu;
    if (idx < nTab) dMu = cosf(tabAngles[idxp - 1]) - cosf(tabAngles[idx - 1]);
    else { dMu = FLT_MAX; idxp = idx; }
    float wgt = 1.0f - (cosf(angles[i]) - cosf(tabAngles[idx - 1])) / dMu;
    table[i] = wgt * tabValues[idx - 1] + (1.0f - wgt) * tabValues[idxp - 1];
  }
  free(angles);
}

/* ===================================================================================================
 * Hybrid phase functions: monteCarloRadiativeTransfer.f95:1925-2039
 * =================================================================================================== */
/* computeNormalization :2000-2023 (1-based transitionIndex) */
static float hybrid_norm(int nA, const float *cosA, const float *val, const float *gau, int ti) {
  float ig = 0.0f, io = 0.0f;
  for (int k = 1; k <= ti - 1; k++) ig += (0.5f * (gau[k - 1] + gau[k])) * (cosA[k - 1] - cosA[k]);
  for (int k = ti; k <= nA - 1; k++) io += (0.5f * (val[k - 1] + val[k])) * (cosA[k - 1] - cosA[k]);
  if (io >= 2.0f) return 1.0f / ig;
  return (2.0f - io) / ig;
}
/* phaseFuncDiff :2025-2039 */
static float hybrid_diff(int nA, const float *cosA, const float *val, const float *gau, int ti) {
  float p0 = hybrid_norm(nA, cosA, val, gau, ti);
  return p0 * gau[ti - 1] - val[ti - 1];
}

void orc_hybrid_phase_functions(int nSteps, int nEntries, const float *values, float widthDeg, float *newValues) {
  int nA = nSteps;
  float *angles = malloc(sizeof(float) * nA), *cosA = malloc(sizeof(float) * nA), *gau = malloc(sizeof(float) * nA);
  forward_angles(nA, angles);
  float w = widthDeg * kPiIntegrator / 180.0f;
  for (int i = 0; i < nA; i++) {
    cosA[i] = cosf(angles[i]);
    float q = angles[i] / w;
    gau[i] = expf(-(q * q)); /* :1947 */
  }
  memcpy(newValues, values, sizeof(float) * (size_t)nA * nEntries);
  for (int e = 0; e < nEntries; e++) {
    const float *val = values + (size_t)e * nA;
    float *out = newValues + (size_t)e * nA;
    int lower = orc_find_index(w, angles, nA, 0) + 1; /* :1954 */
    if (lower >= nA - 2) break;                        /* exit entryLoop */
    float lowDiff = hybrid_diff(nA, cosA, val, gau, lower);
    int inc = 1, upper;
    float upDiff;
    int noRoot = 0;
    for (;;) { /* huntingLoop :1965-1975 */
      upper = lower + inc < nA - 1 ? lower + inc : nA - 1;
      upDiff = hybrid_diff(nA, cosA, val, gau, upper);
      if (lower == nA - 1) { noRoot = 1; break; }
      if (lowDiff * upDiff < 0.0f) break;
      lower = upper;
      lowDiff = upDiff;
      inc *= 2;
    }
    if (noRoot) continue;
    for (;;) { /* bisectionLoop :1979-1990 */
      if (upper <= lower + 1) break;
      int midp = (lower + upper) / 2;
      float midDiff = hybrid_diff(nA, cosA, val, gau, midp);
      if (midDiff * upDiff < 0.0f) { lower = midp; lowDiff = midDiff; }
      else { upper = midp; upDiff = midDiff; }
    }
    int ti = lower;
    float p0 = hybrid_norm(nA, cosA, val, gau, ti);
    for (int k = 1; k <= ti; k++) out[k - 1] = p0 * gau[k - 1];
    for (int k = ti + 1; k <= nA; k++) out[k - 1] = val[k - 1];
  }
  free(angles); free(cosA); free(gau);
}
