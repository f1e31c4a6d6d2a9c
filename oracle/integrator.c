/*
 * oracle/integrator.c -- TEST INFRASTRUCTURE (see i3rc_oracle.h).  CPU restatement of
 * Integrators/monteCarloRadiativeTransfer.f95 computeRT and everything it calls in the photon loop.
 * float32, reference operator order, reference RNG draw order, 1-based cell indices.
 */
#include "i3rc_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float kPi = 3.14159265358979312f; /* :43 */

#define XE(i) p->xEdges[(i)-1]
#define YE(i) p->yEdges[(i)-1]
#define ZE(i) p->zEdges[(i)-1]
#define CELL(ix, iy, iz) ((size_t)((iz)-1) * p->ny * p->nx + (size_t)((iy)-1) * p->nx + ((ix)-1))
#define CELLC(ix, iy, iz, c) ((size_t)((c)-1) * p->nz * p->ny * p->nx + CELL(ix, iy, iz))

/* new_Integrator :193-211 */
void orc_regular_flags(const orc_problem *p, int *xyRegular, int *zRegular) {
  float dx = XE(2) - XE(1), dy = YE(2) - YE(1), dz = ZE(2) - ZE(1);
  int xy = 1, z = 1;
  for (int i = 1; i <= p->nx; i++)
    if (!(fabsf((XE(i + 1) - XE(i)) - dx) <= 2.0f * orc_spacing(XE(i + 1)))) xy = 0;
  for (int i = 1; i <= p->ny; i++)
    if (!(fabsf((YE(i + 1) - YE(i)) - dy) <= 2.0f * orc_spacing(YE(i + 1)))) xy = 0;
  for (int i = 1; i <= p->nz; i++)
    if (!(fabsf((ZE(i + 1) - ZE(i)) - dz) <= orc_spacing(ZE(i + 1)))) z = 0;
  *xyRegular = xy;
  *zRegular = z;
}

typedef struct {
  const orc_problem *p;
  int xyRegular, zRegular;
  float deltaX, deltaY, deltaZ, x0, y0, z0;
} ctx_t;

/* makeDirectionCosines :2041-2059 */
static void make_dircos(float mu, float phi, float s[3]) {
  float sinTheta = sqrtf(1.0f - mu * mu);
  float cosPhi = cosf(phi), sinPhi = sinf(phi);
  s[0] = sinTheta * cosPhi;
  s[1] = sinTheta * sinPhi;
  s[2] = mu;
}

/* makePeriodic :2063-2082 */
static float make_periodic(float a, float aMin, float aMax) {
  for (;;) {
    if (a <= aMax && a > aMin) break;
    float b;
    if (a > aMax) b = a - (aMax - aMin);
    else if (a == aMin) b = aMax;
    else b = a + (aMax - aMin);
    /* no progress: a is NaN, infinite or more than 2^24 widths away (a width no longer changes a float32), where the
       reference's loop never ends (a max-cross-section step-back along a direction that is all but horizontal).
       Such a position carries no information: the boundary value is as good as any (the kernels do the same). */
    if (b == a || b != b) return aMax;
    a = b;
  }
  return a;
}

/* findXYIndicies :1353-1374 */
static void find_xy(const ctx_t *c, float xPos, float yPos, int *ix, int *iy) {
  const orc_problem *p = c->p;
  if (c->xyRegular) {
    int i = (int)((xPos - c->x0) / c->deltaX) + 1; if (i > p->nx) i = p->nx;
    int j = (int)((yPos - c->y0) / c->deltaY) + 1; if (j > p->ny) j = p->ny;
    if (fabsf(XE(i + 1) - xPos) < orc_spacing(xPos)) i = i + 1;
    if (fabsf(YE(j + 1) - yPos) < orc_spacing(yPos)) j = j + 1;
    if (i == p->nx + 1) i = 1;
    if (j == p->ny + 1) j = 1;
    *ix = i; *iy = j;
  } else {
    *ix = orc_find_index(xPos, p->xEdges, p->nx + 1, *ix);
    *iy = orc_find_index(yPos, p->yEdges, p->ny + 1, *iy);
    /* xPos == xMax exactly gives nx + 1: the regular branch wraps it (:1366-1367), the reference's irregular branch
       does not and indexes out of bounds afterwards; this restatement wraps in both (as the kernels do) */
    if (*ix == p->nx + 1) *ix = 1;
    if (*iy == p->ny + 1) *iy = 1;
  }
}

/* findZIndex :1376-1388 */
static void find_z(const ctx_t *c, float zPos, int *iz) {
  const orc_problem *p = c->p;
  if (c->zRegular) {
    int k = (int)((zPos - c->z0) / c->deltaZ) + 1; if (k > p->nz) k = p->nz;
    if (fabsf(ZE(k + 1) - zPos) < orc_spacing(zPos)) k = k + 1;
    *iz = k;
  } else {
    *iz = orc_find_index(zPos, p->zEdges, p->nz + 1, *iz);
  }
}

/* accumulateExtinctionAlongPath :1654-1807 */
float orc_trace(const orc_problem *p, const float dir[3], float pos[3], int idx[3], int hasTarget, float target,
                int64_t *cellSteps) {
  float acc = 0.0f;
  int nx = p->nx, ny = p->ny, nz = p->nz;
  int side[3], inc[3];
  for (int a = 0; a < 3; a++) { side[a] = dir[a] >= 0.0f ? 1 : 0; inc[a] = dir[a] >= 0.0f ? 1 : -1; }
  float z0 = ZE(1), zMax = ZE(nz + 1);
  float xPos = pos[0], yPos = pos[1], zPos = pos[2];
  int ix = idx[0], iy = idx[1], iz = idx[2];
  for (;;) {
    if (cellSteps) (*cellSteps)++;
    float step[3];
    step[0] = fabsf(dir[0]) >= 2.0f * FLT_MIN ? (XE(ix + side[0]) - xPos) / dir[0] : FLT_MAX;
    step[1] = fabsf(dir[1]) >= 2.0f * FLT_MIN ? (YE(iy + side[1]) - yPos) / dir[1] : FLT_MAX;
    step[2] = fabsf(dir[2]) >= 2.0f * FLT_MIN ? (ZE(iz + side[2]) - zPos) / dir[2] : FLT_MAX;
    float thisStep = step[0];
    if (step[1] < thisStep) thisStep = step[1];
    if (step[2] < thisStep) thisStep = step[2];
    if (!(thisStep > 0.0f)) { acc = -2.0f; break; }   /* :1711-1714 `thisStep <= 0.`; a NaN step ends the trace too (the reference loops forever) */

    float ext = p->totalExt[CELL(ix, iy, iz)];
    if (hasTarget) {
      if (acc + thisStep * ext > target) {
        thisStep = (target - acc) / ext;
        xPos = xPos + thisStep * dir[0];
        yPos = yPos + thisStep * dir[1];
        zPos = zPos + thisStep * dir[2];
        acc = target;
        break;
      }
    }
    acc = acc + thisStep * ext;

    if (step[0] <= thisStep) { xPos = XE(ix + side[0]); ix = ix + inc[0]; }
    else {
      xPos = xPos + thisStep * dir[0];
      if (fabsf(XE(ix + side[0]) - xPos) <= 2.0f * orc_spacing(xPos)) ix = ix + inc[0];
    }
    if (step[1] <= thisStep) { yPos = YE(iy + side[1]); iy = iy + inc[1]; }
    else {
      yPos = yPos + thisStep * dir[1];
      if (fabsf(YE(iy + side[1]) - yPos) <= 2.0f * orc_spacing(yPos)) iy = iy + inc[1];
    }
    if (step[2] <= thisStep) { zPos = ZE(iz + side[2]); iz = iz + inc[2]; }
    else {
      zPos = zPos + thisStep * dir[2];
      if (fabsf(ZE(iz + side[2]) - zPos) <= 2.0f * orc_spacing(zPos)) iz = iz + inc[2];
    }

    /* periodicity :1774-1788 -- note y uses inc[0] (x's sign) as the reference does */
    if (ix <= 0) { ix = nx; xPos = XE(ix + 1) + (float)(inc[0] * 2) * orc_spacing(xPos); }
    else if (ix >= nx + 1) { ix = 1; xPos = XE(ix) + (float)(inc[0] * 2) * orc_spacing(xPos); }
    if (iy <= 0) { iy = ny; yPos = YE(iy + 1) + (float)(inc[0] * 2) * orc_spacing(yPos); }
    else if (iy >= ny + 1) { iy = 1; yPos = YE(iy) + (float)(inc[0] * 2) * orc_spacing(yPos); }

    if (iz > nz) { zPos = zMax + 2.0f * orc_spacing(zMax); break; }
    if (iz < 1) { zPos = z0; break; }
  }
  pos[0] = xPos; pos[1] = yPos; pos[2] = zPos;
  idx[0] = ix; idx[1] = iy; idx[2] = iz;
  return acc;
}

/* computeScatteringAngle :1390-1417 */
static float scattering_angle(float r, const float *tab, int n) {
  int k = (int)(r * (float)n) + 1;
  if (k < n) {
    float left = r - (float)(k - 1) / (float)n;
    return (1.0f - left) * tab[k - 1] + left * tab[k];
  }
  return tab[n - 1];
}

/* next_direct :2086-2113 */
static void next_direct(orc_mt *rng, float cosS, float s[3]) {
  float d = 2.0f, ax = 0.0f, ay = 0.0f, b;
  while (d > 1.0f) {
    ax = 1.0f - 2.0f * orc_mt_real(rng);
    ay = 1.0f - 2.0f * orc_mt_real(rng);
    d = ax * ax + ay * ay;
  }
  b = sqrtf((1.0f - cosS * cosS) / d);
  ax = ax * b;
  ay = ay * b;
  b = s[0] * ax - s[1] * ay;
  d = cosS - b / (1.0f + fabsf(s[2]));
  s[0] = s[0] * d + ax;
  s[1] = s[1] * d - ay;
  s[2] = s[2] * cosS - copysignf(fabsf(b), s[2] * b);
}

/* lookUpPhaseFuncValsFromTable :1613-1652 for one angle */
static float lookup_phase(const float *tab, int n, float angle) {
  float dTheta = kPi / (float)(n - 1);
  int k = (int)(angle / dTheta) + 1;
  if (k < n) {
    float w = 1.0f - (angle - (float)(k - 1) * dTheta) / dTheta;
    return w * tab[k - 1] + (1.0f - w) * tab[k];
  }
  return tab[n - 1];
}

/* computeSurfaceReflectance, Code/surfaceProperties.f95:121-148 (Lambertian R :154-162) */
static float surface_reflectance(const orc_problem *p, float xPos, float yPos) {
  float x0 = p->xsEdges[0], xMax = p->xsEdges[p->nxs], y0 = p->ysEdges[0], yMax = p->ysEdges[p->nys];
  int ix = orc_find_index(make_periodic(xPos, x0, xMax), p->xsEdges, p->nxs + 1, 0);
  int iy = orc_find_index(make_periodic(yPos, y0, yMax), p->ysEdges, p->nys + 1, 0);
  return p->brdf[(size_t)(iy - 1) * p->nxs + (ix - 1)];
}

/* ... for m points of a surface given by its arrays: what the loop's own look-up (above) answers, callable from the tests
 * (tests/test_ref_numerics.py holds it against the reference's computeSurfaceReflectance, oracle/_ref) */
void orc_surface_reflectance(int nxs, int nys, const float *xsEdges, const float *ysEdges, const float *brdf, int m,
                             const float *x, const float *y, float *out) {
  orc_problem p;
  memset(&p, 0, sizeof(p));
  p.nxs = nxs; p.nys = nys; p.xsEdges = xsEdges; p.ysEdges = ysEdges; p.brdf = brdf;
  for (int i = 0; i < m; i++) out[i] = surface_reflectance(&p, x[i], y[i]);
}

/* computeIntensityContribution :1419-1611 */
static void intensity_contribution(const ctx_t *c, orc_tallies *t, float weight, const float posI[3], const int idxI[3],
                                   const float dirCos[3], int component, orc_mt *rng, int order,
                                   float *contrib, int *ixF, int *iyF) {
  const orc_problem *p = c->p;
  int nDir = p->nDir;
  int zIndexMax = p->nz + 1;
  for (int d = 0; d < nDir; d++) {
    const float *dI = p->dirCos + 3 * d;
    float normPF;
    if (component < 1) {
      normPF = 1.0f / kPi;
    } else {
      /* matmul(directionCosines, intensityDirections(:, d)) :1487 */
      float proj = 0.0f;
      proj += dirCos[0] * dI[0];
      proj += dirCos[1] * dI[1];
      proj += dirCos[2] * dI[2];
      if (fabsf(proj) > 1.0f) proj = copysignf(1.0f, proj);
      float ang = acosf(proj);
      int pfi = p->pfIndex[CELLC(idxI[0], idxI[1], idxI[2], component)];
      int n = p->nFwdSteps[component - 1];
      const float *tab = (p->useHybrid && order <= p->numOrdersOrig)
                             ? p->forwardOrigTables[component - 1] + (size_t)(pfi - 1) * n
                             : p->forwardTables[component - 1] + (size_t)(pfi - 1) * n;
      float pv = lookup_phase(tab, n, ang);
      normPF = pv / ((4.0f * kPi) * fabsf(dI[2]));
    }
    float pos[3] = {posI[0], posI[1], posI[2]};
    int idx[3] = {idxI[0], idxI[1], idxI[2]};
    float tauB, con;
    if (!p->useRRForIntensity) {
      t->tracerCalls++;
      tauB = orc_trace(p, dI, pos, idx, 0, 0.0f, &t->cellSteps);
      con = tauB >= 0.0f ? (weight * normPF) * expf(-tauB) : 0.0f;
    } else {
      float r = orc_mt_real(rng);
      float tauFree = -logf(r > FLT_MIN ? r : FLT_MIN);
      if (kPi * normPF <= p->zetaMin) {
        t->tracerCalls++;
        tauB = orc_trace(p, dI, pos, idx, 1, tauFree, &t->cellSteps);
        float r2 = orc_mt_real(rng);
        if (r2 <= kPi * normPF / p->zetaMin && idx[2] >= zIndexMax) con = weight * p->zetaMin / kPi;
        else con = 0.0f;
      } else {
        float q = kPi * normPF; if (!(q > FLT_MIN)) q = FLT_MIN;
        float tauMax = -logf(p->zetaMin / q);
        t->tracerCalls++;
        tauB = orc_trace(p, dI, pos, idx, 1, tauMax, &t->cellSteps);
        if (idx[2] >= zIndexMax && tauB >= 0.0f) {
          con = (weight * normPF) * expf(-tauB);
        } else if (tauB >= 0.0f && idx[2] >= 1) {
          t->tracerCalls++;
          tauB = orc_trace(p, dI, pos, idx, 1, tauFree, &t->cellSteps);
          con = idx[2] >= zIndexMax ? weight * p->zetaMin / kPi : 0.0f;
        } else {
          /* Tracer error, or the first leg left through the BOTTOM (idx[2] == 0, a radiance direction pointing down).
             The reference starts its second leg from there all the same (:1576-1587) -- with zIndex 0 it indexes
             zPosition(0) and totalExt(:, :, 0), out of bounds -- and then discards the outcome, because only an exit
             through the top counts (:1583, quirk Q8).  The contribution is 0 either way; this restatement does not
             reproduce the out-of-bounds reads. */
          con = 0.0f;
        }
      }
    }
    if (p->limitContrib && con > p->maxContrib) { /* :1598-1609 */
      t->intensityExcess[(size_t)component * nDir + d] += con - p->maxContrib;
      con = p->maxContrib;
    }
    contrib[d] = con;
    ixF[d] = idx[0];
    iyF[d] = idx[1];
  }
}

static void add_intensity(const orc_problem *p, orc_tallies *t, int component, const float *contrib,
                          const int *ixF, const int *iyF) {
  size_t ncol = (size_t)p->nx * p->ny;
  for (int d = 0; d < p->nDir; d++) {
    size_t col = (size_t)(iyF[d] - 1) * p->nx + (ixF[d] - 1);
    t->intensity[d * ncol + col] += contrib[d];
    t->intensityByComp[((size_t)component * p->nDir + d) * ncol + col] += contrib[d];
  }
}

/* computeRT :400-707 */
int64_t orc_compute_rt(const orc_problem *p, orc_mt *rng, int64_t n,
                       const float *xs, const float *ys, const float *zs, const float *mus, const float *phis,
                       orc_tallies *t) {
  ctx_t cx; cx.p = p;
  orc_regular_flags(p, &cx.xyRegular, &cx.zRegular);
  cx.x0 = XE(1); cx.y0 = YE(1); cx.z0 = ZE(1);
  cx.deltaX = XE(2) - XE(1); cx.deltaY = YE(2) - YE(1); cx.deltaZ = ZE(2) - ZE(1);
  const ctx_t *c = &cx;
  int useRay = p->useRayTracing;
  float maxExt = 0.0f;
  if (!useRay) {
    size_t ncell = (size_t)p->nx * p->ny * p->nz;
    maxExt = p->totalExt[0];
    for (size_t i = 1; i < ncell; i++) if (p->totalExt[i] > maxExt) maxExt = p->totalExt[i];
  }
  /* no extinction anywhere: tau / maxExt is an infinite step and the reference's makePeriodic never returns (:494-496,
     :2063-2082) -- nor does it once the step exceeds 2^24 domain widths (subtracting a width no longer changes a
     float32).  In such an optically empty domain (width * maxExt <= 1e-5) the photon flies straight to the boundary,
     which is what ray tracing gives (the kernels do the same). */
  {
    float wx = XE(p->nx + 1) - XE(1), wy = YE(p->ny + 1) - YE(1);
    if (!useRay && !(maxExt * (wx < wy ? wx : wy) > 1e-5f)) useRay = 1;
  }
  float x0 = cx.x0, xMax = XE(p->nx + 1), y0 = cx.y0, yMax = YE(p->ny + 1), z0 = cx.z0, zMax = ZE(p->nz + 1);
  float *contrib = NULL; int *ixF = NULL, *iyF = NULL;
  float *cum = malloc(sizeof(float) * ((size_t)p->ncomp + 1));   /* (/0, cumulativeExt(ix,iy,iz,:)/) of the event's cell, :637 */
  if (p->nDir > 0) {
    contrib = malloc(sizeof(float) * p->nDir); ixF = malloc(sizeof(int) * p->nDir); iyF = malloc(sizeof(int) * p->nDir);
  }
  int64_t draws0 = rng->draws;
  int64_t nPhotons = 0;

  for (int64_t ip = 0; ip < n; ip++) {
    if (t->drawStart) t->drawStart[ip] = rng->draws - draws0;
    float xPos = xs[ip], yPos = ys[ip], zPos = zs[ip];
    float dir[3];
    int order = 0;
    make_dircos(mus[ip], phis[ip], dir);
    float weight = 1.0f;
    nPhotons++;
    xPos = x0 + xPos * (xMax - x0);
    yPos = y0 + yPos * (yMax - y0);
    zPos = z0 + zPos * (zMax - z0);
    int ix = 1, iy = 1, iz = 1;
    find_xy(c, xPos, yPos, &ix, &iy);
    find_z(c, zPos, &iz);
    int fate = -1, fateCol = -1; float fateW = 0.0f;

    for (;;) { /* scatteringLoop :474-690 */
      float r = orc_mt_real(rng);
      float tau = -logf(r > FLT_MIN ? r : FLT_MIN);
      if (useRay) {
        float pos[3] = {xPos, yPos, zPos}; int idx[3] = {ix, iy, iz};
        t->tracerCalls++;
        float acc = orc_trace(p, dir, pos, idx, 1, tau, &t->cellSteps);
        xPos = pos[0]; yPos = pos[1]; zPos = pos[2]; ix = idx[0]; iy = idx[1]; iz = idx[2];
        if (acc < 0.0f) { t->nBad++; fate = 3; break; }
      } else {
        xPos = make_periodic(xPos + dir[0] * tau / maxExt, x0, xMax);
        yPos = make_periodic(yPos + dir[1] * tau / maxExt, y0, yMax);
        zPos = zPos + dir[2] * tau / maxExt;
      }

      if (zPos >= zMax) { /* :499-514 */
        if (!useRay) {
          xPos = make_periodic(xPos - dir[0] * fabsf((zPos - zMax) / dir[2]), x0, xMax);
          yPos = make_periodic(yPos - dir[1] * fabsf((zPos - zMax) / dir[2]), y0, yMax);
          find_xy(c, xPos, yPos, &ix, &iy);
        }
        size_t col = (size_t)(iy - 1) * p->nx + (ix - 1);
        t->fluxUp[col] += weight;
        t->exitsTop++;
        fate = 0; fateCol = (int)col; fateW = weight;
        break;
      } else if (zPos <= z0 + orc_spacing(z0)) { /* :515-580 */
        order++;
        if (!useRay) {
          xPos = make_periodic(xPos - dir[0] * fabsf((zPos - z0) / dir[2]), x0, xMax);
          yPos = make_periodic(yPos - dir[1] * fabsf((zPos - z0) / dir[2]), y0, yMax);
          find_xy(c, xPos, yPos, &ix, &iy);
        }
        iz = 1;
        zPos = z0 + orc_spacing(z0);
        size_t col = (size_t)(iy - 1) * p->nx + (ix - 1);
        t->fluxDown[col] += weight;
        t->surfaceHits++;
        fateCol = (int)col; fateW = weight;
        float mu, phi;
        do { mu = sqrtf(orc_mt_real(rng)); } while (!(fabsf(mu) > 2.0f * FLT_MIN));
        phi = (2.0f * kPi) * orc_mt_real(rng);
        if (p->useSurfaceBDRF) weight = weight * surface_reflectance(p, xPos, yPos);
        else weight = weight * p->surfaceAlbedo;
        if (weight <= FLT_MIN) { fate = 1; break; }
        make_dircos(mu, phi, dir);
        if (p->nDir > 0) {
          float posI[3] = {xPos, yPos, zPos}; int idxI[3] = {ix, iy, iz};
          intensity_contribution(c, t, weight, posI, idxI, dir, 0, rng, order, contrib, ixF, iyF);
          add_intensity(p, t, 0, contrib, ixF, iyF);
        }
      } else { /* scattering event :581-689 */
        int scatterThis = 1;
        if (!useRay) {
          /* the indices are stale in this mode (no index search after the move, :494-496): those of the start or of
             the last surface hit.  A start index of nz + 1 (photon starting within spacing() of the top) is outside
             the grid and the reference reads totalExt out of bounds; here: no extinction outside the grid. */
          float extHere = (iz >= 1 && iz <= p->nz) ? p->totalExt[CELL(ix, iy, iz)] : 0.0f;
          scatterThis = orc_mt_real(rng) < extHere / maxExt;
        }
        if (useRay || scatterThis) {
          order++;
          t->scatterings++;
          if (p->totalExt[CELL(ix, iy, iz)] <= 0.0f) { /* :606-632, quirks Q2 kept */
            if (xPos - XE(ix) <= 0.0f && dir[0] > 0.0f) {
              xPos = xPos - orc_spacing(xPos);
              ix = ix - 1;
              if (ix <= 0) { ix = p->nx; xPos = XE(ix); xPos = xPos - 2.0f * orc_spacing(xPos); }
            }
            if (yPos - YE(iy) <= 0.0f && dir[1] > 0.0f) {
              yPos = yPos - orc_spacing(yPos);
              iy = iy - 1;
              if (iy <= 0) { iy = p->ny; yPos = XE(iy); yPos = xPos - 2.0f * orc_spacing(yPos); }
            }
            if (zPos - ZE(iz) <= 0.0f && dir[2] > 0.0f) { zPos = zPos - orc_spacing(zPos); iz = iz - 1; }
          }
          /* component :637-638: findIndex(r, (/0, cumExt(ix,iy,iz,:)/)) */
          cum[0] = 0.0f;
          for (int k = 1; k <= p->ncomp; k++) cum[k] = p->cumExt[CELLC(ix, iy, iz, k)];
          int comp = orc_find_index(orc_mt_real(rng), cum, p->ncomp + 1, 0);
          if (comp < 1 || comp > p->ncomp) { /* out-of-bounds in the reference (undefined); drop */
            t->nBad++; fate = 3; break;
          }
          float ssa = p->ssa[CELLC(ix, iy, iz, comp)];
          if (ssa < 1.0f) {
            size_t col = (size_t)(iy - 1) * p->nx + (ix - 1);
            t->fluxAbsorbed[col] += weight * (1.0f - ssa);
            t->volumeAbsorption[CELL(ix, iy, iz)] += weight * (1.0f - ssa);
            weight = weight * ssa;
          }
          if (p->nDir > 0) {
            float posI[3] = {xPos, yPos, zPos}; int idxI[3] = {ix, iy, iz};
            intensity_contribution(c, t, weight, posI, idxI, dir, comp, rng, order, contrib, ixF, iyF);
            add_intensity(p, t, comp, contrib, ixF, iyF);
          }
          if (p->useRussianRoulette && weight < 1.0f / 2.0f) { /* RussianRouletteW = 1 :66 */
            t->roulettePlays++;
            if (orc_mt_real(rng) >= weight / 1.0f) weight = 0.0f; else weight = 1.0f;
          }
          if (weight <= FLT_MIN) { fate = 2; break; }
          int pfi = p->pfIndex[CELLC(ix, iy, iz, comp)];
          int nInv = p->nInvSteps[comp - 1];
          float theta = scattering_angle(orc_mt_real(rng), p->inverseTables[comp - 1] + (size_t)(pfi - 1) * nInv, nInv);
          next_direct(rng, cosf(theta), dir);
        }
      }
    }
    if (t->fate) { t->fate[ip] = fate; t->fateColumn[ip] = fateCol; t->fateWeight[ip] = fateW; t->fateOrder[ip] = order; }
  }
  if (t->drawStart) t->drawStart[n] = rng->draws - draws0;
  t->nPhotons += nPhotons;
  free(contrib); free(ixF); free(iyF); free(cum);
  return nPhotons;
}

/* computeRadiativeTransfer :327-395 */
void orc_normalise(const orc_problem *p, int64_t numPhotonsProcessed, orc_tallies *t) {
  size_t ncol = (size_t)p->nx * p->ny;
  int xyReg, zReg;
  orc_regular_flags(p, &xyReg, &zReg);
  if (p->nDir > 0 && p->limitContrib) {
    for (int j = 0; j <= p->ncomp; j++)
      for (int d = 0; d < p->nDir; d++) {
        float ex = t->intensityExcess[(size_t)j * p->nDir + d];
        if (ex > 0.0f) {
          float *byc = t->intensityByComp + ((size_t)j * p->nDir + d) * ncol;
          float s = 0.0f;
          for (size_t k = 0; k < ncol; k++) s += byc[k];
          for (size_t k = 0; k < ncol; k++) t->intensity[d * ncol + k] += (byc[k] / s) * ex;
          /* the reference re-evaluates sum() after intensity, before updating byComponent: same s */
          for (size_t k = 0; k < ncol; k++) byc[k] = byc[k] + (byc[k] / s) * ex;
        }
      }
  }
  float *perCol = malloc(sizeof(float) * ncol);
  if (xyReg) {
    float v = (float)numPhotonsProcessed / (float)(p->nx * p->ny);
    for (size_t k = 0; k < ncol; k++) perCol[k] = v;
  } else {
    for (int j = 1; j <= p->ny; j++)
      for (int i = 1; i <= p->nx; i++) {
        float a = ((YE(j + 1) - YE(j)) * (XE(i + 1) - XE(i))) / ((XE(p->nx + 1) - XE(1)) * (YE(p->ny + 1) - YE(1)));
        perCol[(size_t)(j - 1) * p->nx + (i - 1)] = a * (float)numPhotonsProcessed;
      }
  }
  for (size_t k = 0; k < ncol; k++) {
    t->fluxUp[k] /= perCol[k];
    t->fluxDown[k] /= perCol[k];
    t->fluxAbsorbed[k] /= perCol[k];
  }
  for (int kz = 1; kz <= p->nz; kz++)
    for (size_t k = 0; k < ncol; k++)
      t->volumeAbsorption[(size_t)(kz - 1) * ncol + k] /= (perCol[k] * (ZE(kz + 1) - ZE(kz)));
  if (p->nDir > 0) {
    for (int d = 0; d < p->nDir; d++)
      for (size_t k = 0; k < ncol; k++) t->intensity[d * ncol + k] /= perCol[k];
    for (int j = 1; j <= p->ncomp; j++) /* NB the reference skips component 0 here (:390) */
      for (int d = 0; d < p->nDir; d++)
        for (size_t k = 0; k < ncol; k++) t->intensityByComp[((size_t)j * p->nDir + d) * ncol + k] /= perCol[k];
  }
  free(perCol);
}

/* newPhotonStream_Directional, Code/monteCarloIllumination.f95:62-104 */
void orc_photons_directional(orc_mt *rng, float solarMu, float solarAzimuthDeg, int64_t n,
                             float *xs, float *ys, float *zs, float *mus, float *phis) {
  for (int64_t i = 0; i < n; i++) {
    xs[i] = orc_mt_real(rng);
    ys[i] = orc_mt_real(rng);
  }
  float z = 1.0f - orc_spacing(1.0f);
  float mu = -fabsf(solarMu);
  float phi = solarAzimuthDeg * acosf(-1.0f) / 180.0f;
  for (int64_t i = 0; i < n; i++) { zs[i] = z; mus[i] = mu; phis[i] = phi; }
}
