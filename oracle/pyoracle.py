"""ctypes front end of the CPU oracle (oracle/liboracle_i3rc.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product path (HIP, through include/i3rc_hip.h) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_i3rc.so")

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("numerics.c", "integrator.c", "illumination.c", "i3rc_oracle.h", "Makefile")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _SO


class MT(C.Structure):
    _fields_ = [("state", C.c_int32 * 624), ("cur", C.c_int32), ("draws", C.c_int64)]


class Problem(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("ncomp", C.c_int),
        ("xEdges", fp), ("yEdges", fp), ("zEdges", fp),
        ("totalExt", fp), ("cumExt", fp), ("ssa", fp), ("pfIndex", ip),
        ("inverseTables", C.POINTER(fp)), ("nInvSteps", C.POINTER(C.c_int)),
        ("forwardTables", C.POINTER(fp)), ("forwardOrigTables", C.POINTER(fp)), ("nFwdSteps", C.POINTER(C.c_int)),
        ("surfaceAlbedo", C.c_float), ("useSurfaceBDRF", C.c_int), ("nxs", C.c_int), ("nys", C.c_int),
        ("xsEdges", fp), ("ysEdges", fp), ("brdf", fp),
        ("useRayTracing", C.c_int), ("useRussianRoulette", C.c_int),
        ("nDir", C.c_int), ("dirCos", fp),
        ("useHybrid", C.c_int), ("numOrdersOrig", C.c_int),
        ("useRRForIntensity", C.c_int), ("zetaMin", C.c_float),
        ("limitContrib", C.c_int), ("maxContrib", C.c_float),
    ]


class Tallies(C.Structure):
    _fields_ = [
        ("fluxUp", fp), ("fluxDown", fp), ("fluxAbsorbed", fp), ("volumeAbsorption", fp),
        ("intensity", fp), ("intensityByComp", fp), ("intensityExcess", fp),
        ("nPhotons", C.c_int64), ("nBad", C.c_int64), ("tracerCalls", C.c_int64), ("cellSteps", C.c_int64),
        ("scatterings", C.c_int64), ("surfaceHits", C.c_int64), ("roulettePlays", C.c_int64), ("exitsTop", C.c_int64),
        ("drawStart", lp), ("fate", ip), ("fateColumn", ip), ("fateWeight", fp), ("fateOrder", ip),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_mt_real.restype = C.c_float
        L.orc_mt_double.restype = C.c_double
        L.orc_mt_int.restype = C.c_int32
        L.orc_spacing.restype = C.c_float
        L.orc_spacing.argtypes = [C.c_float]
        L.orc_find_index.restype = C.c_int
        L.orc_find_index.argtypes = [C.c_float, fp, C.c_int, C.c_int]
        L.orc_compute_rt.restype = C.c_int64
        L.orc_compute_rt.argtypes = [C.POINTER(Problem), C.POINTER(MT), C.c_int64, fp, fp, fp, fp, fp, C.POINTER(Tallies)]
        L.orc_normalise.argtypes = [C.POINTER(Problem), C.c_int64, C.POINTER(Tallies)]
        L.orc_trace.restype = C.c_float
        L.orc_trace.argtypes = [C.POINTER(Problem), fp, fp, C.POINTER(C.c_int), C.c_int, C.c_float, lp]
        L.orc_photons_directional.argtypes = [C.POINTER(MT), C.c_float, C.c_float, C.c_int64, fp, fp, fp, fp, fp]
        L.orc_photons_random_azimuth.argtypes = [C.POINTER(MT), C.c_float, C.c_int64, fp, fp, fp, fp, fp]
        L.orc_photons_flux.argtypes = [C.POINTER(MT), C.c_int64, fp, fp, fp, fp, fp]
        L.orc_photons_spotlight.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, fp, fp, fp, fp, fp]
        L.orc_photons_internal_flux.argtypes = [C.POINTER(MT), C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                                C.c_int64, fp, fp, fp, fp, fp]
        L.orc_photons_internal_intensity.argtypes = [C.POINTER(MT), C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                     C.c_float, C.c_float, C.c_int64, fp, fp, fp, fp, fp]
        _lib = L
    return _lib


def _libm_f(name, x):
    """libm's float32 routine `name` at x (the C library the reference's intrinsics are compiled to)"""
    import ctypes.util
    global _LIBM
    try:
        _LIBM
    except NameError:
        _LIBM = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    f = getattr(_LIBM, name)
    f.restype, f.argtypes = C.c_float, [C.c_float]
    return np.float32(f(float(x)))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pf(a):
    return a.ctypes.data_as(fp)


# ---------------------------------------------------------------------------------------------------------
# RNG
# ---------------------------------------------------------------------------------------------------------
class RandomNumberSequence:
    """MT19937 as in Code/RandomNumbersForMC.f95 (new_RandomNumberSequence scalar / vector seed)."""

    def __init__(self, seed):
        self.t = MT()
        if np.isscalar(seed):
            lib().orc_mt_seed_scalar(C.byref(self.t), C.c_int32(int(seed)))
        else:
            s = (C.c_int32 * len(seed))(*[int(v) for v in seed])
            lib().orc_mt_seed_vector(C.byref(self.t), s, len(seed))

    def real(self):
        return float(lib().orc_mt_real(C.byref(self.t)))

    def int32(self):
        return int(lib().orc_mt_int(C.byref(self.t)))

    def reals(self, n):
        return np.array([lib().orc_mt_real(C.byref(self.t)) for _ in range(n)], dtype=np.float32)

    @property
    def draws(self):
        return int(self.t.draws)


# ---------------------------------------------------------------------------------------------------------
# Phase-function tables
# ---------------------------------------------------------------------------------------------------------
def powi(a, n):
    """real(4)**integer as the Fortran compiler lowers it (square and multiply, float32)."""
    a = np.float32(a)
    r = np.float32(1.0)
    while True:
        if n & 1:
            r = np.float32(r * a)
        n //= 2
        if n == 0:
            break
        a = np.float32(a * a)
    return r


def hg_coefficients(g, n):
    """Henyey-Greenstein Legendre coefficients g**l, l=1..n  (I3RC-Examples/i3rcStepCloud.f95:54-55)."""
    return np.array([powi(g, l) for l in range(1, n + 1)], dtype=np.float32)


def inverse_table_legendre(coef, n_steps):
    coef = _f(coef)
    out = np.zeros(n_steps, np.float32)
    lib().orc_inverse_table_legendre(_pf(coef), len(coef), n_steps, _pf(out))
    return out


def inverse_table_tabulated(angles, values, n_steps):
    angles, values = _f(angles), _f(values)
    out = np.zeros(n_steps, np.float32)
    lib().orc_inverse_table_tabulated(_pf(angles), _pf(values), len(angles), n_steps, _pf(out))
    return out


def forward_table_legendre(coef, n_steps):
    coef = _f(coef)
    out = np.zeros(n_steps, np.float32)
    lib().orc_forward_table_legendre(_pf(coef), len(coef), n_steps, _pf(out))
    return out


def forward_table_tabulated(angles, values, n_steps):
    angles, values = _f(angles), _f(values)
    out = np.zeros(n_steps, np.float32)
    lib().orc_forward_table_tabulated(_pf(angles), _pf(values), len(angles), n_steps, _pf(out))
    return out


def normalize_tabulated(angles, values):
    angles, values = _f(angles), _f(values)
    out = np.zeros_like(values)
    lib().orc_normalize_tabulated(_pf(angles), _pf(values), len(angles), _pf(out))
    return out


def hybrid_tables(values, width_deg):
    values = _f(values)
    if values.ndim == 1:
        values = values[None, :]
    out = np.zeros_like(values)
    lib().orc_hybrid_phase_functions(values.shape[1], values.shape[0], _pf(values), C.c_float(width_deg), _pf(out))
    return out


def lobatto(n):
    mus = np.zeros(n, np.float32)
    w = np.zeros(n, np.float32)
    lib().orc_lobatto(n, _pf(mus), _pf(w))
    return mus, w


def gauss_legendre(n):
    mus = np.zeros(n, np.float32)
    w = np.zeros(n, np.float32)
    lib().orc_gauss_legendre(n, _pf(mus), _pf(w))
    return mus, w


def legendre_polynomials(max_l, mus):
    """P[mu][l], l = 0 .. max_l (computeLegendrePolynomials, Code/numericUtilities.f95:175-193)"""
    mus = _f(mus)
    P = np.zeros((len(mus), max_l + 1), np.float32)
    lib().orc_legendre_polynomials(max_l, _pf(mus), len(mus), _pf(P))
    return P


def find_index(value, table, first_guess=0):
    table = _f(table)
    return int(lib().orc_find_index(C.c_float(value), _pf(table), len(table), int(first_guess)))


def surface_reflectance(xs_edges, ys_edges, brdf, x, y):
    """computeSurfaceReflectance (Code/surfaceProperties.f95:121-162) at the points (x, y); brdf [nys][nxs]"""
    xs, ys, b, x, y = _f(xs_edges), _f(ys_edges), _f(brdf), _f(x), _f(y)
    out = np.zeros(len(x), np.float32)
    lib().orc_surface_reflectance(len(xs) - 1, len(ys) - 1, _pf(xs), _pf(ys), _pf(b), len(x), _pf(x), _pf(y), _pf(out))
    return out


# ---------------------------------------------------------------------------------------------------------
# Integrator
# ---------------------------------------------------------------------------------------------------------
class Integrator:
    """Mirror of the reference integrator object, backed by the C restatement.

    Grids are numpy arrays indexed [z, y, x] (x fastest), per-component arrays [comp, z, y, x].
    inverse_tables / forward_tables: list (one per component) of arrays [nEntries, nSteps].
    """

    def __init__(self, x_edges, y_edges, z_edges, ext, ssa, pf_index, inverse_tables,
                 forward_tables=None, forward_orig_tables=None):
        self.xe, self.ye, self.ze = _f(x_edges), _f(y_edges), _f(z_edges)
        ext = _f(ext)
        if ext.ndim == 3:
            ext = ext[None]
        self.ncomp, self.nz, self.ny, self.nx = ext.shape
        ssa = _f(ssa)
        if ssa.ndim == 3:
            ssa = ssa[None]
        pf = np.ascontiguousarray(pf_index, dtype=np.int32)
        if pf.ndim == 3:
            pf = pf[None]
        # getOpticalPropertiesByComponent (Code/opticalProperties.f95:523-537) + new_Integrator :233-234
        cum = np.cumsum(ext, axis=0, dtype=np.float32).astype(np.float32)
        self.total_ext = np.ascontiguousarray(cum[-1])
        mask = self.total_ext > np.finfo(np.float32).tiny
        cum = np.where(mask[None], cum / np.where(mask, self.total_ext, 1)[None], cum).astype(np.float32)
        last = cum[-1]
        last[np.abs(last - np.float32(1)) <= np.spacing(np.float32(1))] = np.float32(1) + np.spacing(np.float32(1))
        self.cum_ext = np.ascontiguousarray(cum)
        self.ssa = np.ascontiguousarray(ssa)
        self.pf = pf
        self.inv = [np.ascontiguousarray(np.atleast_2d(_f(t))) for t in inverse_tables]
        self.fwd = [np.ascontiguousarray(np.atleast_2d(_f(t))) for t in (forward_tables or [])]
        self.fwd_orig = [np.ascontiguousarray(np.atleast_2d(_f(t))) for t in (forward_orig_tables or forward_tables or [])]
        self.params = dict(surfaceAlbedo=0.0, useRayTracing=1, useRussianRoulette=1, useHybrid=0, numOrdersOrig=0,
                           useRRForIntensity=0, zetaMin=0.3, limitContrib=0, maxContrib=float(np.finfo(np.float32).max))
        self.dirs = np.zeros((0, 3), np.float32)
        self.brdf = None
        self._keep = []

    def specify(self, **kw):
        """specifyParameters (monteCarloRadiativeTransfer.f95:830-1069) -- subset of keywords, same meaning."""
        if "intensityMus" in kw:
            mus = _f(kw.pop("intensityMus"))
            phis = _f(kw.pop("intensityPhis"))
            pi = np.float32(3.14159265358979312)
            d = []
            for m, ph in zip(mus, phis):
                phr = np.float32(np.float32(ph * pi) / np.float32(180.0))
                st = np.sqrt(np.float32(1.0) - m * m, dtype=np.float32)
                # (libm's cosf / sinf, which the reference's cos / sin end in: numpy's float32 routines are SIMD code of their own and
                # differ from them in the last bit here and there -- enough to move a ray: tests/test_ref_loop.py)
                d.append([st * _libm_f("cosf", phr), st * _libm_f("sinf", phr), m])
            self.dirs = np.array(d, np.float32).reshape(-1, 3)
        if "surfaceBDRF" in kw:
            xs, ys, alb = kw.pop("surfaceBDRF")
            self.brdf = (_f(xs), _f(ys), _f(alb))
        self.params.update(kw)

    def _problem(self):
        p = Problem()
        p.nx, p.ny, p.nz, p.ncomp = self.nx, self.ny, self.nz, self.ncomp
        p.xEdges, p.yEdges, p.zEdges = _pf(self.xe), _pf(self.ye), _pf(self.ze)
        p.totalExt, p.cumExt, p.ssa = _pf(self.total_ext), _pf(self.cum_ext), _pf(self.ssa)
        p.pfIndex = self.pf.ctypes.data_as(ip)
        nc = self.ncomp
        inv = (fp * nc)(*[_pf(t) for t in self.inv])
        ninv = (C.c_int * nc)(*[t.shape[1] for t in self.inv])
        p.inverseTables, p.nInvSteps = inv, ninv
        self._keep = [inv, ninv]
        if self.fwd:
            fw = (fp * nc)(*[_pf(t) for t in self.fwd])
            fo = (fp * nc)(*[_pf(t) for t in self.fwd_orig])
            nf = (C.c_int * nc)(*[t.shape[1] for t in self.fwd])
            p.forwardTables, p.forwardOrigTables, p.nFwdSteps = fw, fo, nf
            self._keep += [fw, fo, nf]
        q = self.params
        p.surfaceAlbedo = q["surfaceAlbedo"]
        if self.brdf is not None:
            xs, ys, alb = self.brdf
            p.useSurfaceBDRF = 1
            p.nxs, p.nys = len(xs) - 1, len(ys) - 1
            p.xsEdges, p.ysEdges, p.brdf = _pf(xs), _pf(ys), _pf(alb)
        p.useRayTracing, p.useRussianRoulette = int(q["useRayTracing"]), int(q["useRussianRoulette"])
        p.nDir = len(self.dirs)
        p.dirCos = _pf(self.dirs)
        p.useHybrid, p.numOrdersOrig = int(q["useHybrid"]), int(q["numOrdersOrig"])
        p.useRRForIntensity, p.zetaMin = int(q["useRRForIntensity"]), q["zetaMin"]
        p.limitContrib, p.maxContrib = int(q["limitContrib"]), q["maxContrib"]
        return p

    def compute(self, rng, xs, ys, zs, mus, phis, record=False, normalise=True):
        """computeRadiativeTransfer for one batch; returns dict of (normalised) tallies + counters."""
        p = self._problem()
        n = len(xs)
        nd = len(self.dirs)
        ncol = self.nx * self.ny
        res = dict(
            fluxUp=np.zeros((self.ny, self.nx), np.float32), fluxDown=np.zeros((self.ny, self.nx), np.float32),
            fluxAbsorbed=np.zeros((self.ny, self.nx), np.float32),
            volumeAbsorption=np.zeros((self.nz, self.ny, self.nx), np.float32),
            intensity=np.zeros((max(nd, 1), self.ny, self.nx), np.float32),
            intensityByComp=np.zeros((self.ncomp + 1, max(nd, 1), self.ny, self.nx), np.float32),
            intensityExcess=np.zeros((self.ncomp + 1, max(nd, 1)), np.float32),
        )
        t = Tallies()
        t.fluxUp, t.fluxDown, t.fluxAbsorbed = _pf(res["fluxUp"]), _pf(res["fluxDown"]), _pf(res["fluxAbsorbed"])
        t.volumeAbsorption = _pf(res["volumeAbsorption"])
        t.intensity, t.intensityByComp, t.intensityExcess = _pf(res["intensity"]), _pf(res["intensityByComp"]), _pf(res["intensityExcess"])
        if record:
            res["drawStart"] = np.zeros(n + 1, np.int64)
            res["fate"] = np.zeros(n, np.int32)
            res["fateColumn"] = np.zeros(n, np.int32)
            res["fateWeight"] = np.zeros(n, np.float32)
            res["fateOrder"] = np.zeros(n, np.int32)
            t.drawStart = res["drawStart"].ctypes.data_as(lp)
            t.fate = res["fate"].ctypes.data_as(ip)
            t.fateColumn = res["fateColumn"].ctypes.data_as(ip)
            t.fateWeight = _pf(res["fateWeight"])
            t.fateOrder = res["fateOrder"].ctypes.data_as(ip)
        xs, ys, zs, mus, phis = map(_f, (xs, ys, zs, mus, phis))
        nproc = lib().orc_compute_rt(C.byref(p), C.byref(rng.t), n, _pf(xs), _pf(ys), _pf(zs), _pf(mus), _pf(phis), C.byref(t))
        if normalise:
            lib().orc_normalise(C.byref(p), nproc, C.byref(t))
        for k in ("nPhotons", "nBad", "tracerCalls", "cellSteps", "scatterings", "surfaceHits", "roulettePlays", "exitsTop"):
            res[k] = int(getattr(t, k))
        if nd == 0:
            for k in ("intensity", "intensityByComp", "intensityExcess"):
                res.pop(k)
        res["ncol"] = ncol
        return res

    def trace(self, direction, pos, idx, target=None):
        """One accumulateExtinctionAlongPath call; returns (tau, pos, idx, steps)."""
        p = self._problem()
        d = _f(direction).copy()
        ps = _f(pos).copy()
        ix = (C.c_int * 3)(*[int(v) for v in idx])
        steps = C.c_int64(0)
        tau = lib().orc_trace(C.byref(p), _pf(d), _pf(ps), ix, 0 if target is None else 1,
                              C.c_float(0.0 if target is None else target), C.byref(steps))
        return float(tau), ps, [ix[0], ix[1], ix[2]], steps.value


def photons_directional(rng, solar_mu, solar_azimuth_deg, n):
    """new_PhotonStream(solarMu, solarAzimuth, numberOfPhotons, randomNumbers)."""
    arrs = [np.zeros(n, np.float32) for _ in range(5)]
    lib().orc_photons_directional(C.byref(rng.t), C.c_float(solar_mu), C.c_float(solar_azimuth_deg), n, *[_pf(a) for a in arrs])
    return arrs


def _five(n):
    return [np.zeros(n, np.float32) for _ in range(5)]


def photons_random_azimuth(rng, solar_mu, n):
    """new_PhotonStream(solarMu, numberOfPhotons, randomNumbers): Code/monteCarloIllumination.f95:106-146."""
    a = _five(n)
    lib().orc_photons_random_azimuth(C.byref(rng.t), solar_mu, n, *[_pf(x) for x in a])
    return a


def photons_flux(rng, n):
    """new_PhotonStream(numberOfPhotons, randomNumbers): :148-185 (flux on the horizontal equally weighted in mu)."""
    a = _five(n)
    lib().orc_photons_flux(C.byref(rng.t), n, *[_pf(x) for x in a])
    return a


def photons_spotlight(solar_mu, solar_azimuth_deg, solar_x, solar_y, n):
    """new_PhotonStream(solarMu, solarAzimuth, solarX, solarY, numberOfPhotons): :187-226."""
    a = _five(n)
    lib().orc_photons_spotlight(solar_mu, solar_azimuth_deg, solar_x, solar_y, n, *[_pf(x) for x in a])
    return a


def photons_internal_flux(rng, x, y, z, points_up, n, delta_x=None, delta_y=None):
    """new_PhotonStream(detectorX, detectorY, detectorZ, detectorPointsUp, [deltaX, deltaY,] ...): :228-331."""
    a = _five(n)
    lib().orc_photons_internal_flux(C.byref(rng.t), x, y, z, int(bool(points_up)), -1.0 if delta_x is None else delta_x,
                                    -1.0 if delta_y is None else delta_y, n, *[_pf(v) for v in a])
    return a


def photons_internal_intensity(rng, x, y, z, mu, phi_deg, n, delta_x=None, delta_y=None):
    """new_PhotonStream(detectorX, detectorY, detectorZ, detectorMu, detectorPhi, ...): :333-424 (phi stays in degrees)."""
    a = _five(n)
    lib().orc_photons_internal_intensity(C.byref(rng.t), x, y, z, mu, phi_deg, -1.0 if delta_x is None else delta_x,
                                         -1.0 if delta_y is None else delta_y, n, *[_pf(v) for v in a])
    return a
