"""Setup-time phase-function numerics (host side, numpy float32): the tables the GPU kernels read.

Mirrors, for the hot path only,
  Code/scatteringPhaseFunctions.f95   phaseFunction / phaseFunctionTable, getPhaseFunctionValues (:446-648)
  Code/numericUtilities.f95           computeLobattoTerms (:15-102), computeLegendrePolynomials (:175-193), findIndex
  Code/inversePhaseFunctions.f95      computeInversePhaseFuncTable (:28-176)
  Integrators/monteCarloRadiativeTransfer.f95  tabulateForwardPhaseFunctions (:1863-1923), hybrid tables (:1925-2039)
Arithmetic is float32 in the reference's operator order; numpy's float32 sin/cos/acos may differ from the
Fortran run-time's by an ulp, so tables agree with the reference to ~1e-6, not bitwise (tests state this).
"""
import ctypes
import ctypes.util

import numpy as np

f32 = np.float32
PI_SPF = f32(3.141592654)          # scatteringPhaseFunctions.f95:26
PI_MCRT = f32(3.14159265358979312)  # monteCarloRadiativeTransfer.f95:43


_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]
_sinf = _libm.sinf
for _name in ("cosf", "acosf", "expf", "logf"):
    getattr(_libm, _name).restype = ctypes.c_float
    getattr(_libm, _name).argtypes = [ctypes.c_float]


def libm_f32(name, x):
    """libm's float32 routine `name` -- the C library's cosf, sinf, ... which the reference's intrinsics are compiled to (amdflang and
    gfortran alike); numpy's float32 routines are SIMD code of their own and differ from them in the last bit here and there.
    Scalar or array in, float32 of the same shape out."""
    f = getattr(_libm, name)
    a = np.asarray(x, np.float32)
    if a.ndim == 0:
        return np.float32(f(float(a)))
    return np.fromiter((f(float(v)) for v in a.ravel()), np.float32, a.size).reshape(a.shape)


def cos32(x):
    return libm_f32("cosf", x)


def spacing(x):
    """Fortran SPACING() for real(4) (elementwise)."""
    x = np.asarray(x, dtype=np.float32)
    tiny = np.finfo(np.float32).tiny
    with np.errstate(over="ignore"):
        s = np.spacing(np.abs(x)).astype(np.float32)
    return np.where((x == 0) | (s < tiny), tiny, s).astype(np.float32)


def find_index(value, table, first_guess=0):
    """findIndex (numericUtilities.f95:195-248): 1-based i with table(i) <= value < table(i+1)."""
    n = len(table)
    T = lambda i: table[i - 1]
    if first_guess > 0:
        lower, inc = first_guess, 1
        while True:
            upper = min(lower + inc, n)
            if lower == n or (T(lower) <= value and T(upper) > value):
                break
            if T(lower) > value:
                upper = lower
                lower = max(upper - inc, 1)
            else:
                lower = upper
            inc *= 2
    else:
        lower, upper = 0, n
    while not (lower == n or upper <= lower + 1):
        mid = (lower + upper) // 2
        if value >= T(mid):
            lower = mid
        else:
            upper = mid
    return lower


def legendre_polynomials(max_l, mus):
    """computeLegendrePolynomials: array [max_l+1, len(mus)] (float32 recursion)."""
    mus = np.asarray(mus, dtype=np.float32)
    P = np.empty((max_l + 1, mus.size), np.float32)
    P[0] = 1
    if max_l >= 1:
        P[1] = mus
    for l in range(1, max_l):
        P[l + 1] = ((f32(2 * l + 1) * mus) * P[l] - f32(l) * P[l - 1]) / f32(l + 1)
    return P


def lobatto(n):
    """computeLobattoTerms: abscissas (increasing from -1 to 1) and weights of n-point Lobatto quadrature."""
    pi = np.arccos(f32(-1.0))
    mid = (n + 1) // 2
    m = mid - 1
    c1 = f32(1.0) if n % 2 == 1 else f32(0.5)
    i = np.arange(1, m + 1, dtype=np.float32)
    # (libm's sinf, the routine the reference's `sin` ends in: numpy's float32 sine is a SIMD routine of its own and the correctly
    # rounded sine is a third answer -- glibc's sinf is faithful, not correctly rounded -- and a first guess one ulp away let the
    # 128-point rule settle on two other float32 roots; tests/test_ref_numerics.py holds this function against the reference's own)
    arg = (pi * (i - c1) / (f32(n) - f32(1.0) + f32(0.5))).astype(np.float32)
    trial = np.array([_sinf(float(a)) for a in arg], np.float32)
    last = trial.copy()

    def update(mask):
        nonlocal trial, last, P
        mu = trial
        d1 = (f32(n - 1) * (mu * P[n - 1] - P[n - 2])) / (mu * mu - f32(1.0))
        d2 = ((f32(2.0) * mu) * d1 - (f32(n * (n - 1)) * P[n - 1])) / (f32(1.0) - mu * mu)
        new = mu - d1 / d2
        last = np.where(mask, mu, last).astype(np.float32)
        trial = np.where(mask, new, mu).astype(np.float32)

    with np.errstate(all="ignore"):
        P = legendre_polynomials(n - 1, trial)
        update(np.ones(m, bool))
        it = 0
        while True:
            tol = f32(3.0) * spacing(trial)
            if np.all(np.abs(trial - last) <= tol):
                break
            P = legendre_polynomials(n - 1, trial)
            update(np.abs(trial - last) > tol)
            it += 1
            if it > 25:
                break
    mus = np.zeros(n, np.float32)
    w = np.zeros(n, np.float32)
    mus[0] = -1
    w[0] = f32(2.0) / f32(n * (n - 1))
    for j in range(m):
        mus[mid - 1 - j] = -trial[j]
        w[mid - 1 - j] = f32(2.0) / (f32(n * (n - 1)) * (P[n - 1, j] * P[n - 1, j]))
    if n % 2 == 0:
        mus[mid:] = -mus[mid - 1::-1][:n - mid]
        w[mid:] = w[mid - 1::-1][:n - mid]
    else:
        tm, tw = -mus[mid - 1::-1].copy(), w[mid - 1::-1].copy()
        mus[mid - 1:] = tm[:n - mid + 1]
        w[mid - 1:] = tw[:n - mid + 1]
    return mus, w


class PhaseFunction:
    """type(phaseFunction) (scatteringPhaseFunctions.f95:34-46): Legendre coefficients l=1.. OR angle/value pairs."""

    def __init__(self, legendre=None, angles=None, values=None):
        if (legendre is None) == (angles is None):
            raise ValueError("new_PhaseFunction: give Legendre coefficients or angle/value pairs")
        self.legendre = None if legendre is None else np.ascontiguousarray(legendre, np.float32)
        self.angles = None if angles is None else np.ascontiguousarray(angles, np.float32)
        self.values_ = None if values is None else np.ascontiguousarray(values, np.float32)
        if self.angles is not None:
            if self.angles.shape != self.values_.shape:
                raise ValueError("new_PhaseFunction: scatteringAngle and value must be the same length")
            if np.any(np.diff(self.angles) <= 0):
                raise ValueError("new_PhaseFunction: scattering angles must be increasing, unique")
            if np.any(self.angles < 0) or np.any(self.angles > PI_SPF):
                raise ValueError("new_PhaseFunction: scattering angle out of bounds")
            if np.any(self.values_ < 0):
                raise ValueError("newPhaseFunction: Negative phase function values supplied.")
            # normalizePhaseFunction (:1329-1345): integral over cos(angle) = 2, as the constructors do (:145)
            cosa = cos32(self.angles)
            terms = (cosa[1:] - cosa[:-1]) * (f32(0.5) * (self.values_[1:] + self.values_[:-1]))
            dot = f32(0.0)
            for t in terms:
                dot = f32(dot + t)
            self.values_ = ((-self.values_ * f32(2.0)) / dot).astype(np.float32)

    @property
    def stored_as_legendre(self):
        return self.legendre is not None

    def values(self, angles):
        """getPhaseFunctionValues_one (:446-529)."""
        angles = np.asarray(angles, np.float32)
        if self.stored_as_legendre:
            c = self.legendre
            if c.size == 0:
                return np.full(angles.shape, f32(0.5), np.float32)
            P = legendre_polynomials(c.size, cos32(angles))
            wts = np.concatenate([[f32(1.0)], c * (2 * np.arange(1, c.size + 1) + 1).astype(np.float32)]).astype(np.float32)
            s = np.zeros(angles.shape, np.float32)
            for l in range(c.size + 1):
                s = s + wts[l] * P[l]
            return s
        return _interp_tabulated(self.angles, self.values_, angles, chained_guess=False)


def _interp_tabulated(tab_angles, tab_values, angles, chained_guess):
    """Linear interpolation in cos(angle) (:497-524 and :581-609)."""
    n = tab_angles.size
    out = np.empty(angles.size, np.float32)
    cos_tab = cos32(tab_angles)
    cos_a = cos32(angles)
    prev = 0
    huge = np.finfo(np.float32).max
    for i, a in enumerate(angles):
        k = find_index(a, tab_angles, prev if chained_guess else 0)
        if chained_guess:
            prev = k
        kp = k + 1
        if k < n:
            dmu = cos_tab[kp - 1] - cos_tab[k - 1]
        else:
            dmu = huge
            kp = k
        with np.errstate(all="ignore"):
            wgt = f32(1.0) - (cos_a[i] - cos_tab[k - 1]) / dmu
        out[i] = wgt * tab_values[k - 1] + (f32(1.0) - wgt) * tab_values[kp - 1]
    return out


class PhaseFunctionTable:
    """type(phaseFunctionTable) (:48-58): a keyed list of phase functions."""

    def __init__(self, phase_functions, key=None, description=""):
        self.entries = list(phase_functions)
        if not self.entries:
            raise ValueError("new_PhaseFunctionTable: no phase functions")
        self.key = np.arange(1, len(self.entries) + 1, dtype=np.float32) if key is None else np.asarray(key, np.float32)
        self.description = description

    @property
    def n_entries(self):
        return len(self.entries)

    def inverse_table(self, n_steps):
        """computeInversePhaseFuncTable: [nEntries, nSteps] scattering angle (rad) vs cumulative probability."""
        return np.stack([inverse_phase_function(p, n_steps) for p in self.entries])

    def forward_table(self, n_steps):
        """tabulateForwardPhaseFunctions :1896-1901: [nEntries, nSteps] values on angles j*pi/(nSteps-1)."""
        angles = ((np.arange(n_steps, dtype=np.float32) / f32(n_steps - 1)) * PI_MCRT).astype(np.float32)
        out = []
        for p in self.entries:
            if p.stored_as_legendre:  # getPhaseFunctionValues_table :575-580,:621-626: (2l+1) folded into P first
                c = p.legendre
                if c.size == 0:
                    out.append(np.full(n_steps, f32(0.5), np.float32))
                    continue
                P = legendre_polynomials(c.size, cos32(angles))
                s = np.zeros(n_steps, np.float32)
                for l in range(c.size + 1):
                    cl = f32(1.0) if l == 0 else c[l - 1]
                    s = s + cl * (f32(2 * l + 1) * P[l])
                out.append(s)
            else:
                out.append(_interp_tabulated(p.angles, p.values_, angles, chained_guess=True))
        return np.stack(out)


def inverse_phase_function(p, n_steps):
    """computeInversePhaseFunction (inversePhaseFunctions.f95:68-176)."""
    if p.stored_as_legendre:
        n = max(p.legendre.size, 2)
        mus, _ = lobatto(n)
        vals = p.values(libm_f32("acosf", mus[::-1]))[::-1].copy()
    else:
        n = p.angles.size
        vals = p.values(p.angles)[::-1].copy()
        mus = cos32(p.angles[::-1])
    incr = ((mus[1:] - mus[:-1]) * f32(0.5)) * (vals[1:] + vals[:-1])
    cdf = np.concatenate([[f32(0.0)], np.cumsum(incr, dtype=np.float32)]).astype(np.float32)
    cdf = (cdf / cdf[-1]).astype(np.float32)
    probs = (np.arange(n_steps, dtype=np.float32) / f32(n_steps - 1)).astype(np.float32)
    ind = np.empty(n_steps, np.int64)
    ind[0] = find_index(f32(0.0), cdf)
    for i in range(1, n_steps):
        ind[i] = find_index(probs[i], cdf, ind[i - 1])
    k = ind[:-1] - 1  # 0-based lower index
    pr = probs[:-1]
    c0, c1, m0, m1, v0, v1 = cdf[k], cdf[k + 1], mus[k], mus[k + 1], vals[k], vals[k + 1]
    with np.errstate(all="ignore"):
        flat = (c1 - c0) <= spacing(c0)
        const = np.abs(v0 - v1) <= spacing(v0)
        a_flat = m0
        a_const = m0 + (m1 - m0) * (pr - c0) / (c1 - c0)
        rad = ((c1 - pr) * (v0 * v0) + (pr - c0) * (v1 * v1)) / (c1 - c0)
        a_gen = m0 + (m1 - m0) / (v0 - v1) * (v0 - np.sqrt(rad, dtype=np.float32))
        arg = np.where(flat, a_flat, np.where(const, a_const, a_gen)).astype(np.float32)
        table = libm_f32("acosf", arg)
    return np.concatenate([table, [f32(0.0)]]).astype(np.float32)


def hybrid_phase_functions(values, width_deg):
    """computeHydridPhaseFunctions (:1925-2039): Gaussian forward peak spliced onto the tabulated functions.

    values: [nEntries, nSteps] on equally spaced angles 0..pi."""
    values = np.atleast_2d(np.asarray(values, np.float32))
    n = values.shape[1]
    angles = ((np.arange(n, dtype=np.float32) / f32(n - 1)) * PI_MCRT).astype(np.float32)
    cos_a = cos32(angles)
    w = f32(width_deg) * PI_MCRT / f32(180.0)
    gau = libm_f32("expf", (-((angles / w) ** 2)).astype(np.float32))
    dcos = (cos_a[:-1] - cos_a[1:]).astype(np.float32)
    new = values.copy()

    def norm(val, ti):  # computeNormalization :2000-2023 (sequential float32 dot products)
        ig = f32(0.0)
        for t in (f32(0.5) * (gau[:ti - 1] + gau[1:ti])) * dcos[:ti - 1]:
            ig = f32(ig + t)
        io = f32(0.0)
        for t in (f32(0.5) * (val[ti - 1:n - 1] + val[ti:n])) * dcos[ti - 1:]:
            io = f32(io + t)
        return f32(1.0) / ig if io >= 2.0 else (f32(2.0) - io) / ig

    def diff(val, ti):
        return norm(val, ti) * gau[ti - 1] - val[ti - 1]

    for e in range(values.shape[0]):
        val = values[e]
        lower = find_index(w, angles) + 1
        if lower >= n - 2:
            break
        low_diff = diff(val, lower)
        inc, no_root = 1, False
        while True:
            upper = min(lower + inc, n - 1)
            up_diff = diff(val, upper)
            if lower == n - 1:
                no_root = True
                break
            if low_diff * up_diff < 0:
                break
            lower, low_diff = upper, up_diff
            inc *= 2
        if no_root:
            continue
        while upper > lower + 1:
            mid = (lower + upper) // 2
            mid_diff = diff(val, mid)
            if mid_diff * up_diff < 0:
                lower, low_diff = mid, mid_diff
            else:
                upper, up_diff = mid, mid_diff
        ti = lower
        p0 = norm(val, ti)
        new[e, :ti] = p0 * gau[:ti]
        new[e, ti:] = val[ti:]
    return new


def powi(a, n):
    """real(4)**integer as compilers lower it (square and multiply in float32)."""
    a = f32(a)
    r = f32(1.0)
    while True:
        if n & 1:
            r = f32(r * a)
        n //= 2
        if n == 0:
            break
        a = f32(a * a)
    return r


def henyey_greenstein(g, n_coefficients):
    """HG phase function by Legendre moments g**l, l = 1..n (I3RC-Examples/i3rcStepCloud.f95:54-55)."""
    return PhaseFunction(legendre=np.array([powi(g, l) for l in range(1, n_coefficients + 1)], np.float32))
