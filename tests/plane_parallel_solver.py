"""An INDEPENDENT deterministic solver for the plane-parallel problem Example-Drivers/planeParallel.f95 sets up (a homogeneous
slab, Henyey-Greenstein phase function, collimated sun, Lambertian surface): adding-doubling in float64 numpy, written from
the textbook method (van de Hulst 1963; Hansen & Travis 1974, Space Sci. Rev. 16, section 4; Liou, "An Introduction to
Atmospheric Radiation", section 6.4).  It shares no code, table or random number with oracle/ or the HIP kernels and does
not trace photons at all: it pins MULTIPLE scattering -- what the closed forms of tests/test_closed_form.py (first order
only) cannot.

Method.  The radiance is expanded in azimuth, I(mu, phi) = sum_m I_m(mu) cos m (phi - phi_sun); for every Fourier mode the
slab is built by doubling from a layer so thin (optical depth < 2e-7) that single scattering describes it, on n Gauss
nodes per hemisphere plus zero-weight nodes at the directions radiances are wanted in.  The phase function enters through
its Legendre coefficients beta_l = (2 l + 1) g^l (l <= 2 n - 1: exactly integrable on the nodes) and the addition theorem
with normalised associated Legendre functions.  The collimated beam is carried as a source (never as a node); a
Lambertian surface is added underneath as a reflecting layer of mode 0.

Normalisation = the reference's: unit flux through a horizontal plane at the top (F_sun = 1 / mu0), so that fluxUp / fluxDown /
fluxAbsorbed and `intensity` compare directly with reportResults (a Lambertian surface of albedo a alone gives a / pi).
fluxDown counts every arrival at the surface, reflected light coming down again included."""
import numpy as np


def _gauss_half(n):
    """n Gauss-Legendre nodes and weights on (0, 1)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _norm_assoc_legendre(lmax, m, mu):
    """Y[l] = sqrt((l-m)!/(l+m)!) P_l^m(mu) for l = m .. lmax (rows), mu an array (columns); sign convention without the
    Condon-Shortley phase -- only products Y(mu) Y(mu') are used."""
    mu = np.asarray(mu, np.float64)
    Y = np.zeros((lmax + 1, mu.size))
    s = np.sqrt(np.maximum(0.0, 1.0 - mu * mu))
    y = np.ones_like(mu)
    for k in range(1, m + 1):
        y = y * np.sqrt((2.0 * k - 1.0) / (2.0 * k)) * s
    if m <= lmax:
        Y[m] = y
    if m + 1 <= lmax:
        Y[m + 1] = np.sqrt(2.0 * m + 1.0) * mu * y
    for l in range(m + 2, lmax + 1):
        Y[l] = ((2.0 * l - 1.0) * mu * Y[l - 1] - np.sqrt((l - 1.0) ** 2 - m * m) * Y[l - 2]) / np.sqrt(float(l * l - m * m))
    return Y


def solve(tau, omega, g, mu0, albedo=0.0, radiance_mus=(), radiance_dphis_deg=(), n=64, max_mode=None, moments=None, chi=None):
    """Fluxes and upward radiances at the top of a homogeneous slab over a Lambertian surface.
    radiance_mus: cosines (> 0) of the viewing zenith angles; radiance_dphis_deg: azimuth of each viewing direction minus the
    azimuth of the sun's PROPAGATION direction, both as the reference defines them (makeDirectionCosines of (mu, phi)).
    moments: the phase function is the Legendre series truncated after this many moments (l = 1 .. moments), as the drivers
    build it (planeParallel.f95:340-352: 64 by default); None = the series up to l = 2 n - 1.
    chi: Legendre moments chi_l (l = 0 ..) of the phase function to use instead of g**l, e.g. those of the distribution a
    tabulated sampler actually draws from.
    Returns dict(fluxUp, fluxDown, fluxAbsorbed, intensity[list])."""
    mu0 = abs(float(mu0))
    lmax = 2 * n - 1
    beta = (2.0 * np.arange(lmax + 1) + 1.0) * float(g) ** np.arange(lmax + 1)
    if moments is not None:
        beta[int(moments) + 1:] = 0.0
    if chi is not None:
        chi = np.asarray(chi, np.float64)[:lmax + 1]
        beta = np.zeros(lmax + 1)
        beta[:chi.size] = (2.0 * np.arange(chi.size) + 1.0) * chi
    gm, gw = _gauss_half(n)
    out_mu = np.asarray(radiance_mus, np.float64)
    mus = np.concatenate([gm, out_mu])
    w = np.concatenate([gw, np.zeros(out_mu.size)])
    N = mus.size
    F0 = 1.0 / mu0
    # thin starting layer: tau / 2^K below 2e-7
    K = max(0, int(np.ceil(np.log2(max(tau, 1e-300) / 2e-7))))
    d0 = tau / 2.0 ** K
    want_radiance = out_mu.size > 0
    modes = range(0, (lmax if max_mode is None else max_mode) + 1) if want_radiance else range(0, 1)
    inten = np.zeros(out_mu.size)
    dphi = np.deg2rad(np.asarray(radiance_dphis_deg, np.float64)) if want_radiance else None
    small = 0
    res = {}
    for m in modes:
        Yp, Ym, Y0 = _norm_assoc_legendre(lmax, m, mus), _norm_assoc_legendre(lmax, m, -mus), _norm_assoc_legendre(lmax, m, np.array([-mu0]))
        b = beta[:, None]
        pSame = (Yp * b).T @ Yp            # p_m(mu_i, mu_j): same hemisphere (= p_m(-mu_i, -mu_j))
        pOpp = (Yp * b).T @ Ym             # p_m(mu_i, -mu_j): hemispheres exchanged
        pSunUp = ((Yp * b).T @ Y0)[:, 0]   # p_m(+mu_i, -mu0): beam into an upward direction
        pSunDn = ((Ym * b).T @ Y0)[:, 0]   # p_m(-mu_i, -mu0): beam into a downward direction
        # single-scattering layer of optical depth d0, in operator form (the quadrature weights inside the matrices)
        # (the attenuation of a grazing node's own radiance is taken exactly, 1 - exp(-d0 / mu), instead of d0 / mu)
        path = -np.expm1(-d0 / mus)
        c = path * omega / 2.0
        R = c[:, None] * pOpp * w[None, :]
        T = np.diag(1.0 - path) + c[:, None] * pSame * w[None, :]
        fac = (2.0 if m > 0 else 1.0) * omega * F0 / (4.0 * np.pi)
        sUp = path * fac * pSunUp
        sDn = path * fac * pSunDn
        E = np.exp(-d0 / mu0)
        I = np.eye(N)
        for _ in range(K):   # doubling: the layer on top of a copy of itself (the copy sees the beam attenuated by E)
            G = np.linalg.inv(I - R @ R)
            u = G @ (R @ sDn + E * sUp)          # upward radiance at the interface
            d = sDn + R @ u                      # downward radiance at the interface
            sUp, sDn = sUp + T @ u, E * sDn + T @ d
            R, T = R + T @ G @ R @ T, T @ G @ T
            E = E * E
        if m == 0:
            # Lambertian surface underneath: down-welling radiance -> isotropic up-welling radiance (albedo / pi) x flux
            Rs = np.tile((2.0 * albedo * mus * w)[None, :], (N, 1))
            sSurf = np.full(N, albedo / np.pi * mu0 * F0)
            G = np.linalg.inv(I - Rs @ R)
            u = G @ (Rs @ sDn + E * sSurf)
            d = sDn + R @ u
            top = sUp + T @ u
            res["fluxUp"] = float(2.0 * np.pi * np.sum(w * mus * top))
            res["fluxDown"] = float(mu0 * F0 * E + 2.0 * np.pi * np.sum(w * mus * d))
            res["fluxAbsorbed"] = 1.0 - res["fluxUp"] - (1.0 - albedo) * res["fluxDown"]
        else:
            top = sUp
        if want_radiance:
            term = top[n:] * np.cos(m * dphi)
            inten += term
            small = small + 1 if np.all(np.abs(term) < 1e-9 * (1.0 + np.abs(inten))) else 0
            if small >= 3:
                break
    res["intensity"] = [float(v) for v in inten]
    return res
