"""Pins that do not come from this repository's own restatement: closed-form radiative-transfer results of the very
loop under test (Integrators/monteCarloRadiativeTransfer.f95 computeRT :452-691), checked on the CPU oracle (not-gpu
tests) and on the HIP path (gpu tests) with the same assertions.

  * omega = 0 (every first interaction absorbs: :642-649 takes the whole weight, the roulette :673-680 ends the photon):
    direct-beam transmission -- Beer-Lambert per column, per layer, for vertical and slant sun;
  * an empty domain over a Lambertian surface: fluxDown = 1, fluxUp = albedo, radiance albedo / pi in every direction
    (:515-580, :1479: the surface contribution is weight / pi), with and without the local estimate's roulette;
  * first-order scattering of a homogeneous slab at omega << 1: I = omega P(Theta) / (4 pi) (1 - exp(-tau (1/mu0 + 1/mu))) / (mu0 + mu)
    per unit flux on the horizontal -- pins the phase-function table, the 1 / (4 pi |mu|) normalisation (:1509) and the
    transmission of the local estimate (:1517-1535).

Tolerances: binomial / batch-scatter standard errors written next to each assertion (fixed seeds: the tests are
deterministic, the bounds say how far a correct implementation may be)."""
import numpy as np
import pytest

import i3rc_monte_carlo_model_amd as M
from tools import cases

f32 = np.float32
SIGMAS = 4.5   # per-column bounds over ~100 cells and quantities: false alarm of a correct code < 1e-3


# ---- the two implementations behind one face ------------------------------------------------------------------
class OracleRunner:
    def __init__(self, oracle):
        self.O = oracle

    def run(self, d, g_moments, n, nb, mu0, az=0.0, albedo=0.0, dirs=None, rri=False, roulette=True):
        O = self.O
        coef = O.hg_coefficients(0.85, g_moments)
        fwd = [O.forward_table_legendre(coef, 10001)] if dirs else None
        o = O.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], d["pf"], [O.inverse_table_legendre(coef, 10001)], fwd, fwd)
        kw = dict(surfaceAlbedo=albedo, useRussianRoulette=int(roulette))
        if dirs:
            kw.update(intensityMus=dirs[0], intensityPhis=dirs[1], useRRForIntensity=int(rri), zetaMin=0.3)
        o.specify(**kw)
        out = []
        for b in range(1, nb + 1):
            rng = O.RandomNumberSequence([10, b])
            r = o.compute(rng, *O.photons_directional(rng, mu0, az, n))
            r["dropped"] = r["nBad"]
            out.append(r)
        return out


class GpuRunner:
    def run(self, d, g_moments, n, nb, mu0, az=0.0, albedo=0.0, dirs=None, rri=False, roulette=True):
        dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
        dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, g_moments)]))
        g = M.new_Integrator(dom)
        kw = dict(surfaceAlbedo=albedo, useRussianRoulette=roulette, minInverseTableSize=10001, minForwardTableSize=10001)
        if dirs:
            kw.update(intensityMus=dirs[0], intensityPhis=dirs[1], useRussianRouletteForIntensity=rri, zetaMin=0.3)
        g.specifyParameters(**kw)
        out = []
        for b in range(1, nb + 1):
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(mu0, az, n))
            r["dropped"] = r["counters"]["dropped"]
            out.append(r)
        g.finalize_Integrator()
        return out


@pytest.fixture(params=["oracle", pytest.param("gpu", marks=pytest.mark.gpu)])
def runner(request):
    if request.param == "oracle":
        from oracle import pyoracle

        pyoracle.build()
        return OracleRunner(pyoracle)
    return GpuRunner()


def _mean(rs, key):
    return np.mean([r[key].astype(np.float64) for r in rs], axis=0)


# ---- omega = 0: Beer-Lambert ----------------------------------------------------------------------------------
def test_black_cloud_vertical_sun_beer_lambert_per_column_and_layer(runner):
    nl = 16
    d = cases.step_cloud(ssa=0.0, nlayers=nl)
    n, nb = 400_000, 2
    rs = runner.run(d, 64, n, nb, mu0=1.0)
    N = n * nb
    ncol = 32
    tau = d["ext"][:, 0, :].astype(np.float64).sum(0) * (250.0 / nl)          # 2 (columns 1-16) and 18 (17-32)
    T = np.exp(-tau)
    down, absd, up = _mean(rs, "fluxDown")[0], _mean(rs, "fluxAbsorbed")[0], _mean(rs, "fluxUp")[0]
    assert np.all(up == 0.0)                                                    # nothing survives its first interaction
    # counts per column are binomial(N, p / ncol); the fields are counts / (N / ncol)
    for got, p in ((down, T), (absd, 1.0 - T)):
        sigma = np.sqrt(ncol * p * (1.0 - p / ncol) / N)
        assert np.all(np.abs(got - p) <= SIGMAS * sigma + 1e-6), (got, p, sigma)
    # every photon is absorbed or reaches the surface (the vertical beam never meets the tracer's drop)
    drops = sum(r["dropped"] for r in rs)
    assert abs((down + absd).mean() - (1.0 - drops / N)) < 2e-6 and drops <= 2e-5 * N
    # absorption per layer (volumeAbsorption is per unit depth, :378-381): column-mean of exp(-tau_above) - exp(-tau_above - dtau)
    vol = _mean(rs, "volumeAbsorption")[:, 0, :] * (250.0 / nl)                # [layer, column]; layer 0 is the lowest
    dt = tau / nl
    above = dt[None, :] * (nl - 1 - np.arange(nl))[:, None]
    want = np.exp(-above) - np.exp(-above - dt[None, :])
    sigma = np.sqrt(ncol * want * (1.0 - want / ncol) / N)
    one_count = ncol / N                                                        # deep layers expect < 1 photon: Poisson tail, 4 counts of slack
    assert np.all(np.abs(vol - want) <= SIGMAS * sigma + 4 * one_count), np.max(np.abs(vol - want) / (sigma + one_count))
    assert abs(vol.sum(0).mean() - absd.mean()) < 1e-5


def test_black_slab_slant_sun_beer_lambert(runner):
    nl, tau_slab, mu0 = 4, 1.0, 0.5
    d = cases.plane_parallel(optical_depth=tau_slab, ssa=0.0, nx=3, ny=2, nlayers=nl)
    n, nb = 300_000, 2
    rs = runner.run(d, 64, n, nb, mu0=mu0, az=30.0)
    N = n * nb
    T = np.exp(-tau_slab / mu0)
    down, absd = _mean(rs, "fluxDown").mean(), _mean(rs, "fluxAbsorbed").mean()
    s = np.sqrt(T * (1 - T) / N)
    assert abs(down - T) <= SIGMAS * s and abs(absd - (1 - T)) <= SIGMAS * s, (down, absd, T)
    prof = _mean(rs, "volumeAbsorption").mean(axis=(1, 2)) * (250.0 / nl)       # absorbed fraction per layer, lowest first
    k = nl - 1 - np.arange(nl)                                                  # layers above
    want = np.exp(-k * tau_slab / nl / mu0) - np.exp(-(k + 1) * tau_slab / nl / mu0)
    assert np.all(np.abs(prof - want) <= SIGMAS * np.sqrt(want * (1 - want) / N) + 4.0 / N), (prof, want)


# ---- empty domain over a Lambertian surface --------------------------------------------------------------------
@pytest.mark.parametrize("albedo,rri", [(0.3, False), (1.0, False), (0.3, True)])
def test_empty_domain_lambertian_surface(runner, albedo, rri):
    d = cases.plane_parallel(optical_depth=0.0, nx=4, ny=3, nlayers=3)
    mus, phis = [1.0, 0.5, 0.2, -0.4], [0.0, 70.0, 200.0, 10.0]
    n, nb = 100_000, 2
    rs = runner.run(d, 64, n, nb, mu0=0.6, az=15.0, albedo=albedo, dirs=(mus, phis), rri=rri)
    a = float(f32(albedo))
    # (the reference -- and the oracle -- add 1e5 weights per column in float32: 1e-4 relative is its rounding, not noise)
    tol = 1e-4
    assert abs(_mean(rs, "fluxDown").mean() - 1.0) < tol                        # every photon arrives, once
    assert abs(_mean(rs, "fluxUp").mean() - a) < tol * max(a, 1.0)              # ... and leaves with weight albedo
    assert abs(_mean(rs, "fluxAbsorbed").mean()) == 0.0
    inten = _mean(rs, "intensity").mean(axis=(1, 2))
    # the local estimate at the surface is weight / pi along every direction that leaves through the top (:1479, :1517-1535);
    # a downward direction never does: with the roulette only an exit through the top counts (quirk Q8), without it the
    # ray ends at the surface it starts from with optical path 0 and counts as well
    want_up = a / np.pi
    assert np.all(np.abs(inten[:3] - want_up) < tol * want_up), (inten, want_up)
    assert abs(inten[3] - (0.0 if rri else want_up)) < tol * want_up, inten


# ---- first-order scattering ----------------------------------------------------------------------------------
def _hg_legendre_phase(cos_theta, g=0.85, n=64):
    """P(Theta) = sum_l (2 l + 1) chi_l P_l(cos Theta), chi_l = g**l (float32, as the table builder stores them), l = 0..n."""
    chi = np.concatenate([[1.0], cases.hg_coefficients(g, n).astype(np.float64)])
    pl = np.polynomial.legendre.legval(cos_theta, chi * (2 * np.arange(n + 1) + 1))
    return pl


def test_first_order_scattering_radiance_of_a_homogeneous_slab(runner):
    omega, tau, mu0 = 0.01, 0.8, 0.5
    d = cases.plane_parallel(optical_depth=tau, ssa=omega, nx=2, ny=2, nlayers=2)
    mus, phis = [1.0, 0.5, 0.3], [0.0, 90.0, 180.0]
    n, nb = 250_000, 8
    rs = runner.run(d, 64, n, nb, mu0=mu0, az=0.0, albedo=0.0, dirs=(mus, phis), rri=False, roulette=False)
    per_batch = np.array([r["intensity"].astype(np.float64).mean(axis=(1, 2)) for r in rs])
    got, se = per_batch.mean(0), per_batch.std(0, ddof=1) / np.sqrt(nb)
    s0 = np.sqrt(1 - mu0 ** 2)
    for k, (mu, phi) in enumerate(zip(mus, phis)):
        st = np.sqrt(1 - mu ** 2)
        cos_t = s0 * st * np.cos(np.radians(phi)) + (-mu0) * mu               # incoming (s0, 0, -mu0) . outgoing (st cos phi, st sin phi, mu)
        first = omega * _hg_legendre_phase(cos_t) / (4 * np.pi) * (1 - np.exp(-tau * (1 / mu0 + 1 / mu))) / (mu0 + mu)
        # higher orders add O(omega) relative to the first (bounded here by 2 omega); Monte Carlo noise 3 sigma from the batches
        assert first * (1 - 1e-3) - 3 * se[k] <= got[k] <= first * (1 + 2 * omega) + 3 * se[k], (k, got[k], first, se[k])
