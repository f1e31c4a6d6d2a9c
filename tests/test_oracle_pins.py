"""Pins of the CPU oracle against reference-run numbers recorded in SURVEY.md 6 / 8c and BASELINE.md 2
(the reference itself cannot be built in this image: it needs netCDF-Fortran) and against the canonical
MT19937 known answers.  CPU only."""
import numpy as np
import pytest

from tools import cases

f32 = np.float32


def test_mt19937_scalar_seed_kat(oracle):
    # seed=100 -> canonical genrand_real1 stream
    r = oracle.RandomNumberSequence(100).reals(5)
    want = [0.543404937, 0.671155632, 0.278369397, 0.412046403, 0.424517602]
    assert np.allclose(r, want, rtol=0, atol=5e-10 + 6e-8)
    assert [f"{v:.7f}" for v in r] == [f"{v:.7f}" for v in np.float32(want)]


def test_mt19937_vector_seed_kat(oracle):
    # seed=(/10,1/)
    g = oracle.RandomNumberSequence([10, 1])
    assert [g.int32() for _ in range(5)] == [168774779, 197189863, 1009846582, -287080014, -590412790]
    r = oracle.RandomNumberSequence([10, 1]).reals(5)
    want = np.float32([0.039295942, 0.045911841, 0.235123232, 0.933158994, 0.862533808])
    assert np.array_equal(r, want)


def test_random_real_is_unsigned_over_2p32m1(oracle):
    g = oracle.RandomNumberSequence([10, 1])
    h = oracle.RandomNumberSequence([10, 1])
    for _ in range(1000):
        i = g.int32() & 0xFFFFFFFF
        assert h.real() == float(f32(np.float64(i) / np.float64(4294967295.0)))


def test_inverse_table_hg_64_moments(oracle):
    # SURVEY.md 8c(2)
    t = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 64), 10001)
    want = {1: 3.141593, 2: 3.045351, 2501: 0.497639, 5001: 0.252000, 7501: 0.134744, 10000: 0.002211, 10001: 0.0}
    for k, v in want.items():
        assert f"{t[k - 1]:.6f}" == f"{v:.6f}", (k, t[k - 1], v)
    assert abs(np.cos(t.astype(np.float64)).mean() - 0.851692) < 2e-6
    assert np.all(np.diff(t) <= 0)


def test_forward_table_hg_64_moments(oracle):
    # P(0)=82.1977, P(1)=0.385320, P(2)=0.0731810, P(pi)=0.0456438
    n = 10001
    t = oracle.forward_table_legendre(cases.hg_coefficients(0.85, 64), n)
    d = np.pi / (n - 1)

    def at(a):
        k = int(a / d)
        w = 1 - (a - k * d) / d
        return w * t[k] + (1 - w) * t[min(k + 1, n - 1)]

    assert f"{t[0]:.4f}" == "82.1977"
    assert abs(at(0.1) - 50.8531) < 2e-3
    assert f"{at(1.0):.6f}" == "0.385320"
    assert abs(at(2.0) - 0.0731810) < 2e-7
    assert f"{t[-1]:.7f}" == "0.0456438"


def test_plane_parallel_shipped_namelist(oracle):
    # Example-Drivers/planeParallel.nml as shipped: tau=1, g=.85 (64 moments), omega=1, mu0=.5, 4 x 1e4
    # photons, seeds (/batch, iseed=10/), default 9001-entry inverse table.
    # Reference output recorded in SURVEY.md 6 / 8c: Fup 0.16420, Fdown 0.83580, std dev 0.00363.
    d = cases.plane_parallel()
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 64), 9001)
    integ = oracle.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], d["pf"], [inv])
    fu, fd = [], []
    for b in range(1, 5):
        rng = oracle.RandomNumberSequence([b, 10])
        ph = oracle.photons_directional(rng, 0.5, 0.0, 10000)
        r = integ.compute(rng, *ph)
        fu.append(r["fluxUp"][0, 0])
        fd.append(r["fluxDown"][0, 0])
    fu, fd = np.float32(fu), np.float32(fd)
    mu, md = fu.sum(dtype=np.float32) / f32(4), fd.sum(dtype=np.float32) / f32(4)
    sd = np.sqrt(((fu - mu) ** 2).sum() / 3)
    assert f"{mu:.5f}" == "0.16420"
    assert f"{md:.5f}" == "0.83580"
    assert f"{sd:.5f}" == "0.00363"


def test_step_cloud_work_counters_and_fluxes(oracle):
    # step cloud 32x1x32, mu0=1, omega=1, 1e6 photons (10 x 1e5, seeds (/10,b/)):
    # Fup 0.3254, Fdn 0.6746; 18.0 tracer calls / 63.0 cell steps / 17.0 scatterings per photon;
    # dropped fraction 3.1e-5; 99 draws per photon.   Here: 2 batches (fast) for the counters.
    d = cases.step_cloud()
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 64), 10001)
    integ = oracle.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], d["pf"], [inv])
    tot = {}
    fu, fd = [], []
    nb = 4
    for b in range(1, nb + 1):
        rng = oracle.RandomNumberSequence([10, b])
        ph = oracle.photons_directional(rng, 1.0, 0.0, 100000)
        r = integ.compute(rng, *ph)
        fu.append(r["fluxUp"].mean())
        fd.append(r["fluxDown"].mean())
        for k in ("nPhotons", "nBad", "tracerCalls", "cellSteps", "scatterings"):
            tot[k] = tot.get(k, 0) + r[k]
        tot["draws"] = tot.get("draws", 0) + rng.draws
    n = tot["nPhotons"]
    assert n == nb * 100000
    assert abs(tot["tracerCalls"] / n - 18.0) < 0.1
    assert abs(tot["cellSteps"] / n - 63.0) < 0.2
    assert abs(tot["scatterings"] / n - 17.0) < 0.1
    assert abs(tot["draws"] / n - 99) < 1
    assert 0 < tot["nBad"] / n < 1e-4
    assert abs(np.mean(fu) - 0.3254) < 3 * 4.7e-4 * np.sqrt(10 / nb)
    assert abs(np.mean(fd) - 0.6746) < 3 * 4.7e-4 * np.sqrt(10 / nb)
    # energy closure with the dropped-photon deficit (quirk Q4)
    assert abs((np.mean(fu) + np.mean(fd)) - (1 - tot["nBad"] / n)) < 2e-6


def _oracle_run(oracle, d, inv, mu0, nb, n, fwd=None, specify=None, tables=None):
    kw = {}
    integ = oracle.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], np.maximum(d["pf"], 1), [inv], *([[fwd], [fwd]] if fwd is not None else []))
    if specify:
        integ.specify(**specify)
    tot, flux = {}, {"fluxUp": [], "fluxDown": [], "fluxAbsorbed": [], "intensity": []}
    for b in range(1, nb + 1):
        rng = oracle.RandomNumberSequence([10, b])
        r = integ.compute(rng, *oracle.photons_directional(rng, mu0, 0.0, n))
        for k in ("nPhotons", "nBad", "tracerCalls", "cellSteps", "scatterings"):
            tot[k] = tot.get(k, 0) + r[k]
        for k in flux:
            if k in r:
                flux[k].append(np.asarray(r[k], np.float64).reshape(r[k].shape[0] if k == "intensity" else 1, -1).mean(1))
    per = {k: tot[k] / tot["nPhotons"] for k in tot}
    return per, {k: np.mean(v, axis=0) for k, v in flux.items() if v}


def _near(value, recorded, n, extra=0.0):
    """Recorded reference result (4 digits) against an independent sample of n photons: 3 sigma of a [0, 1] tally."""
    return abs(value - recorded) < 3 * np.sqrt(max(recorded * (1 - min(recorded, 1.0)), 1e-3) / n) * np.sqrt(2) + 5e-5 + extra


# the reference's own code run at survey time on the I3RC cases -- tracer calls / cell steps / scatterings
# per photon and domain-mean fluxes.  The work counters are properties of the algorithm (how the tracer counts
# iterations, when it stops, what a scattering is): a restatement that walks a different path does not reproduce them
# to three digits; the fluxes pin the physics of each variant.
def test_step_cloud_slant_sun_and_absorbing_variants(oracle):
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 64), 10001)
    per, flux = _oracle_run(oracle, cases.step_cloud(), inv, 0.5, 3, 100000)
    assert abs(per["tracerCalls"] - 25.1) < 0.2 and abs(per["cellSteps"] - 77.5) < 0.5 and abs(per["scatterings"] - 24.1) < 0.2
    assert _near(flux["fluxUp"][0], 0.5793, 3e5) and _near(flux["fluxDown"][0], 0.4206, 3e5)
    per, flux = _oracle_run(oracle, cases.step_cloud(ssa=0.99), inv, 1.0, 3, 100000)
    assert abs(per["tracerCalls"] - 17.7) < 0.2 and abs(per["cellSteps"] - 62.0) < 0.5 and abs(per["scatterings"] - 16.7) < 0.2
    assert _near(flux["fluxUp"][0], 0.2587, 3e5) and _near(flux["fluxDown"][0], 0.6004, 3e5)
    assert _near(flux["fluxAbsorbed"][0], 0.1408, 3e5)


def test_radar_cloud_recorded_results(oracle):
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 299), 10001)
    d = cases.radar_cloud()
    per, flux = _oracle_run(oracle, d, inv, 1.0, 2, 50000)
    assert abs(per["tracerCalls"] - 45.3) < 0.5 and abs(per["cellSteps"] - 119.0) < 1.2 and abs(per["scatterings"] - 44.3) < 0.5
    assert _near(flux["fluxUp"][0], 0.5579, 1e5, 1e-3) and _near(flux["fluxDown"][0], 0.4414, 1e5, 1e-3)
    assert 4e-4 < per["nBad"] < 1.1e-3                                  # 7.0e-4 on the radar field
    # + nadir radiance with Iwabuchi roulette (zeta_min 0.3): 92.6 tracer calls / 222.8 cell steps, I(mu=1) = 0.1821
    fwd = oracle.forward_table_legendre(cases.hg_coefficients(0.85, 299), 10001)
    per, flux = _oracle_run(oracle, d, inv, 1.0, 2, 40000, fwd=fwd,
                            specify=dict(intensityMus=[1.0], intensityPhis=[0.0], useRRForIntensity=1, zetaMin=0.3))
    assert abs(per["tracerCalls"] - 92.6) < 1.0 and abs(per["cellSteps"] - 222.8) < 2.5 and abs(per["scatterings"] - 44.4) < 0.5
    assert abs(flux["intensity"][0] - 0.1821) < 0.004


def test_landsat_scene_recorded_results(oracle):
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 299), 10001)
    d = cases.landsat_cloud()
    per, flux = _oracle_run(oracle, d, inv, 1.0, 2, 40000)
    assert abs(per["tracerCalls"] - 16.7) < 0.2 and abs(per["cellSteps"] - 241.2) < 2.5 and abs(per["scatterings"] - 15.7) < 0.2
    assert _near(flux["fluxUp"][0], 0.3045, 8e4, 1e-3) and _near(flux["fluxDown"][0], 0.6953, 8e4, 1e-3)
    assert 5e-5 < per["nBad"] < 3e-4                                    # 1.5e-4
    per, flux = _oracle_run(oracle, d, inv, 0.5, 2, 40000)
    assert abs(per["tracerCalls"] - 21.8) < 0.3 and abs(per["cellSteps"] - 343.5) < 3.5 and abs(per["scatterings"] - 20.8) < 0.3
    assert _near(flux["fluxUp"][0], 0.5170, 8e4, 1e-3) and _near(flux["fluxDown"][0], 0.4828, 8e4, 1e-3)
    # 7 radiance directions + Lambertian 0.2 through a surface object: 214 tracer calls / 3194 cell steps / 23.1 scatterings
    fwd = oracle.forward_table_legendre(cases.hg_coefficients(0.85, 299), 10001)
    huge = np.finfo(np.float32).max
    per, _ = _oracle_run(oracle, d, inv, 0.5, 1, 16000, fwd=fwd, specify=dict(
        intensityMus=[1, .5, .5, .8, .8, .3, .3], intensityPhis=[0, 0, 180, 90, 270, 45, 225], useRRForIntensity=1, zetaMin=0.3,
        surfaceBDRF=(np.array([0.0, huge], np.float32), np.array([0.0, huge], np.float32), np.array([[0.2]], np.float32))))
    # (cell steps per photon scatter with a standard deviation of ~2000: +-16 at this sample size, +-10 in the recorded run;
    # the GPU's 2e7-photon mean is 3205)
    assert abs(per["tracerCalls"] - 214) < 4 and abs(per["cellSteps"] - 3194) < 60 and abs(per["scatterings"] - 23.1) < 0.5


def test_photon_sources_have_the_distributions_the_reference_documents(oracle):
    """Code/monteCarloIllumination.f95:106-424 restated in oracle/illumination.c: what each source promises.
    Flux: mu = -sqrt(r) makes the flux on the horizontal equally weighted in mu (density 2 |mu|: mean |mu| = 2/3,
    mean mu^2 = 1/2); azimuths uniform in [0, 2 pi]; RandomAzimuth keeps mu; the internal flux detector draws the same
    cosine law upward or downward; the finite detector moves x by delta (1 - r / 2), i.e. into [x + delta / 2, x + delta]
    (the reference's formula, not a centred detector); Internal_Intensity stores its azimuth in degrees."""
    O = oracle
    n = 200000
    x, y, z, mu, phi = O.photons_flux(O.RandomNumberSequence([3, 4]), n)
    assert np.all(z == np.float32(1.0) - np.spacing(np.float32(1.0)))
    assert abs(np.abs(mu).mean() - 2 / 3) < 4 * np.sqrt(1 / 18 / n) and abs((mu.astype(np.float64) ** 2).mean() - 0.5) < 4 * np.sqrt(1 / 12 / n)
    assert mu.max() <= 0.0 and 0.0 <= phi.min() and phi.max() <= np.float32(2 * np.pi) * (1 + 1e-6)
    assert abs(phi.mean() - np.pi) < 4 * 2 * np.pi / np.sqrt(12 * n) and abs(x.mean() - 0.5) < 4 / np.sqrt(12 * n)
    x, y, z, mu, phi = O.photons_random_azimuth(O.RandomNumberSequence([3, 4]), 0.3, n)
    assert np.all(mu == np.float32(-0.3)) and abs(phi.mean() - np.pi) < 4 * 2 * np.pi / np.sqrt(12 * n)
    x, y, z, mu, phi = O.photons_spotlight(-0.3, 90.0, 0.2, 0.9, 10)
    assert np.all(x == np.float32(0.2)) and np.all(y == np.float32(0.9)) and np.all(mu == np.float32(-0.3))
    assert np.all(phi == np.float32(90.0) * np.arccos(np.float32(-1.0)) / np.float32(180.0))
    for up in (True, False):
        x, y, z, mu, phi = O.photons_internal_flux(O.RandomNumberSequence([3, 4]), 0.5, 0.5, 0.25, up, n, delta_x=0.2)
        assert np.all(mu > 0) if up else np.all(mu < 0)
        assert abs(np.abs(mu).mean() - 2 / 3) < 4 * np.sqrt(1 / 18 / n)
        assert np.all(z == np.float32(0.25)) and np.all(y == np.float32(0.5))
        assert x.min() >= np.float32(0.5) + np.float32(0.1) * (1 - 1e-6) and x.max() <= np.float32(0.7) * (1 + 1e-6)
        assert abs(x.mean() - 0.65) < 4 * 0.1 / np.sqrt(12 * n)
    x, y, z, mu, phi = O.photons_internal_intensity(O.RandomNumberSequence([3, 4]), 0.5, 0.5, 1.0, -0.2, 270.0, 5)
    assert np.all(phi == np.float32(270.0)) and np.all(mu == np.float32(-0.2))          # degrees, as the reference stores it
    assert np.all(z == np.float32(1.0) - np.spacing(np.float32(1.0)))                   # a downward detector at the top is nudged inside


def test_cpu_baseline_tool_reports_work_counters_and_saves_batches(tmp_path):
    """tools/cpu_baseline.py is bench.py's cpu_baseline leg and the oracle child of the big GPU parity tests: its JSON
    line carries the reference algorithm's work per photon (what SURVEY.md 8(d)'s byte formula is evaluated with) and
    --save writes the per-batch results those tests read."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "batches.npz")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "cpu_baseline.py"), "--config", "radar64_nadir", "--cores", "2",
                        "--batches-per-core", "2", "--photons", "500", "--save", out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["kind"] == "port" and j["cores"] == 2 and j["value"] > 0
    w = j["oracle_per_photon"]
    assert 100 < w["S"] < 400 and 20 < w["K"] < 80 and 0.9 < w["E"] <= 1.0 + 1e-6    # radar field + nadir shadow rays
    z = np.load(out)
    assert z["means"].shape[0] == 4 and z["intensityMeans"].shape[0] == 4 and int(z["photonsPerBatch"]) == 500
    assert float(z["cellSteps"].sum()) / 2000 == pytest.approx(w["S"])


def test_many_components_share_one_draw_each(oracle):
    """Component selection (:637-638) with more components than the restatement's former fixed buffer of 64 took (the product
    takes 255): a domain with 99 empty components behind its cloud is the one-component domain -- the same deviates are drawn
    in the same order and every scattering picks component 1 -- so every counter and tally agrees to the last bit."""
    d = cases.step_cloud(ssa=0.9, nlayers=8)
    inv = oracle.inverse_table_legendre(cases.hg_coefficients(0.85, 64), 10001)
    nc = 100
    zero = np.zeros_like(d["ext"])
    one = oracle.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], np.maximum(d["pf"], 1), [inv])
    many = oracle.Integrator(d["xe"], d["ye"], d["ze"], np.stack([d["ext"]] + [zero] * (nc - 1)), np.stack([d["ssa"]] + [zero] * (nc - 1)),
                             np.stack([np.maximum(d["pf"], 1)] * nc), [inv] * nc)
    res = []
    for integ in (one, many):
        rng = oracle.RandomNumberSequence([10, 3])
        res.append(integ.compute(rng, *oracle.photons_directional(rng, 1.0, 0.0, 20000)))
    a, b = res
    assert a["scatterings"] == b["scatterings"] > 1e5 and a["cellSteps"] == b["cellSteps"] and a["nBad"] == b["nBad"]
    for k in ("fluxUp", "fluxDown", "fluxAbsorbed"):
        assert np.array_equal(a[k], b[k]), k
