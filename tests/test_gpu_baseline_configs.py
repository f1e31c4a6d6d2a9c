"""BASELINE.json configs[2..4] at test size (configs[1], the step cloud, is test_gpu_parity.py; the reference-exact
radar 640 x 1 x 54 and Landsat 128 x 128 x 119 fields are test_gpu_features.py): the labelled synthetic shapes of
SURVEY.md 8d, HIP path against the oracle on the same seeds / batch scheme, domain means within
3 sqrt(se_gpu^2 + se_ref^2) (standard errors from the batch-to-batch variance, monteCarloDriver.f95:358-378)."""
import numpy as np
import pytest

import i3rc_monte_carlo_model_amd as M
from tools import cases
from tests.test_gpu_parity import _batches_gpu, _batches_oracle, hg_table, make_gpu, make_oracle

pytestmark = pytest.mark.gpu
DIRS7_MU = [1.0, 0.5, 0.5, 0.8, 0.8, 0.3, 0.3]
DIRS7_PHI = [0.0, 0.0, 180.0, 90.0, 270.0, 45.0, 225.0]


def _means_agree(gr, orr, keys):
    nb = len(gr)
    for key in keys:
        dg = np.array([r[key].mean(dtype=np.float64) for r in gr])
        dr = np.array([r[key].mean(dtype=np.float64) for r in orr])
        tol = 3.0 * np.sqrt(dg.var(ddof=1) / nb + dr.var(ddof=1) / nb) + 1e-6
        assert abs(dg.mean() - dr.mean()) <= tol, (key, dg.mean(), dr.mean(), tol)


def _two_stage(oracle, g, o, nb, n, mu0, keys, per_direction=False):
    """As test_gpu_parity._parity: a first failure is re-examined once on an independent sample twice as large."""
    def run(nbatches, iseed):
        gr, orr = _batches_gpu(g, nbatches, n, mu0, iseed=iseed), _batches_oracle(oracle, o, nbatches, n, mu0, iseed=iseed)
        _means_agree(gr, orr, keys)
        if per_direction:   # every radiance direction on its own, not only their mean
            nd = gr[0]["intensity"].shape[0]
            for k in range(nd):
                dg = np.array([r["intensity"][k].mean(dtype=np.float64) for r in gr])
                dr = np.array([r["intensity"][k].mean(dtype=np.float64) for r in orr])
                tol = 3.0 * np.sqrt(dg.var(ddof=1) / nbatches + dr.var(ddof=1) / nbatches) + 1e-6
                assert abs(dg.mean() - dr.mean()) <= tol, ("intensity", k, dg.mean(), dr.mean(), tol)
        return gr, orr
    try:
        return run(nb, 10)
    except AssertionError as first:
        import inspect

        from tests.conftest import record_stage1_miss
        caller = next((f.function for f in inspect.stack()[1:] if f.function.startswith("test_")), "?")
        record_stage1_miss(caller, first.args)
        try:
            return run(2 * nb, 11)
        except AssertionError as second:
            raise AssertionError(f"failed twice: {first.args} then {second.args}")


def test_config2_radar_64x64x54_flux_and_nadir_radiance(oracle):
    d = cases.radar_cloud_64()
    tab = hg_table(0.85, 299)
    inv, fwd = tab.inverse_table(10001), tab.forward_table(10001)
    g = make_gpu(d, tab, intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
    g.set_tables(1, inverse=inv, forward=fwd, forward_orig=fwd)
    o = make_oracle(oracle, d, [inv], [fwd], [fwd])
    o.specify(intensityMus=[1.0], intensityPhis=[0.0], useRRForIntensity=1, zetaMin=0.3)
    gr, orr = _two_stage(oracle, g, o, 8, 25000, 1.0, ("fluxUp", "fluxDown", "intensity"))
    # dropped-photon deficit (quirk Q4) is part of the result: same rate on both sides
    n = 25000 * len(gr)
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n, sum(r["nBad"] for r in orr) / n
    assert abs(dg - do) < 3 * np.sqrt((do + 1e-5) / n) + 2e-4
    # the nadir radiance field has the field's structure: brighter over the thick columns
    tau = d["ext"].sum(0) * 45.0
    inten = np.stack([r["intensity"][0] for r in gr]).mean(0)
    assert inten[tau > np.median(tau)].mean() > inten[tau <= np.median(tau)].mean()


def test_config3_landsat_128x128x36_flux(oracle):
    d = cases.landsat_cloud(nlayers=36)
    tab = hg_table(0.85, 299)
    inv = tab.inverse_table(10001)
    g = make_gpu(d, tab)
    g.set_tables(1, inverse=inv)
    o = make_oracle(oracle, d, [inv])
    gr, orr = _two_stage(oracle, g, o, 4, 40000, 1.0, ("fluxUp", "fluxDown"))
    r0 = gr[0]   # black surface, no absorption: every photon is tallied once or dropped by the tracer
    closure = r0["fluxUp"].mean(dtype=np.float64) + r0["fluxDown"].mean(dtype=np.float64)
    assert abs(closure - (1 - r0["counters"]["dropped"] / 40000)) < 2e-6
    # sharding the batch over 8 ranks by photon range (multigpu.shard_photons) and summing the raw tallies is the
    # same computation: BASELINE.json runs this case on 8 GPUs
    from i3rc_monte_carlo_model_amd import multigpu
    raw = np.zeros_like(r0["raw"])
    for rank in range(8):
        first, count = multigpu.shard_photons(40000, 8, rank)
        g.launch(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, count), firstPhoton=first)
        raw += g.fetch()
    lay = g.layout()
    assert np.array_equal(raw[lay.counters:lay.counters + 10], r0["raw"][lay.counters:lay.counters + 10])
    assert np.allclose(raw, r0["raw"], rtol=1e-9, atol=1e-9)


def test_config4_landsat_seven_radiances_lambertian_surface_object(oracle):
    d = cases.landsat_cloud()
    tab = hg_table(0.85, 299)
    inv, fwd = tab.inverse_table(10001), tab.forward_table(10001)
    g = make_gpu(d, tab, intensityMus=DIRS7_MU, intensityPhis=DIRS7_PHI, useRussianRouletteForIntensity=True, zetaMin=0.3,
                 surfaceBDRF=M.new_SurfaceDescription([0.2]))
    g.set_tables(1, inverse=inv, forward=fwd, forward_orig=fwd)
    o = make_oracle(oracle, d, [inv], [fwd], [fwd])
    o.specify(intensityMus=DIRS7_MU, intensityPhis=DIRS7_PHI, useRRForIntensity=1, zetaMin=0.3,
              surfaceBDRF=(np.array([0.0, np.finfo(np.float32).max], np.float32), np.array([0.0, np.finfo(np.float32).max], np.float32),
                           np.array([[0.2]], np.float32)))
    gr, orr = _two_stage(oracle, g, o, 8, 10000, 0.5, ("fluxUp", "fluxDown", "intensity"), per_direction=True)
    # ~3200 cell steps per photon in the reference, almost all of them shadow rays.  Rays whose roulette is
    # lost before the trace (small contributions, :1554) are not traced here: fewer steps, the same radiances (above)
    sh = sum(r["counters"]["shadowSteps"] for r in gr) / (10000 * len(gr))
    skipped = sum(r["counters"]["raysSkipped"] for r in gr) / (10000 * len(gr))
    calls = sum(r["counters"]["tracerCalls"] for r in gr) / (10000 * len(gr))
    sc = sum(r["counters"]["scatterings"] + r["counters"]["surfaceHits"] for r in gr) / (10000 * len(gr))
    assert 300 < sh < 3300 and skipped > 10
    assert skipped < 7 * sc * 1.001     # at most every ray of every event


# ---- configs 2 and 4 at 1e6+ photons: the oracle's sample is produced by a child program on the host cores while the
#      GPU traces its own (as the 1e8 step-cloud test does) -------------------------------------------------------------
def _oracle_child(tmp_path, config, cores, per_core, photons, columns=False):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"oracle_{config}.npz")
    child = subprocess.Popen([sys.executable, os.path.join(root, "tools", "cpu_baseline.py"), "--config", config, "--cores", str(cores),
                              "--batches-per-core", str(per_core), "--photons", str(photons), "--save", out] + (["--save-columns"] if columns else []),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return child, out


def _gpu_batches(config, nb, n):
    from tools import workloads as W

    name, w = W.get(config)
    g, d = W.make_integrator(w)
    rs = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((77, b)), M.new_PhotonStream(w["mu0"], 0.0, n)) for b in range(1, nb + 1)]
    g.finalize_Integrator()
    return rs


def _assert_means_3sigma(gpu_vals, ref_vals, what):
    gpu_vals, ref_vals = np.asarray(gpu_vals, np.float64), np.asarray(ref_vals, np.float64)
    tol = 3.0 * np.sqrt(gpu_vals.var(ddof=1) / len(gpu_vals) + ref_vals.var(ddof=1) / len(ref_vals)) + 1e-7
    assert abs(gpu_vals.mean() - ref_vals.mean()) <= tol, (what, gpu_vals.mean(), ref_vals.mean(), tol)
    return abs(gpu_vals.mean() - ref_vals.mean()) / tol


def test_config2_radar_64_nadir_radiance_at_2e6_photons(tmp_path):
    import os

    cores = min(16, len(os.sched_getaffinity(0)))
    per_core, n_ref = 4, 2_000_000 // (cores * 4)
    child, out = _oracle_child(tmp_path, "radar64_nadir", cores, per_core, n_ref)
    gr = _gpu_batches("radar64_nadir", 40, 50_000)                        # 2e6 photons on the GPU meanwhile
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, so + se
    z = np.load(out)
    assert len(z["means"]) == cores * per_core
    for k, key in enumerate(("fluxUp", "fluxDown")):
        _assert_means_3sigma([r[key].mean(dtype=np.float64) for r in gr], z["means"][:, k], key)
    _assert_means_3sigma([r["intensity"][0].mean(dtype=np.float64) for r in gr], z["intensityMeans"][:, 0], "nadir radiance")
    # per column: the same 3-sigma statistic over the 4096 columns (Student-t allowance as in _assert_3sigma)
    from tests.test_gpu_parity import _assert_3sigma
    orr = [dict(fluxUp=u, fluxDown=dn, intensity=i) for u, dn, i in zip(z["fluxUp"], z["fluxDown"], z["intensity"])]
    for key in ("fluxUp", "fluxDown", "intensity"):
        _assert_3sigma(gr, orr, key, floor=1e-7)
    # dropped-photon deficit (quirk Q4) and work per photon: same on both sides
    n_g, n_o = 50_000 * len(gr), n_ref * len(z["nBad"])
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n_g, z["nBad"].sum() / n_o
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / n_o) + 1e-5, (dg, do)
    # work per photon: the same scatterings; the oracle counts photon and shadow-ray steps in one figure, of which the HIP
    # path leaves out the rays whose roulette is lost before the trace
    kg = sum(r["counters"]["scatterings"] for r in gr) / n_g
    assert abs(kg - z["scatterings"].sum() / n_o) < 0.01 * kg
    sg = sum(r["counters"]["cellSteps"] + r["counters"]["shadowSteps"] for r in gr) / n_g
    assert sum(r["counters"]["cellSteps"] for r in gr) / n_g < sg <= z["cellSteps"].sum() / n_o * 1.01
    assert sum(r["counters"]["raysSkipped"] for r in gr) > 0


def test_config4_landsat_seven_radiances_at_1e6_photons(tmp_path):
    import os

    cores = min(16, len(os.sched_getaffinity(0)))
    per_core, n_ref = 4, 2_000_000 // (cores * 4)
    child, out = _oracle_child(tmp_path, "landsat119_7dir", cores, per_core, n_ref, columns=True)
    gr = _gpu_batches("landsat119_7dir", 40, 200_000)                     # 8e6 photons on the GPU meanwhile
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, so + se
    z = np.load(out)
    for k, key in enumerate(("fluxUp", "fluxDown")):
        _assert_means_3sigma([r[key].mean(dtype=np.float64) for r in gr], z["means"][:, k], key)
    for d in range(7):                                                      # every direction on its own
        _assert_means_3sigma([r["intensity"][d].mean(dtype=np.float64) for r in gr], z["intensityMeans"][:, d], f"radiance {d}")
    # per column: the 3-sigma statistic of config 2 over the 7 x 16384 radiance columns and the flux fields
    from tests.test_gpu_parity import _assert_3sigma
    # The fields, column by column.  Config 2's per-cell statistic (_assert_3sigma: excursions beyond 3 sigma counted against
    # Student's t) assumes Gaussian batch means; here a column sees two photons per oracle batch and a local-estimate field is
    # made of a few large contributions among many small ones: on numbers that agree the count of 3-sigma excursions is 2.5
    # times what t allows.  So: (1) that statistic on BLOCKS of 4 x 4 columns (8 x 8 for the fluxes), batches pooled four /
    # eight to a group, where the means are Gaussian; (2) per column, statistics that do not lean on the tails -- the mean
    # of z^2 (1.14 for t with 16 degrees of freedom), and no structure in the difference field: it must not correlate with
    # the field's gradient along x or y (a radiance tallied one column off would).
    def pooled(rs, k, block):
        def f(a, key):
            a = a.astype(np.float64)
            b = block if key == "intensity" else 2 * block
            return a if block == 1 else a.reshape(a.shape[:-2] + (128 // b, b, 128 // b, b)).mean(axis=(-3, -1))
        return [{key: np.mean([f(r[key], key) for r in rs[i:i + k]], axis=0) for key in ("fluxUp", "fluxDown", "intensity")}
                for i in range(0, len(rs) - k + 1, k)]
    orr = [dict(fluxUp=u, fluxDown=dn, intensity=i) for u, dn, i in zip(z["fluxUp"], z["fluxDown"], z["intensity"])]
    gp, op = pooled(gr, 4, 4), pooled(orr, 8, 4)
    assert len(gp) == 10 and len(op) == cores * per_core // 8 and gp[0]["intensity"].shape == (7, 32, 32) and gp[0]["fluxUp"].shape == (16, 16)
    for key in ("intensity", "fluxUp", "fluxDown"):
        _assert_3sigma(gp, op, key, floor=1e-7)
    from tests.test_gpu_parity import _mean_se
    g1, o1 = pooled(gr, 4, 1), pooled(orr, 8, 1)
    mg, sg = _mean_se(g1, "intensity")
    mo, so_ = _mean_se(o1, "intensity")
    zz = (mg - mo) / np.sqrt(sg ** 2 + so_ ** 2 + 1e-14)
    assert 0.8 < np.mean(zz ** 2) < 1.6, np.mean(zz ** 2)
    for k in range(7):
        diff, field = (mg[k] - mo[k]), 0.5 * (mg[k] + mo[k])
        lim = 6.0 / np.sqrt(diff.size)
        # (the gradient at a column is made of its NEIGHBOURS: independent of the column's own noise, which the field itself is not)
        for what, other in (("x gradient", np.roll(field, 1, axis=1) - np.roll(field, -1, axis=1)),
                            ("y gradient", np.roll(field, 1, axis=0) - np.roll(field, -1, axis=0))):
            c = np.corrcoef(diff.ravel(), other.ravel())[0, 1]
            assert abs(c) < lim, (k, what, c, lim)
    n_g, n_o = 200_000 * len(gr), n_ref * len(z["nBad"])
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n_g, z["nBad"].sum() / n_o
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / n_o) + 1e-5, (dg, do)


def test_config3_landsat36_column_by_column_with_absorption(tmp_path):
    """Config 3 (Landsat 128 x 128 x 36) COLUMN BY COLUMN, with omega = 0.99 in the cloudy cells -- so that the value the cells with
    extinction share (passed in the kernel arguments: i3rc_hip_create), the absorption tallies and the table-in-LDS kernel of
    1024-thread workgroups are all in it: 4e6 photons on the GPU against 4e6 of the oracle (a child process on the host cores
    meanwhile), the per-column 3-sigma statistic (_assert_3sigma: Student's t allowance) on fluxUp, fluxDown and fluxAbsorbed
    of the 16 384 columns, and the absorbed profile layer by layer."""
    import os

    from tests.test_gpu_parity import _assert_3sigma
    cores = min(16, len(os.sched_getaffinity(0)))
    per_core, n_ref = 2, 4_000_000 // (cores * 2)
    child, out = _oracle_child(tmp_path, "landsat36_absorbing", cores, per_core, n_ref, columns=True)
    from tools import workloads as W

    name, w = W.get("landsat36_absorbing")
    g, d = W.make_integrator(w)
    gr = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((77, b)), M.new_PhotonStream(1.0, 0.0, 125_000)) for b in range(1, 33)]
    assert "table in LDS" in g.kernel_name() and "GRID_COLUMNS" in g.kernel_name(), g.kernel_name()   # (the scene has column records)
    g.finalize_Integrator()
    for r in gr:
        r["absorbedProfile"] = r["volumeAbsorption"].reshape(36, -1).mean(axis=1, dtype=np.float64)
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, so + se
    z = np.load(out)
    assert len(z["means"]) == cores * per_core and z["fluxAbsorbed"].shape[1:] == (128, 128)
    orr = [dict(fluxUp=u, fluxDown=dn, fluxAbsorbed=a, absorbedProfile=p) for u, dn, a, p in zip(z["fluxUp"], z["fluxDown"], z["fluxAbsorbed"], z["absorbedProfile"])]
    for key in ("fluxUp", "fluxDown", "fluxAbsorbed", "absorbedProfile"):
        _assert_3sigma(gr, orr, key, floor=1e-7)
    # energy: what is not dropped is reflected, transmitted or absorbed (black surface) -- in expectation only: below a weight of
    # 0.5 a photon plays Russian roulette (:673-680), which conserves energy on average, not photon by photon
    n_g = 125_000 * len(gr)
    tot = np.mean([r["fluxUp"].mean(dtype=np.float64) + r["fluxDown"].mean(dtype=np.float64) + r["fluxAbsorbed"].mean(dtype=np.float64) for r in gr])
    assert abs(tot - (1 - sum(r["counters"]["dropped"] for r in gr) / n_g)) < 3e-4
    assert 0.02 < np.mean([r["fluxAbsorbed"].mean() for r in gr]) < 0.3
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n_g, z["nBad"].sum() / (n_ref * len(z["nBad"]))
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / (n_ref * len(z["nBad"]))) + 1e-5, (dg, do)


def test_two_components_column_by_column_with_absorption(tmp_path):
    """A domain of two components COLUMN BY COLUMN against the oracle: the Landsat 128 x 128 x 36 scene (omega = 0.99) plus a
    horizontally uniform gas (omega = 0.9) -- two components, two phase-function tables, every cell optically active, omega and the
    table entry read per cell and component (:637-649, :684-686) -- 4e6 photons a side: the per-column 3-sigma statistic on fluxUp,
    fluxDown, fluxAbsorbed of the 16 384 columns and the absorbed profile layer by layer (where the gas absorbs above the clouds)."""
    import os

    from tests.test_gpu_parity import _assert_3sigma
    cores = min(16, len(os.sched_getaffinity(0)))
    per_core, n_ref = 2, 4_000_000 // (cores * 2)
    child, out = _oracle_child(tmp_path, "landsat36_gas_absorbing", cores, per_core, n_ref, columns=True)
    from tools import workloads as W

    name, w = W.get("landsat36_gas_absorbing")
    g, d = W.make_integrator(w)
    gr = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((77, b)), M.new_PhotonStream(1.0, 0.0, 125_000)) for b in range(1, 33)]
    assert "false, true" in g.kernel_name(), g.kernel_name()   # (flux: the general flux kernel; the several-components kernels are radiance kernels)
    g.finalize_Integrator()
    for r in gr:
        r["absorbedProfile"] = r["volumeAbsorption"].reshape(36, -1).mean(axis=1, dtype=np.float64)
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, so + se
    z = np.load(out)
    assert len(z["means"]) == cores * per_core and z["fluxAbsorbed"].shape[1:] == (128, 128)
    orr = [dict(fluxUp=u, fluxDown=dn, fluxAbsorbed=a, absorbedProfile=p) for u, dn, a, p in zip(z["fluxUp"], z["fluxDown"], z["fluxAbsorbed"], z["absorbedProfile"])]
    for key in ("fluxUp", "fluxDown", "fluxAbsorbed", "absorbedProfile"):
        _assert_3sigma(gr, orr, key, floor=1e-7)
    # the top layers hold gas only: what is absorbed there is the gas's (omega = 0.9), and it is not nothing
    top = np.mean([r["absorbedProfile"][-3:].sum() for r in gr])
    assert top > 0
    n_g = 125_000 * len(gr)
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n_g, z["nBad"].sum() / (n_ref * len(z["nBad"]))
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / (n_ref * len(z["nBad"]))) + 1e-5, (dg, do)
    kg = sum(r["counters"]["scatterings"] for r in gr) / n_g
    assert abs(kg - z["scatterings"].sum() / (n_ref * len(z["nBad"]))) < 0.01 * kg


def test_tool_chain_les_stratocumulus_with_rayleigh_against_the_oracle(oracle, tmp_path):
    """A production-style domain against the oracle: the LES stratocumulus field + Rayleigh scattering as the REFERENCE'S OWN
    TOOLS wrote it (MakeMieTable -> PhysicalPropertiesToDomain on Tools/Examples, tests/golden/make_tool_domains.py): two
    components -- cloud droplets with a Mie table of 35 effective radii (18 of them in use, up to 1381 Legendre coefficients: forward
    peaks of 3e4), a horizontally uniform gas in 18 irregular layers --, two radiance directions, a Lambertian surface, sun at 60
    degrees.  Both sides get the same tables; domain means of the fluxes and of each radiance direction within
    3 sqrt(se_gpu^2 + se_ref^2), and the kernels are the widened class's."""
    from tests.test_fortran_shell import _tool_chain_domain
    dom = M.read_Domain(_tool_chain_domain("les_stcu_rayleigh", tmp_path))
    nz, ny, nx = dom.shape

    def full(c, key, dtype):   # a component's array on the whole grid (level base, horizontally uniform components)
        a = np.zeros((nz, ny, nx), dtype)
        z0 = c["zbase"] - 1
        a[z0:z0 + c[key].shape[0]] = np.broadcast_to(c[key], (c[key].shape[0], ny, nx))
        return a
    d = dict(xe=dom.x, ye=dom.y, ze=dom.z, ext=[full(c, "ext", np.float32) for c in dom.components],
             ssa=[full(c, "ssa", np.float32) for c in dom.components], pf=[full(c, "pfi", np.int32) for c in dom.components])
    inv = [c["table"].inverse_table(10001) for c in dom.components]
    fwd = [c["table"].forward_table(10001) for c in dom.components]
    mus, phis = [1.0, 0.6], [0.0, 135.0]
    g = M.new_Integrator(dom)
    # (the droplets' forward peak is 3e4: a plain local estimate of such a phase function has a long-tailed error, which twelve batches do not
    # measure.  The reference's own remedy -- hybrid phase functions, a Gaussian peak of 7 degrees for every order of scattering,
    # :1925-1998 -- is what a user of such a table would switch on, and what both sides estimate with here.)
    hyb = [oracle.hybrid_tables(f, 7.0) for f in fwd]
    g.specifyParameters(surfaceAlbedo=0.06, minInverseTableSize=10001, intensityMus=mus, intensityPhis=phis,
                        useRussianRouletteForIntensity=True, zetaMin=0.3, useHybridPhaseFunsForIntenCalcs=True, hybridPhaseFunWidth=7.0,
                        numOrdersOrigPhaseFunIntenCalcs=0)
    for k in range(len(dom.components)):
        g.set_tables(k + 1, inverse=inv[k], forward=hyb[k], forward_orig=fwd[k])
    o = make_oracle(oracle, d, inv, hyb, fwd)
    o.specify(intensityMus=mus, intensityPhis=phis, useRRForIntensity=1, zetaMin=0.3, surfaceAlbedo=0.06, useHybrid=1, numOrdersOrig=0)
    # (at the largest sample -- 1e7 photons against 2.4e6, tests/manual/les_parity_large.py, profiles/r05_parity_large_les.txt -- the five means
    # agree within 1.4 combined standard errors of 0.1 ... 0.5 %)
    gr, orr = _two_stage(oracle, g, o, 16, 100000, 0.5, ("fluxUp", "fluxDown", "fluxAbsorbed", "intensity"), per_direction=True)
    assert "wide" in g.kernel_name(), g.kernel_name()
    # the same work on both sides: scatterings per photon (the Mie table's entries are chosen per cell, the gas by the compare chain)
    kg = sum(r["counters"]["scatterings"] for r in gr) / (len(gr) * 100000)
    ko = sum(r["scatterings"] for r in orr) / (len(orr) * 100000)
    assert abs(kg - ko) < 0.02 * ko, (kg, ko)
    g.finalize_Integrator()


def test_tool_chain_three_component_column_against_the_oracle(oracle, tmp_path):
    """The other domain of the reference's tool chain (Tools/Examples/cloudAndDust_to_domain.nml through PhysicalPropertiesToDomain,
    tests/golden/tools_mixture.dom.gz): ONE column of eleven irregular layers with THREE components -- droplets (35 Mie entries), absorbing
    dust (omega 0.87 ... 0.92, 35 entries) and molecular absorption with omega = 0 in every layer, so that a "scattering" by the third
    component ends the photon (:642-649: the weight times omega is 0).  Fluxes, the absorbed profile layer by layer and a radiance
    direction against the oracle; what comes in is reflected, transmitted or absorbed."""
    from tests.test_fortran_shell import _tool_chain_domain
    dom = M.read_Domain(_tool_chain_domain("mixture", tmp_path))
    nz, ny, nx = dom.shape
    assert (nz, ny, nx) == (11, 1, 1) and len(dom.components) == 3

    def full(c, key, dtype):
        a = np.zeros((nz, ny, nx), dtype)
        z0 = c["zbase"] - 1
        a[z0:z0 + c[key].shape[0]] = np.broadcast_to(c[key], (c[key].shape[0], ny, nx))
        return a
    d = dict(xe=dom.x, ye=dom.y, ze=dom.z, ext=[full(c, "ext", np.float32) for c in dom.components],
             ssa=[full(c, "ssa", np.float32) for c in dom.components], pf=[full(c, "pfi", np.int32) for c in dom.components])
    inv = [c["table"].inverse_table(10001) for c in dom.components]
    fwd = [c["table"].forward_table(10001) for c in dom.components]
    g = M.new_Integrator(dom)
    g.specifyParameters(surfaceAlbedo=0.1, minInverseTableSize=10001, intensityMus=[0.8], intensityPhis=[30.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
    for k in range(3):
        g.set_tables(k + 1, inverse=inv[k], forward=fwd[k], forward_orig=fwd[k])
    o = make_oracle(oracle, d, inv, fwd, fwd)
    o.specify(intensityMus=[0.8], intensityPhis=[30.0], useRRForIntensity=1, zetaMin=0.3, surfaceAlbedo=0.1)
    gr, orr = _two_stage(oracle, g, o, 12, 50000, 0.6, ("fluxUp", "fluxDown", "fluxAbsorbed", "intensity"))
    dz = np.diff(dom.z).astype(np.float64)
    pg = np.array([(r["volumeAbsorption"].astype(np.float64)[:, 0, 0] * dz) for r in gr])
    po = np.array([(r["volumeAbsorption"].astype(np.float64)[:, 0, 0] * dz) for r in orr])
    tol = 3.0 * np.sqrt(pg.var(0, ddof=1) / len(gr) + po.var(0, ddof=1) / len(orr)) + 1e-6
    assert np.all(np.abs(pg.mean(0) - po.mean(0)) <= 1.35 * tol), (pg.mean(0), po.mean(0), tol)     # (eleven layers at once: 4 sigma each)
    assert pg.mean(0).min() > 0                                                                   # every layer absorbs: the gas is in all of them
    up, dn, ab = (np.mean([float(r[k].mean()) for r in gr]) for k in ("fluxUp", "fluxDown", "fluxAbsorbed"))
    assert abs(up + ab + dn * (1 - 0.1) - 1.0) < 3e-3 and ab > 0.05
    g.finalize_Integrator()


def test_config4_column_by_column_against_the_oracles_fixture():
    """Config 4 (Landsat 128 x 128 x 119 + 7 radiance directions + the surface object) COLUMN BY COLUMN against a fixture the oracle
    wrote in the build container (tests/golden/make_config4_columns.py: 48 batches of 5e5 photons, 2.4e7 in all -- thirty photons
    per column and batch, where the batch means of a column are near enough to Gaussian for the per-column statistic that two
    photons per column were not): per column the mean and the batch standard error of fluxUp, fluxDown and the seven radiance
    fields.  Here: 32 batches of 1e6 photons on the GPU, the per-column 3-sigma statistic with its Student-t allowance
    (test_gpu_parity._assert_3sigma's rule, the reference side from its recorded mean / standard error) on all nine fields
    of 16 384 columns, the domain means of each, the dropped-photon rate and the scatterings per photon."""
    import os

    from scipy import stats
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config4_columns.npz"))
    nb_ref, n_ref = int(z["batches"]), int(z["photonsPerBatch"])
    assert str(z["config"]) == "landsat119_7dir" and z["mean"].shape == (9, 128, 128) and nb_ref >= 40 and nb_ref * n_ref >= 20_000_000
    gr = _gpu_batches("landsat119_7dir", 32, 1_000_000)
    fields = np.stack([np.concatenate([r["fluxUp"][None], r["fluxDown"][None], r["intensity"]]).astype(np.float64) for r in gr])   # [32][9][128][128]
    mg, sg = fields.mean(0), fields.std(0, ddof=1) / np.sqrt(len(fields))
    dof = len(fields) + nb_ref - 2
    p3 = 2 * stats.t.sf(3.0, dof)
    for k, name in enumerate(z["fieldNames"]):
        zz = np.abs(mg[k] - z["mean"][k]) / (np.sqrt(sg[k] ** 2 + z["stderr"][k].astype(np.float64) ** 2) + 1e-7)
        expected = p3 * zz.size
        allowed = int(np.ceil(expected + 3 * np.sqrt(expected) + 1))
        assert (zz > 3.0).sum() <= allowed, (str(name), int((zz > 3.0).sum()), allowed, zz.max())
        assert zz.max() < stats.t.isf(1e-3 / (2 * zz.size), dof), (str(name), zz.max())
        assert 0.85 < np.mean(zz ** 2) < 1.25, (str(name), np.mean(zz ** 2))     # (E[t^2] = dof / (dof - 2) = 1.03 here)
        # the domain mean of the field, from the batch means of both sides
        _assert_means_3sigma(fields[:, k].mean(axis=(1, 2)), z["batchMeans"][:, k], str(name))
    c = dict(zip((str(x) for x in z["counterNames"]), z["counters"].sum(0)))
    n_g, n_o = 1_000_000 * len(gr), nb_ref * n_ref
    dg, do = sum(r["counters"]["dropped"] for r in gr) / n_g, c["nBad"] / n_o
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / n_o) + 1e-5, (dg, do)
    kg = sum(r["counters"]["scatterings"] for r in gr) / n_g
    assert abs(kg - c["scatterings"] / n_o) < 0.003 * kg


@pytest.mark.parametrize("config,nb,n,place", [("landsat_tiled", 10, 40_000, "bricks"), ("landsat_tiled_7dir", 10, 20_000, "bricks"),
                                               ("landsat_tiled_7dir", 10, 20_000, "auto")])
def test_fields_beyond_16_MB_against_the_oracle(tmp_path, config, nb, n, place):
    """The Landsat scene tiled 2 x 2 (256 x 256 x 119: 31 MB of extinction) runs the instantiation and the launch set-up of fields
    beyond 16 MB -- bricks, four workgroups per CU, the XCD-aware photon order also for radiance runs: here against the
    oracle on the same domain (4e5 / 2e5 photons a side), domain means of the fluxes and of every radiance direction, the
    dropped-photon rate, the work per photon."""
    import os

    cores = min(16, len(os.sched_getaffinity(0)))
    per_core = 2
    n_ref = nb * n // (cores * per_core)
    child, out = _oracle_child(tmp_path, config, cores, per_core, n_ref)
    from tools import workloads as W

    name, w = W.get(config)
    g, d = W.make_integrator(w)
    # (the tiled scene has column records, which AUTO reads -- 0.5 MB instead of 31: the third case; the bricks are asked for by name)
    g.select_grid_place(place)
    rs = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((77, b)), M.new_PhotonStream(w["mu0"], 0.0, n)) for b in range(1, nb + 1)]
    assert ("GRID_BRICKS" if place == "bricks" else "GRID_COLUMNS") in g.kernel_name()
    g.finalize_Integrator()
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, so + se
    z = np.load(out)
    for k, key in enumerate(("fluxUp", "fluxDown")):
        _assert_means_3sigma([r[key].mean(dtype=np.float64) for r in rs], z["means"][:, k], key)
    for k in range(W.n_dir(w)):
        _assert_means_3sigma([r["intensity"][k].mean(dtype=np.float64) for r in rs], z["intensityMeans"][:, k], f"radiance {k}")
    n_g, n_o = n * nb, n_ref * len(z["nBad"])
    dg, do = sum(r["counters"]["dropped"] for r in rs) / n_g, z["nBad"].sum() / n_o
    assert abs(dg - do) < 3 * np.sqrt(do / n_g + do / n_o) + 2e-5, (dg, do)
    kg = sum(r["counters"]["scatterings"] for r in rs) / n_g
    assert abs(kg - z["scatterings"].sum() / n_o) < 0.02 * kg
