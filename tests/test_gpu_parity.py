"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Levels of evidence, strongest first:
  1. tracer calls (accumulateExtinctionAlongPath) on identical inputs: BIT-EXACT position, cell, tau, steps;
  2. whole photons replayed with the reference's own MT19937 deviates (same draw order): identical fate,
     exit column, scattering order and draw count for (almost) every photon -- the only float differences
     are libm vs device log/cos/sin/acos/exp, which flip a handful of trajectories in 1e4;
  3. production RNG (Philox): fluxes / absorption / radiances within 3 sigma of the oracle, per-photon work
     counters equal within Monte-Carlo noise.
"""
import numpy as np
import pytest

import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import binding as B
from tools import cases
from tests.philox_ref import philox4x32_10
from tests.sums import assert_same_sums

pytestmark = pytest.mark.gpu
f32 = np.float32


def make_gpu(d, table, **params):
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    ext = d["ext"] if isinstance(d["ext"], list) else [d["ext"]]
    ssa = d["ssa"] if isinstance(d["ssa"], list) else [d["ssa"]]
    pf = d["pf"] if isinstance(d["pf"], list) else [d["pf"]]
    tables = table if isinstance(table, list) else [table] * len(ext)
    for i, (e, s, p, t) in enumerate(zip(ext, ssa, pf, tables)):
        dom.addOpticalComponent(f"component {i + 1}", e, s, p, t)
    integ = M.new_Integrator(dom)
    if params:
        integ.specifyParameters(**params)
    return integ


def make_oracle(oracle, d, inv, fwd=None, fwd_orig=None):
    ext = np.stack(d["ext"]) if isinstance(d["ext"], list) else d["ext"]
    ssa = np.stack(d["ssa"]) if isinstance(d["ssa"], list) else d["ssa"]
    pf = np.stack(d["pf"]) if isinstance(d["pf"], list) else d["pf"]
    return oracle.Integrator(d["xe"], d["ye"], d["ze"], ext, ssa, pf, inv, fwd, fwd_orig)


def hg_table(g=0.85, n=64):
    return M.PhaseFunctionTable([M.henyey_greenstein(g, n)])


# ---------------------------------------------------------------------------------------------------------
def test_philox_device_matches_reference_implementation():
    d = cases.plane_parallel()
    g = make_gpu(d, hg_table())
    seed = (10, 7)
    first = (1 << 33) + 12345  # exercises the high counter word
    out, outf = g.philox_blocks(seed, first, 300, 3)
    for i in (0, 1, 63, 64, 299):
        for b in range(3):
            pid = first + i
            want = philox4x32_10((pid & 0xFFFFFFFF, pid >> 32, b, 0), seed)
            assert tuple(int(v) for v in out[i, b]) == want
    # deviates: real(dble(u)/(2**32-1)) as in getRandomReal (Code/RandomNumbersForMC.f95:275-299)
    want_f = (out.astype(np.float64) / 4294967295.0).astype(np.float32)
    assert np.array_equal(outf, want_f)
    assert outf.min() >= 0.0 and outf.max() <= 1.0


def test_exact_arithmetic_helpers():
    """The kernels replace IEEE `/` (by a direction cosine that is constant along a trace) and sqrtf by hardware
    approximations plus fused corrections; the results must be the correctly rounded ones, bit for bit."""
    g = make_gpu(cases.plane_parallel(), hg_table())
    rng = np.random.default_rng(7)
    n = 4_000_000
    # numerators: distances to cell faces (incl. 0 and tiny), denominators: direction cosines / extinctions
    num = (rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-6, 4, n)).astype(np.float32)
    num[:1000] = 0.0
    den = (rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-19, 0, n)).astype(np.float32)
    den[1000:2000] = rng.uniform(0.5, 1.0, 1000).astype(np.float32)
    bad_div, bad_sqrt = g.arith_check(num, den)
    assert bad_div == 0 and bad_sqrt == 0, (bad_div, bad_sqrt)
    # unit-interval arguments as they occur in makeDirectionCosines / next_direct / the Lambertian surface
    u = rng.random(n).astype(np.float32)
    bad_div, bad_sqrt = g.arith_check(u, (1.0 + u).astype(np.float32))
    assert bad_div == 0 and bad_sqrt == 0, (bad_div, bad_sqrt)


def _random_rays(rng, d, n, oracle_integ):
    nz, ny, nx = d["ext"].shape
    ix = rng.integers(1, nx + 1, n)
    iy = rng.integers(1, ny + 1, n)
    iz = rng.integers(1, nz + 1, n)
    u = rng.random((n, 3)).astype(np.float32)
    pos = np.stack([d["xe"][ix - 1] + u[:, 0] * (d["xe"][ix] - d["xe"][ix - 1]),
                    d["ye"][iy - 1] + u[:, 1] * (d["ye"][iy] - d["ye"][iy - 1]),
                    d["ze"][iz - 1] + u[:, 2] * (d["ze"][iz] - d["ze"][iz - 1])], axis=1).astype(np.float32)
    mu = (2 * rng.random(n) - 1).astype(np.float32)
    phi = (2 * np.pi * rng.random(n)).astype(np.float32)
    st = np.sqrt(1 - mu * mu, dtype=np.float32)
    dirs = np.stack([st * np.cos(phi), st * np.sin(phi), mu], axis=1).astype(np.float32)
    # some axis-aligned and grazing rays
    dirs[:50] = [0, 0, -1]
    dirs[50:100] = [0, 0, 1]
    dirs[120:140, 2] = f32(1e-7)   # grazing rays (always given a finite optical path below)
    # direction cosines the reference still divides by (2 tiny <= |cosine|: :1697-1704) although the kernels' reciprocal-based
    # division does not reach down there (< 1e-20: the guarded IEEE division behind its own uniform test), and one below 2 tiny
    dirs[140:150, 0] = f32(1e-25)
    dirs[150:160, 1] = f32(-3e-30)
    dirs[160:170, 0] = f32(1e-39)
    idx = np.stack([ix, iy, iz], axis=1).astype(np.int32)
    target = np.where(rng.random(n) < 0.7, -np.log(np.maximum(rng.random(n), 1e-12)), -1.0).astype(np.float32)
    target[120:170] = f32(0.5)
    return dirs, pos, idx, target


@pytest.mark.parametrize("case,place", [("step", "auto"), ("irregular", "auto"), ("columns", "columns"), ("columns", "linear"),
                                        ("columns", "auto"), ("step", "columns"), ("irregular", "linear"),
                                        ("columns over gas", "columns"), ("columns over gas", "auto")])
def test_tracer_bit_exact(oracle, case, place):
    """accumulateExtinctionAlongPath (:1654-1807) ray by ray against the oracle, every bit of position, cell, optical path and step
    count -- from the bricked copy of the field (what the hook reads by default), from the plain field, from the column
    records of a field whose columns each hold one run of one value (cases.column_clouds, the step cloud), and (round 5) from column
    records OVER A BASE PROFILE: the same clouds plus a horizontally uniform second component."""
    rng = np.random.default_rng(42)
    tab = hg_table()
    if case == "step":
        d = cases.step_cloud()
    elif case == "columns":
        d = cases.column_clouds()
    elif case == "columns over gas":
        c = cases.column_clouds()
        gas = np.broadcast_to(np.linspace(3.0e-3, 1.0e-4, c["ext"].shape[0], dtype=np.float32)[:, None, None], c["ext"].shape).copy()
        d = dict(c, ext=[c["ext"], gas], ssa=[c["ssa"], np.ones_like(gas)], pf=[c["pf"], np.ones(gas.shape, np.int32)])
    else:
        d = cases.irregular_domain()
    g = make_gpu(d, tab)
    assert g.has_column_records() == (case != "irregular")
    if place != "auto":
        g.select_grid_place(place)
    o = make_oracle(oracle, d, [tab.inverse_table(9001)] * (2 if case == "columns over gas" else 1))
    if case == "columns over gas":
        d = dict(d, ext=d["ext"][0])   # (_random_rays only asks for the grid)
    n = 4000
    dirs, pos, idx, target = _random_rays(rng, d, n, o)
    tau, p2, i2, steps = g.trace_rays(dirs, pos, idx, target)
    nerr = 0
    for k in range(n):
        t, pp, ii, ss = o.trace(dirs[k], pos[k], idx[k], None if target[k] < 0 else float(target[k]))
        assert f32(t) == tau[k], (k, t, tau[k])
        assert np.array_equal(pp, p2[k]) and list(i2[k]) == ii and ss == steps[k], (k, pp, p2[k], ii, i2[k])
        nerr += t < 0
    assert nerr < n // 20


def test_a_field_without_column_records_refuses_them():
    g = make_gpu(cases.irregular_domain(), hg_table())
    with pytest.raises(Exception, match="no column records"):
        g.select_grid_place("columns")


def _replay(oracle, d, tab, n, seed, solar_mu, **params):
    inv = [tab.inverse_table(10001)]
    o = make_oracle(oracle, d, inv)
    o.specify(**{k: v for k, v in params.items()})
    rng = oracle.RandomNumberSequence(seed)
    ph = oracle.photons_directional(rng, solar_mu, 0.0, n)
    draws_before = rng.draws
    ref = o.compute(rng, *ph, record=True, normalise=False)
    ndraw = rng.draws - draws_before
    rng2 = oracle.RandomNumberSequence(seed)
    rng2.reals(draws_before)
    randoms = rng2.reals(ndraw)
    g = make_gpu(d, tab)
    g.set_tables(1, inverse=inv[0])
    gp = {}
    for k, v in params.items():
        gp[{"useRR": "useRussianRoulette"}.get(k, k)] = v
    if gp:
        g.specifyParameters(**gp)
    out = g.run_replay(M.PhotonStream(arrays=ph), randoms, ref["drawStart"])
    return ref, out, g


def test_replay_reference_random_stream_step_cloud(oracle):
    d = cases.step_cloud(ssa=0.99)
    n = 20000
    ref, out, g = _replay(oracle, d, hg_table(), n, [10, 1], 0.5, surfaceAlbedo=0.3)
    used_ref = np.diff(ref["drawStart"])
    same = (out["fate"] == ref["fate"]) & (out["fateColumn"] == ref["fateColumn"]) & \
           (out["fateOrder"] == ref["fateOrder"]) & (out["drawsUsed"] == used_ref)
    # device libm differs from glibc by an ulp in log/cos/sin: a few trajectories in 1e4 diverge
    assert same.mean() > 0.995, same.mean()
    assert np.array_equal(out["fateWeight"][same], ref["fateWeight"][same])
    c = out["counters"]
    assert c["photons"] == n
    assert abs(c["cellSteps"] - ref["cellSteps"]) <= 0.002 * ref["cellSteps"]
    assert abs(c["scatterings"] - ref["scatterings"]) <= 0.002 * ref["scatterings"]
    lay = g.layout()
    raw = out["raw"]
    for name, off in (("fluxUp", lay.fluxUp), ("fluxDown", lay.fluxDown), ("fluxAbsorbed", lay.fluxAbsorbed)):
        a = raw[off:off + 32]
        b = ref[name].ravel().astype(np.float64)
        assert abs(a.sum() - b.sum()) <= 0.003 * max(b.sum(), 1.0), name
    vol = raw[lay.volumeAbsorption:lay.volumeAbsorption + 32 * 32]
    assert abs(vol.sum() - ref["volumeAbsorption"].sum()) <= 0.003 * ref["volumeAbsorption"].sum()


def _batches_gpu(g, nb, n, mu0, az=0.0, iseed=10):
    res = []
    for b in range(1, nb + 1):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((iseed, b)), M.new_PhotonStream(mu0, az, n))
        res.append(r)
    return res


def _batches_oracle(oracle, o, nb, n, mu0, az=0.0, iseed=10):
    res = []
    for b in range(1, nb + 1):
        rng = oracle.RandomNumberSequence([iseed, b])
        ph = oracle.photons_directional(rng, mu0, az, n)
        res.append(o.compute(rng, *ph))
    return res


def _mean_se(rs, key):
    a = np.stack([r[key].astype(np.float64) for r in rs])
    return a.mean(0), a.std(0, ddof=1) / np.sqrt(len(rs))


def _assert_3sigma(gpu, ref, key, floor=1e-9, frac_ok=None):
    """Monte-Carlo tolerance (written here, as BASELINE.md 3(4) states it): the DOMAIN MEAN must agree within
    3*sqrt(se_gpu^2 + se_ref^2), standard errors from the batch-to-batch variance as in
    monteCarloDriver.f95:358-378.  Per column / cell the same statistic z = |diff| / sigma is formed; with
    standard errors estimated from nb batches each, z follows Student's t with ~(nb_gpu + nb_ref - 2) degrees
    of freedom, so over n cells the number of 3-sigma excursions is binomial with p = P(|t| > 3): it must not
    exceed its expectation by more than 3 standard deviations (+1), and no cell may exceed the value a
    t-variable passes with probability 1e-3 / n."""
    from scipy import stats

    mg, sg = _mean_se(gpu, key)
    mr, sr = _mean_se(ref, key)
    sig = np.sqrt(sg ** 2 + sr ** 2) + floor
    z = np.abs(mg - mr) / sig
    dof = len(gpu) + len(ref) - 2
    p3 = 2 * stats.t.sf(3.0, dof)
    expected = p3 * z.size
    allowed = int(np.ceil(expected + 3 * np.sqrt(expected) + 1))
    zmax = stats.t.isf(1e-3 / (2 * z.size), dof)
    assert (z > 3.0).sum() <= allowed, (key, int((z > 3.0).sum()), allowed, z.max())
    assert z.max() < zmax, (key, z.max(), zmax)
    dg = np.array([r[key].mean(dtype=np.float64) for r in gpu])
    dr = np.array([r[key].mean(dtype=np.float64) for r in ref])
    t = 3.0 * np.sqrt(dg.var(ddof=1) / len(dg) + dr.var(ddof=1) / len(dr)) + floor
    assert abs(dg.mean() - dr.mean()) <= t, (key, dg.mean(), dr.mean(), t)


def _parity(oracle, g, o, nb, n, mu0, az=0.0, keys=("fluxUp", "fluxDown"), floor=1e-9, iseed=10):
    """Two-stage 3-sigma test.  Stage 1: nb batches of n photons on both sides (seeds (10, b)).  A suite of ~60
    such assertions would raise a false alarm every few runs (P(|t| > 3) ~ 1 % at these batch counts), so a
    failure is re-examined once with an independent, twice as large sample (seeds (11, b)); a real bias fails both
    stages, a fluctuation passes the second with probability > 99 %.  Returns the batches that were accepted."""
    try:
        gr, orr = _batches_gpu(g, nb, n, mu0, az, iseed=iseed), _batches_oracle(oracle, o, nb, n, mu0, az, iseed=iseed)
        for key in keys:
            _assert_3sigma(gr, orr, key, floor=floor)
        return gr, orr
    except AssertionError as first:
        import inspect

        from tests.conftest import record_stage1_miss
        caller = next((f.function for f in inspect.stack()[1:] if f.function.startswith("test_")), "?")
        record_stage1_miss(caller, first.args)
        gr = _batches_gpu(g, 2 * nb, n, mu0, az, iseed=iseed + 1)
        orr = _batches_oracle(oracle, o, 2 * nb, n, mu0, az, iseed=iseed + 1)
        try:
            for key in keys:
                _assert_3sigma(gr, orr, key, floor=floor)
        except AssertionError as second:
            raise AssertionError(f"failed twice: {first.args} then {second.args}")
        return gr, orr


@pytest.mark.parametrize("mu0,ssa,albedo", [(1.0, 1.0, 0.0), (0.5, 0.99, 0.2)])
def test_step_cloud_flux_parity(oracle, mu0, ssa, albedo):
    d = cases.step_cloud(ssa=ssa)
    tab = hg_table()
    g = make_gpu(d, tab, surfaceAlbedo=albedo, minInverseTableSize=10001)
    o = make_oracle(oracle, d, [tab.inverse_table(10001)])
    o.specify(surfaceAlbedo=albedo)
    nb, n = 10, 40000
    keys = ("fluxUp", "fluxDown") + (("fluxAbsorbed", "volumeAbsorption") if ssa < 1 else ())
    gr, orr = _parity(oracle, g, o, nb, n, mu0, keys=keys)
    nb = len(gr)
    # per-photon work counters (what the roofline's algorithmic bytes are computed from)
    cs = sum(r["counters"]["cellSteps"] for r in gr) / (nb * n)
    ks = sum(r["counters"]["scatterings"] for r in gr) / (nb * n)
    cs_o = sum(r["cellSteps"] for r in orr) / (nb * n)
    ks_o = sum(r["scatterings"] for r in orr) / (nb * n)
    assert abs(cs - cs_o) < 0.02 * cs_o and abs(ks - ks_o) < 0.02 * ks_o
    # energy closure incl. the dropped-photon deficit (quirk Q4); exact only without roulette plays
    r0 = gr[0]
    if ssa == 1.0 and albedo == 0.0:
        closure = r0["fluxUp"].mean(dtype=np.float64) + r0["fluxDown"].mean(dtype=np.float64)
        assert abs(closure - (1 - r0["counters"]["dropped"] / n)) < 2e-6


def test_results_do_not_depend_on_launch_geometry():
    # per-photon Philox streams: same photons whatever the grid / threshold; only the float64 summation order moves
    d = cases.step_cloud(ssa=0.99)
    g = make_gpu(d, hg_table(), surfaceAlbedo=0.1)
    n = 30000
    g.set_tuning(evThreshold=32, blocksPerCU=0)
    a = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, n))
    g.set_tuning(evThreshold=8, blocksPerCU=1)
    b = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, n))
    assert a["counters"] == b["counters"]
    assert_same_sums(a["raw"], b["raw"], a["counters"])
    # split into two launches with counter offsets = one launch
    g.set_tuning(evThreshold=32, blocksPerCU=0)
    g.launch(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, n // 2), firstPhoton=0)
    g.launch(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, n - n // 2), firstPhoton=n // 2, zero=False)
    c = g.finish()
    assert c["counters"] == a["counters"]
    assert_same_sums(c["raw"], a["raw"], a["counters"])
    # the library cuts very long batches into several launches by itself (float32 partial sums per workgroup must
    # stay below 2^24): with the limit lowered to 7000 photons this batch runs as five launches
    from i3rc_monte_carlo_model_amd import binding as B
    assert B.load().i3rc_hip_set_launch_limit(g._h, 7000) == 0
    e = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, n))
    assert B.load().i3rc_hip_set_launch_limit(g._h, 0) == 0
    assert e["counters"] == a["counters"]
    assert_same_sums(e["raw"], a["raw"], a["counters"])


def test_a_sequence_used_twice_goes_on_with_fresh_photons():
    # the reference's Mersenne Twister simply goes on when a sequence is used for a second computeRadiativeTransfer; here a
    # sequence hands out per-photon Philox streams by number, and a second call must get the NEXT numbers, not the same
    # photons again (round-1 advisor finding): two calls of n photons = one call of 2 n photons
    d = cases.step_cloud(ssa=0.99)
    g = make_gpu(d, hg_table(), surfaceAlbedo=0.1)
    n = 20000
    seq = M.new_RandomNumberSequence((3, 4))
    a = g.computeRadiativeTransfer(seq, M.new_PhotonStream(0.7, 30.0, n))
    b = g.computeRadiativeTransfer(seq, M.new_PhotonStream(0.7, 30.0, n))
    assert a["counters"] != b["counters"] and not np.array_equal(a["raw"], b["raw"])
    c = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 4)), M.new_PhotonStream(0.7, 30.0, 2 * n))
    for k in ("photons", "cellSteps", "scatterings", "surfaceHits", "exitsTop", "roulette", "dropped"):
        assert a["counters"][k] + b["counters"][k] == c["counters"][k], k
    assert_same_sums((a["raw"] + b["raw"])[:g.layout().counters], c["raw"][:g.layout().counters], c["counters"])


def test_specialised_and_general_kernels_trace_the_same_photons():
    # the launch picks a kernel specialised for regular grid / ray tracing / one component / Directional source;
    # the general kernel must give the same photons the same fate
    for ssa, albedo, dirs in ((1.0, 0.0, None), (0.98, 0.3, ([1.0, 0.6], [0.0, 120.0]))):
        d = cases.step_cloud(ssa=ssa, nlayers=16)
        kw = dict(surfaceAlbedo=albedo)
        if dirs:
            kw.update(intensityMus=dirs[0], intensityPhis=dirs[1])
        g = make_gpu(d, hg_table(), **kw)
        n = 50000
        g.set_tuning(40, 0, forceGeneral=False)
        a = g.computeRadiativeTransfer(M.new_RandomNumberSequence((5, 6)), M.new_PhotonStream(0.6, 10.0, n))
        g.set_tuning(40, 0, forceGeneral=True)
        b = g.computeRadiativeTransfer(M.new_RandomNumberSequence((5, 6)), M.new_PhotonStream(0.6, 10.0, n))
        assert a["counters"] == b["counters"]
        assert_same_sums(a["raw"], b["raw"], a["counters"], directions=len(dirs[0]) if dirs else 0)


def test_lane_and_general_kernels_trace_the_same_photons_for_any_batch_size():
    # the specialised one-photon-per-lane kernel and the general kernel must give every photon the same fate:
    # identical work counters, tallies equal up to the float32 order of the LDS partial sums -- grids and tallies in
    # LDS or in HBM, reflecting surfaces, batches smaller than a wave / ending in the middle of a reservoir refill
    cfgs = [(cases.step_cloud(ssa=1.0, nlayers=16), 0.0, 1.0, 0.0, 200000),
            (cases.step_cloud(ssa=0.97, nlayers=32), 0.4, 0.5, 75.0, 60000),
            (cases.plane_parallel(optical_depth=0.3), 1.0, 0.3, 0.0, 20000),      # mirror-white surface, thin layer
            (cases.radar_cloud_64(), 0.1, 0.8, 200.0, 20000)]                      # tallies and grid in HBM, not in LDS
    for d, albedo, mu0, az, n in cfgs:
        g = make_gpu(d, hg_table(), surfaceAlbedo=albedo)
        out = {}
        for kernel in ("lane", "general"):
            g.set_tuning(40, 0, kernel=kernel)
            out[kernel] = g.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 2)), M.new_PhotonStream(mu0, az, n))
        assert out["lane"]["counters"] == out["general"]["counters"]
        assert_same_sums(out["lane"]["raw"], out["general"]["raw"], out["lane"]["counters"])
        for m in (1, 63, 129, 5000):
            g.set_tuning(40, 0, kernel="general")
            a = g.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 3)), M.new_PhotonStream(mu0, az, m))
            g.set_tuning(40, 0, kernel="lane")
            b = g.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 3)), M.new_PhotonStream(mu0, az, m))
            assert a["counters"] == b["counters"] and a["counters"]["photons"] == m
            assert_same_sums(a["raw"], b["raw"], a["counters"])
        g.set_tuning(40, 0, kernel="auto")


def test_edge_cases_and_errors():
    d = cases.plane_parallel(optical_depth=0.0)  # empty domain: everything reaches the black surface
    g = make_gpu(d, hg_table())
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), M.new_PhotonStream(1.0, 0.0, 1000))
    assert r["fluxDown"][0, 0] == 1.0 and r["fluxUp"][0, 0] == 0.0 and r["counters"]["scatterings"] == 0
    g.specifyParameters(surfaceAlbedo=1.0)  # mirror-white surface: everything comes back out
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), M.new_PhotonStream(1.0, 0.0, 1000))
    assert r["fluxUp"][0, 0] == 1.0
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), M.new_PhotonStream(1.0, 0.0, 1))  # 1-photon warm-up
    assert r["counters"]["photons"] == 1
    with pytest.raises(M.I3RCError):
        g.specifyParameters(surfaceAlbedo=1.5)
    with pytest.raises(M.I3RCError):
        g.specifyParameters(surfaceAlbedo=0.1, surfaceBDRF=M.new_SurfaceDescription([0.2]))
    with pytest.raises(M.I3RCError):
        g.specifyParameters(intensityMus=[0.5])
    with pytest.raises(M.I3RCError):
        g.specifyParameters(intensityMus=[0.0], intensityPhis=[0.0])
    s = M.new_PhotonStream(1.0, 0.0, 10)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), s)
    assert not s.morePhotonsExist()  # the stream is consumed, as in the reference
    with pytest.raises(M.I3RCError):
        g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), s)
    # explicit photons: a horizontal / NaN direction or a start outside the domain is refused, as the reference's
    # photon-stream constructors do (such a photon would never leave a periodic domain)
    ok = [np.full(4, 0.5, np.float32), np.full(4, 0.5, np.float32), np.full(4, 0.9, np.float32), np.full(4, -0.7, np.float32), np.zeros(4, np.float32)]
    assert g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), M.PhotonStream(arrays=ok))["counters"]["photons"] == 4
    for k, badv in ((3, 0.0), (3, np.nan), (3, 1.5), (0, 1.2), (2, -0.1), (4, np.inf)):
        arr = [a.copy() for a in ok]
        arr[k][2] = badv
        with pytest.raises(M.I3RCError):
            g.computeRadiativeTransfer(M.new_RandomNumberSequence((1, 1)), M.PhotonStream(arrays=arr))
    g.finalize_Integrator()
    assert not g.isReady_Integrator()


def test_step_cloud_at_1e8_photons_size_independent_properties(oracle):
    # BASELINE.json configs[1] at its full size: 1e8 photons as 20 batches of 5e6 (float64 tallies: no 2^24
    # saturation).  The oracle cannot follow to this size, so the checks are the ones that do not depend on it:
    # exact bookkeeping per batch, the batch-to-batch scatter predicted by binomial statistics, and agreement with
    # the oracle's (necessarily smaller) sample and with the reference's recorded 1e6-photon result within 3 sigma.
    d = cases.step_cloud(nlayers=16)
    tab = hg_table()
    g = make_gpu(d, tab, minInverseTableSize=10001)
    nb, n = 20, 5_000_000
    ups, dns, drops = [], [], []
    cols = np.zeros((1, 32))
    for b in range(1, nb + 1):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(1.0, 0.0, n))
        c = r["counters"]
        assert c["photons"] == n and c["exitsTop"] + c["surfaceHits"] + c["dropped"] == n    # every photon ends once
        up, dn = r["fluxUp"].mean(dtype=np.float64), r["fluxDown"].mean(dtype=np.float64)
        assert abs(up + dn + c["dropped"] / n - 1.0) < 1e-6
        assert abs(up * n - c["exitsTop"]) < 1e-3 * n ** 0.5 + 40     # unit weights: the flux tally counts photons (float32 file format)
        ups.append(up); dns.append(dn); drops.append(c["dropped"] / n)
        cols += r["fluxUp"].astype(np.float64)
    ups, dns = np.array(ups), np.array(dns)
    p = ups.mean()
    expected_sd = np.sqrt(p * (1 - p) / n)          # a photon leaves through the top or it does not
    assert 0.6 * expected_sd < ups.std(ddof=1) < 1.5 * expected_sd, (ups.std(ddof=1), expected_sd)
    se = ups.std(ddof=1) / np.sqrt(nb)               # ~4.7e-5 at 1e8 photons
    assert 3e-5 < se < 7e-5
    # the reference's own 1e6-photon run: Fup 0.3254, Fdown 0.6746 (printed to 4 digits; its standard error 4.7e-4)
    assert abs(p - 0.3254) < 3 * 4.7e-4 + 5e-5 and abs(dns.mean() - 0.6746) < 3 * 4.7e-4 + 5e-5
    assert 1.5e-5 < np.mean(drops) < 4e-5            # quirk Q4: tracer drops, 2e-5...3e-5 on this case
    # the oracle on 10 x 2e5 photons
    o = make_oracle(oracle, d, [tab.inverse_table(10001)])
    orr = _batches_oracle(oracle, o, 10, 200000, 1.0)
    ou = np.array([r["fluxUp"].mean(dtype=np.float64) for r in orr])
    assert abs(p - ou.mean()) < 3 * np.sqrt(se ** 2 + ou.var(ddof=1) / 10)
    # per column: thin half (columns 1-16, optical depth 2) darker than the thick half (optical depth 18)
    col = cols[0] / nb
    assert col[:12].max() < col[20:].min()


def test_absorption_is_tallied_per_cell_and_the_column_sums_follow():
    """:644-647 add one increment to fluxAbsorbed(ix, iy) and to volumeAbsorption(ix, iy, iz).  The kernels tally the cell only (one
    scattered float64 atomic per scattering instead of two: the absorbing I3RC cases ran two to nineteen times slower than the
    conservative ones on them) -- in LDS where the domain has few cells -- and fluxAbsorbed is the sum of its column's cells, formed
    on the device after the launch (absorbed_columns_kernel): exactly that sum, in the raw block of a plain launch, of a launch that
    adds to a block already summed, and of every batch of a fused group; and the same photons give the same sums whether the cells are
    gathered in LDS or not (the order of the float64 additions: tests/sums.py)."""
    n = 200_000
    for d, kw in ((cases.step_cloud(ssa=0.99), {}), (cases.step_cloud(ssa=0.9, nlayers=3), dict(surfaceAlbedo=0.3)), (cases.two_component(seed=5, nx=6, ny=4, nz=8), {})):
        tab = ([M.PhaseFunctionTable([M.henyey_greenstein(0.85, 32), M.henyey_greenstein(0.6, 16)]),
                M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])] if isinstance(d["ext"], list) else hg_table())
        g = make_gpu(d, tab, **kw)
        lay, (nz, ny, nx) = g.layout(), (len(d["ze"]) - 1, len(d["ye"]) - 1, len(d["xe"]) - 1)
        ncol = nx * ny

        def columns(raw):
            vol = raw[lay.volumeAbsorption:lay.volumeAbsorption + nz * ncol].reshape(nz, ncol)
            s = np.zeros(ncol)
            for k in range(nz):
                s = s + vol[k]
            return s, raw[lay.fluxAbsorbed:lay.fluxAbsorbed + ncol]
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 1)), M.new_PhotonStream(0.7, 10.0, n))
        want, got = columns(r["raw"])
        assert want.sum() > 0 and np.array_equal(want, got)
        # a second launch adding to the same block (the photon stream's second half)
        g.launch(M.new_RandomNumberSequence((3, 2)), M.new_PhotonStream(0.7, 10.0, n // 2))
        g.launch(M.new_RandomNumberSequence((3, 2)), M.new_PhotonStream(0.7, 10.0, n // 2), firstPhoton=n // 2, zero=False)
        both = g.finish()
        want, got = columns(both["raw"])
        assert np.array_equal(want, got)
        whole = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 2)), M.new_PhotonStream(0.7, 10.0, n))
        assert_same_sums(both["raw"][:lay.counters], whole["raw"][:lay.counters], whole["counters"], what="two launches into one block")
        # the cells gathered in LDS or not
        g.set_lds_tallies(False)
        plain = g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 1)), M.new_PhotonStream(0.7, 10.0, n))
        g.set_lds_tallies(True)
        assert plain["counters"] == r["counters"]
        assert_same_sums(plain["raw"][:lay.counters], r["raw"][:lay.counters], r["counters"], what="volume absorption in LDS or not")
        want, got = columns(plain["raw"])
        assert np.array_equal(want, got)
        # every batch of a fused group
        for b, res in enumerate(g.computeRadiativeTransferBatches((3, 1), 3, 0.7, 10.0, n // 4)):
            want, got = columns(res["raw"])
            assert want.sum() > 0 and np.array_equal(want, got), b
        # the normalised fields say the same: fluxAbsorbed = sum over the layers of volumeAbsorption x depth
        dz = np.diff(d["ze"]).astype(np.float64)
        assert np.allclose((r["volumeAbsorption"].astype(np.float64) * dz[:, None, None]).sum(0), r["fluxAbsorbed"], rtol=2e-6, atol=1e-12)
        g.finalize_Integrator()


def test_step_cloud_flux_fields_at_1e8_photons_against_the_oracle_at_1e8(tmp_path):
    # The north star's acceptance line taken literally: flux tallies within 3 sigma of the reference at 1e8 photons.
    # BASELINE.json configs[1] on the GPU (20 batches of 5e6) against the CPU restatement at the same 1e8 photons
    # (16 processes x 10 batches x 625000 photons on the box's host cores, started as a child program: ~15 s),
    # column by column, standard errors from the batch-to-batch scatter on both sides (_assert_3sigma).
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "oracle_1e8.npz")
    cores = min(16, len(os.sched_getaffinity(0)))
    per_core = 10
    photons = 100_000_000 // (cores * per_core)
    child = subprocess.Popen([sys.executable, os.path.join(root, "tools", "cpu_baseline.py"), "--cores", str(cores), "--batches-per-core", str(per_core),
                              "--photons", str(photons), "--save", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    g = make_gpu(cases.step_cloud(nlayers=16), hg_table(), minInverseTableSize=10001)   # the GPU works meanwhile
    gr = _batches_gpu(g, 20, 5_000_000, 1.0)
    so, se = child.communicate(timeout=600)
    assert child.returncode == 0, so + se
    z = np.load(out)
    orr = [dict(fluxUp=u, fluxDown=d) for u, d in zip(z["fluxUp"], z["fluxDown"])]
    assert len(orr) == cores * per_core and sum(r["counters"]["photons"] for r in gr) == 100_000_000
    for key in ("fluxUp", "fluxDown"):
        _assert_3sigma(gr, orr, key)
    mg, mo = np.mean([r["fluxUp"].mean() for r in gr]), np.mean([r["fluxUp"].mean() for r in orr])
    print(f"mean upward flux at 1e8 photons: gpu {mg:.6f} oracle {mo:.6f} (difference {mg - mo:+.2e})")


def _random_regular_case(rng, case):
    nx, ny, nz = (int(rng.choice([1, 2, 5, 17, 40])), int(rng.choice([1, 1, 3, 24])), int(rng.choice([1, 4, 9, 33])))
    if case % 5 == 4:
        nx, ny, nz = 48, 40, 12        # 23040 cells: the grid stays in global memory
    xe = (np.float32(rng.uniform(20, 200)) * np.arange(nx + 1)).astype(np.float32)
    ye = (np.float32(rng.uniform(20, 200)) * np.arange(ny + 1)).astype(np.float32)
    ze = (np.float32(rng.uniform(10, 100)) * np.arange(nz + 1)).astype(np.float32) + np.float32(rng.choice([0.0, 150.0]))
    ext = rng.uniform(0.0, 0.05, (nz, ny, nx)).astype(np.float32)
    ext[rng.uniform(size=ext.shape) < 0.3] = 0.0
    if rng.uniform() < 0.5:
        ssa = np.full(ext.shape, np.float32(rng.choice([1.0, 0.9, 0.5])), np.float32)
    else:
        ssa = rng.uniform(0.3, 1.0, ext.shape).astype(np.float32)
    d = dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=ssa, pf=np.ones(ext.shape, np.int32))
    albedo, mu0, az = float(rng.choice([0.0, 0.3, 1.0])), float(rng.uniform(0.05, 1.0)), float(rng.uniform(0, 360))
    return d, albedo, mu0, az


def test_random_regular_domains_against_the_oracle(oracle):
    # the same random domains against the CPU restatement (production RNG against MT19937: statistical parity)
    rng = np.random.default_rng(2024)
    inv = hg_table().inverse_table(9001)
    for case in range(16):
        d, albedo, mu0, az = _random_regular_case(rng, case)
        g = make_gpu(d, hg_table(), surfaceAlbedo=albedo)
        g.set_tables(1, inverse=inv)
        o = make_oracle(oracle, d, [inv])
        o.specify(surfaceAlbedo=albedo)
        gr, orr = _parity(oracle, g, o, 6, 5000, mu0, az=az, keys=("fluxUp", "fluxDown", "fluxAbsorbed"), floor=1e-6)
        dg = sum(r["counters"]["dropped"] for r in gr) / (5000 * len(gr))
        do = sum(r["nBad"] for r in orr) / (5000 * len(orr))
        assert abs(dg - do) < 3 * np.sqrt((do * (1 - do) + 1e-4) / (5000 * len(gr))) + 1e-3, (case, dg, do)


def test_kernels_agree_on_random_regular_domains():
    # randomised cross-check of the two kernels: regular grids of random shape (1-D, 2-D, 3-D; grid in LDS or in
    # global memory), random extinction fields with holes, random single-scattering albedo (uniform or per cell),
    # surface albedo, sun position -- identical work counters, tallies equal up to float32 summation order
    rng = np.random.default_rng(2024)
    dropped_all = 0
    seen_kernels = set()
    for case in range(16):
        d, albedo, mu0, az = _random_regular_case(rng, case)
        nx, ny, nz = len(d["xe"]) - 1, len(d["ye"]) - 1, len(d["ze"]) - 1
        g = make_gpu(d, hg_table(), surfaceAlbedo=albedo)
        n = 20000
        out = {}
        for kernel in ("lane", "general", "auto"):   # (auto: small domains take the instantiation that keeps the inverse table in LDS)
            g.set_tuning(0, 0, kernel=kernel)
            out[kernel] = g.computeRadiativeTransfer(M.new_RandomNumberSequence((77, case)), M.new_PhotonStream(mu0, az, n))
            seen_kernels.add(g.kernel_name())
        for other in ("general", "auto"):
            assert out["lane"]["counters"] == out[other]["counters"], (case, other, nx, ny, nz)
            assert_same_sums(out["lane"]["raw"], out[other]["raw"], out["lane"]["counters"], what=(case, other, nx, ny, nz))
        c = out["lane"]["counters"]
        assert c["photons"] == n
        # (cases 6 and 7 -- one layer, base at z = 150 -- lose EVERY photon to the tracer's "step <= 0" escape, in the
        # oracle too: the start height z0 + (1 - spacing(1)) (zMax - z0) rounds to zMax itself there; reference behaviour)
        dropped_all += c["dropped"] == n
    assert dropped_all == 2
    assert any("table in LDS" in k for k in seen_kernels), seen_kernels


def test_pipelined_batches_equal_one_call_per_batch():
    """i3rc_hip_run_batches (several batches on the device at a time, each on a stream and in a tally buffer of its own)
    gives every batch exactly what zero + launch + fetch gives it: same photons (integer work counters), same tallies up
    to the order of the float64 additions -- for a flux run with LDS tallies and for a radiance run with a surface."""
    for d, params, mu0 in ((cases.step_cloud(ssa=0.99), dict(surfaceAlbedo=0.1), 0.7),
                           (cases.step_cloud(ssa=1.0, nlayers=8), dict(intensityMus=[1.0, 0.4], intensityPhis=[0.0, 80.0],
                                                                       useRussianRouletteForIntensity=True, zetaMin=0.3, surfaceAlbedo=0.3), 0.8)):
        g = make_gpu(d, hg_table(), **params)
        n, nb = 20000, 7
        one = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((5, 3 + b)), M.new_PhotonStream(mu0, 25.0, n)) for b in range(nb)]
        before = g.fetch().copy()                                   # the handle's own tallies: the last of those calls
        for in_flight in (1, 3, 8):
            many = g.computeRadiativeTransferBatches((5, 3), nb, mu0, 25.0, n, inFlight=in_flight)
            assert len(many) == nb
            for a, b in zip(one, many):
                assert a["counters"] == b["counters"]
                assert_same_sums(a["raw"], b["raw"], a["counters"], directions=2)
        assert np.array_equal(g.fetch(), before)                   # ... which the pipelined call leaves alone
        g.finalize_Integrator()
    # errors: explicit streams are refused
    from i3rc_monte_carlo_model_amd import binding as B
    import ctypes as C
    g = make_gpu(cases.step_cloud(), hg_table())
    g._ensure_tables()
    s = B.Source(); s.kind = 1
    out = np.zeros(g.layout().total, np.float64)
    assert B.load().i3rc_hip_run_batches(g._h, 1, 2, 1, 100, C.byref(s), 0, out.ctypes.data_as(B.dp)) != 0
    assert b"Directional" in B.load().i3rc_hip_last_error(g._h)


def test_looking_ahead_never_changes_a_batch():
    """i3rc_hip_compute_batch launches the following batches of a driver's loop ahead of the caller: whatever the caller does
    next -- goes on with the loop, jumps to another seed, changes the photon count, the sun, a parameter, the directions --
    every batch equals the plain zero + launch + fetch of the same batch (same photons, tallies up to summation order)."""
    d = cases.step_cloud(ssa=0.99, nlayers=8)
    params = dict(intensityMus=[1.0, 0.4], intensityPhis=[0.0, 80.0], useRussianRouletteForIntensity=True, zetaMin=0.3, surfaceAlbedo=0.3)
    plain, ahead = make_gpu(d, hg_table(), **params), make_gpu(d, hg_table(), **params)

    def both(seed, n, mu0=0.8, az=25.0, look=3):
        a = plain.computeRadiativeTransfer(M.new_RandomNumberSequence(seed), M.new_PhotonStream(mu0, az, n))
        b = ahead.computeRadiativeTransferLookingAhead(M.new_RandomNumberSequence(seed), M.new_PhotonStream(mu0, az, n), lookAhead=look)
        # (since round 4 a radiance problem's batches are looked ahead in fused groups as well: photons and dropped photons are
        # counted per batch, the radiance kernels' other counters per group -- _same_batches)
        assert all(a["counters"][k] == b["counters"][k] for k in ("photons", "dropped")), (seed, n)
        assert_same_sums(_tallies_only(plain, a), _tallies_only(ahead, b), a["counters"], directions=2, what=(seed, n))

    both((7, 0), 1)                                     # the drivers' one-photon warm-up
    for b in range(1, 9):
        both((7, b), 20000)                             # the loop: from the second batch on the next ones are under way
    both((7, 20), 20000)                                # a jump: the batches launched ahead are discarded
    both((7, 21), 20000); both((7, 22), 20000)
    both((7, 23), 5000)                                 # another photon count
    both((7, 24), 5000); both((7, 25), 5000, mu0=0.6)   # another sun in mid-loop
    both((7, 26), 5000, mu0=0.6); both((7, 27), 5000, mu0=0.6)
    for g in (plain, ahead):                            # a parameter changes while batches are under way
        g.specifyParameters(surfaceAlbedo=0.6)
    both((7, 28), 5000, mu0=0.6); both((7, 29), 5000, mu0=0.6); both((7, 30), 5000, mu0=0.6)
    for g in (plain, ahead):                            # ... and the directions (another tally layout)
        g.specifyParameters(intensityMus=[0.9], intensityPhis=[10.0])
    both((7, 31), 5000, mu0=0.6); both((7, 32), 5000, mu0=0.6); both((7, 33), 5000, mu0=0.6, look=7)
    both((8, 34), 5000, mu0=0.6, look=0); both((8, 35), 5000, mu0=0.6, look=0)
    # the plain path on the same handle is not disturbed by batches launched ahead
    r = ahead.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 35)), M.new_PhotonStream(0.6, 25.0, 5000))
    q = plain.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 35)), M.new_PhotonStream(0.6, 25.0, 5000))
    assert r["counters"] == q["counters"]
    plain.finalize_Integrator(); ahead.finalize_Integrator()


def _tallies_only(g, r):
    """The raw tally block without its counter words."""
    lay = g.layout()
    return np.delete(r["raw"], np.s_[lay.counters:lay.counters + B.NUM_COUNTERS])


def _same_batches(one, many, what, g=None):
    """Batch by batch: the same photons traced.  Flux problems: every integer work counter identical per batch.  Radiance
    problems in a fused launch (g given): the radiance kernels count per wave and hand their counts to the batch a wave was
    given last -- photons and dropped photons (what the normalisation needs) are exact per batch, the other counters over the
    batches together; the tallies are per batch either way."""
    assert len(one) == len(many), what
    for b, (a, f) in enumerate(zip(one, many)):
        if g is None:
            assert a["counters"] == f["counters"], (what, b, {k: (a["counters"][k], f["counters"][k]) for k in a["counters"] if a["counters"][k] != f["counters"][k]})
            assert_same_sums(a["raw"], f["raw"], a["counters"], directions=8, what=(what, b))
        else:
            assert all(a["counters"][k] == f["counters"][k] for k in ("photons", "dropped")), (what, b, a["counters"], f["counters"])
            assert_same_sums(_tallies_only(g, a), _tallies_only(g, f), a["counters"], directions=8, what=(what, b))
    if g is not None:
        for k in one[0]["counters"]:
            assert sum(r["counters"][k] for r in one) == sum(r["counters"][k] for r in many), (what, k)


def test_fused_batches_equal_one_launch_per_batch():
    """A fused multi-batch launch (photon_kernel<PhiloxBatchStream, ...>: the batches of a loop in ONE grid, every lane with
    the batch of its photon in its Philox key, per-batch tally blocks in global memory) gives every batch exactly what a launch
    of its own gives it -- integer work counters identical, tallies equal to the order of the additions -- whatever the
    batch size (shorter than a wavefront, not a multiple of a chunk, one batch only), with absorption (volume tallies), a
    reflecting surface, a slant sun, for the extinction grid in LDS, in global memory, in bricks and as column records."""
    # (round 4) ... and for radiance problems: a local-estimate ray carries its batch, radiances go to the batch's block -- with
    # the roulette, hybrid tables and the contribution limit, through the event ring (several directions) and without it (one)
    rri = dict(useRussianRouletteForIntensity=True, zetaMin=0.3)
    full = dict(rri, intensityMus=[1.0, 0.4, 0.7], intensityPhis=[0.0, 80.0, 250.0], surfaceAlbedo=0.3, useHybridPhaseFunsForIntenCalcs=True,
                hybridPhaseFunWidth=7.0, numOrdersOrigPhaseFunIntenCalcs=1, limitIntensityContributions=True, maxIntensityContribution=0.4)
    problems = [("step cloud, absorbing, surface", cases.step_cloud(ssa=0.9), dict(surfaceAlbedo=0.3), 0.7, [(20000, 7), (50, 5), (1000, 1), (777, 33)]),
                ("step cloud, conservative", cases.step_cloud(nlayers=16), dict(), 1.0, [(30011, 12)]),
                ("radar field (grid in global memory)", cases.radar_cloud(), dict(surfaceAlbedo=0.1), 0.9, [(20000, 5)]),
                ("Landsat field (bricks), absorbing", cases.landsat_cloud(ssa=0.98), dict(), 0.5, [(15000, 3)]),
                ("Landsat field (column records), absorbing", cases.landsat_cloud(ssa=0.98), dict(), 0.5, [(15000, 3)]),
                ("Landsat field (column records), two radiances", cases.landsat_cloud(ssa=0.98), dict(rri, intensityMus=[0.8, 0.3], intensityPhis=[90.0, 225.0], surfaceAlbedo=0.2), 0.5, [(15000, 3)]),
                ("step cloud, three radiances, hybrid tables, limit", cases.step_cloud(ssa=0.95, nlayers=8), full, 0.7, [(20000, 7), (50, 5), (777, 33)]),
                ("step cloud, nadir radiance, plain local estimate", cases.step_cloud(ssa=0.95, nlayers=8), dict(intensityMus=[1.0], intensityPhis=[0.0], surfaceAlbedo=0.2), 0.7, [(20000, 7), (60, 9)]),
                ("radar field, nadir radiance (one direction: no ring)", cases.radar_cloud(), dict(rri, intensityMus=[1.0], intensityPhis=[0.0], surfaceAlbedo=0.1), 0.9, [(20000, 5), (3000, 24)]),
                ("Landsat field (bricks), two radiances", cases.landsat_cloud(ssa=0.98), dict(rri, intensityMus=[0.8, 0.3], intensityPhis=[90.0, 225.0], surfaceAlbedo=0.2), 0.5, [(15000, 3)]),
                # (a ray carries its batch in 13 bits: a loop of more than 8192 tiny batches on a tiny domain is cut into groups within that)
                ("one column, nadir radiance, 8300 batches of 33 photons", cases.plane_parallel(optical_depth=2.0, ssa=0.9), dict(rri, intensityMus=[1.0], intensityPhis=[0.0], surfaceAlbedo=0.3), 0.6, [(33, 8300)])]
    # (round 5) ... and for the widened class: two components (flux: a fused kernel of its own where plain launches run the general flux
    # kernel; radiances with and without the ring), column clouds over a gas (records over a base profile) on an irregular grid, a gridded surface
    t2 = [M.PhaseFunctionTable([M.henyey_greenstein(0.85, 32), M.henyey_greenstein(0.6, 16)]),
          M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])]
    cc = cases.column_clouds()
    gas = np.broadcast_to(np.linspace(3.0e-3, 1.0e-4, cc["ext"].shape[0], dtype=np.float32)[:, None, None], cc["ext"].shape).copy()
    colgas = dict(cc, ext=[cc["ext"], gas], ssa=[cc["ssa"], np.full_like(gas, f32(0.9))], pf=[cc["pf"], np.ones(gas.shape, np.int32)])
    grid = M.new_SurfaceDescription(np.array([[0.1, 0.5], [0.3, 0.7]], np.float32), np.array([0.0, 250.0, 500.0], np.float32), np.array([0.0, 250.0, 500.0], np.float32))
    problems += [("two components, flux (wide)", cases.two_component(), dict(surfaceAlbedo=0.3), 0.7, [(20000, 7), (50, 5), (777, 33)], t2),
                 ("two components, two radiances (wide)", cases.two_component(), dict(rri, intensityMus=[1.0, 0.4], intensityPhis=[0.0, 100.0], surfaceAlbedo=0.3), 0.7, [(15000, 5)], t2),
                 ("two components, one radiance (wide)", cases.two_component(), dict(rri, intensityMus=[0.8], intensityPhis=[200.0], surfaceAlbedo=0.2), 0.7, [(15000, 5), (60, 9)], t2),
                 ("column clouds over a gas, irregular grid, flux (wide)", colgas, dict(surfaceAlbedo=0.2), 0.6, [(20000, 5)], [hg_table(), t2[1]]),
                 ("column clouds over a gas, irregular grid, two radiances (wide)", colgas, dict(rri, intensityMus=[0.6, 1.0], intensityPhis=[20.0, 0.0], surfaceAlbedo=0.3), 0.6, [(15000, 4)], [hg_table(), t2[1]]),
                 ("step cloud over a gridded surface, flux (wide)", cases.step_cloud(ssa=0.95, nlayers=8), dict(surfaceBDRF=grid), 0.7, [(20000, 6)], None),
                 ("step cloud over a gridded surface, nadir radiance (wide)", cases.step_cloud(ssa=0.95, nlayers=8), dict(rri, surfaceBDRF=grid, intensityMus=[1.0], intensityPhis=[0.0]), 0.7, [(20000, 6)], None)]
    for what, d, params, mu0, runs, *tabs in problems:
        g = make_gpu(d, tabs[0] if tabs and tabs[0] is not None else hg_table(), **params)
        if "(bricks)" in what:
            g.select_grid_place("bricks")   # (the scene has column records, which AUTO would read)
        rad = g if "intensityMus" in params else None
        for n, nb in runs:
            g.set_batch_fusion(0)
            one = g.computeRadiativeTransferBatches((5, 3), nb, mu0, 25.0, n)
            assert "PhiloxStream" in g.kernel_name()
            g.set_batch_fusion(1)
            fused = g.computeRadiativeTransferBatches((5, 3), nb, mu0, 25.0, n)
            assert "PhiloxBatchStream" in g.kernel_name() and ("wide" in g.kernel_name()) == ("(wide)" in what), g.kernel_name()
            _same_batches(one, fused, (what, n, nb), rad)
            plain = g.computeRadiativeTransfer(M.new_RandomNumberSequence((5, 3 + nb - 1)), M.new_PhotonStream(mu0, 25.0, n))
            _same_batches([plain], [one[-1]], (what, n, nb, "plain call"))
            if rad is None: _same_batches([plain], [fused[-1]], (what, n, nb, "plain call, fused"))
            assert all(r["counters"]["photons"] == n for r in fused)
        g.finalize_Integrator()


def test_fused_groups_and_chunks_do_not_matter(monkeypatch):
    """Group size (batches per fused launch) and chunk size (photons a wave takes from one batch at a time) only schedule
    work: the same batches come out, whether the loop is one launch or many, the chunks long or short -- for a flux problem
    (every counter per batch) and for a radiance problem (tallies, photons and dropped photons per batch; the radiance kernels'
    other counters over the loop: _same_batches)."""
    import subprocess, sys, json, os
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "import i3rc_monte_carlo_model_amd as M\nfrom tools import cases\nfrom tests.test_gpu_parity import make_gpu, hg_table\n"
            "out = []\n"
            "for params in (dict(surfaceAlbedo=0.2), dict(surfaceAlbedo=0.2, intensityMus=[1.0, 0.5], intensityPhis=[0.0, 70.0], useRussianRouletteForIntensity=True, zetaMin=0.3)):\n"
            "    g = make_gpu(cases.step_cloud(ssa=0.95), hg_table(), **params)\n    g.set_batch_fusion(1)\n"
            "    r = g.computeRadiativeTransferBatches((9, 1), 21, 0.8, 10.0, 4000)\n"
            "    assert 'PhiloxBatchStream' in g.kernel_name()\n"
            "    out.append([int(g.layout().counters), [[x['counters'], [float(v) for v in x['raw']]] for x in r]])\n"
            "print(json.dumps(out))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {"I3RC_FUSED_GROUP_PHOTONS": "9000", "I3RC_FUSED_CHUNK": "64"}, {"I3RC_FUSED_GROUP_PHOTONS": "30000", "I3RC_FUSED_CHUNK": "4096"}):
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    for other in outs[1:]:
        for problem, ((cnt0, base), (cnt1, run)) in enumerate(zip(outs[0], other)):
            for (c0, r0), (c1, r1) in zip(base, run):
                if problem == 0:
                    assert c0 == c1
                    assert_same_sums(r0, r1, c0)
                else:
                    assert c0["photons"] == c1["photons"] == 4000 and c0["dropped"] == c1["dropped"]
                    whole = {k: sum(c[k] for c, _ in base) for k in c0}   # (radiance kernels count per group: the group's counters bound a batch's)
                    assert_same_sums(r0[:cnt0], r1[:cnt1], whole, directions=2)   # (the counter words are the block's last)
            for k in base[0][0]:
                assert sum(c[k] for c, _ in base) == sum(c[k] for c, _ in run), (problem, k)


def test_looking_ahead_in_fused_groups_never_changes_a_batch():
    """The look-ahead of i3rc_hip_compute_batch on a flux problem launches GROUPS of following batches as fused launches
    (8, 16, 32 ... batches): a loop, a jump (the groups are called off through their abort word), another photon count, another
    sun, a parameter change in mid-loop -- every batch equals the plain zero + launch + fetch of the same batch."""
    d = cases.step_cloud(ssa=0.99, nlayers=8)
    params = dict(surfaceAlbedo=0.3)
    plain, ahead = make_gpu(d, hg_table(), **params), make_gpu(d, hg_table(), **params)
    plain.set_batch_fusion(0)

    def both(seed, n, mu0=0.8, az=25.0, look=3):
        a = plain.computeRadiativeTransfer(M.new_RandomNumberSequence(seed), M.new_PhotonStream(mu0, az, n))
        b = ahead.computeRadiativeTransferLookingAhead(M.new_RandomNumberSequence(seed), M.new_PhotonStream(mu0, az, n), lookAhead=look)
        assert a["counters"] == b["counters"], (seed, n)
        # (float64 everywhere, the plain launch's partial sums in LDS included: tests/sums.py.  Round 4 had float32 partial sums there
        # and this line at rtol = 1e-4 after one miss at 1e-5 in ten runs of the suite -- profiles/r05_lds_tallies_f32_vs_f64.txt)
        assert_same_sums(a["raw"], b["raw"], a["counters"], what=(seed, n))

    both((7, 0), 1)                                     # the drivers' one-photon warm-up
    for b in range(1, 40):
        both((7, b), 20000)                             # the loop: served from the first group, then the second, the third
    both((7, 100), 20000)                               # a jump: the groups launched ahead are called off
    both((7, 101), 20000); both((7, 102), 20000); both((7, 103), 20000)
    both((7, 104), 5000)                                # another photon count
    both((7, 105), 5000); both((7, 106), 5000, mu0=0.6)   # another sun in mid-loop
    both((7, 107), 5000, mu0=0.6); both((7, 108), 5000, mu0=0.6)
    for g in (plain, ahead):                            # a parameter changes while groups are under way
        g.specifyParameters(surfaceAlbedo=0.6)
    for b in range(109, 120):
        both((7, b), 5000, mu0=0.6)
    both((8, 120), 5000, mu0=0.6, look=0); both((8, 121), 5000, mu0=0.6, look=0)
    r = ahead.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 121)), M.new_PhotonStream(0.6, 25.0, 5000))
    q = plain.computeRadiativeTransfer(M.new_RandomNumberSequence((8, 121)), M.new_PhotonStream(0.6, 25.0, 5000))
    assert r["counters"] == q["counters"]
    plain.finalize_Integrator(); ahead.finalize_Integrator()


def test_phase_function_entries_beyond_32767():
    """Radiance runs pack the phase-function table entry into the upper half of a ray record's info word: an entry beyond
    32767 must come back as itself (unsigned), not as a negative offset.  A table of 40000 identical entries with every cell
    pointing at the last one gives exactly the run with one entry."""
    d = cases.step_cloud(ssa=1.0, nlayers=8)
    t = hg_table()
    inv, fwd = t.inverse_table(129), t.forward_table(181)
    params = dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 30.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
    res = []
    hg = M.henyey_greenstein(0.85, 64)
    for entries in (1, 40000):
        dd = dict(d, pf=np.full(d["ext"].shape, entries, np.int32))
        g = make_gpu(dd, M.PhaseFunctionTable([hg] * entries), **params)   # (the tables themselves are handed over ready-made)
        g.set_tables(1, inverse=np.repeat(inv, entries, axis=0), forward=np.repeat(fwd, entries, axis=0))
        res.append(g.computeRadiativeTransfer(M.new_RandomNumberSequence((3, 1)), M.new_PhotonStream(0.8, 0.0, 30000)))
        g.finalize_Integrator()
    assert res[0]["counters"] == res[1]["counters"]
    assert_same_sums(res[0]["raw"], res[1]["raw"], res[0]["counters"])



def test_an_announced_loop_is_streamed_and_never_overshot():
    """i3rc_hip_expect_batches tells the library a driver's loop in advance (the shell's computeRadiativeTransferBatches does): the
    batches are traced at once in fused groups and handed over by i3rc_hip_compute_batch one by one -- each equal to the plain
    call of the same batch --; leaving the announced order in mid-loop (another seed) calls the rest off and still gives the
    right batch; a radiance problem is not accepted (its batches go through i3rc_hip_run_batches)."""
    import ctypes as C
    from i3rc_monte_carlo_model_amd import binding as B
    lib = B.load()
    d = cases.step_cloud(ssa=0.97, nlayers=8)
    plain, stream = make_gpu(d, hg_table(), surfaceAlbedo=0.2), make_gpu(d, hg_table(), surfaceAlbedo=0.2)
    plain.set_batch_fusion(0)
    stream._ensure_tables()
    n, nb = 9000, 45
    s = B.Source(); s.kind, s.solarMu, s.solarAzimuth = 0, 0.7, 40.0
    acc = C.c_int(-1)
    assert lib.i3rc_hip_expect_batches(stream._h, 21, 5, nb, n, C.byref(s), C.byref(acc)) == 0 and acc.value == 1
    raw = np.zeros(stream.layout().total, np.float64)

    def check(seed1):
        assert lib.i3rc_hip_compute_batch(stream._h, 21, seed1, n, C.byref(s), 3, raw.ctypes.data_as(B.dp)) == 0, lib.i3rc_hip_last_error(stream._h)
        got = stream.finish(raw.copy())
        want = plain.computeRadiativeTransfer(M.new_RandomNumberSequence((21, seed1)), M.new_PhotonStream(0.7, 40.0, n))
        assert got["counters"] == want["counters"], seed1
        assert_same_sums(got["raw"], want["raw"], want["counters"])

    for b in range(nb):
        check(5 + b)
    check(5 + nb)                       # beyond the announced loop: an ordinary call again
    assert lib.i3rc_hip_expect_batches(stream._h, 21, 100, 40, n, C.byref(s), C.byref(acc)) == 0 and acc.value == 1
    for b in range(7):
        check(100 + b)
    check(300)                          # out of order: the announced rest is called off
    check(301); check(302)
    # a radiance problem is streamed in the same way (round 4: fused launches for radiance problems)
    for g in (plain, stream):
        g.specifyParameters(intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
        g._ensure_tables()
    raw = np.zeros(stream.layout().total, np.float64)
    assert lib.i3rc_hip_expect_batches(stream._h, 21, 5, nb, n, C.byref(s), C.byref(acc)) == 0 and acc.value == 1
    for b in range(nb):
        assert lib.i3rc_hip_compute_batch(stream._h, 21, 5 + b, n, C.byref(s), 3, raw.ctypes.data_as(B.dp)) == 0, lib.i3rc_hip_last_error(stream._h)
        got = stream.finish(raw.copy())
        want = plain.computeRadiativeTransfer(M.new_RandomNumberSequence((21, 5 + b)), M.new_PhotonStream(0.7, 40.0, n))
        assert got["counters"]["photons"] == want["counters"]["photons"] == n
        assert_same_sums(_tallies_only(stream, got), _tallies_only(plain, want), want["counters"], directions=8, what=b)
        assert got["intensity"].mean() > 0
    # a gridded surface is streamed too since round 5 (the widened class has fused kernels of its own) ...
    for g in (plain, stream):
        g.specifyParameters(surfaceBDRF=M.new_SurfaceDescription(np.array([[0.1, 0.3], [0.2, 0.4]], np.float32), np.array([0.0, 250.0, 500.0], np.float32),
                                                                 np.array([0.0, 250.0, 500.0], np.float32)))
        g._ensure_tables()
    assert lib.i3rc_hip_expect_batches(stream._h, 21, 5, 6, n, C.byref(s), C.byref(acc)) == 0 and acc.value == 1
    for b in range(6):
        assert lib.i3rc_hip_compute_batch(stream._h, 21, 5 + b, n, C.byref(s), 3, raw.ctypes.data_as(B.dp)) == 0, lib.i3rc_hip_last_error(stream._h)
        got = stream.finish(raw.copy())
        assert "wide" in stream.kernel_name(), stream.kernel_name()
        want = plain.computeRadiativeTransfer(M.new_RandomNumberSequence((21, 5 + b)), M.new_PhotonStream(0.7, 40.0, n))
        assert got["counters"]["photons"] == want["counters"]["photons"] == n
        assert_same_sums(_tallies_only(stream, got), _tallies_only(plain, want), want["counters"], directions=8, what=("gridded surface", b))
    # ... a problem of the general kernels (max cross-section) is left to i3rc_hip_run_batches
    stream.specifyParameters(useRayTracing=False)
    stream._ensure_tables()
    assert lib.i3rc_hip_expect_batches(stream._h, 21, 5, nb, n, C.byref(s), C.byref(acc)) == 0 and acc.value == 0
    plain.finalize_Integrator(); stream.finalize_Integrator()


def test_batch_moments_on_the_device_equal_the_drivers_sums():
    """i3rc_hip_run_batches_moments: the sums and sums of squares over a loop's batches of everything reportResults hands out,
    normalised and added up on the device, against the same batches fetched block by block (i3rc_hip_run_batches), normalised
    by i3rc_hip_normalise and summed the way the drivers do (monteCarloDriver.f95:300-321) -- for a flux problem in fused
    launches (LDS grid), a radiance problem with hybrid tables and limited contributions (the excess redistribution :327-347),
    a one-direction radiance problem on the radar field, and a two-component domain (the widened-class kernels: one launch per
    batch)."""
    rri = dict(useRussianRouletteForIntensity=True, zetaMin=0.3)
    full = dict(rri, intensityMus=[1.0, 0.4, 0.7], intensityPhis=[0.0, 80.0, 250.0], surfaceAlbedo=0.3, useHybridPhaseFunsForIntenCalcs=True,
                hybridPhaseFunWidth=7.0, numOrdersOrigPhaseFunIntenCalcs=1, limitIntensityContributions=True, maxIntensityContribution=0.4)
    t2 = [M.PhaseFunctionTable([M.henyey_greenstein(0.85, 32), M.henyey_greenstein(0.6, 16)]),
          M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])]
    problems = [("step cloud, absorbing, surface", cases.step_cloud(ssa=0.9), hg_table(), dict(surfaceAlbedo=0.3), 20000, 11, "PhiloxBatchStream, false"),
                ("step cloud, radiances, hybrid tables, limit", cases.step_cloud(ssa=0.95, nlayers=8), hg_table(), full, 8000, 9, "PhiloxBatchStream, true"),
                ("radar field, nadir", cases.radar_cloud(), hg_table(), dict(rri, intensityMus=[1.0], intensityPhis=[0.0], surfaceAlbedo=0.1), 15000, 5, "one direction"),
                ("two components, irregular grid", cases.two_component(), t2, dict(rri, intensityMus=[0.6], intensityPhis=[30.0], surfaceAlbedo=0.2), 6000, 7, "one direction, wide")]
    for what, d, tab, params, n, nb, kernel in problems:
        g = make_gpu(d, tab, **params)
        s1, s2, cnt = g.computeRadiativeTransferBatchMoments((5, 3), nb, 0.7, 25.0, n)
        assert kernel in g.kernel_name(), (what, g.kernel_name())
        assert cnt["photons"] == n * nb
        rs = g.computeRadiativeTransferBatches((5, 3), nb, 0.7, 25.0, n)
        per = []
        for r in rs:
            g._results = r
            per.append(g.reportResults())
        assert sum(r["counters"]["scatterings"] for r in rs) == cnt["scatterings"]
        for key in s1:
            x = np.stack([np.asarray(p[key], np.float64) for p in per])
            want1, want2 = x.sum(0), (x * x).sum(0)
            # (the device sums the same float32 values in float64; domain means: Fortran's sum() in real(4) against float64 rounded once)
            # (float64 all the way, a workgroup's partial sums in LDS included: round 5)
            tol = 2e-6 if key.startswith("mean") or key == "absorbedProfile" else 1e-9
            assert np.allclose(s1[key], want1, rtol=tol, atol=1e-12), (what, key, np.abs(s1[key] - want1).max())
            assert np.allclose(s2[key], want2, rtol=2 * tol, atol=1e-12), (what, key)
        g.finalize_Integrator()
