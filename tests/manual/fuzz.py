"""Fuzz the C ABI on the GPU box: random domains (regular / irregular, on the ground / thin and elevated, 1-3 D, holes,
one or two components), random parameters (ray tracing / max cross-section, roulettes, hybrid phase function,
contribution limit, Lambertian albedo or BRDF grid, radiance directions up and down), random sources (directional or
explicit photons anywhere in the domain).  Every configuration is written to gpurun_out/fuzz.log BEFORE its launch, so a
fault names its configuration; run with I3RC_POISON=1 so that reads of unwritten device memory show.  Checked per
configuration: energy conservation of the tallies, finite radiances, and identical integer work counters under a
second schedule.  usage: fuzz.py [first seed] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: F401
import i3rc_monte_carlo_model_amd as M
from tools import cases
from tests.sums import order_rtol
from tests.test_gpu_parity import hg_table, make_gpu

if os.environ.get('I3RC_LIB'):   # another build of the library
    M.build.LIB = os.path.abspath(os.environ['I3RC_LIB']); M.build.needs_build = lambda: False
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
log = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "fuzz.log"), "a")
def note(*a):
    print(*a, file=log, flush=True); os.fsync(log.fileno())

t_cloud = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 32), M.henyey_greenstein(0.6, 16)])
t_gas = M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])
KEYS = ("cellSteps", "scatterings", "surfaceHits", "exitsTop", "roulette", "shadowSteps", "tracerCalls", "dropped")
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kind = rng.choice(["regular", "irregular", "two"])
    big = os.environ.get("BIG") == "1" or rng.random() < 0.04   # grids that live in global memory / in bricks (BIG=1: all)
    if kind == "two":
        d = cases.two_component(seed=seed, nx=int(rng.integers(1, 9)), ny=int(rng.integers(1, 6)), nz=8); tab = [t_cloud, t_gas]
    elif kind == "irregular":
        d = cases.irregular_domain(seed=seed, nx=int(rng.integers(1, 12)), ny=int(rng.integers(1, 8)), nz=int(rng.integers(1, 14)),
                                   ssa=float(rng.choice([1.0, 0.95, 0.5])), z0=float(rng.choice([0.0, 0.0, 100.0, 5000.0])))
        tab = hg_table(float(rng.choice([0.0, 0.85, 0.95])), 64)
    else:
        d = cases.step_cloud(ssa=float(rng.choice([1.0, 0.99, 0.6])), nlayers=int(rng.integers(1, 20)))
        if rng.random() < 0.3:   # lift it: the thin elevated case on a regular grid
            d["ze"] = (d["ze"] + np.float32(rng.choice([300.0, 20000.0]))).astype(np.float32)
        tab = hg_table(float(rng.choice([0.0, 0.85])), 64)
    if big:
        kind = "regular"
        nx, ny, nz = [(48, 40, 12), (97, 3, 40), (110, 100, 100), (64, 64, 54)][int(rng.integers(0, 4))]   # 92 KB ... 4.4 MB of extinction
        d = dict(xe=(np.float32(rng.uniform(20, 200)) * np.arange(nx + 1)).astype(np.float32),
                 ye=(np.float32(rng.uniform(20, 200)) * np.arange(ny + 1)).astype(np.float32),
                 ze=((np.float32(rng.uniform(10, 100)) * np.arange(nz + 1)) + np.float32(rng.choice([0.0, 0.0, 20000.0]))).astype(np.float32))
        ext = rng.uniform(0.0, 0.02, (nz, ny, nx)).astype(np.float32)
        ext[rng.uniform(size=ext.shape) < 0.4] = 0.0
        w0 = np.float32(rng.choice([1.0, 0.9]))
        d.update(ext=ext, ssa=np.where(ext > 0, w0, np.float32(0)).astype(np.float32), pf=np.where(ext > 0, 1, 0).astype(np.int32))
        tab = hg_table(float(rng.choice([0.0, 0.85])), 64)
    # (round 4) where the kernels read the extinction field from -- LDS / plain / bricks / column records, i3rc_hip_select_grid_place --
    # is drawn from a stream of its own (the configurations of earlier rounds keep their seeds): one seed in seven traces a field of
    # column clouds (one run of one value per column: the form that has column records), and the two schedules below read the field
    # from two places drawn at random: the integer work counters must not notice
    aux = np.random.default_rng(seed + 7_000_003)
    if not big and kind != "two" and aux.random() < 0.15:
        kind = "irregular"
        d = cases.column_clouds(seed=seed, nx=int(aux.integers(1, 12)), ny=int(aux.integers(1, 8)), nz=int(aux.integers(1, 14)), ssa=float(aux.choice([1.0, 0.95, 0.5])))
        tab = hg_table(float(aux.choice([0.0, 0.85, 0.95])), 64)
    places = [str(aux.choice(["auto", "auto", "linear", "bricks", "columns"])) for _ in range(2)]
    # (round 5) from a third stream: one seed in eight traces column clouds OVER A GAS (two components whose sum has column records over a
    # base profile), and the second schedule of every configuration runs the GENERAL kernels one time in three -- the widened-class
    # kernels (several components, irregular x / y, gridded surface) must trace the general kernels' photons: the counters say
    aux5 = np.random.default_rng(seed + 9_000_011)
    unequal_columns = kind == "irregular"
    if not big and kind != "two" and aux5.random() < 0.125:
        kind = "two"; unequal_columns = True   # (column_clouds is irregularly spaced: column means do not add up to the energy)
        c = cases.column_clouds(seed=seed, nx=int(aux5.integers(1, 12)), ny=int(aux5.integers(1, 8)), nz=int(aux5.integers(2, 14)), ssa=float(aux5.choice([1.0, 0.95])))
        gas = np.broadcast_to(np.linspace(float(aux5.choice([3.0e-3, 2.0e-5])), 1.0e-5, c["ext"].shape[0], dtype=np.float32)[:, None, None], c["ext"].shape).copy()
        d = dict(c, ext=[c["ext"], gas], ssa=[c["ssa"], np.full_like(gas, np.float32(aux5.choice([1.0, 0.8])))], pf=[c["pf"], np.ones(gas.shape, np.int32)])
        tab = [hg_table(0.85, 64), t_gas]
    second_kernel = "general" if aux5.random() < 0.34 else "auto"
    p = {}
    if rng.random() < 0.4: p["useRayTracing"] = False
    if rng.random() < 0.3: p["useRussianRoulette"] = False
    nd = int(rng.choice([0, 0, 1, 2, 4]))
    if nd:
        p["intensityMus"] = [float(v) for v in rng.uniform(0.05, 1.0, nd) * rng.choice([-1, 1], nd)]
        p["intensityPhis"] = [float(v) for v in rng.uniform(0, 360, nd)]
        if rng.random() < 0.5: p.update(useRussianRouletteForIntensity=True, zetaMin=float(rng.choice([0.1, 0.3, 1.0])))
        if rng.random() < 0.3: p.update(useHybridPhaseFunsForIntenCalcs=True, hybridPhaseFunWidth=7.0, numOrdersOrigPhaseFunIntenCalcs=int(rng.integers(0, 3)))
        if rng.random() < 0.3: p.update(limitIntensityContributions=True, maxIntensityContribution=float(rng.choice([0.1, 1.0])))
    s = rng.random()
    if s < 0.5: p["surfaceAlbedo"] = float(rng.choice([0.0, 0.3, 1.0]))
    elif s < 0.8:
        nxs, nys = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        xs = np.linspace(d["xe"][0], d["xe"][-1], nxs + 1).astype(np.float32); ys = np.linspace(d["ye"][0], d["ye"][-1], nys + 1).astype(np.float32)
        alb = rng.uniform(0, 1, (1, nxs, nys)).astype(np.float32)
        p["surfaceBDRF"] = M.new_SurfaceDescription(alb, xs, ys) if nxs * nys > 1 else M.new_SurfaceDescription(alb.reshape(1))
    for k in os.environ.get("DROP", "").split(","):   # bisecting a configuration: DROP=key,key  NDIRS=k
        if k == "dirs" and "intensityMus" in p: nd = 0; p.pop("intensityMus"); p.pop("intensityPhis")
        elif k in p: p.pop(k)
    if "NDIRS" in os.environ and nd:
        keep = [int(v) for v in os.environ["NDIRS"].split(",")]
        p["intensityMus"] = [p["intensityMus"][i] for i in keep]; p["intensityPhis"] = [p["intensityPhis"][i] for i in keep]; nd = len(keep)
    n = int(rng.choice([1, 63, 1000, 30000]))
    if "N" in os.environ: n = int(os.environ["N"])   # a closer look at one seed with a larger sample
    mu0, az = float(rng.uniform(0.05, 1.0)), float(rng.uniform(0, 360))
    explicit = rng.random() < 0.3
    if explicit:   # photons anywhere: relative positions in [0, 1], any direction but horizontal
        arr = [rng.random(n), rng.random(n), rng.random(n), rng.uniform(0.05, 1.0, n) * rng.choice([-1, 1], n), rng.uniform(0, 2 * np.pi, n)]
    note("seed", seed, kind, places, {k: (v if not hasattr(v, "albedo") else "BRDF") for k, v in p.items()}, "n", n, "explicit" if explicit else (mu0, az),
         "shape", d["ext"][0].shape if isinstance(d["ext"], list) else d["ext"].shape, "z", float(d["ze"][0]), float(d["ze"][-1]))
    if os.environ.get("REPLAY") == "1":
        # the replay build against the oracle photon by photon (reference deviates in reference order): same fate, exit
        # column, scattering order, number of deviates; radiance sums within 2 % + the share of diverged photons
        from oracle import pyoracle as O
        from tests.test_gpu_features import _intensity_pair, _replay_pair
        O.build()
        gp = {k: v for k, v in p.items() if k not in ("intensityMus", "intensityPhis", "surfaceBDRF")}
        op = dict(surfaceAlbedo=gp.get("surfaceAlbedo", 0.0), useRayTracing=int(gp.get("useRayTracing", True)),
                  useRussianRoulette=int(gp.get("useRussianRoulette", True)), useRRForIntensity=int(gp.get("useRussianRouletteForIntensity", False)),
                  zetaMin=gp.get("zetaMin", 0.3), useHybrid=int(gp.get("useHybridPhaseFunsForIntenCalcs", False)),
                  numOrdersOrig=gp.get("numOrdersOrigPhaseFunIntenCalcs", 0), limitContrib=int(gp.get("limitIntensityContributions", False)))
        if "maxIntensityContribution" in gp: op["maxContrib"] = gp["maxIntensityContribution"]
        if "surfaceBDRF" in p:
            sd = p["surfaceBDRF"]
            if nxs * nys > 1:
                gp["surfaceBDRF"] = sd; op["surfaceBDRF"] = (xs, ys, np.ascontiguousarray(alb[0].T)); op.pop("surfaceAlbedo")
            else:
                gp["surfaceAlbedo"] = op["surfaceAlbedo"] = float(alb.reshape(-1)[0])
        mus = p.get("intensityMus", [1.0]); phis = p.get("intensityPhis", [0.0])
        try:
            g, o = _intensity_pair(O, d, tab, n_table=2001, gpu_params=gp, oracle_params=op, mus=mus, phis=phis,
                                   hybrid_width=7.0 if op["useHybrid"] else None)
        except M.I3RCError as e:
            note("   rejected:", e); continue
        m = min(n, 1500)
        ref, out = _replay_pair(O, g, o, m, [seed, 3], mu0, az)
        same = (out["fate"] == ref["fate"]) & (out["fateColumn"] == ref["fateColumn"]) & (out["fateOrder"] == ref["fateOrder"]) & \
               (out["drawsUsed"] == np.diff(ref["drawStart"]))
        lay = g.layout(); ncol = g.nx * g.ny
        gi = float(out["raw"][lay.intensityByComponent:lay.intensityByComponent + (g.ncomp + 1) * len(mus) * ncol].sum())
        ri = float(np.asarray(ref["intensityByComp"], np.float64).sum())
        problems = []
        # (photons part ways through 1-ulp differences of log / cos / acos between device and glibc: the more events a
        # photon lives through -- mirror surfaces, no roulette -- the more of them do; "same" cannot see a photon that
        # chose another component and still ended alike, hence the weights are compared as a share too)
        events = max(1.0, float(np.mean(np.diff(ref["drawStart"]))) / 4.0)
        if os.environ.get("SHOW") == "1":
            for i in np.nonzero(~same)[0][:16]:
                note("      photon", int(i), "fate", int(out["fate"][i]), int(ref["fate"][i]), "column", int(out["fateColumn"][i]), int(ref["fateColumn"][i]),
                     "order", int(out["fateOrder"][i]), int(ref["fateOrder"][i]), "draws", int(out["drawsUsed"][i]), int(np.diff(ref["drawStart"])[i]))
        if os.environ.get("SHOW") == "1":   # photons that ended alike but with another weight
            for i in np.nonzero(same & (out["fateWeight"] != ref["fateWeight"]))[0][:12]:
                note("      photon", int(i), "fate", int(ref["fate"][i]), "order", int(ref["fateOrder"][i]), "draws", int(out["drawsUsed"][i]),
                     "weight gpu %.9g ref %.9g ratio %.7f" % (out["fateWeight"][i], ref["fateWeight"][i], out["fateWeight"][i] / max(ref["fateWeight"][i], 1e-38)))
        same &= out["fateWeight"] == ref["fateWeight"]
        # (optically thin media amplify a 1-ulp difference of the sampled optical depth to 1e-4 m of path: after some
        # tens of events the positions are decimetres apart and a photon in fifty leaves through a neighbouring column)
        # (...and where thin and thick cells alternate the separation grows by the ratio of the extinctions at every
        # event: half of the photons of a 40-event life may end in another column.  Large grids are such media here.)
        if not big and same.mean() < 1.0 - 0.006 * events - 0.01 and (~same).sum() > 6: problems.append(("replay agreement", float(same.mean()), events))
        # (a photon that parted ways carries its own contributions: up to 1 / mu of a grazing direction each)
        # (sums of contributions of both signs -- truncated Legendre series go negative -- are not compared: they cancel)
        if ri > 0 and gi > 0 and abs(gi - ri) > (0.02 + 40.0 * (1.0 - same.mean())) * max(abs(ri), 1e-3) + 1e-6 + (0.5 * ri if m <= 1000 else 0.0):
            problems.append(("radiance sums", gi, ri))   # (small samples: one diverged photon can carry a third of the sum)
        note("   ok" if not problems else "   PROBLEM", problems, "identical %.4f of %d, radiance sums %.6g %.6g" % (same.mean(), m, gi, ri))
        bad += bool(problems)
        continue
    res = []
    for place, tune in zip(places, (dict(evThreshold=0), dict(evThreshold=int(rng.choice([1, 8, 64])), blocksPerCU=1, lightThreshold=int(rng.choice([1, 16, 64]))))):
        try:
            g = make_gpu(d, tab, **p)
        except M.I3RCError as e:
            note("   rejected:", e); res = None; break
        g.set_tuning(**tune)
        if len(res) == 1: g.set_tuning(tune["evThreshold"], tune.get("blocksPerCU", 0), kernel=second_kernel, lightThreshold=tune.get("lightThreshold"))
        g.select_grid_place(place if place != "columns" or g.has_column_records() else "linear")
        src = M.PhotonStream(arrays=arr) if explicit else M.new_PhotonStream(mu0, az, n)
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((seed, 1)), src)
        res.append(r)
    if not res: continue
    r = res[0]
    c0, c1 = ({k: x["counters"][k] for k in KEYS} for x in res)
    entry = []
    if not explicit:   # the same batch through the pipelined and the looking-ahead entry points (g: the second schedule's handle)
        many = g.computeRadiativeTransferBatches((seed, 0), 3, mu0, az, n, inFlight=int(rng.integers(1, 5)))
        # (radiance problems in fused launches count per wave: photons and dropped photons are exact per batch, the other counters
        # over the batches of a group -- include/i3rc_hip.h, i3rc_hip_run_batches; the tallies are per batch either way)
        keys = ("dropped",) if nd > 0 else KEYS
        lay = g.layout()
        for b in (0, 1, 2):
            q = g.computeRadiativeTransferLookingAhead(M.new_RandomNumberSequence((seed, b)), M.new_PhotonStream(mu0, az, n), lookAhead=int(rng.integers(0, 4)))
            if {k: q["counters"][k] for k in keys} != {k: many[b]["counters"][k] for k in keys}: entry.append(("look-ahead / pipelined", b))
            # (float64 all the way, a workgroup's partial sums in LDS included: two runs of a batch differ by the order of float64
            # additions -- tests/sums.py; the plain run's counters, times four, bound the additions of any batch of this size)
            qa, qb = q["raw"][:lay.counters], many[b]["raw"][:lay.counters]
            if (np.abs(qa - qb) > 4.0 * order_rtol(dict(c0, photons=n), nd) * (np.abs(qa) + 1e-3 * np.abs(qa).max())).any(): entry.append(("look-ahead / pipelined tallies", b, float(np.abs(qa - qb).max())))
        if many[1]["counters"]["photons"] != n or {k: many[1]["counters"][k] for k in keys} != {k: c0[k] for k in keys}: entry.append(("pipelined / plain", {k: many[1]["counters"][k] for k in keys}, c0))
    tot = float(r["fluxUp"].mean() + r["fluxAbsorbed"].mean()) + (float(r["fluxDown"].mean()) * (1.0 - p["surfaceAlbedo"]) if "surfaceAlbedo" in p else 0.0)
    problems = list(entry)
    if c0 != c1: problems.append(("schedule", c0, c1))
    if not all(np.isfinite(r[k]).all() for k in ("fluxUp", "fluxDown", "fluxAbsorbed", "volumeAbsorption")): problems.append("non-finite flux")
    if nd and not np.isfinite(r["intensity"]).all(): problems.append("non-finite radiance")
    drop = c0["dropped"] / n
    # (column means add up to the photons' energy on grids of equal columns only: not on the irregular ones)
    if not unequal_columns and "surfaceAlbedo" in p and p.get("useRussianRoulette", True) is False and abs(tot + drop - 1.0) > 0.02 + 3.0 / np.sqrt(n):
        problems.append(("energy", tot, drop))
    if os.environ.get("ORACLE") == "1" and not explicit and "surfaceBDRF" not in p:
        # domain-mean fluxes against the CPU oracle (its own MT19937 stream: 5 sigma of the two samples)
        from oracle import pyoracle as O
        from tests.test_gpu_parity import make_oracle
        O.build()
        tabs = tab if isinstance(tab, list) else [tab]
        o = make_oracle(O, d, [t.inverse_table(2001) for t in tabs])
        o.specify(useRayTracing=int(p.get("useRayTracing", True)), useRussianRoulette=int(p.get("useRussianRoulette", True)), surfaceAlbedo=p.get("surfaceAlbedo", 0.0))
        m = max(n, 4000)
        rr = O.RandomNumberSequence([seed, 7]); ph = O.photons_directional(rr, mu0, az, m)
        ro = o.compute(rr, *ph)
        for k in ("fluxUp", "fluxDown", "fluxAbsorbed"):
            a, b = float(r[k].mean()), float(ro[k].mean())
            # (crude standard errors: binomial-like, widened where the tallies are heavy-tailed -- irregular columns, and
            # the downward flux over a bright surface, which counts every one of a photon's surface hits)
            tol = 5.0 * np.sqrt(max(a, b, 0.02) * (1.0 / n + 1.0 / m)) * (3.0 if unequal_columns else 1.0) + 1e-3
            if k == "fluxDown" and p.get("surfaceAlbedo", 0.0) > 0.5: tol *= 4.0
            if abs(a - b) > tol: problems.append(("oracle", k, a, b, tol))
    note("   ok" if not problems else "   PROBLEM", problems, "up %.4f down %.4f abs %.4f dropped %.4f" % (r["fluxUp"].mean(), r["fluxDown"].mean(), r["fluxAbsorbed"].mean(), drop))
    bad += bool(problems)
note("fuzz done", first, count, "problems", bad)
print("fuzz done", first, count, "problems", bad)
