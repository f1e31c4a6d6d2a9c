"""Large-sample parity report: every BASELINE workload on the GPU at (or near) its BASELINE.json size against the CPU
oracle on all host cores, the oracle's sample produced by a child program while the GPU traces its own.  Prints, per
workload, the domain means with their standard errors, the z score of every difference (GPU - oracle, in units of the
combined standard error), the share of columns within 3 sigma and the largest column |z|.  The tests do the same at
1e6-2e6 photons (tests/test_gpu_baseline_configs.py); this is the long version, run by hand on the GPU box:
    python tests/manual/parity_large.py [workload ...] > gpurun_out/parity_large.txt      (PARITY_SCALE=4: four times the sample)
(test infrastructure: the only place the oracle is used is as the checker.)"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import workloads as W  # noqa: E402

# workload -> (GPU batches, photons per GPU batch, oracle batches per core, photons per oracle batch): 16 cores assumed
PLAN = {
    "step16": (100, 1_000_000, 8, 400_000),          # 1e8 on the GPU, 5.1e7 in the oracle (~7 s)
    "radar64_nadir": (100, 1_000_000, 8, 150_000),   # 1e8 / 1.9e7 (~10 s)
    "landsat36": (100, 1_000_000, 8, 250_000),       # 1e8 / 3.2e7 (~9 s)
    "landsat36_absorbing": (100, 1_000_000, 8, 250_000),   # omega = 0.99 in the cloudy cells: absorption tallies, the shared-omega argument
    "landsat119_7dir": (100, 250_000, 8, 40_000),    # 2.5e7 / 5.1e6 (~20 s)
    # (round 5) beyond the common class: two components (general flux kernel / widened-class radiance kernels, column records over a base
    # profile), an irregular x / y grid, a gridded surface
    "landsat36_gas_absorbing": (100, 1_000_000, 8, 200_000),
    "landsat119_gas": (100, 1_000_000, 8, 150_000),
    "landsat119_gas_7dir": (100, 250_000, 8, 30_000),
    "landsat119_irregular_7dir": (100, 250_000, 8, 40_000),
    "landsat119_brdfgrid_7dir": (100, 250_000, 8, 40_000),
    # (round 5, after the absorption tallies were rewritten: the cell's tally only, in LDS or merged per run, column sums formed afterwards)
    # (small oracle batches: the reference -- and so the oracle -- tallies in float32, and adding increments of 0.01 to a column's sum of
    # 2000 rounds each of them to 2.4e-4: batches of 4e5 photons on these 32 columns read fluxAbsorbed 1.2e-4 too high, the energy balance
    # says -- up + down + absorbed + dropped = 1.000121, 1.000012 with batches of 5e4, 0.999999 with 5e3 -- and the GPU's float64 sums,
    # which close to 1 - dropped at any size, were 5 sigma away from them at 6e8 photons: profiles/r05_parity_large_absorbing.txt)
    "step16_absorbing": (100, 1_000_000, 640, 5_000),
    "landsat119_absorbing": (100, 1_000_000, 8, 150_000),
    "les_stcu_rayleigh": (100, 1_000_000, 8, 300_000),
    "landsat36_aerosol_gas": (100, 1_000_000, 8, 200_000),   # three components: cell records of 32 bytes
}
# (round 4) per-column fields of every workload (the oracle child saves them whatever the size of the domain), and config 4 also
# against the oracle's committed fixture of 2.4e7 photons (tests/golden/config4_columns.npz), with 1e8 photons on the GPU


def z_of(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    se = np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b))
    return a.mean(), b.mean(), se, (a.mean() - b.mean()) / se if se > 0 else 0.0


def columns(gpu, ref, floor=1e-7):
    g, r = np.stack(gpu).astype(np.float64), np.stack(ref).astype(np.float64)
    se = np.sqrt(g.var(0, ddof=1) / len(g) + r.var(0, ddof=1) / len(r))
    z = np.abs(g.mean(0) - r.mean(0)) / np.maximum(se, floor)
    return float((z <= 3.0).mean()), float(z.max()), z.size


def main():
    import i3rc_monte_carlo_model_amd as M

    names = sys.argv[1:] or list(PLAN)
    cores = min(16, len(os.sched_getaffinity(0)))
    worst = 0.0
    scale = int(os.environ.get("PARITY_SCALE", "1"))   # PARITY_SCALE=4: four times the batches on both sides (half the standard errors)
    for name in names:
        nb, n, per_core, n_ref = PLAN[name]
        nb, per_core = nb * scale, per_core * scale
        _, w = W.get(name)
        out = os.path.join(tempfile.mkdtemp(), f"oracle_{name}.npz")
        t0 = time.perf_counter()
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--config", name, "--cores", str(cores),
                                  "--batches-per-core", str(per_core), "--photons", str(n_ref), "--save", out, "--save-columns"],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        g, _ = W.make_integrator(w)
        rs = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((91, b)), M.new_PhotonStream(w["mu0"], 0.0, n)) for b in range(1, nb + 1)]
        kernel = g.kernel_name()
        g.finalize_Integrator()
        t_gpu = time.perf_counter() - t0
        so, se = child.communicate(timeout=1500)
        if child.returncode != 0:
            print(f"{name}: oracle child failed\n{so}{se}")
            sys.exit(1)
        t_all = time.perf_counter() - t0
        z = np.load(out)
        nd = len(w["params"].get("intensityMus", []))
        print(f"== {name}: GPU {nb} x {n} = {nb * n:.3g} photons ({kernel}, {t_gpu:.1f} s incl. set-up); "
              f"oracle {len(z['means'])} x {n_ref} = {len(z['means']) * n_ref:.3g} photons on {cores} cores ({t_all:.1f} s)")
        for k, key in enumerate(("fluxUp", "fluxDown", "fluxAbsorbed")):
            a, b, s, zz = z_of([r[key].mean(dtype=np.float64) for r in rs], z["means"][:, k])
            worst = max(worst, abs(zz))
            print(f"   mean {key:12s} GPU {a:.6f}  oracle {b:.6f}  combined se {s:.2e}  z {zz:+.2f}")
        for d in range(nd):
            a, b, s, zz = z_of([r["intensity"][d].mean(dtype=np.float64) for r in rs], z["intensityMeans"][:, d])
            worst = max(worst, abs(zz))
            print(f"   mean radiance {d} (mu {w['params']['intensityMus'][d]:+.1f}, phi {w['params']['intensityPhis'][d]:5.1f})"
                  f"  GPU {a:.6f}  oracle {b:.6f}  combined se {s:.2e}  z {zz:+.2f}")
        for key in ("fluxUp", "fluxDown", "fluxAbsorbed") + (("intensity",) if nd else ()):
            if key in z.files:
                frac, zmax, cnt = columns([r[key] for r in rs], list(z[key]))
                print(f"   columns {key:10s} {cnt} values: {100 * frac:.2f} % within 3 sigma, max |z| {zmax:.2f}")
        n_g, n_o = nb * n, n_ref * len(z["nBad"])
        dg, do = sum(r["counters"]["dropped"] for r in rs) / n_g, z["nBad"].sum() / n_o
        kg, ko = sum(r["counters"]["scatterings"] for r in rs) / n_g, z["scatterings"].sum() / n_o
        sg = sum(r["counters"]["cellSteps"] + r["counters"]["shadowSteps"] for r in rs) / n_g
        print(f"   per photon: dropped GPU {dg:.3e} oracle {do:.3e}; scatterings {kg:.4f} / {ko:.4f}; tracer steps {sg:.2f} / "
              f"{z['cellSteps'].sum() / n_o:.2f} (GPU leaves out rays whose roulette is lost before the trace: "
              f"{sum(r['counters']['raysSkipped'] for r in rs) / n_g:.3f} per photon)", flush=True)
    if "landsat119_7dir" in names and scale == 1:
        fixture_columns(M)
    print(f"largest |z| of a domain mean: {worst:.2f}")


def fixture_columns(M):
    """Config 4 at 1e8 photons on the GPU against the oracle's fixture, column by column."""
    from scipy import stats
    z = np.load(os.path.join(ROOT, "tests", "golden", "config4_columns.npz"))
    _, w = W.get("landsat119_7dir")
    g, _ = W.make_integrator(w)
    nb, n = 100, 1_000_000
    t0 = time.perf_counter()
    acc1 = acc2 = 0.0
    for b in range(1, nb + 1):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((92, b)), M.new_PhotonStream(w["mu0"], 0.0, n))
        f = np.concatenate([r["fluxUp"][None], r["fluxDown"][None], r["intensity"]]).astype(np.float64)
        acc1 = acc1 + f; acc2 = acc2 + f * f
    g.finalize_Integrator()
    mg = acc1 / nb
    sg = np.sqrt(np.maximum(acc2 / nb - mg * mg, 0.0) / (nb - 1))
    # The same statistic with the GPU in the fixture's place: 48 batches of 5e5 GPU photons (other seeds) against the 1e8 -- two
    # samples of ONE code, as unequal as fixture and GPU are.  A sample of 48 skewed batch means against a much larger one is nearly a
    # one-sample t statistic, which skewness inflates (the balanced odd-against-even check in the fixture, selfCheck, is not):
    # what this pair shows is what the statistic does by itself; what the fixture shows beyond it would be a difference.
    g, _ = W.make_integrator(w)
    nbf, nf = int(z["batches"]), int(z["photonsPerBatch"])
    f1 = f2 = 0.0
    for b in range(1, nbf + 1):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((93, b)), M.new_PhotonStream(w["mu0"], 0.0, nf))
        f = np.concatenate([r["fluxUp"][None], r["fluxDown"][None], r["intensity"]]).astype(np.float64)
        f1 = f1 + f; f2 = f2 + f * f
    g.finalize_Integrator()
    mf = f1 / nbf
    sf = np.sqrt(np.maximum(f2 / nbf - mf * mf, 0.0) / (nbf - 1))
    # Welch-Satterthwaite: with the GPU's standard errors half the fixture's, the fixture's 47 degrees of freedom dominate
    dof = nbf - 1
    print(f"== landsat119_7dir against tests/golden/config4_columns.npz: GPU {nb} x {n} = {nb * n:.3g} photons ({time.perf_counter() - t0:.1f} s), "
          f"oracle fixture {nbf} x {nf} = {nbf * nf:.3g} photons; Student's t with about {dof} degrees of freedom expects "
          f"{100 * (1 - 2 * stats.t.sf(3.0, dof)):.2f} % within 3 sigma and mean z^2 {dof / (dof - 2):.3f} of Gaussian batch means")
    for k, name in enumerate(z["fieldNames"]):
        zz = np.abs(mg[k] - z["mean"][k]) / (np.sqrt(sg[k] ** 2 + z["stderr"][k].astype(np.float64) ** 2) + 1e-7)
        zs = np.abs(mg[k] - mf[k]) / (np.sqrt(sg[k] ** 2 + sf[k] ** 2) + 1e-7)
        sc = z["selfCheck"][k] if "selfCheck" in z.files else (np.nan, np.nan, np.nan)
        print(f"   columns {str(name):10s} {zz.size} values: {100 * (zz <= 3.0).mean():.2f} % within 3 sigma, max |z| {zz.max():.2f}, mean z^2 {np.mean(zz ** 2):.3f} "
              f"| GPU sample of the fixture's size in its place: {100 * (zs <= 3.0).mean():.2f} %, {zs.max():.2f}, {np.mean(zs ** 2):.3f} "
              f"| oracle odd against even batches: {100 * sc[0]:.2f} %, {sc[2]:.2f}, {sc[1]:.3f}; domain mean GPU {mg[k].mean():.6f} fixture {z['mean'][k].mean(dtype=np.float64):.6f}")


def full_size():
    """BASELINE.json configs[3] and [4] at their FULL size -- 1e9 photons, as 1000 batches of 1e6 through the batch-moments entry
    point (no per-batch block leaves the device) -- on one GPU: size-independent properties (photon count, energy closure with the
    dropped-photon deficit) and, for config 4, every column against the oracle's fixture."""
    import i3rc_monte_carlo_model_amd as M
    from scipy import stats

    nb, n = 1000, 1_000_000
    for name in ("landsat36", "landsat119_7dir"):
        _, w = W.get(name)
        g, _ = W.make_integrator(w)
        t0 = time.perf_counter()
        s1, s2, cnt = g.computeRadiativeTransferBatchMoments((10, 1), nb, w["mu0"], 0.0, n)
        dt = time.perf_counter() - t0
        kernel = g.kernel_name()
        g.finalize_Integrator()
        mean = {k: np.asarray(v, np.float64) / nb for k, v in s1.items()}
        se = {k: np.sqrt(np.maximum(np.asarray(s2[k], np.float64) / nb - mean[k] ** 2, 0.0) / (nb - 1)) for k in s1}
        albedo = w.get("surface", 0.0)
        closure = float(mean["meanFluxUp"] + mean["meanFluxAbsorbed"] + (1.0 - albedo) * mean["meanFluxDown"]) + cnt["dropped"] / cnt["photons"]
        print(f"== {name} at BASELINE.json's full size: {nb} x {n} = {cnt['photons']:.4g} photons in {dt:.2f} s = {cnt['photons'] / dt:.3e} photons/s through "
              f"computeRadiativeTransferBatchMoments ({kernel}); meanFluxUp {float(mean['meanFluxUp']):.6f} +- {float(se['meanFluxUp']):.1e}, "
              f"meanFluxDown {float(mean['meanFluxDown']):.6f} +- {float(se['meanFluxDown']):.1e}; up + absorbed + (1 - albedo) down + dropped = {closure:.6f}; "
              f"dropped {cnt['dropped'] / cnt['photons']:.3e}, scatterings per photon {cnt['scatterings'] / cnt['photons']:.4f}")
        if name == "landsat119_7dir":
            z = np.load(os.path.join(ROOT, "tests", "golden", "config4_columns.npz"))
            mg = np.concatenate([mean["fluxUp"][None], mean["fluxDown"][None], mean["intensity"]])
            sg = np.concatenate([se["fluxUp"][None], se["fluxDown"][None], se["intensity"]])
            dof = int(z["batches"]) - 1
            print(f"   against tests/golden/config4_columns.npz ({int(z['batches'])} x {int(z['photonsPerBatch'])} oracle photons; the GPU's standard errors are a sixth of the fixture's: "
                  f"nearly a one-sample statistic, see above; t with {dof} degrees of freedom: {100 * (1 - 2 * stats.t.sf(3.0, dof)):.2f} %, mean z^2 {dof / (dof - 2):.3f}):")
            for k, fname in enumerate(z["fieldNames"]):
                zz = np.abs(mg[k] - z["mean"][k]) / (np.sqrt(sg[k] ** 2 + z["stderr"][k].astype(np.float64) ** 2) + 1e-7)
                dm, fm = mg[k].mean(), z["mean"][k].mean(dtype=np.float64)
                sem = np.std(z["batchMeans"][:, k], ddof=1) / np.sqrt(len(z["batchMeans"]))
                print(f"   columns {str(fname):10s} {100 * (zz <= 3.0).mean():.2f} % within 3 sigma, max |z| {zz.max():.2f}, mean z^2 {np.mean(zz ** 2):.3f}; "
                      f"domain mean GPU {dm:.6f} fixture {fm:.6f} +- {sem:.1e} (z {(dm - fm) / sem:+.2f})")


if __name__ == "__main__":
    if sys.argv[1:] == ["full"]:
        full_size()
    else:
        main()
