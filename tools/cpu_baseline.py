"""CPU baseline leg of bench.py: times the CPU oracle (oracle/, the C restatement of the reference's
computeRT) on the host cores of the box, one process per core over disjoint batches -- the reference's own
decomposition (Example-Drivers/monteCarloDriver.f95:264-274).  Never touches the GPU.  Prints one JSON line.

kind "port": this is the C restatement, NOT the reference's Fortran (netCDF-Fortran is absent from this image and the rules call
a reference that needs it unbuildable: DESIGN.md section 2).  `reference_note` carries the survey-time figure of the reference
itself for the nearest shape; `reference_loop` (round 5) is the reference's own loop timed HERE, on one core,
beside the port on the same batches -- oracle/_ref/ref_loop, the reference's sources unmodified over this repository's netCDF module --
where the tree holds that binary: the ratio of the two is the calibration, and their fields are compared bit for bit on the way."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the reference's own code, amdflang -O2, one core of an Intel Xeon @ 2.10 GHz (survey container)
REFERENCE_NOTES = {
    "step16": "reference Fortran (survey-time, SURVEY.md s6): 1.7-2.0e5 photons/s/core on step cloud 32x1x32 (S=63); this port timed here on 32x1x16 (S=45.8)",
    "step32": "reference Fortran (survey-time, SURVEY.md s6): 1.7-2.0e5 photons/s/core on this shape (32x1x32, mu0=1)",
    "radar640": "reference Fortran (survey-time): 7.9e4 photons/s/core on this shape (640x1x54 flux)",
    "radar640_nadir": "reference Fortran (survey-time): 5.1e4 photons/s/core on this shape (640x1x54 + nadir radiance)",
    "radar64_nadir": "reference Fortran (survey-time): 5.1e4 photons/s/core on the 640x1x54 field + nadir radiance (same field statistics)",
    "landsat119": "reference Fortran (survey-time): 5.1e4 photons/s/core on this shape (128x128x119, mu0=1, flux)",
    "landsat36": "reference Fortran (survey-time): 5.1e4 photons/s/core on 128x128x119 (S=241); this port timed here on 128x128x36 (S=150)",
    "landsat119_7dir": "reference Fortran (survey-time): 7.2e3 photons/s/core on this shape (128x128x119, mu0=0.5, 7 directions, Lambertian 0.2)",
    "landsat36_7dir": "reference Fortran (survey-time): 7.2e3 photons/s/core on 128x128x119 with the same 7 directions",
}


def _worker(args):
    first_batch, n_batches, photons, config, nlayers, mu0, columns = args
    import numpy as np

    from oracle import pyoracle as O
    from tools import workloads as W

    name, w = W.get(config)
    if nlayers and name.startswith("step"):
        w = dict(w, domain=("step_cloud", dict(nlayers=nlayers)))
    integ, _ = W.make_oracle(w)
    mu0 = w["mu0"] if mu0 is None else mu0
    t0 = time.perf_counter()
    fu = 0.0
    cols = []
    for b in range(first_batch, first_batch + n_batches):
        rng = O.RandomNumberSequence([10, b])
        ph = O.photons_directional(rng, mu0, 0.0, photons)
        r = integ.compute(rng, *ph)
        fu += float(r["fluxUp"].mean())
        small = columns or r["fluxUp"].size <= 5000     # per-column fields of small domains (or on request); domain means (float64) of all
        inten = r.get("intensity")
        cols.append(dict(fluxUp=r["fluxUp"].copy() if small else None, fluxDown=r["fluxDown"].copy() if small else None,
                         fluxAbsorbed=r["fluxAbsorbed"].copy() if small else None,
                         absorbedProfile=r["volumeAbsorption"].reshape(r["volumeAbsorption"].shape[0], -1).mean(axis=1, dtype=np.float64),
                         intensity=inten.copy() if (small and inten is not None) else None,
                         means=[float(r[k].mean(dtype=np.float64)) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")],
                         intensityMeans=[] if inten is None else [float(v) for v in inten.mean(axis=(1, 2), dtype=np.float64)],
                         nBad=int(r["nBad"]), cellSteps=int(r["cellSteps"]), scatterings=int(r["scatterings"]),
                         exits=int(r["exitsTop"]) + int(r["surfaceHits"])))
    return time.perf_counter() - t0, fu / n_batches, cols


def reference_loop_rate(name, w, photons):
    """The REFERENCE'S OWN loop on this workload, one core: oracle/_ref/ref_loop (the whole of the reference's Code/ and its integrator
    compiled unmodified over this repository's netCDF module -- oracle/ref_loop.f95 says what that build is), where the tree holds it and
    the workload is one it can be handed (Henyey-Greenstein components from recipes, a uniform surface), timed beside the port on the
    SAME batches on the same core: the ratio bench.py's calibration used to take from survey-time figures.  None where it cannot be run."""
    import tempfile

    import numpy as np

    from oracle import pyoracle as O
    from oracle import ref_loop_io as R
    from tools import workloads as W

    if not os.path.exists(R.REF_LOOP) or "domain_file" in w or "surface_grid" in w or "irregular" in w:
        return None
    d = W.domain(w)
    comps = [dict(coefficients=[O.hg_coefficients(0.85, w["moments"])], ext=d["ext"], ssa=d["ssa"], pf=d["pf"])]
    if "gas" in w:
        gas = W.gas_component(w, d)
        comps.append(dict(coefficients=[np.array(W.GAS_LEGENDRE, np.float32)], ext=gas, ssa=np.full_like(gas, np.float32(w.get("gas_ssa", 1.0))), pf=np.ones(gas.shape, np.int32)))
    p = w["params"]
    case = dict(xe=d["xe"], ye=d["ye"], ze=d["ze"], components=comps, solarMu=w["mu0"], nBatches=2, nPhotons=int(photons), seed=(10, 1), dumpTables=0,
                mus=p.get("intensityMus", ()), phis=p.get("intensityPhis", ()), useRRForIntensity=int(bool(p.get("useRussianRouletteForIntensity"))), zetaMin=p.get("zetaMin", 0.3))
    if "surface" in w:
        huge = np.finfo(np.float32).max
        case["surface"] = (np.array([0.0, huge], np.float32), np.array([0.0, huge], np.float32), np.array([[w["surface"]]], np.float32))
    with tempfile.TemporaryDirectory() as tmp:
        # a first call of one photon per batch: tables and start-up, taken off the timed call's wall time
        t0 = time.perf_counter(); R.run(dict(case, nPhotons=1), tmp); setup = time.perf_counter() - t0
        t0 = time.perf_counter(); batches, _ = R.run(case, tmp); ref_s = time.perf_counter() - t0 - setup
    integ, _ = W.make_oracle(w)
    t0 = time.perf_counter()
    same = True
    for b in range(2):
        rng = O.RandomNumberSequence([10, 1 + b])
        r = integ.compute(rng, *O.photons_directional(rng, w["mu0"], 0.0, int(photons)))
        same = same and np.array_equal(r["fluxUp"], batches[b]["fluxUp"]) and np.array_equal(r["fluxDown"], batches[b]["fluxDown"])
    port_s = time.perf_counter() - t0
    return {"photons_per_s_per_core": 2 * photons / max(ref_s, 1e-9), "port_photons_per_s_per_core_same_batches": 2 * photons / port_s,
            "port_over_reference": ref_s / port_s, "fields_bit_identical": bool(same),
            "what": "oracle/_ref/ref_loop: the reference's Code/ + integrator, unmodified, amdflang -O2, over this repository's netCDF module; "
                    f"2 batches x {int(photons)} photons of this workload on one core, start-up and tables taken off"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="step16")
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--batches-per-core", type=int, default=40)   # 10-16 s on the GPU box's 16 cores (bench.py: "a bounded sample")
    ap.add_argument("--photons", type=int, default=0, help="photons per batch (0 = the workload's bounded sample)")
    ap.add_argument("--nlayers", type=int, default=0, help="step cloud only: 16 = BASELINE.json label, 32 = reference generator")
    ap.add_argument("--mu0", type=float, default=None)
    ap.add_argument("--first-batch", type=int, default=1, help="batch number (second seed word) of the first batch: other numbers, another sample")
    ap.add_argument("--save-columns", action="store_true", help="with --save: per-column fields whatever the size of the domain")
    ap.add_argument("--save", default="", help="write per-batch results to this .npz: domain means of every batch, per-column fields of small domains")
    a = ap.parse_args()
    from oracle import pyoracle as O
    from tools import workloads as W

    O.build()
    name, w = W.get(a.config)
    if a.nlayers == 32 and name == "step16":
        name, w = W.get("step32")
    photons = a.photons or w["cpu_photons"]
    cores = a.cores or min(16, len(os.sched_getaffinity(0)))
    jobs = [(a.first_batch + i * a.batches_per_core, a.batches_per_core, photons, name, a.nlayers, a.mu0, a.save_columns) for i in range(cores)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_worker, jobs))
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    allb = [c for r in res for c in r[2]]
    if a.save:
        import numpy as np

        out = dict(means=np.array([c["means"] for c in allb]), intensityMeans=np.array([c["intensityMeans"] for c in allb]),
                   nBad=np.array([c["nBad"] for c in allb]), cellSteps=np.array([c["cellSteps"] for c in allb]),
                   scatterings=np.array([c["scatterings"] for c in allb]), photonsPerBatch=np.array(photons))
        for k in ("fluxUp", "fluxDown", "fluxAbsorbed", "intensity", "absorbedProfile"):
            if allb[0][k] is not None:
                out[k] = np.stack([c[k] for c in allb])
        np.savez(a.save, **out)
    total = cores * a.batches_per_core * photons
    # work per photon of the reference's algorithm on this workload (the oracle's own counters: tracer iterations incl.
    # local-estimate rays, scatterings, boundary tallies) -- what SURVEY.md 8(d)'s byte formula is evaluated with
    per_photon = {"S": sum(c["cellSteps"] for c in allb) / total, "K": sum(c["scatterings"] for c in allb) / total,
                  "E": sum(c["exits"] for c in allb) / total}
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown CPU"
    try:
        reference_loop = reference_loop_rate(name, w, photons)
    except Exception as e:   # noqa: BLE001  (a report beside the baseline, never a reason to lose it)
        reference_loop = {"error": str(e)[:200]}
    print(json.dumps({"value": total / busy, "unit": "photons/s", "cores": cores, "kind": "port", "reference_loop": reference_loop,
                      "sample": f"{cores} processes x {a.batches_per_core} batches x {photons} photons of the same workload "
                                f"({name}), max busy time {busy:.1f} s, wall {wall:.1f} s, host {model}",
                      "per_core": total / busy / cores, "oracle_per_photon": per_photon,
                      "reference_note": REFERENCE_NOTES.get(name, "") + "; kind 'port' = oracle/ C restatement, not the reference binary",
                      "meanFluxUp": sum(r[1] for r in res) / len(res)}))


if __name__ == "__main__":
    main()
