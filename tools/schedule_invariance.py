"""A photon's path must not depend on scheduling (per-photon Philox streams): the integer work counters of one batch
are compared across event thresholds, kernels and block counts.  Run on the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import i3rc_monte_carlo_model_amd as M
from tools import cases
from tests.test_gpu_parity import hg_table, make_gpu

CASES = {"landsat": lambda: cases.landsat_cloud(ssa=0.99), "radar": cases.radar_cloud, "step": lambda: cases.step_cloud(ssa=0.99, nlayers=8)}
RAD = dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0], useRussianRouletteForIntensity=True, zetaMin=0.3)

for name in sys.argv[1:] or ["landsat", "radar", "step"]:
    d = CASES[name]()
    for label, params in [("max-xsec radiance+RR", dict(RAD, useRayTracing=False)), ("max-xsec radiance", dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0], useRayTracing=False)), ("flux", {}), ("radiance+RR", RAD), ("radiance", dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0]))]:
        base = None
        for tune in [dict(evThreshold=8), dict(evThreshold=24), dict(evThreshold=44), dict(evThreshold=0), dict(evThreshold=24, blocksPerCU=1),
                     dict(evThreshold=24, forceGeneral=True)]:
            g = make_gpu(d, hg_table(0.85, 299), **params)
            g.set_tuning(**tune)
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(0.7, 25.0, 200000))
            c = {k: r["counters"][k] for k in ("cellSteps", "scatterings", "surfaceHits", "exitsTop", "roulette", "shadowSteps", "tracerCalls")}
            rad = float(np.asarray(r["intensity"], np.float64).sum()) if "intensity" in r and params else 0.0
            if base is None: base = c
            print(name, label, tune, "OK" if c == base else "DIFFERS", c if c != base or tune == dict(evThreshold=8) else "", "radiance sum %.6g" % rad, flush=True)
