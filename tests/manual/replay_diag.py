"""Replay diagnostics (run on the GPU box): the replay kernel fed with the reference's MT19937 deviates against the
oracle on the I3RC Landsat scene with the local estimate's roulette, for growing photon counts, optionally with another
build of the library (I3RC_LIB), the event threshold fixed (EVTHR) or one photon per launch (ONE=1).  With one photon
per launch only one lane of the chip works: a disagreement that needs several photons in a batch is an interaction
between lanes (this is how the miscompiled grid-place flag was found; see tracer.hpp, GridPlace)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: F401  (first: one HIP runtime per process, see tests/conftest.py)
import i3rc_monte_carlo_model_amd as M
from oracle import pyoracle as O
from tools import cases
from tests.test_gpu_features import _intensity_pair, _replay_pair, hg_table

O.build()
if os.environ.get("I3RC_LIB"):   # another build of the library (A/B comparisons)
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False
d = cases.landsat_cloud(ssa=0.99)
gp = dict(useRussianRouletteForIntensity=True, zetaMin=0.3); op = dict(useRRForIntensity=1, zetaMin=0.3)
g, o = _intensity_pair(O, d, hg_table(0.85, 299), gpu_params=gp, oracle_params=op, mus=[0.5], phis=[40.0])
if os.environ.get("EVTHR"): g.set_tuning(evThreshold=int(os.environ["EVTHR"]))
lay = g.layout(); ncol = g.nx * g.ny
seed = [10, 3]
radiance = lambda out: out["raw"][lay.intensityByComponent:lay.intensityByComponent + 2 * ncol].astype(np.float64)

if os.environ.get("ONE"):
    n = int(os.environ.get("N", "200"))
    rng = O.RandomNumberSequence(seed)
    ph = O.photons_directional(rng, 0.7, 25.0, n)
    rng2 = O.RandomNumberSequence(seed); rng2.reals(rng.draws)
    gsum = rsum = 0.0
    for i in range(n):
        one = tuple(np.ascontiguousarray(a[i:i + 1]) for a in ph)
        before = rng.draws
        ref = o.compute(rng, *one, record=True, normalise=False)
        out = g.run_replay(M.PhotonStream(arrays=one), rng2.reals(rng.draws - before), ref["drawStart"])
        gsum += radiance(out).sum(); rsum += np.asarray(ref["intensityByComp"], np.float64).sum()
        if i % 50 == 0: print("photon", i, "gpu", gsum, "ref", rsum, flush=True)
    print("one photon per launch: gpu", gsum, "ref", rsum)
else:
    for n in [6, 8, 64, 1024, 3000]:
        ref, out = _replay_pair(O, g, o, n, seed, 0.7, 25.0)
        print(n, "photons: radiance sum gpu", radiance(out).sum(), "ref", np.asarray(ref["intensityByComp"], np.float64).sum(),
              "| steps gpu", out["counters"]["shadowSteps"] + out["counters"]["cellSteps"], "ref", ref["cellSteps"],
              "| tracer calls", out["counters"]["tracerCalls"], ref["tracerCalls"], flush=True)
