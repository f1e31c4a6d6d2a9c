"""How much of a radiance run's time is the phase-function tables missing L2?  The same workload with its 10001-step tables and with
tables of 257 steps (other physics in the third digit; the work per photon stays the same): if the small tables are much
faster, the big ones do not stay in L2 beside the field.  usage: tests/manual/table_size_experiment.py [workload] [photons]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "landsat119_7dir")
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
tab = M.PhaseFunctionTable([M.henyey_greenstein(0.85, w["moments"])])
for steps_inv, steps_fwd in ((10001, 10001), (10001, 257), (257, 10001), (257, 257)):
    g, _ = W.make_integrator(w)
    g.set_tables(1, inverse=tab.inverse_table(steps_inv), forward=tab.forward_table(steps_fwd), forward_orig=tab.forward_table(steps_fwd))
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 100000))
    best = 1e30
    for b in (1, 2):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(w["mu0"], 0.0, n))
        best = min(best, g.kernel_ms())
    c = r["counters"]
    print(f"{name}: inverse table {steps_inv} steps, forward tables {steps_fwd}: {n / best * 1e3:.3e} photons/s ({best:.1f} ms) S={(c['cellSteps'] + c['shadowSteps']) / n:.0f} K={c['scatterings'] / n:.1f}", flush=True)
    g.finalize_Integrator()
