"""A production-like multi-component domain: the Landsat cloud field plus a horizontally uniform Rayleigh-like gas (every cell
optically active, two components, two phase-function tables) -- the general kernels on a field beyond L2.
usage: multicomponent_bench.py [photons] [directions 0|2|7]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import cases
from tools.workloads import DIRS7
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = cases.landsat_cloud()
nz = d["ext"].shape[0]
gas = np.broadcast_to(np.linspace(2.0e-5, 1.5e-5, nz, dtype=np.float32)[:, None, None], d["ext"].shape).copy()   # tau ~ 0.04
tabs = [M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)]), M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])]
dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], tabs[0])
dom.addOpticalComponent("gas", gas, np.ones_like(gas), np.ones(gas.shape, np.int32), tabs[1])
g = M.new_Integrator(dom)
kw = dict(minInverseTableSize=10001, minForwardTableSize=10001, surfaceAlbedo=0.2)
if nd == 2: kw.update(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
if nd == 7: kw.update(useRussianRouletteForIntensity=True, zetaMin=0.3, **DIRS7)
g.specifyParameters(**kw)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 100000))
best = 1e9
for b in (1, 2, 3):
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(1.0, 0.0, n)); best = min(best, g.kernel_ms())
c = r["counters"]
print(f"Landsat cloud + gas, {nd} directions: {n / best * 1e3:.3e} photons/s ({best:.1f} ms for {n:.3g}) {g.kernel_name()} "
      f"S={(c['cellSteps'] + c['shadowSteps']) / n:.1f} K={c['scatterings'] / n:.1f} Fup={r['fluxUp'].mean():.4f}")
