"""One launch of one BASELINE configuration (for profiling): python3 tools/run_case.py <case> [photons]
cases: step16 step32 radar radar_nadir radar64_nadir landsat landsat36 landsat7"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tests import cases

if os.environ.get("I3RC_LIB"):   # another build of the library (A/B comparisons)
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False

hg64 = lambda: M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)])
hg299 = lambda: M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)])
nadir = dict(intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
dirs7 = dict(intensityMus=[1, .5, .5, .8, .8, .3, .3], intensityPhis=[0, 0, 180, 90, 270, 45, 225],
             useRussianRouletteForIntensity=True, zetaMin=0.3)
CASES = {
    "step16": (lambda: cases.step_cloud(nlayers=16), hg64, {}, 1.0, 20_000_000),
    "step32": (lambda: cases.step_cloud(nlayers=32), hg64, {}, 1.0, 20_000_000),
    "radar": (cases.radar_cloud, hg299, {}, 1.0, 10_000_000),
    "radar_nadir": (cases.radar_cloud, hg299, nadir, 1.0, 5_000_000),
    "radar64_nadir": (cases.radar_cloud_64, hg299, nadir, 1.0, 5_000_000),
    "landsat": (cases.landsat_cloud, hg299, {}, 1.0, 10_000_000),
    "landsat36": (lambda: cases.landsat_cloud(nlayers=36), hg299, {}, 1.0, 10_000_000),
    "landsat7": (cases.landsat_cloud, hg299, dict(surfaceBDRF=None, **dirs7), 0.5, 1_000_000),
}
name = sys.argv[1]
make, table, kw, mu0, n = CASES[name]
if len(sys.argv) > 2:
    n = int(float(sys.argv[2]))
if "surfaceBDRF" in kw:
    kw = dict(kw, surfaceBDRF=M.new_SurfaceDescription([0.2]))
d = make()
dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], table())
g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001, minForwardTableSize=10001, **kw)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(mu0, 0.0, 1))   # tables
seed = int(os.environ.get("BATCH", "1"))
r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, seed)), M.new_PhotonStream(mu0, 0.0, n))
c = r["counters"]
print(f"{name}: {n / g.kernel_ms() * 1e3:.3e} photons/s ({g.kernel_ms():.2f} ms, {n} photons) S={(c['cellSteps'] + c['shadowSteps']) / n:.1f} K={c['scatterings'] / n:.1f}")
