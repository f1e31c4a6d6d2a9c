"""Event-threshold sweep per configuration: python3 tools/threshold_sweep.py case [case ...] (cases of tools/run_case.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tools import cases

hg64 = lambda: M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)])
hg299 = lambda: M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)])
CASES = {"step16": (lambda: cases.step_cloud(nlayers=16), hg64, 1e8), "step32": (lambda: cases.step_cloud(nlayers=32), hg64, 2e7),
         "radar": (cases.radar_cloud, hg299, 1e7), "landsat": (cases.landsat_cloud, hg299, 1e7),
         "landsat36": (lambda: cases.landsat_cloud(nlayers=36), hg299, 1e7), "radar64": (cases.radar_cloud_64, hg299, 1e7)}
for name in sys.argv[1:]:
    make, table, n = CASES[name]; n = int(n)
    d = make()
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], table())
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 100000))
    out = []
    for thr in (0, 8, 16, 24, 32, 36, 40, 44, 48, 52):   # 0 = the per-wave adaptive default
        g.set_tuning(thr, 0)
        best = 0
        for rep in range(2):
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, n))
            best = max(best, n / g.kernel_ms() * 1e3)
        out.append(f"{thr}:{best:.3e}")
    c = r["counters"]
    print(f"{name} (steps/event {c['cellSteps'] / (c['scatterings'] + c['photons']):.1f}): " + "  ".join(out), flush=True)
