"""Threshold sweep of the radiance state machine (event / light thresholds) on the radiance configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tools import cases

def build(d, table, **kw):
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], table)
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001, minForwardTableSize=10001, **kw); return g

hg299 = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)])
dirs7 = dict(intensityMus=[1, .5, .5, .8, .8, .3, .3], intensityPhis=[0, 0, 180, 90, 270, 45, 225])
runs = [("radar+nadir", cases.radar_cloud(), dict(intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3), 1.0, 30_000_000),
        ("landsat+7", cases.landsat_cloud(), dict(surfaceBDRF=M.new_SurfaceDescription([0.2]), useRussianRouletteForIntensity=True, zetaMin=0.3, **dirs7), 0.5, 10_000_000)]
for name, d, kw, mu0, n in runs:
    g = build(d, hg299, **kw)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(mu0, 0.0, 100000))
    for ev in (16, 24, 32, 40):
        out = []
        for li in (12, 16, 20, 24, 32, 40):
            g.set_tuning(ev, 0, lightThreshold=li)
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(mu0, 0.0, n))
            out.append(f"L{li}:{n / g.kernel_ms() * 1e3:.3e}")
        print(f"{name} ev {ev}: " + "  ".join(out), flush=True)
