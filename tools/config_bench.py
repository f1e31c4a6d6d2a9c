"""Single-GPU throughput of the BASELINE.json configurations (bench.py measures configs[1] only), at photon counts
of the order BASELINE.json names per GPU (small launches are dominated by the ramp and tail of the persistent kernel:
radar flux 6.8e8 photons/s at 2e7 photons, 1.0e9 at 1e8)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import cases

def build(d, table, **kw):
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], table)
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001, minForwardTableSize=10001, **kw); return g

hg64 = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)])
hg299 = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)])
dirs7 = dict(intensityMus=[1, .5, .5, .8, .8, .3, .3], intensityPhis=[0, 0, 180, 90, 270, 45, 225])
runs = [
  ("step cloud 32x1x16 flux", cases.step_cloud(nlayers=16), hg64, {}, 1.0, 100_000_000),
  ("step cloud 32x1x32 flux", cases.step_cloud(nlayers=32), hg64, {}, 1.0, 100_000_000),
  ("radar 640x1x54 flux", cases.radar_cloud(), hg299, {}, 1.0, 100_000_000),
  ("radar 640x1x54 flux + nadir radiance (RR, zeta 0.3)", cases.radar_cloud(), hg299,
     dict(intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3), 1.0, 50_000_000),
  ("radar-64 64x64x54 flux + nadir radiance", cases.radar_cloud_64(), hg299,
     dict(intensityMus=[1.0], intensityPhis=[0.0], useRussianRouletteForIntensity=True, zetaMin=0.3), 1.0, 50_000_000),
  ("landsat 128x128x119 flux mu0=1", cases.landsat_cloud(), hg299, {}, 1.0, 100_000_000),
  ("landsat 128x128x36 flux mu0=1", cases.landsat_cloud(nlayers=36), hg299, {}, 1.0, 100_000_000),
  ("landsat 128x128x119 + 7 radiances + Lambertian 0.2 (BRDF object), mu0=.5", cases.landsat_cloud(), hg299,
     dict(surfaceBDRF=M.new_SurfaceDescription([0.2]), useRussianRouletteForIntensity=True, zetaMin=0.3, **dirs7), 0.5, 20_000_000),
]
for name, d, tab, kw, mu0, n in runs:
    g = build(d, tab, **kw)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(mu0, 0.0, 100000))
    # two batches, the faster one is reported: a launch ends with its longest photon, and in the radar field a rare
    # photon scatters tens of thousands of times (batch (10, 1) of 5e7 photons holds one that adds 48 ms)
    times = []
    for batch in (1, 2):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, batch)), M.new_PhotonStream(mu0, 0.0, n))
        times.append(g.kernel_ms())
    ms = min(times); c = r["counters"]
    S = (c["cellSteps"] + c["shadowSteps"]) / n; K = c["scatterings"] / n
    extra = f" I={r['intensity'].mean(axis=(1,2))[:2]}" if "intensity" in r else ""
    print(f"{name}: {n/ms*1e3:.3e} photons/s ({ms:.1f} ms for {n:.0e}) S={S:.1f} K={K:.1f} Fup={r['fluxUp'].mean():.4f} Fdn={r['fluxDown'].mean():.4f} drop={c['dropped']/n:.1e}{extra} [batches: {', '.join(f'{t:.1f} ms' for t in times)}]", flush=True)
    g.finalize_Integrator()
