"""What costs the LES stratocumulus + Rayleigh workload (tools/workloads.py: les_stcu_rayleigh, the domain the reference's tool chain wrote) its
rate: the same field with the surface / sun / table / gas taken away one at a time.  usage: python3 tools/les_experiments.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W
name, w = W.get("les_stcu_rayleigh")
dom, d = W.domain_from_file(w)
def run(label, comps, albedo=0.06, mu0=0.5, n=20_000_000, dirs=None):
    dm = M.new_Domain(dom.x, dom.y, dom.z)
    for c in comps:
        dm.addOpticalComponent(c["name"], c["ext"][:, 0, 0] if c["uniform"] else c["ext"], c["ssa"][:, 0, 0] if c["uniform"] else c["ssa"], c["pfi"][:, 0, 0] if c["uniform"] else c["pfi"], c["table"], zLevelBase=c["zbase"])
    g = M.new_Integrator(dm)
    kw = dict(intensityMus=dirs[0], intensityPhis=dirs[1], useRussianRouletteForIntensity=True, zetaMin=0.3) if dirs else {}
    g.specifyParameters(surfaceAlbedo=albedo, minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True, **kw)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(mu0, 0.0, 1))
    for k in range(2):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1 + k)), M.new_PhotonStream(mu0, 0.0, n))
    c = r["counters"]
    print(f"{label:60s} {n / g.kernel_ms() * 1e3:.3e} photons/s  S={c['cellSteps']/n:.1f} K={c['scatterings']/n:.1f} surf={c['surfaceHits']/n:.2f}  {g.kernel_name()}", flush=True)
    g.finalize_Integrator()
cloud, gas = dom.components
one = dict(cloud); one["pfi"] = np.where(cloud["pfi"] > 0, 25, 0).astype(np.int32)
tab1 = M.PhaseFunctionTable([cloud["table"].entries[24]])
single = dict(cloud); single["pfi"] = np.where(cloud["pfi"] > 0, 1, 0).astype(np.int32); single["table"] = tab1
run("two components, 35-entry table (as the tools wrote it)", [cloud, gas])
lossless = dict(cloud); lossless["ssa"] = np.where(cloud["ext"] > 0, np.float32(1), cloud["ssa"]).astype(np.float32)
run("two components, the droplets' omega = 0.999996 -> 1", [lossless, gas])
run("cloud only, omega -> 1", [lossless])
run("two components, albedo 0", [cloud, gas], albedo=0.0)
run("two components, mu0 = 1", [cloud, gas], mu0=1.0)
run("two components, every cloudy cell entry 25", [one, gas])
run("two components, a table of one entry", [single, gas])
run("cloud only, 35-entry table", [cloud])
run("cloud only, every cloudy cell entry 25", [one])
run("cloud only, a table of one entry", [single])
two = ([1.0, 0.6], [0.0, 135.0])
run("two components + 2 radiance directions", [cloud, gas], n=10_000_000, dirs=two)
run("two components + 2 radiance directions, omega -> 1", [lossless, gas], n=10_000_000, dirs=two)
