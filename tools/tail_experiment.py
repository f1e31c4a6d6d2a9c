"""Does a launch's END cost a domain with cloud-free layers and no gas its rate?  The LES stratocumulus field WITHOUT its Rayleigh component
(the reference's Tools/Examples/cloud_to_domain.nml as shipped makes that: one component) at three launch sizes: kernel time against
photons -- a launch that ends in a tail of few long-lived photons has a time a + b N with a large a.
usage: python3 tools/tail_experiment.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get("les_stcu_rayleigh")
dom, d = W.domain_from_file(w)
for label, comps in (("cloud + Rayleigh gas", dom.components), ("cloud only", dom.components[:1])):
    dm = M.new_Domain(dom.x, dom.y, dom.z)
    for c in comps:
        u = c["uniform"]
        dm.addOpticalComponent(c["name"], c["ext"][:, 0, 0] if u else c["ext"], c["ssa"][:, 0, 0] if u else c["ssa"], c["pfi"][:, 0, 0] if u else c["pfi"], c["table"], zLevelBase=c["zbase"])
    g = M.new_Integrator(dm)
    g.specifyParameters(surfaceAlbedo=0.06, minInverseTableSize=10001)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(0.5, 0.0, 1))
    for n in (2_000_000, 10_000_000, 50_000_000, 200_000_000):
        for k in range(2):
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1 + k)), M.new_PhotonStream(0.5, 0.0, n))
        c = r["counters"]
        print(f"{label:22s} {n:>11d} photons: {g.kernel_ms():9.2f} ms = {n / g.kernel_ms() * 1e3:.3e} photons/s  S={c['cellSteps'] / n:.1f} K={c['scatterings'] / n:.1f}  {g.kernel_name()}", flush=True)
    g.finalize_Integrator()
