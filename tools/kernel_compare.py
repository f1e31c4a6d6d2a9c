"""Time the kernel variants (pool / lane / general) on the step cloud; prints photons/s and the work counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tests import cases

n = int(float(os.environ.get("PHOTONS", "5e7")))
for nl in (16, 32):
    d = cases.step_cloud(nlayers=nl)
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 100000))
    for kernel in sys.argv[1:] or ("pool", "lane", "general"):
        for bpc in (0,):
            g.set_tuning(40, bpc, kernel=kernel)
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, n))
            c = r["counters"]
            print(f"nlayers {nl} {kernel:8s}: {g.kernel_ms():8.2f} ms {n / g.kernel_ms() * 1e3:.3e} photons/s  Fup {r['fluxUp'].mean():.5f} "
                  f"steps/ph {c['cellSteps'] / n:.2f} scat/ph {c['scatterings'] / n:.2f} draws/ph {c['rngDraws'] / n:.1f}", flush=True)
