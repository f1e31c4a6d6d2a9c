import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tests import cases
d = cases.step_cloud(nlayers=16)
dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 100000))
n = 100_000_000
for bpc in (0, 3, 4, 5, 6, 7, 8, 0):
    g.set_tuning(0, bpc)
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, n))
    print(f"blocks/CU {bpc}: {g.kernel_ms():.2f} ms {n/g.kernel_ms()*1e3:.3e} photons/s", flush=True)
