import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W
name, w = W.get(sys.argv[1]); n = int(sys.argv[2])
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 100000))
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, n))
t0 = time.perf_counter(); k = 0.0
for b in range(4):
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b + 1)), M.new_PhotonStream(w["mu0"], 0.0, n)); k += g.kernel_ms()
wall = (time.perf_counter() - t0) / 4 * 1e3
print(f"{name} I3RC_SLABS={os.environ.get('I3RC_SLABS','1')}: wall {wall:.2f} ms per launch, photon kernel {k/4:.2f} ms; {n/wall*1e3:.3e} photons/s end to end")
