"""Workgroups per CU (= waves per SIMD) against throughput: tools/blocks_sweep.py [workload] [photons]; 0 = the library's
own choice (occupancy query)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "step16")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 100000))
for bpc in (0, 3, 4, 5, 6, 7, 8, 0):
    g.set_tuning(0, bpc)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(w["mu0"], 0.0, n))
    print(f"{name} blocks/CU {bpc}: {g.kernel_ms():.2f} ms {n / g.kernel_ms() * 1e3:.3e} photons/s  {g.kernel_name()}", flush=True)
