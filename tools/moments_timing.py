"""A loop's batches with their statistics gathered on the device (i3rc_hip_run_batches_moments) against the same loop with every
batch's tally block brought to the host (i3rc_hip_run_batches): wall time per batch through the C ABI.
usage: tools/moments_timing.py [workload] [photons per batch] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "landsat36")
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 100
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1000))
def moments():
    t0 = time.perf_counter()
    s1, s2, cnt = g.computeRadiativeTransferBatchMoments((10, 1), nb, w["mu0"], 0.0, n)
    assert cnt["photons"] == n * nb
    return time.perf_counter() - t0, s1
def blocks():
    import ctypes as C
    from i3rc_monte_carlo_model_amd import binding as B
    lay = g.layout()
    raw = np.zeros((nb, lay.total), np.float64)
    s = B.Source(); s.kind, s.solarMu, s.solarAzimuth = 0, w["mu0"], 0.0
    t0 = time.perf_counter()
    assert B.load().i3rc_hip_run_batches(g._h, 10, 1, nb, n, C.byref(s), 0, raw.ctypes.data_as(B.dp)) == 0
    return time.perf_counter() - t0
moments(); blocks()
tm = min(moments()[0] for _ in range(3)); tb = min(blocks() for _ in range(3))
s1 = moments()[1]
print(f"{name} {nb} x {n:.0e} photons: batch moments on the device {tm / nb * 1e3:.4f} ms per batch ({nb * n / tm:.3e} photons/s), "
      f"per-batch blocks to the host {tb / nb * 1e3:.4f} ms per batch ({nb * n / tb:.3e} photons/s); mean fluxUp {s1['meanFluxUp'] / nb:.5f} "
      f"({g.kernel_name()})", flush=True)
