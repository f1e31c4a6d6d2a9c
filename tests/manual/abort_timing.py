import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import binding as B
from tools import workloads as W
name, w = W.get("step16")
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 1))
lib = B.load(); raw = np.zeros(g.layout().total, np.float64)
s = B.Source(); s.kind, s.solarMu, s.solarAzimuth = 0, 1.0, 0.0
t0 = time.perf_counter()
for b in range(1, 300):
    assert lib.i3rc_hip_compute_batch(g._h, 10, b, 1000000, C.byref(s), 3, raw.ctypes.data_as(B.dp)) == 0
t1 = time.perf_counter()
g.finalize_Integrator()
t2 = time.perf_counter()
print(f"299 batches {1e3*(t1-t0):.1f} ms; finalize with groups launched ahead: {1e3*(t2-t1):.1f} ms")
