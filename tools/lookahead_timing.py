"""A driver's loop through i3rc_hip_compute_batch (what the unchanged reference driver's computeRadiativeTransfer calls end in):
one call per batch with seed (iseed, batch), the library looking ahead in fused groups.  Wall time per batch at the C ABI.
usage: tools/lookahead_timing.py [workload] [photons per batch] [batches]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import binding as B
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "step16")
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1))
lib = B.load()
raw = np.zeros(g.layout().total, np.float64)
s = B.Source(); s.kind, s.solarMu, s.solarAzimuth = 0, w["mu0"], 0.0
for look in (3, 0):
    t0 = time.perf_counter(); marks = []
    for b in range(1, nb + 1):
        assert lib.i3rc_hip_compute_batch(g._h, 10, b, n, C.byref(s), look, raw.ctypes.data_as(B.dp)) == 0
        if b in (10, 100, nb): marks.append((b, time.perf_counter() - t0))
    txt = ", ".join(f"{b} batches {t * 1e3:.1f} ms" for b, t in marks)
    (b0, t0_), (b1, t1_) = marks[0], marks[-1]
    print(f"{name} {n:.0e} photons per batch, look-ahead {look}: {txt}; steady state {(t1_ - t0_) / (b1 - b0) * 1e3:.3f} ms per batch = {n * (b1 - b0) / (t1_ - t0_):.3e} photons/s", flush=True)
    nb = min(nb, 200)
g.finalize_Integrator()
