"""Fused multi-batch launches: wall time per batch of computeRadiativeTransferBatches and the kernels' own time.
usage: tools/fused_timing.py [workload] [photons per batch] [batches]   (I3RC_FUSED_CHUNK / I3RC_FUSED_GROUP_PHOTONS / I3RC_FUSED=0 in the environment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
if os.environ.get("I3RC_LIB"):
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "step16")
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 300
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1000))
if os.environ.get("FUSION"):
    g.set_batch_fusion(int(os.environ["FUSION"]))
lay = g.layout()
raw = np.zeros((nb, lay.total), np.float64)
from i3rc_monte_carlo_model_amd import binding as B
import ctypes as C
s = B.Source(); s.kind, s.solarMu, s.solarAzimuth = 0, w["mu0"], 0.0
lib = B.load()
def run():
    t0 = time.perf_counter()
    rc = lib.i3rc_hip_run_batches(g._h, 10, 1, nb, n, C.byref(s), 0, raw.ctypes.data_as(B.dp))
    assert rc == 0, lib.i3rc_hip_last_error(g._h)
    return time.perf_counter() - t0
reps = int(os.environ.get("REPS", "3"))
if reps > 1:
    run()
l0 = g.timed_launches()
dt = min(run() for _ in range(reps))
nl = (g.timed_launches() - l0) // reps
kms = g.kernel_ms_history(min(nl, 64))
cnt = raw[:, lay.counters]
assert np.all(cnt == n), cnt[:5]
print(f"{name} {nb} x {n:.0e} photons chunk={os.environ.get('I3RC_FUSED_CHUNK','-')} group={os.environ.get('I3RC_FUSED_GROUP_PHOTONS','-')} fused={os.environ.get('I3RC_FUSED','1')} lib={os.path.basename(os.environ.get('I3RC_LIB','default'))}: "
      f"{dt / nb * 1e3:.4f} ms per batch, {nb * n / dt:.3e} photons/s through the C ABI; {nl} launches per call, kernel sum {kms.sum() * (nl / len(kms)):.2f} ms "
      f"= {nb * n / (kms.sum() * (nl / len(kms))) * 1e3:.3e} photons/s ({g.kernel_name()})", flush=True)
