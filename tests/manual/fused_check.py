"""Fused multi-batch launches against one launch per batch: counters identical, tallies equal to summation order; timing.
usage: tests/manual/fused_check.py [workload] [photons per batch] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "step16")
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 7
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1000))
out = {}
for mode in (0, 1):
    g.set_batch_fusion(mode)
    g.computeRadiativeTransferBatches((5, 3), min(nb, 4), w["mu0"], 0.0, n)   # warm-up (buffers, streams)
    t0 = time.perf_counter()
    out[mode] = g.computeRadiativeTransferBatches((5, 3), nb, w["mu0"], 0.0, n)
    dt = time.perf_counter() - t0
    print(f"{name} {nb} batches of {n} photons, fusion {mode}: {dt / nb * 1e3:.3f} ms per batch, {nb * n / dt:.3e} photons/s ({g.kernel_name()})", flush=True)
bad = 0
for b, (a, f) in enumerate(zip(out[0], out[1])):
    if a["counters"] != f["counters"]:
        bad += 1
        if bad < 4:
            print("batch", b, "counters differ:", {k: (a["counters"][k], f["counters"][k]) for k in a["counters"] if a["counters"][k] != f["counters"][k]})
    if not np.allclose(a["raw"], f["raw"], rtol=1e-5, atol=1e-6):
        bad += 1
        if bad < 4:
            i = np.argmax(np.abs(a["raw"] - f["raw"]))
            print("batch", b, "tallies differ at", i, a["raw"][i], f["raw"][i])
print("FAILED" if bad else "fused == one launch per batch: ok", flush=True)
sys.exit(1 if bad else 0)
