"""Per-batch domain means of one workload on the GPU, saved for a comparison with an oracle sample made elsewhere (tools/cpu_baseline.py
--save, e.g. on the build container's cores over an hour):  gpu_means.py <workload> <batches> <photons per batch> <out.npz> [first seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W
if os.environ.get("I3RC_LIB"):
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False
name, w = W.get(sys.argv[1]); nb = int(sys.argv[2]); n = int(float(sys.argv[3])); out = sys.argv[4]; first = int(sys.argv[5]) if len(sys.argv) > 5 else 1
g, _ = W.make_integrator(w)
if os.environ.get("MAXCS") == "1":   # max cross-section: the general kernel, the local estimate in the reference's nested order (trace, then roulette)
    g.specifyParameters(useRayTracing=False)
if os.environ.get("GENERAL") == "1":   # the general kernel through the ray queue
    g.set_tuning(kernel="general")
means, inten = [], []
for b in range(first, first + nb):
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((int(os.environ.get("SEED0", "191")), b)), M.new_PhotonStream(w["mu0"], 0.0, n))
    means.append([float(r[k].mean(dtype=np.float64)) for k in ("fluxUp", "fluxDown", "fluxAbsorbed")])
    if "intensity" in r:
        inten.append([float(v) for v in r["intensity"].mean(axis=(1, 2), dtype=np.float64)])
np.savez(out, means=np.array(means), intensityMeans=np.array(inten), photonsPerBatch=n, kernel=g.kernel_name())
print(name, nb, "x", n, "photons:", g.kernel_name(), "mean fluxUp %.6f" % np.mean([m[0] for m in means]))
