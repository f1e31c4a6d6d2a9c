import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"])
import torch
import numpy as np
import i3rc_monte_carlo_model_amd as M
if os.environ.get("I3RC_LIB"):
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False
from tools import cases
from tests.test_gpu_parity import hg_table, make_gpu
rad = dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
t2 = [M.PhaseFunctionTable([M.henyey_greenstein(0.85, 32), M.henyey_greenstein(0.6, 16)]),
      M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array([0.0, 0.1], np.float32))])]
for name, d, tab, n in (("landsat general", cases.landsat_cloud(ssa=0.99), hg_table(0.85, 299), 2000000), ("radar general", cases.radar_cloud(), hg_table(0.85, 299), 5000000),
                        ("two components", cases.two_component(nx=48, ny=32, nz=16), t2, 5000000), ("irregular", cases.irregular_domain(nx=40, ny=30, nz=20), hg_table(), 5000000)):
    g = make_gpu(d, tab, surfaceAlbedo=0.2, **rad); g.set_tuning(forceGeneral=True)
    best = 1e9
    for b in (1, 2, 3):
        g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(0.7, 25.0, n))
        best = min(best, g.kernel_ms())
    print(name, "%.2f ms  %.3g photons/s" % (best, n / best * 1e3), flush=True)
