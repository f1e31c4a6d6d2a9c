"""The general kernels (every path behind run-time switches: what irregular grids, several components, BRDF grids and explicit
sources run) against the specialised ones on problems both can run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import cases
from tests.test_gpu_parity import hg_table, make_gpu
rad = dict(intensityMus=[1.0, 0.5], intensityPhis=[0.0, 40.0], useRussianRouletteForIntensity=True, zetaMin=0.3)
for name, d, tab, n, params in (("step cloud flux", cases.step_cloud(), hg_table(), 50000000, {}),
                                ("landsat-36 flux", cases.landsat_cloud(nlayers=36), hg_table(0.85, 299), 20000000, {}),
                                ("radar 640 flux", cases.radar_cloud(), hg_table(0.85, 299), 20000000, {}),
                                ("landsat-119 + 2 directions", cases.landsat_cloud(ssa=0.99), hg_table(0.85, 299), 2000000, rad),
                                ("radar 640 + 2 directions", cases.radar_cloud(), hg_table(0.85, 299), 5000000, rad)):
    out = []
    for general in (False, True):
        g = make_gpu(d, tab, surfaceAlbedo=0.2, **params); g.set_tuning(forceGeneral=general)
        best = 1e9
        for b in (1, 2, 3):
            g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(0.7, 25.0, n))
            best = min(best, g.kernel_ms())
        out.append(n / best * 1e3)
        g.finalize_Integrator()
    print(f"{name:28s} specialised {out[0]:.3e}  general {out[1]:.3e} photons/s  ({out[1] / out[0]:.2f})", flush=True)
