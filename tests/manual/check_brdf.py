import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from oracle import pyoracle as O
from tools import cases
from tests.test_gpu_parity import make_gpu, make_oracle, hg_table, _batches_gpu, _batches_oracle
d = cases.step_cloud(ssa=1.0, nlayers=8)
xs = np.array([0.0, 100.0, 350.0, 500.0], np.float32); ys = np.array([0.0, 500.0], np.float32)
alb = np.array([[0.1, 0.6, 0.9]], np.float32)
g = make_gpu(d, hg_table(), surfaceBDRF=M.new_SurfaceDescription(alb.T[None].copy(), xs, ys))
o = make_oracle(O, d, [hg_table().inverse_table(9001)]); o.specify(surfaceBDRF=(xs, ys, alb))
for thr in (1, 40, 64):
    g.set_tuning(thr, 0)
    gr = _batches_gpu(g, 16, 400000, 0.8, az=45.0)
    for key in ("fluxUp", "fluxDown"):
        a = np.array([r[key].mean(dtype=np.float64) for r in gr])
        print("gpu thr", thr, key, a.mean(), a.std(ddof=1)/4)
    c = gr[0]["counters"]; print({k: v/400000 for k, v in c.items()})
orr = _batches_oracle(O, o, 16, 50000, 0.8, az=45.0)
for key in ("fluxUp", "fluxDown"):
    a = np.array([r[key].mean(dtype=np.float64) for r in orr]); print("oracle", key, a.mean(), a.std(ddof=1)/4)
print({k: orr[0][k]/50000 for k in ("cellSteps","scatterings","surfaceHits","roulettePlays","exitsTop","nBad")})
