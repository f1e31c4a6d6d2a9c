"""What a workgroup's partial tally sums in LDS do to a batch's tallies: the batches of
tests/test_gpu_parity.py::test_looking_ahead_in_fused_groups_never_changes_a_batch -- the test that failed once in ten runs of the
round-4 suite at rtol = 1e-5 (seed (7, 27)) -- through a plain launch WITH partial sums in LDS against the same launch adding every
tally straight to the float64 buffer (i3rc_hip_set_lds_tallies(h, 0)): the same increments, float64 atomics, nothing else.
Run once per build:   python tools/lds_tallies_ab.py [variant]      (variant: a side build, tools/variant_bench.py build ldsf32="-DI3RC_LDS_F32")
Prints, per batch, the largest relative difference of a tally word and which word it is."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import i3rc_monte_carlo_model_amd as M

if len(sys.argv) > 1 and sys.argv[1] != "default":
    M.build.LIB = M.build.variant_lib(sys.argv[1]); M.build.needs_build = lambda: False
from tools import cases
from tests.sums import order_rtol


def make(d, **params):
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
    g = M.new_Integrator(dom)
    g.specifyParameters(**params)
    g.set_batch_fusion(0)
    return g


d = cases.step_cloud(ssa=0.99, nlayers=8)
lds, wide = make(d, surfaceAlbedo=0.3), make(d, surfaceAlbedo=0.3)
wide.set_lds_tallies(False)
lay = lds.layout()
names = [(lay.fluxUp, "fluxUp"), (lay.fluxDown, "fluxDown"), (lay.fluxAbsorbed, "fluxAbsorbed"), (lay.volumeAbsorption, "volumeAbsorption")]
print("build:", sys.argv[1] if len(sys.argv) > 1 else "default", "| library:", os.path.basename(M.build.LIB))
worst = (0.0, None)
for repeat in range(3):
    for b in list(range(1, 40)) + [100, 101, 102, 103]:
        a = lds.computeRadiativeTransfer(M.new_RandomNumberSequence((7, b)), M.new_PhotonStream(0.8, 25.0, 20000))
        w = wide.computeRadiativeTransfer(M.new_RandomNumberSequence((7, b)), M.new_PhotonStream(0.8, 25.0, 20000))
        assert a["counters"] == w["counters"]
        x, y = a["raw"][:lay.counters], w["raw"][:lay.counters]
        rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-300)
        rel[y == 0] = 0.0
        i = int(np.argmax(rel))
        field = [nm for off, nm in names if off <= i][-1]
        off = [o for o, nm in names if nm == field][0]
        if repeat == 0 and (b in (27, 1, 2) or rel[i] > 1e-6):
            print(f"  seed (7, {b:3d}): largest relative difference {rel[i]:.3e} at {field}[{i - off}] ({x[i]:.9f} against {y[i]:.9f}; {int(a['counters']['scatterings'])} scatterings)")
        if rel[i] > worst[0]:
            worst = (float(rel[i]), (repeat, b, field, i - off, float(x[i]), float(y[i])))
print("kernels:", lds.kernel_name(), "|", wide.kernel_name())
print(f"largest over 3 x 43 batches: {worst[0]:.3e} {worst[1]}")
print(f"order-of-float64-additions bound of tests/sums.py for these batches: {order_rtol(a['counters']):.3e}")
