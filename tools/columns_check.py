"""Column records against the plain / bricked field: the same photons, the same counters, the same tallies (to the order of the additions).
  python3 tools/columns_check.py [workload ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

for name in sys.argv[1:] or ["landsat36", "landsat119", "landsat36_absorbing", "landsat119_7dir"]:
    name, w = W.get(name)
    n = 200_000 if W.n_dir(w) else 2_000_000
    out = {}
    for place in ("columns", "linear", "bricks"):
        g, d = W.make_integrator(w)
        assert g.has_column_records(), name
        g.select_grid_place(place)
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 3)), M.new_PhotonStream(w["mu0"], 0.0, n))
        out[place] = (r, g.kernel_name())
    ref, kname = out["columns"]
    for place in ("linear", "bricks"):
        r, kn = out[place]
        assert r["counters"] == ref["counters"], (name, place, r["counters"], ref["counters"])
        for k in ("fluxUp", "fluxDown", "fluxAbsorbed", "intensity"):
            if k in ref:
                np.testing.assert_allclose(r[k], ref[k], rtol=2e-6, atol=1e-9, err_msg=f"{name} {place} {k}")
        print(f"{name}: {kname} == {kn}: counters identical, tallies equal", flush=True)
