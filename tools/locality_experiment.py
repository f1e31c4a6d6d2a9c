"""What would XCD-resident slabs of the extinction field give?  The Landsat field's first 16 rows tiled 8 times along y make
a 128 x 128 domain that is periodic with period 16 rows: photons started anywhere (default library) and photons started
in the first eighth only (library built with -DI3RC_EXPERIMENT_SLAB) then do the same physics -- same steps and
scatterings per photon -- on a working set of the whole field against one eighth of it.
  python3 tools/variant_bench.py build slab="-DI3RC_EXPERIMENT_SLAB";  python3 tools/locality_experiment.py [nlayers] [photons]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    import i3rc_monte_carlo_model_amd as M
    from tests import cases
    lib, nl, n = sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4]))
    if lib != "default":
        M.build.LIB = os.path.join(M.build.CSRC, f"libi3rc_hip_var_{lib}.so"); M.build.needs_build = lambda: False
    d = cases.landsat_cloud(nlayers=nl)
    for k in ("ext", "ssa", "pf"):
        d[k] = np.ascontiguousarray(np.tile(d[k][:, :16, :], (1, 8, 1)))
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)]))
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 1000))
    best = 1e9
    for b in (1, 2, 3):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(1.0, 0.0, n)); best = min(best, g.kernel_ms())
    c = r["counters"]
    print(f"{lib:8s} {nl} layers: {n / best * 1e3:.3e} photons/s  S={c['cellSteps'] / n:.1f} K={c['scatterings'] / n:.1f} kernel {g.kernel_name()}", flush=True)
else:
    nl = sys.argv[1] if len(sys.argv) > 1 else "36"; n = sys.argv[2] if len(sys.argv) > 2 else "1e8"
    for lib in ("default", "slab"):
        subprocess.call([sys.executable, __file__, "one", lib, nl, n])
