"""What would XCD-resident slabs of the extinction field give?  `rows` rows of the Landsat field (from row `first`) tiled 8
times along y make a domain that is periodic with that period: photons started anywhere (default library) and photons
started in the first eighth only (library built with -DI3RC_EXPERIMENT_SLAB) then do the same physics -- same steps and
scatterings per photon -- on a working set of the whole field against one eighth of it.  rows = 16: the 7.8 MB field's
size; rows = 128: the whole scene tiled 8 times, a 62 MB field whose eighth is the scene itself.
  python3 tools/variant_bench.py build slab="-DI3RC_EXPERIMENT_SLAB"
  python3 tools/locality_experiment.py [nlayers] [photons] [first row] [rows]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    import i3rc_monte_carlo_model_amd as M
    from tools import cases
    lib, nl, n = sys.argv[2], int(sys.argv[3]), int(float(sys.argv[4]))
    first, rows = int(sys.argv[5]), int(sys.argv[6])
    if lib != "default":
        M.build.LIB = os.path.join(M.build.CSRC, f"libi3rc_hip_var_{lib}.so"); M.build.needs_build = lambda: False
    d = cases.landsat_cloud(nlayers=nl)
    for k in ("ext", "ssa", "pf"):
        d[k] = np.ascontiguousarray(np.tile(d[k][:, first:first + rows, :], (1, 8, 1)))
    d["ye"] = (np.float32(30.0) * np.arange(0, 8 * rows + 1, dtype=np.float32)).astype(np.float32)
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)]))
    g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 1000))
    best = 1e9
    for b in (1, 2, 3):
        r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b)), M.new_PhotonStream(1.0, 0.0, n)); best = min(best, g.kernel_ms())
    c = r["counters"]
    print(f"{lib:8s} {nl} layers, rows {first}..{first + rows - 1} x 8 ({d['ext'].nbytes / 1e6:.1f} MB): {n / best * 1e3:.3e} photons/s  S={c['cellSteps'] / n:.1f} K={c['scatterings'] / n:.1f} kernel {g.kernel_name()}", flush=True)
else:
    nl = sys.argv[1] if len(sys.argv) > 1 else "36"; n = sys.argv[2] if len(sys.argv) > 2 else "1e8"
    first = sys.argv[3] if len(sys.argv) > 3 else "0"; rows = sys.argv[4] if len(sys.argv) > 4 else "16"
    for lib in ("default", "slab"):
        subprocess.call([sys.executable, __file__, "one", lib, nl, n, first, rows])
