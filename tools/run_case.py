"""One launch of one benchmark workload (the program tools/pmc_profile.sh profiles): no torch, no oracle.
  python3 tools/run_case.py <workload> [photons]        workloads: tools/workloads.py (step16 radar64_nadir landsat36 ...)
env: GRID=auto|linear|bricks|columns where the extinction field is read from, KERNEL=auto|lane|general|ring the kernel family, EVTHR= / LITHR= fixed event / service thresholds, BATCH= batch number of the seed (default 1), I3RC_LIB= another build of the library (A/B comparisons), REPEAT= launches"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

if os.environ.get("I3RC_LIB"):
    M.build.LIB = os.path.abspath(os.environ["I3RC_LIB"]); M.build.needs_build = lambda: False

PROFILE_PHOTONS = {"step16": 20_000_000, "step32": 20_000_000, "radar640": 10_000_000, "radar640_nadir": 5_000_000,
                   "radar64_nadir": 5_000_000, "landsat119": 10_000_000, "landsat36": 10_000_000, "landsat119_7dir": 1_000_000,
                   "landsat36_7dir": 1_000_000, "landsat119_gas": 5_000_000, "landsat36_gas": 5_000_000, "landsat119_gas_7dir": 1_000_000,
                   "landsat119_irregular_7dir": 1_000_000, "landsat119_brdfgrid_7dir": 1_000_000, "les_stcu_rayleigh": 10_000_000, "step16_absorbing": 20_000_000, "landsat36_aerosol_gas": 10_000_000,
                   "landsat119_absorbing": 10_000_000, "landsat36_absorbing": 10_000_000,
                   "les_stcu_rayleigh_2dir": 5_000_000}
name, w = W.get(sys.argv[1])
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else PROFILE_PHOTONS[name]
g, d = W.make_integrator(w)
if os.environ.get("EVTHR") or os.environ.get("LITHR"):   # fixed phase thresholds instead of the per-wave adaptive ones
    g.set_tuning(evThreshold=int(os.environ.get("EVTHR", "0")), lightThreshold=int(os.environ.get("LITHR", "0")))
if os.environ.get("GRID"):
    g.select_grid_place(os.environ["GRID"])
if os.environ.get("KERNEL"):   # auto | lane | general | ring: which kernel family runs the launch (i3rc_hip_select_kernel)
    g.set_tuning(kernel=os.environ["KERNEL"])
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1))   # tables
seed = int(os.environ.get("BATCH", "1"))
for k in range(int(os.environ.get("REPEAT", "1"))):
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, seed + k)), M.new_PhotonStream(w["mu0"], 0.0, n))
    c = r["counters"]
    print(f"{name}: {n / g.kernel_ms() * 1e3:.3e} photons/s ({g.kernel_ms():.2f} ms, {n} photons) kernel {g.kernel_name()} "
          f"S={(c['cellSteps'] + c['shadowSteps']) / n:.1f} K={c['scatterings'] / n:.1f}", flush=True)
