"""Diagnostic (-DI3RC_PROFILE_PHASES build of the library): where a wave's cycles go -- event / voxel-step (photons) /
voxel-step (shadow rays) / expand / service -- with the number of phases, cycles per phase and lanes
served per phase.  s_memtime instrumentation costs ~10 % itself: read shares, not absolute rates.
  python3 tools/phase_profile.py [--build] <workload> [photons] ...      (workloads: tools/workloads.py; omega = 1 only)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import build as BLD
from tools import workloads as W

prof = os.path.join(BLD.CSRC, os.environ.get("PROF_LIB", "libi3rc_hip_prof.so"))
args = [a for a in sys.argv[1:] if a != "--build"]
if "--build" in sys.argv or not os.path.exists(prof):
    subprocess.check_call([BLD.hipcc()] + BLD.HIPCC_FLAGS + ["-DI3RC_PROFILE_PHASES"] + os.environ.get("EXTRA", "").split() +
                          ["-o", prof, os.path.join(BLD.CSRC, "i3rc_hip.hip")])
    if not args:
        sys.exit(0)
BLD.LIB = prof
BLD.needs_build = lambda: False
KINDS = ["event", "step(photon)", "step(ray)", "expand", "service"]
i = 0
while i < len(args):
    name, w = W.get(args[i]); i += 1
    n = w["photons"] // 10
    if i < len(args) and args[i][0].isdigit():
        n = int(float(args[i])); i += 1
    g, d = W.make_integrator(w)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 1))
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(w["mu0"], 0.0, n))
    lay = g.layout()
    v = r["raw"][lay.volumeAbsorption:lay.volumeAbsorption + 3 * len(KINDS) + 1]
    total = v[3 * len(KINDS)]
    c = r["counters"]
    print(f"{name}: {n / g.kernel_ms() * 1e3:.3e} photons/s (profiling build), {n} photons, kernel {g.kernel_name()}")
    acc = 0.0
    for k, kind in enumerate(KINDS):
        cyc, cnt, lanes = v[3 * k:3 * k + 3]
        acc += cyc
        if cnt:
            print(f"  {kind:13s} {cyc / total * 100:5.1f} % of wave time  {cnt / n:9.3f} phases/photon  {cyc / cnt:7.0f} cycles/phase  {lanes / cnt:5.1f} lanes/phase")
    print(f"  {'(loop logic)':13s} {(total - acc) / total * 100:5.1f} %")
    g.finalize_Integrator()
