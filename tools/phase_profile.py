"""Diagnostic: builds csrc with -DI3RC_PROFILE_PHASES into a separate .so, runs the step cloud and prints where a
wave's cycles go (event phase vs voxel-step phase, lanes active in each).  Not used by tests or bench."""
import os, subprocess, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import binding as B, build as BLD
from tests import cases

prof = os.path.join(BLD.CSRC, "libi3rc_hip_prof.so")
if "--build" in sys.argv or not os.path.exists(prof):
    subprocess.check_call([BLD.hipcc()] + BLD.HIPCC_FLAGS + ["-DI3RC_PROFILE_PHASES", "-o", prof, os.path.join(BLD.CSRC, "i3rc_hip.hip")])
    if "--build" in sys.argv: sys.exit(0)
BLD.LIB = prof
nl = int(os.environ.get("NLAYERS", "32"))
d = cases.step_cloud(nlayers=nl)
dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
n = 20_000_000
for thr in (40,):
    g.set_tuning(thr, 0)
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, n))
    raw = r["raw"]; lay = g.layout(); c = raw[lay.counters:lay.counters + 16]
    ev, st, nev, nst, lev, lst = c[10:16]
    seg = raw[lay.volumeAbsorption:lay.volumeAbsorption + 6]
    names = ["A ends", "B indices", "C philox", "C new photon", "C scatter/surface", "C tau+rcp"]
    print("   event-phase segments (cycles per phase): " + ", ".join(f"{nm} {v/nev:.0f}" for nm, v in zip(names, seg)))
    print(f"thr {thr}: {g.kernel_ms():.1f} ms | event phase {ev/(ev+st)*100:.0f}% of wave time, {ev/nev:.0f} cyc/phase, {lev/nev:.1f} lanes | "
          f"step {st/nst:.0f} cyc/iter, {lst/nst:.1f} lanes | per photon: {nev*64/n/64:.4f} ev-phases/lane-photon, iters/phase {nst/nev:.2f}, events/photon {lev/n:.2f}")
