"""Diagnostic (-DI3RC_PROFILE_PHASES build): share of a wave's cycles in light / event / voxel-step phases of the
radiance state machine and the lanes each phase serves.  python3 tools/phase_profile_radiance.py [--build]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd import build as BLD
from tests import cases

prof = os.path.join(BLD.CSRC, os.environ.get("PROF_LIB", "libi3rc_hip_prof.so"))
if "--build" in sys.argv or not os.path.exists(prof):
    subprocess.check_call([BLD.hipcc()] + BLD.HIPCC_FLAGS + ["-DI3RC_PROFILE_PHASES", "-o", prof, os.path.join(BLD.CSRC, "i3rc_hip.hip")])
    if "--build" in sys.argv:
        sys.exit(0)
BLD.LIB = prof
hg299 = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)])
dirs7 = dict(intensityMus=[1, .5, .5, .8, .8, .3, .3], intensityPhis=[0, 0, 180, 90, 270, 45, 225])
runs = [("radar+nadir", cases.radar_cloud(), dict(intensityMus=[1.0], intensityPhis=[0.0]), 1.0, 20_000_000),
        ("landsat+7", cases.landsat_cloud(), dict(surfaceBDRF=M.new_SurfaceDescription([0.2]), **dirs7), 0.5, 5_000_000)]
for name, d, kw, mu0, n in runs:
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], hg299)
    g = M.new_Integrator(dom)
    g.specifyParameters(minInverseTableSize=10001, minForwardTableSize=10001, useRussianRouletteForIntensity=True, zetaMin=0.3, **kw)
    g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(mu0, 0.0, 1))
    r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(mu0, 0.0, n))
    raw = r["raw"]; lay = g.layout(); c = raw[lay.counters:lay.counters + 16]
    ev, st, nev, nst, lev, lst = c[10:16]
    seg = raw[lay.volumeAbsorption:lay.volumeAbsorption + 9]
    li_lanes, li_cyc, nli = seg[6], seg[7], seg[8]
    tot = ev + st + li_cyc
    print(f"{name}: {n / g.kernel_ms() * 1e3:.3e} photons/s (profiling build) | light {li_cyc / tot * 100:.0f}% of wave time, {li_cyc / max(nli, 1):.0f} cyc/phase, "
          f"{li_lanes / max(nli, 1):.1f} lanes | event {ev / tot * 100:.0f}%, {ev / max(nev, 1):.0f} cyc/phase, {lev / max(nev, 1):.1f} lanes | "
          f"step {st / tot * 100:.0f}%, {st / nst:.0f} cyc/iter, {lst / nst:.1f} lanes | iterations per light phase {nst / max(nli, 1):.2f}")
