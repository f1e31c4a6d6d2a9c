"""Compile csrc with extra -D flags into side libraries (here, on the build host) and time each on the GPU box.
   build : python tools/variant_bench.py build name1="-DX=1 -DY" name2=...
   run   : python tools/variant_bench.py run name1 name2 ...   (each in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module

def lib(name):
    BLD = import_module("i3rc_monte_carlo_model_amd").build
    return os.path.join(BLD.CSRC, f"libi3rc_hip_var_{name}.so")

if sys.argv[1] == "build":
    BLD = import_module("i3rc_monte_carlo_model_amd").build
    for spec in sys.argv[2:]:
        name, flags = spec.split("=", 1)
        subprocess.check_call([BLD.hipcc()] + BLD.HIPCC_FLAGS + flags.split() + ["-o", lib(name), os.path.join(BLD.CSRC, "i3rc_hip.hip")])
        print("built", name)
elif sys.argv[1] == "run":
    for name in sys.argv[2:]:
        subprocess.call([sys.executable, __file__, "one", name])
else:
    name = sys.argv[2]
    M = import_module("i3rc_monte_carlo_model_amd")
    M.build.LIB = lib(name) if name != "default" else M.build.LIB
    from tools import cases
    for nl in (16, 32):
        d = cases.step_cloud(nlayers=nl)
        dom = M.new_Domain(d["xe"], d["ye"], d["ze"]); dom.addOpticalComponent("c", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
        g = M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
        g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(1.0, 0.0, 100000))
        n = 100_000_000
        out = []
        for thr in (0, 0):
            g.set_tuning(thr, 0)
            r = g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 1)), M.new_PhotonStream(1.0, 0.0, n))
            out.append(f"thr{thr} {n / g.kernel_ms() * 1e3:.3e}")
        print(f"{name:12s} nlayers {nl}: " + "  ".join(out) + f"  Fup {r['fluxUp'].mean():.5f}", flush=True)
