import sys, time; sys.path.insert(0,'.')
import numpy as np
import i3rc_monte_carlo_model_amd as M
from tests import cases
d=cases.step_cloud()
table=M.PhaseFunctionTable([M.henyey_greenstein(0.85,64)])
dom=M.new_Domain(d["xe"],d["ye"],d["ze"]); dom.addOpticalComponent("c",d["ext"],d["ssa"],d["pf"],table)
g=M.new_Integrator(dom); g.specifyParameters(minInverseTableSize=10001)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10,0)),M.new_PhotonStream(1.0,0.0,100000))
for thr in (1,8,16,24,32,40,48,56,64):
  for bpc in (0,):
    g.set_tuning(thr,bpc)
    n=20_000_000
    t0=time.time(); r=g.computeRadiativeTransfer(M.new_RandomNumberSequence((10,1)),M.new_PhotonStream(1.0,0.0,n)); dt=time.time()-t0
    ms=g.kernel_ms()
    c=r["counters"]
    print(f"thr {thr} bpc {bpc}: kernel {ms:.1f} ms  {n/ms*1e3:.3e} photons/s  wall {dt:.2f}s  Fup {r['fluxUp'].mean():.5f} steps/ph {c['cellSteps']/n:.2f} scat/ph {c['scatterings']/n:.2f} draws/ph {c['rngDraws']/n:.1f} dropped {c['dropped']/n:.2e}", flush=True)
