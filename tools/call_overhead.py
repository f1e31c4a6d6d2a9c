"""Wall time of one computeRadiativeTransfer call against the batch size (the reference's drivers use batches of 1e5 ...
1e6 photons): kernel time (HIP events) and everything around it (zero, launch, normalise, copy back).
With `pipe`: the same batches through i3rc_hip_run_batches (several batches on the device at a time).
usage: tools/call_overhead.py [workload] [pipe]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import i3rc_monte_carlo_model_amd as M
from tools import workloads as W

name, w = W.get(sys.argv[1] if len(sys.argv) > 1 else "step32")
g, _ = W.make_integrator(w)
g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, 0)), M.new_PhotonStream(w["mu0"], 0.0, 100000))
for n in (10_000, 100_000, 1_000_000, 10_000_000, 100_000_000):
    reps = 200 if n <= 1_000_000 else 20
    t0 = time.perf_counter(); kms = 0.0
    for b in range(reps):
        g.computeRadiativeTransfer(M.new_RandomNumberSequence((10, b + 1)), M.new_PhotonStream(w["mu0"], 0.0, n))
        kms += g.kernel_ms()
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name} {n:>11,d} photons per call: wall {wall:8.3f} ms, kernel {kms / reps:8.3f} ms, around it {wall - kms / reps:6.3f} ms; "
          f"{n / wall * 1e3:.3e} photons/s end to end, {n / (kms / reps) * 1e3:.3e} in the kernel", flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "pipe":
    for n in (100_000, 1_000_000, 10_000_000):
        nb = 200 if n <= 1_000_000 else 40
        for k in (1, 2, 3, 4, 6, 8):
            t0 = time.perf_counter()
            rs = g.computeRadiativeTransferBatches((10, 1), nb, w["mu0"], 0.0, n, inFlight=k)
            wall = (time.perf_counter() - t0) / nb * 1e3
            print(f"{name} {n:>11,d} photons per batch, {k} in flight: {wall:8.3f} ms per batch, {n / wall * 1e3:.3e} photons/s end to end "
                  f"(incl. normalising every batch on the host)", flush=True)
