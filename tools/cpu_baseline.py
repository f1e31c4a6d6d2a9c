"""CPU baseline leg of bench.py: times the CPU oracle (oracle/, the C restatement of the reference's
computeRT) on the host cores of the box, one process per core over disjoint batches -- the reference's own
decomposition (Example-Drivers/monteCarloDriver.f95:264-274).  Never touches the GPU.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(args):
    first_batch, n_batches, photons, nlayers, mu0 = args
    import numpy as np  # noqa: F401

    from oracle import pyoracle as O
    from tests import cases

    d = cases.step_cloud(nlayers=nlayers)
    inv = O.inverse_table_legendre(O.hg_coefficients(0.85, 64), 10001)
    integ = O.Integrator(d["xe"], d["ye"], d["ze"], d["ext"], d["ssa"], d["pf"], [inv])
    t0 = time.perf_counter()
    fu = 0.0
    cols = []
    for b in range(first_batch, first_batch + n_batches):
        rng = O.RandomNumberSequence([10, b])
        ph = O.photons_directional(rng, mu0, 0.0, photons)
        r = integ.compute(rng, *ph)
        fu += float(r["fluxUp"].mean())
        cols.append((r["fluxUp"].copy(), r["fluxDown"].copy()))
    return time.perf_counter() - t0, fu / n_batches, cols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--batches-per-core", type=int, default=10)
    ap.add_argument("--photons", type=int, default=200000)
    ap.add_argument("--nlayers", type=int, default=16)
    ap.add_argument("--mu0", type=float, default=1.0)
    ap.add_argument("--save", default="", help="write the per-batch flux fields (fluxUp, fluxDown: batch x ny x nx) to this .npz")
    a = ap.parse_args()
    from oracle import pyoracle as O

    O.build()
    cores = a.cores or min(16, len(os.sched_getaffinity(0)))
    jobs = [(1 + i * a.batches_per_core, a.batches_per_core, a.photons, a.nlayers, a.mu0) for i in range(cores)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_worker, jobs))
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    if a.save:
        import numpy as np

        np.savez(a.save, fluxUp=np.stack([c[0] for r in res for c in r[2]]), fluxDown=np.stack([c[1] for r in res for c in r[2]]))
    total = cores * a.batches_per_core * a.photons
    print(json.dumps({"value": total / busy, "unit": "photons/s", "cores": cores, "kind": "port",
                      "sample": f"{cores} processes x {a.batches_per_core} batches x {a.photons} photons of the same step cloud "
                                f"(32x1x{a.nlayers}, mu0={a.mu0}), max busy time {busy:.1f} s, wall {wall:.1f} s",
                      "meanFluxUp": sum(r[1] for r in res) / len(res)}))


if __name__ == "__main__":
    main()
