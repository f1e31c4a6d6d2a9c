"""Generates tests/golden/config4_columns.npz: BASELINE.json configs[4] (i3rcLandsatCloud 128 x 128 x 119 + 7 radiance directions +
Lambertian surface 0.2 through the surfaceProperties object, mu0 = 0.5, roulette with zetaMin 0.3: tools/workloads.py
"landsat119_7dir") traced by the CPU ORACLE (oracle/, the C restatement of the reference's computeRT) -- per column the mean over
the batches and its batch standard error (monteCarloDriver.f95:358-378) of fluxUp, fluxDown and the seven radiance fields, the
per-batch domain means and the work counters.  tests/test_gpu_baseline_configs.py compares the HIP path with it column by column.

Run in the build container (all cores, about six minutes):   python tests/golden/make_config4_columns.py [batches] [photons per batch]
The oracle's MT19937 streams are seeded (/ 10, batch /), batch = 1 ... batches, as tools/cpu_baseline.py does."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CONFIG = "landsat119_7dir"


def _one(args):
    first, count, photons = args
    from oracle import pyoracle as O
    from tools import workloads as W

    name, w = W.get(CONFIG)
    integ, _ = W.make_oracle(w)
    out = []
    for b in range(first, first + count):
        rng = O.RandomNumberSequence([10, b])
        r = integ.compute(rng, *O.photons_directional(rng, w["mu0"], 0.0, photons))
        f = np.concatenate([r["fluxUp"][None], r["fluxDown"][None], r["intensity"]]).astype(np.float64)   # [9][ny][nx]
        out.append((f, [int(r[k]) for k in ("nBad", "cellSteps", "scatterings", "tracerCalls")]))
    return out


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    photons = int(float(sys.argv[2])) if len(sys.argv) > 2 else 500_000
    from oracle import pyoracle as O

    O.build()
    cores = len(os.sched_getaffinity(0))
    per = (nb + cores - 1) // cores
    jobs = [(1 + i * per, min(per, nb - i * per), photons) for i in range(cores) if i * per < nb]
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        res = [x for part in ex.map(_one, jobs) for x in part]
    fields = np.stack([f for f, _ in res])                      # [nb][9][128][128]
    counters = np.array([c for _, c in res], np.int64)
    mean = fields.mean(0)
    se = fields.std(0, ddof=1) / np.sqrt(len(fields))
    # The per-column statistic on the oracle against ITSELF (odd batches against even ones): what z = |difference| / combined
    # standard error looks like on two samples of the same code when a column sees thirty photons per batch and a radiance is made
    # of a few large contributions among many small ones -- the baseline the GPU-against-fixture figures are read against.
    # Per field: share of columns within 3 sigma, mean of z^2, largest z.
    a, b = fields[0::2], fields[1::2]
    zz = np.abs(a.mean(0) - b.mean(0)) / (np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) + 1e-7)
    self_check = np.stack([(zz <= 3.0).mean(axis=(1, 2)), (zz ** 2).mean(axis=(1, 2)), zz.max(axis=(1, 2))], axis=1)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config4_columns.npz"),
                        mean=mean.astype(np.float32), stderr=se.astype(np.float32),
                        batchMeans=fields.mean(axis=(2, 3)),   # [nb][9]: fluxUp, fluxDown, radiance 1..7
                        selfCheck=self_check,                  # [9][3]: odd against even batches (see above)
                        counters=counters, counterNames=np.array(["nBad", "cellSteps", "scatterings", "tracerCalls"]),
                        photonsPerBatch=np.int64(photons), batches=np.int64(len(fields)), config=np.array(CONFIG),
                        fieldNames=np.array(["fluxUp", "fluxDown"] + [f"intensity{d + 1}" for d in range(7)]))
    print("odd against even batches, per field (share within 3 sigma, mean z^2, max z):", np.round(self_check, 3).tolist())
    print(f"{len(fields)} batches x {photons} photons in {time.time() - t0:.0f} s on {cores} cores; mean fluxUp {mean[0].mean():.5f} "
          f"fluxDown {mean[1].mean():.5f} radiances {[round(float(v), 5) for v in mean[2:].mean(axis=(1, 2))]}")


if __name__ == "__main__":
    main()
