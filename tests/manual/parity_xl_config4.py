"""Config 4 (Landsat-119 + 7 directions + surface) at the largest sample of round 4: per-batch domain means of four independent samples,
kept in profiles/r04_parity_xl_config4_means.npz, and the table made from them (profiles/r04_parity_xl.txt, last block):

  oracle_local  tools/cpu_baseline.py --config landsat119_7dir --cores 7 --batches-per-core 586 --photons 40000 --first-batch 100001 --save ...
                (the build container's cores, 52 minutes: 1.64e8 photons)
  oracle_box    the same on the GPU box's 16 cores, --batches-per-core 128 --first-batch 200001 (8.2e7 photons)
  gpu_queue     tools/gpu_means.py landsat119_7dir 1600 1e6 (the production kernels: ray queue, lazy roulette; seeds (191, b))
  gpu_nested    the same program, 600 batches, on the measurement build -DI3RC_NESTED_BUILD: the general kernels with the local estimate
                in the reference's nested order (every ray traced, roulette after the trace, libm in the weights), production random
                streams; seeds (291, b)

Columns: fluxUp, fluxDown, radiance 0 ... 6 (domain means of one batch each).  Why: at 8.2e7 oracle photons (batches 1 ... 2048) one of
ten means, the radiance at mu = 0.5, phi = 0, had stood 3.0 combined standard errors above the GPU's -- the samples here are the ones
drawn AFTER that, to see whether it was the estimate or the estimator.   usage: python tests/manual/parity_xl_config4.py"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = np.load(os.path.join(ROOT, "profiles", "r04_parity_xl_config4_means.npz"))
names = ["fluxUp", "fluxDown"] + [f"radiance {d}" for d in range(7)]


def ms(x):
    return x.mean(0), x.std(0, ddof=1) / np.sqrt(len(x))


oracle = np.concatenate([z["oracle_local"], z["oracle_box"]])
mo, so = ms(oracle)
mq, sq = ms(z["gpu_queue"])
mn, sn = ms(z["gpu_nested"])
print(f"oracle {len(oracle)} batches of 40000 = {len(oracle) * 40000:.3g} photons; GPU ray queue {len(z['gpu_queue'])} x 1e6; GPU nested order {len(z['gpu_nested'])} x 1e6")
worst = 0.0
for k, n in enumerate(names):
    zq, zn, zqn = (mq[k] - mo[k]) / np.hypot(sq[k], so[k]), (mn[k] - mo[k]) / np.hypot(sn[k], so[k]), (mq[k] - mn[k]) / np.hypot(sq[k], sn[k])
    worst = max(worst, abs(zq), abs(zn), abs(zqn))
    print(f"{n:11s} oracle {mo[k]:.6f} ({so[k]:.1e}) | GPU queue {mq[k]:.6f} ({sq[k]:.1e}) z {zq:+.2f} | GPU nested {mn[k]:.6f} ({sn[k]:.1e}) z {zn:+.2f} | queue - nested z {zqn:+.2f}")
print(f"largest |z|: {worst:.2f}")

# ---- radar-64 + nadir (config 2): three implementations of the local estimate on the device, per-batch means of 10^6 photons each in
# profiles/r04_parity_xl_radar64_means.npz (columns fluxUp, fluxDown, nadir radiance): direct = the production kernels without an event
# ring (4e9 photons, seeds (391, b)), ring = the same problem through the event ring (I3RC_DIRECT=0; 4e9, (491, b)), nested = the
# measurement build -DI3RC_NESTED_BUILD (3e9, (591, b))
r = np.load(os.path.join(ROOT, "profiles", "r04_parity_xl_radar64_means.npz"))
print("radar-64 + nadir: " + ", ".join(f"{k} {len(r[k])} x 1e6" for k in ("direct", "ring", "nested")))
for k, n in enumerate(("fluxUp", "fluxDown", "nadir radiance")):
    m = {key: (r[key][:, k].astype(np.float64).mean(), r[key][:, k].astype(np.float64).std(ddof=1) / np.sqrt(len(r[key]))) for key in ("direct", "ring", "nested")}
    zs = {f"{a} - {b}": (m[a][0] - m[b][0]) / np.hypot(m[a][1], m[b][1]) for a, b in (("direct", "nested"), ("ring", "nested"), ("direct", "ring"))}
    print(f"{n:15s} " + " | ".join(f"{key} {v[0]:.6f} ({v[1]:.1e})" for key, v in m.items()) + " | " + ", ".join(f"{key} z {v:+.2f}" for key, v in zs.items()))
