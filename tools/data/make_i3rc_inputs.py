"""Generates tools/data/i3rc_phase1_inputs.npz from the I3RC phase-1 DATA files shipped with the reference
(I3RC-Examples/Data/: radar optical depths, Landsat optical depth / thickness fields, C1 phase function).
Data only -- no reference source text is copied.  Run in the build container (needs /root/reference)."""
import os

import numpy as np

REF = "/root/reference/I3RC-Examples/Data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "i3rc_phase1_inputs.npz")


def fixed(path, width, per_row):
    rows = []
    for line in open(path):
        line = line.rstrip("\n")
        rows.append([float(line[i * width:(i + 1) * width]) for i in range(per_row)])
    return np.array(rows, dtype=np.float32)


def main():
    mmcr = fixed(os.path.join(REF, "mmcr_tau_32km_020898"), 8, 640)       # i3rcRadarCloud.f95:107-110 '(640f8.3)'
    tau = fixed(os.path.join(REF, "scene43.tau.128x128"), 7, 128)         # i3rcLandsatCloud.f95:69-72 '(128f7.2)'
    dz = fixed(os.path.join(REF, "scene43.dz.128x128"), 7, 128)
    c1 = np.loadtxt(os.path.join(REF, "C.1_PF"), dtype=np.float64).astype(np.float32)   # angle (deg), value
    leg = np.loadtxt(os.path.join(REF, "C.1_leg_coef"), dtype=np.float64).astype(np.float32)
    assert mmcr.shape == (54, 640) and tau.shape == (128, 128) and dz.shape == (128, 128) and c1.shape == (1801, 2)
    np.savez_compressed(OUT, mmcr_tau=mmcr, landsat_tau=tau, landsat_dz_km=dz, c1_angle_deg=c1[:, 0], c1_value=c1[:, 1],
                        c1_legendre=leg)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
