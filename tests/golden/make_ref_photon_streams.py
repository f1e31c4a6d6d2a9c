"""Makes tests/golden/ref_photon_streams.npz: the five arrays (x, y, z, mu, phi) of the six photon-stream constructors of the REFERENCE'S
Code/monteCarloIllumination.f95, as bit patterns -- this repository's fortran/tools/dumpPhotonStreams.f95 compiled against the reference's
unmodified modules (oracle/Makefile, target _ref_loop: oracle/_ref/ref_streams; what that build is: oracle/ref_loop.f95's header), seed
(/ 10, 1 /), 500 photons a stream.  Only runs where /root/reference exists.   usage: python3 tests/golden/make_ref_photon_streams.py"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
N = 500

if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref_loop"])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_streams"), str(N)], capture_output=True, text=True, check=True).stdout
    got = {}
    for line in out.splitlines():
        f = line.split()
        got.setdefault(f[0], []).append([int(h, 16) for h in f[2:7]])
    arrays = {k: np.array(v, np.uint32) for k, v in got.items()}
    assert all(a.shape == (N, 5) for a in arrays.values()), {k: a.shape for k, a in arrays.items()}
    np.savez_compressed(os.path.join(HERE, "ref_photon_streams.npz"), **arrays)
    print("wrote ref_photon_streams.npz:", sorted(arrays))
