"""Makes tests/golden/ref_loop.npz: what the REFERENCE'S OWN PHOTON LOOP gives on twelve small problems -- Integrators/monteCarloRadiativeTransfer.f95
and every module of Code/, compiled unmodified and in place behind this repository's caller oracle/ref_loop.f95 (oracle/Makefile, target
_ref_loop; the header of ref_loop.f95 says what that build is and what it is not: its `module netcdf` is this repository's, never called).
Per case and batch: fluxUp, fluxDown, fluxAbsorbed, the absorbed profile, volumeAbsorption and the radiances, as reportResults hands them
out (float32); per component the SHA-256 of the inverse and forward tables the reference's own routines make (the tables themselves where
a restatement cannot be expected to reproduce them bit for bit: the tabulated phase function, whose normalisation is a DOT_PRODUCT in a
compiler-defined order).  The inputs are not stored: tests/golden/ref_loop_io.py regenerates them from tools/cases.py.
Only runs where /root/reference exists.   usage: python3 tests/golden/make_ref_loop.py"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import ref_loop_io as R  # noqa: E402

if __name__ == "__main__":
    if not os.path.isdir("/root/reference/Integrators"):
        raise SystemExit("needs /root/reference")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref_loop"])
    out = {"compiler": np.array(subprocess.run(["/opt/rocm/bin/amdflang", "--version"], capture_output=True, text=True).stdout.splitlines()[0]),
           "cases": np.array(sorted(R.cases()))}
    for name, case in R.cases().items():
        with tempfile.TemporaryDirectory() as tmp:
            batches, tables = R.run(case, tmp)
        for b, res in enumerate(batches):
            for k, v in res.items():
                out[f"{name}/batch{b}/{k}"] = v
        for c, (comp, t) in enumerate(zip(case["components"], tables)):
            out[f"{name}/tables{c}/sha256"] = np.array([hashlib.sha256(t["inverse"].tobytes()).hexdigest(), hashlib.sha256(t["forward"].tobytes()).hexdigest()])
            if any(isinstance(e, tuple) for e in comp["coefficients"]):
                out[f"{name}/tables{c}/inverse"], out[f"{name}/tables{c}/forward"] = t["inverse"], t["forward"]
        print(name, "fluxUp", float(batches[0]["fluxUp"].mean()), "fluxDown", float(batches[0]["fluxDown"].mean()))
    for name, case in R.big_cases().items():        # BASELINE.json's configurations at their full grids: a hash per field
        with tempfile.TemporaryDirectory() as tmp:
            batches, _ = R.run(case, tmp)
        for b, res in enumerate(batches):
            for k, v in res.items():
                out[f"{name}/batch{b}/{k}/sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).hexdigest())
        out[f"{name}/means"] = np.array([[float(r[k].mean(dtype=np.float64)) for k in ("fluxUp", "fluxDown")] for r in batches])
        print(name, "fluxUp", float(batches[0]["fluxUp"].mean()), "fluxDown", float(batches[0]["fluxDown"].mean()))
    out["big_cases"] = np.array(sorted(R.big_cases()))
    path = os.path.join(HERE, "ref_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
