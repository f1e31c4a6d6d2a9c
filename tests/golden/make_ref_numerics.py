"""Generates tests/golden/ref_numerics.npz from the REFERENCE ITSELF: oracle/_ref/ref_dump is this repository's caller
(oracle/ref_dump.f95) linked with /root/reference/Code/{ErrorMessages,numericUtilities,surfaceProperties,characterUtils}.f95 compiled
unmodified and in place with amdflang -O2 (oracle/Makefile, target _ref) -- the modules on the hot path and its boundary that
need no netCDF.  The file holds the inputs (tests/golden/ref_numerics_io.py: cases) and what the reference answered:

  findIndex                   Code/numericUtilities.f95:195-248   
  computeLobattoTerms         :15-102                             (a13: nodes of the inverse tables' CDF)
  computeGaussLegendreTerms   :104-173
  computeLegendrePolynomials  :175-193                            (a13: phase-function values from moments)
  computeSurfaceReflectance   Code/surfaceProperties.f95:121-162  (a11)
  ErrorMessages               Code/ErrorMessages.f95:92-293       (b: the status object every boundary procedure reports through)
  CharacterUtils              Code/characterUtils.f95:14-65

Only numbers are stored; no reference source is copied.  Run in the build container:   python tests/golden/make_ref_numerics.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_numerics_io as io   # noqa: E402


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref"])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    cs = io.cases()
    out = subprocess.run([exe], input=io.script(cs), capture_output=True, text=True, check=True).stdout
    res = io.parse(out, cs)
    flang = subprocess.run(["/opt/rocm/bin/amdflang", "--version"], capture_output=True, text=True).stdout.split("\n")[0]
    header = ("outputs of the reference's numericUtilities / surfaceProperties / ErrorMessages / CharacterUtils (RobertPincus/i3rc-monte-carlo-model, Code/*.f95 unmodified), "
              f"compiled -O2 with {flang}; caller oracle/ref_dump.f95; generator tests/golden/make_ref_numerics.py")
    path = os.path.join(HERE, "ref_numerics.npz")
    io.save(path, cs, res, header)
    n = {k: sum(c["kind"] == k for c in cs) for k in io.INPUT_KEYS}
    print(path, os.path.getsize(path), "bytes;", n)


if __name__ == "__main__":
    main()
