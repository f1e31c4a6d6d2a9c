"""The DEVICE's findIndex and computeSurfaceReflectance (csrc/tracer.hpp find_index, surface_reflectance -- rows a5, a9, a11 of
SURVEY.md section 8) against the reference's own routines, through the C ABI: tests/golden/ref_numerics.npz holds the answers of
Code/numericUtilities.f95:195-248 and Code/surfaceProperties.f95:121-162 compiled unmodified (tests/golden/make_ref_numerics.py).
Integer / bit equality; the CPU side of the same pin is tests/test_ref_numerics.py."""
import os
import sys

import numpy as np
import pytest

import i3rc_monte_carlo_model_amd as M
from tools import cases

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_numerics_io as io   # noqa: E402

pytestmark = pytest.mark.gpu
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_numerics.npz")


def _integrator():
    d = cases.plane_parallel()
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)]))
    return M.new_Integrator(dom)


def test_device_find_index_equals_the_reference():
    cs, rs, _ = io.load(FIXTURE)
    g = _integrator()
    n = 0
    for c, r in zip(cs, rs):
        if c["kind"] != "findIndex":
            continue
        got = g.find_index(c["values"], c["table"], c["guess"])
        bad = np.nonzero(got != r["index"])[0]
        assert len(bad) == 0, (c["name"], len(bad), c["values"][bad[:5]], got[bad[:5]], r["index"][bad[:5]])
        n += len(got)
    assert n > 5000


def test_device_surface_reflectance_equals_the_reference():
    cs, rs, _ = io.load(FIXTURE)
    g = _integrator()
    seen = 0
    for c, r in zip(cs, rs):
        if c["kind"] == "surface" and int(r["refused"]) == 0:
            g.specifyParameters(surfaceBDRF=M.new_SurfaceDescription(c["R"][None].transpose(0, 2, 1), c["xs"], c["ys"]))
        elif c["kind"] == "uniform":
            g.specifyParameters(surfaceBDRF=M.new_SurfaceDescription(c["R"]))
        else:
            continue
        got = g.surface_reflectance(c["x"], c["y"])
        assert np.array_equal(got.view(np.int32), r["reflectance"].view(np.int32)), (c["name"], got[:5], r["reflectance"][:5])
        seen += 1
    assert seen == 4
