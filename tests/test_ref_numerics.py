"""The oracle, the shell and the host mirror against THE REFERENCE ITSELF, bit for bit (and, for the status object and the string
conversions of the boundary -- ErrorMessages, CharacterUtils --, line for line).

tests/golden/ref_numerics.npz holds the answers of the reference's own numericUtilities / surfaceProperties -- the three
netCDF-free modules of the hot path, compiled unmodified and in place (oracle/Makefile, target _ref; the generator is
tests/golden/make_ref_numerics.py): findIndex (Code/numericUtilities.f95:195-248: SURVEY.md section 8 rows a5, a9),
computeLobattoTerms / computeGaussLegendreTerms / computeLegendrePolynomials (:15-193: the quadrature and recurrence behind
a13's tables) and computeSurfaceReflectance (Code/surfaceProperties.f95:121-162: a11).  Integer and bit-pattern equality, no
tolerance anywhere."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_numerics_io as io   # noqa: E402

from oracle import pyoracle as O   # noqa: E402

FIXTURE = os.path.join(HERE, "golden", "ref_numerics.npz")
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
SHELL_DUMP = os.path.join(ROOT, "i3rc-monte-carlo-model_amd", "fortran", "build", "shellNumericsDump")


@pytest.fixture(scope="module")
def ref():
    cs, rs, header = io.load(FIXTURE)
    assert "unmodified" in header and "flang" in header
    return cs, rs


def _same_bits(a, b, what):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.nonzero(a.view(np.int32) != b.view(np.int32))[0] if a.ndim == 1 else np.argwhere(a.view(np.int32) != b.view(np.int32))
    assert len(bad) == 0, (what, len(bad), bad[:5], a[tuple(bad[0])] if a.ndim > 1 else a[bad[0]], b[tuple(bad[0])] if b.ndim > 1 else b[bad[0]])


def test_the_fixture_covers_what_it_says(ref):
    cs, rs = ref
    kinds = [c["kind"] for c in cs]
    assert kinds.count("findIndex") >= 20 and kinds.count("lobatto") >= 15 and kinds.count("gauss") >= 15
    assert {c["maxL"] for c in cs if c["kind"] == "legendre"} >= {64, 299}     # the moment counts of the BASELINE tables
    assert {c["n"] for c in cs if c["kind"] == "lobatto"} >= {2, 64, 299}
    names = {c["name"] for c in cs if c["kind"] == "findIndex"}
    assert {"cum3", "cum3+guess", "stepX", "irregularZ+guess", "cdf+chain"} <= names   # (/0, cum/) of :637, edges with and without firstGuess
    n = sum(len(r["index"]) for c, r in zip(cs, rs) if c["kind"] == "findIndex")
    assert n > 5000, n
    # positions outside the surface's period are in it
    s = [c for c in cs if c["kind"] == "surface" and c["name"] == "grid4x3"][0]
    assert (s["x"] > s["xs"][-1]).any() and (s["x"] < s["xs"][0]).any() and (s["y"] > s["ys"][-1]).any() and (s["y"] < s["ys"][0]).any()
    assert [int(r["refused"]) for c, r in zip(cs, rs) if c["kind"] == "surface"] == [0, 0, 0, 1, 1]
    # the status object of the boundary: a history that overflows (100 messages), a text longer than a message (256), every state
    e = [(c, r) for c, r in zip(cs, rs) if c["kind"] == "errors"][0]
    assert len(e[0]["ops"]) > 110 and "limits 100 256" in list(e[1]["lines"]) and sum(l.startswith("[overflow") for l in e[1]["lines"]) > 3000
    ch = [(c, r) for c, r in zip(cs, rs) if c["kind"] == "chars"][0]
    assert list(ch[1]["lines"])[3] == "[42] 25"


def test_oracle_find_index_equals_the_reference(ref):
    cs, rs = ref
    for c, r in zip(cs, rs):
        if c["kind"] != "findIndex":
            continue
        g = c["guess"] if c["guess"] is not None else np.zeros(len(c["values"]), np.int32)
        got = np.array([O.find_index(v, c["table"], int(k)) for v, k in zip(c["values"], g)], np.int32)
        bad = np.nonzero(got != r["index"])[0]
        assert len(bad) == 0, (c["name"], len(bad), c["values"][bad[:5]], got[bad[:5]], r["index"][bad[:5]])


def test_host_mirror_find_index_equals_the_reference(ref):
    import i3rc_monte_carlo_model_amd as M

    cs, rs = ref
    for c, r in zip(cs, rs):
        if c["kind"] != "findIndex":
            continue
        g = c["guess"] if c["guess"] is not None else np.zeros(len(c["values"]), np.int32)
        step = max(1, len(g) // 400)   # (pure Python: a sample of every case)
        got = np.array([M.phasefunctions.find_index(np.float32(v), c["table"], int(k)) for v, k in zip(c["values"][::step], g[::step])], np.int32)
        assert (got == r["index"][::step]).all(), c["name"]


def test_oracle_quadratures_and_legendre_recurrence_equal_the_reference(ref):
    cs, rs = ref
    for c, r in zip(cs, rs):
        if c["kind"] == "lobatto":
            mus, w = O.lobatto(c["n"])
            _same_bits(mus, r["mus"], ("lobatto mus", c["n"])); _same_bits(w, r["weights"], ("lobatto weights", c["n"]))
        elif c["kind"] == "gauss":
            mus, w = O.gauss_legendre(c["n"])
            _same_bits(mus, r["mus"], ("gauss mus", c["n"])); _same_bits(w, r["weights"], ("gauss weights", c["n"]))
        elif c["kind"] == "legendre":
            _same_bits(O.legendre_polynomials(c["maxL"], c["mus"]), r["P"], ("legendre", c["maxL"]))


def test_host_mirror_lobatto_and_legendre_equal_the_reference(ref):
    import i3rc_monte_carlo_model_amd as M

    cs, rs = ref
    for c, r in zip(cs, rs):
        if c["kind"] == "lobatto":
            mus, w = M.phasefunctions.lobatto(c["n"])
            _same_bits(mus, r["mus"], ("lobatto mus", c["n"])); _same_bits(w, r["weights"], ("lobatto weights", c["n"]))
        elif c["kind"] == "legendre":
            P = np.asarray(M.phasefunctions.legendre_polynomials(c["maxL"], c["mus"]), np.float32)
            if P.shape != r["P"].shape:
                P = P.T
            _same_bits(P, r["P"], ("legendre", c["maxL"]))


def test_oracle_surface_reflectance_equals_the_reference(ref):
    cs, rs = ref
    seen = 0
    for c, r in zip(cs, rs):
        if c["kind"] == "surface" and int(r["refused"]) == 0:
            _same_bits(O.surface_reflectance(c["xs"], c["ys"], c["R"], c["x"], c["y"]), r["reflectance"], c["name"])
            seen += 1
        elif c["kind"] == "uniform":
            big = np.float32(np.finfo(np.float32).max)   # newSurfaceUniform: positions (/0, huge/) (Code/surfaceProperties.f95:106-107)
            _same_bits(O.surface_reflectance([0.0, big], [0.0, big], c["R"].reshape(1, 1), c["x"], c["y"]), r["reflectance"], "uniform")
            assert (r["reflectance"] == c["R"][0]).all()
            seen += 1
    assert seen == 4


def _run_dump(exe, cs):
    out = subprocess.run([exe], input=io.script(cs), capture_output=True, text=True, check=True, timeout=300).stdout
    return io.parse(out, cs)


def _equal_results(cs, rs, got, who):
    for c, r, g in zip(cs, rs, got):
        for k in r:
            a, b = np.asarray(r[k]), np.asarray(g[k])
            if a.dtype == np.float32:
                _same_bits(b.ravel(), a.ravel(), (who, c["kind"], c["name"], k))
            else:
                assert (a == b).all(), (who, c["kind"], c["name"], k)


def test_shell_modules_equal_the_reference(ref):
    """oracle/ref_dump.f95 linked with the shell's numericUtilities / surfaceProperties / ErrorMessages
    (fortran/build/shellNumericsDump) answers what it answered linked with the reference's."""
    if not os.path.exists(SHELL_DUMP):
        pytest.skip("fortran/build/shellNumericsDump not built (__graft_entry__.build())")
    cs, rs = ref
    _equal_results(cs, rs, _run_dump(SHELL_DUMP, cs), "shell")


def test_the_reference_binary_still_answers_the_fixture(ref):
    """Where oracle/_ref/ref_dump exists (the build container makes it in build(); it travels with the tree), the fixture is what
    it says now -- so the committed numbers cannot drift from the reference they are quoted from."""
    if not os.path.exists(REF_DUMP):
        pytest.skip("oracle/_ref/ref_dump not built (needs /root/reference: make -C oracle _ref)")
    cs, rs = ref
    _equal_results(cs, rs, _run_dump(REF_DUMP, cs), "oracle/_ref")
    fresh = io.cases()   # ... and the committed inputs are what the generator makes
    assert [f"{c['kind']}|{c['name']}" for c in fresh] == [f"{c['kind']}|{c['name']}" for c in cs]


def test_what_the_reference_refuses_the_host_refuses():
    import i3rc_monte_carlo_model_amd as M

    cs, rs, _ = io.load(FIXTURE)
    for c, r in zip(cs, rs):
        if c["kind"] == "surface" and int(r["refused"]) == 1:
            with pytest.raises(M.I3RCError):
                M.new_SurfaceDescription(c["R"][None].transpose(0, 2, 1), c["xs"], c["ys"])
