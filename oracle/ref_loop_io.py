"""The case / result files of oracle/_ref/ref_loop (oracle/ref_loop.f95's header has the layout): writer, reader, runner.
Test infrastructure, like everything under oracle/: used by tests/golden/make_ref_loop.py (which runs the reference's loop, build container
only), by tests/test_ref_loop.py and by bench.py's cpu_baseline leg (tools/cpu_baseline.py: the reference's own rate beside the port's)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LOOP = os.path.join(ROOT, "oracle", "_ref", "ref_loop")
f4, i4 = np.dtype("<f4"), np.dtype("<i4")

DEFAULTS = dict(surfaceAlbedo=0.0, useRayTracing=1, useRussianRoulette=1, useRRForIntensity=0, zetaMin=0.3, useHybrid=0, hybridWidth=7.0,
                numOrdersOrig=0, limitContrib=0, maxContrib=77.0, nInverse=10001, nForward=10001, mus=(), phis=(), surface=None,
                solarMu=1.0, solarAzimuth=0.0, nBatches=2, nPhotons=10000, seed=(10, 1), dumpTables=1)


def write_case(path, case):
    """case: dict(xe, ye, ze, components=[dict(coefficients=[array, ...], ext[z,y,x], ssa, pf)], **DEFAULTS overrides)"""
    c = dict(DEFAULTS, **case)
    nz, ny, nx = c["components"][0]["ext"].shape
    with open(path, "wb") as f:
        def w(a, t):
            f.write(np.ascontiguousarray(a, t).tobytes())
        w([nx, ny, nz, len(c["components"])], i4)
        w(c["xe"], f4), w(c["ye"], f4), w(c["ze"], f4)
        for comp in c["components"]:
            w([len(comp["coefficients"])], i4)
            for coef in comp["coefficients"]:                                     # Legendre coefficients, or (angles, values) of a tabulated one
                if isinstance(coef, tuple):
                    w([-len(coef[0])], i4), w(coef[0], f4), w(coef[1], f4)
                else:
                    w([len(coef)], i4), w(coef, f4)
            w(comp["ext"], f4), w(comp["ssa"], f4), w(comp["pf"], i4)          # [z][y][x] in C order = (x, y, z) in Fortran order
        w([c["surfaceAlbedo"]], f4), w([c["useRayTracing"], c["useRussianRoulette"], c["useRRForIntensity"]], i4), w([c["zetaMin"]], f4)
        w([c["useHybrid"]], i4), w([c["hybridWidth"]], f4), w([c["numOrdersOrig"], c["limitContrib"]], i4), w([c["maxContrib"]], f4)
        w([c["nInverse"], c["nForward"], len(c["mus"])], i4), w(c["mus"], f4), w(c["phis"], f4)
        if c["surface"] is None:
            w([0, 0], i4)
        else:
            xs, ys, refl = c["surface"]                                           # refl[y][x]
            w([len(xs) - 1, len(ys) - 1], i4), w(xs, f4), w(ys, f4), w(refl, f4)
        w([c["solarMu"], c["solarAzimuth"]], f4), w([c["nBatches"], c["nPhotons"], c["seed"][0], c["seed"][1], c["dumpTables"]], i4)
    return c


def read_result(path, c):
    nz, ny, nx = c["components"][0]["ext"].shape
    nd = len(c["mus"])
    raw = np.fromfile(path, f4)
    at = 0

    def take(*shape):
        nonlocal at
        n = int(np.prod(shape))
        out = raw[at:at + n].reshape(shape).copy()
        at += n
        return out
    batches = []
    for _ in range(c["nBatches"]):
        b = dict(fluxUp=take(ny, nx), fluxDown=take(ny, nx), fluxAbsorbed=take(ny, nx), absorbedProfile=take(nz), volumeAbsorption=take(nz, ny, nx))
        if nd:
            b["intensity"] = take(nd, ny, nx)
        batches.append(b)
    tables = []
    if c["dumpTables"]:
        for comp in c["components"]:
            ne = len(comp["coefficients"])
            tables.append(dict(inverse=take(ne, c["nInverse"]), forward=take(ne, c["nForward"])))
    assert at == raw.size, (at, raw.size)
    return batches, tables


def run(case, directory):
    cf, rf = os.path.join(directory, "case.bin"), os.path.join(directory, "result.bin")
    c = write_case(cf, case)
    r = subprocess.run([REF_LOOP, cf, rf], capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError(f"ref_loop failed: {r.stdout}\n{r.stderr}")
    return read_result(rf, c)
