"""The case / result files of oracle/_ref/ref_loop (oracle/ref_loop.f95's header has the layout): writer, reader, runner.
Test infrastructure, shared by tests/golden/make_ref_loop.py (which runs the reference's loop, build container only) and the tests."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


# ---------------------------------------------------------------------------------------------------------------------------------
# The cases: inputs from the recipes of tools/cases.py (regenerated wherever they are needed), so that the fixture holds outputs only
def cases():
    from oracle import pyoracle as O
    from tools import cases as K

    hg64, hg299 = O.hg_coefficients(0.85, 64), O.hg_coefficients(0.85, 299)
    dirs7 = dict(mus=[1.0, 0.5, 0.5, 0.8, 0.8, 0.3, 0.3], phis=[0.0, 0.0, 180.0, 90.0, 270.0, 45.0, 225.0])

    def one(d, coef, **kw):
        return dict(xe=d["xe"], ye=d["ye"], ze=d["ze"], components=[dict(coefficients=[coef], ext=d["ext"], ssa=d["ssa"], pf=d["pf"])], **kw)
    out = {}
    out["step16"] = one(K.step_cloud(nlayers=16), hg64, nPhotons=20000)
    out["step16_absorbing_surface"] = one(K.step_cloud(ssa=0.99, nlayers=16), hg64, solarMu=0.5, solarAzimuth=30.0, surfaceAlbedo=0.3, nPhotons=20000)
    d = K.two_component(seed=5, nx=6, ny=4, nz=8)
    out["two_components"] = dict(xe=d["xe"], ye=d["ye"], ze=d["ze"], solarMu=0.7, solarAzimuth=10.0, surfaceAlbedo=0.2, nPhotons=20000,
                                 mus=[1.0, 0.6, 0.4], phis=[0.0, 135.0, 300.0],
                                 components=[dict(coefficients=[O.hg_coefficients(0.85, 32), O.hg_coefficients(0.6, 16)], ext=d["ext"][0], ssa=d["ssa"][0], pf=d["pf"][0]),
                                             dict(coefficients=[np.array([0.0, 0.1], np.float32)], ext=d["ext"][1], ssa=d["ssa"][1], pf=d["pf"][1])])
    out["irregular_ground"] = one(K.irregular_domain(seed=3, nx=7, ny=5, nz=9, ssa=0.95, z0=0.0), hg64, solarMu=0.6, solarAzimuth=45.0, surfaceAlbedo=0.5,
                                  mus=[0.9, 0.3], phis=[20.0, 200.0], useRRForIntensity=1, nPhotons=20000)
    # (the same grid lifted by 100: the photons' start height, z0 + (1 - spacing(1)) (zMax - z0), rounds to zMax in float32, the start layer is
    # nz + 1, the tracer reports an error and EVERY photon is dropped -- the reference's own answer is zero everywhere, and so is the oracle's)
    out["irregular"] = one(K.irregular_domain(seed=3, nx=7, ny=5, nz=9, ssa=0.95, z0=100.0), hg64, solarMu=0.6, solarAzimuth=45.0, surfaceAlbedo=0.5,
                           mus=[0.9, 0.3], phis=[20.0, 200.0], useRRForIntensity=1, nPhotons=20000)
    out["hybrid_limit"] = one(K.step_cloud(nlayers=16), hg64, solarMu=0.8, mus=[1.0, 0.5], phis=[0.0, 60.0], useRRForIntensity=1, useHybrid=1, hybridWidth=7.0,
                              numOrdersOrig=1, limitContrib=1, maxContrib=0.5, nPhotons=20000)
    out["max_cross_section"] = one(K.step_cloud(ssa=0.95, nlayers=8), hg64, useRayTracing=0, surfaceAlbedo=0.1, solarMu=0.9, nPhotons=20000)
    d = K.step_cloud(nlayers=16)
    xs = np.linspace(d["xe"][0], d["xe"][-1], 5).astype(np.float32)
    out["surface_grid"] = one(d, hg64, solarMu=0.7, mus=[0.8], phis=[120.0], useRRForIntensity=1, nPhotons=20000,
                              surface=(xs, np.array([d["ye"][0], d["ye"][-1]], np.float32), np.array([[0.1, 0.4, 0.0, 0.8]], np.float32)))
    out["thin_elevated"] = one(dict(K.step_cloud(nlayers=4), ze=(K.step_cloud(nlayers=4)["ze"] + np.float32(20000.0)).astype(np.float32)), hg64, solarMu=0.5, nPhotons=20000)
    out["radar640_nadir"] = one(K.radar_cloud(), hg299, mus=[1.0], phis=[0.0], useRRForIntensity=1, nPhotons=5000)
    ang, val = K.c1_phase_function()
    out["radar640_c1"] = one(K.radar_cloud(ssa=0.99), (ang, val), mus=[1.0], phis=[0.0], useRRForIntensity=1, nPhotons=5000)
    d = K.landsat_cloud(nlayers=36)
    crop = dict(xe=d["xe"][:33], ye=d["ye"][:33], ze=d["ze"], ext=d["ext"][:, 40:72, 40:72].copy(), ssa=d["ssa"][:, 40:72, 40:72].copy(), pf=d["pf"][:, 40:72, 40:72].copy())
    out["landsat_crop_7dir"] = one(crop, hg299, solarMu=0.5, surfaceAlbedo=0.2, useRRForIntensity=1, nPhotons=5000, **dirs7)
    return out
