"""The problems of tests/golden/ref_loop.npz (the reference's own photon loop: oracle/ref_loop.f95); the file formats and the runner are
oracle/ref_loop_io.py's."""
import numpy as np

from oracle.ref_loop_io import DEFAULTS, REF_LOOP, ROOT, read_result, run, write_case  # noqa: F401
import oracle.ref_loop_io as _io


def __getattr__(name):   # (REF_LOOP is a setting of the runner: tests point it at the shell's build of the same caller)
    return getattr(_io, name)


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


def big_cases():
    """BASELINE.json's configurations at their full grids (the fixture holds the SHA-256 of each field, not the field)"""
    from oracle import pyoracle as O
    from tools import cases as K

    hg64, hg299 = O.hg_coefficients(0.85, 64), O.hg_coefficients(0.85, 299)
    dirs7 = dict(mus=[1.0, 0.5, 0.5, 0.8, 0.8, 0.3, 0.3], phis=[0.0, 0.0, 180.0, 90.0, 270.0, 45.0, 225.0])

    def one(d, coef, **kw):
        return dict(xe=d["xe"], ye=d["ye"], ze=d["ze"], components=[dict(coefficients=[coef], ext=d["ext"], ssa=d["ssa"], pf=d["pf"])], dumpTables=0, **kw)
    return {"config0_step32": one(K.step_cloud(nlayers=32), hg64, solarMu=0.5, nPhotons=100000),                          # the reference generator's own shape
            "config2_radar64_nadir": one(K.radar_cloud_64(), hg299, mus=[1.0], phis=[0.0], useRRForIntensity=1, nPhotons=20000),
            "config3_landsat36": one(K.landsat_cloud(nlayers=36), hg299, nPhotons=50000),
            "config4_landsat119_7dir": one(K.landsat_cloud(), hg299, solarMu=0.5, useRRForIntensity=1, nPhotons=10000,
                                           surface=(np.array([0.0, np.finfo(np.float32).max], np.float32), np.array([0.0, np.finfo(np.float32).max], np.float32),
                                                    np.array([[0.2]], np.float32)), **dirs7)}
