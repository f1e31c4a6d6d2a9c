"""Workload recipes -- the domains bench.py, the tools, smoke() and the tests run (moved out of tests/ in round 5: the bench no
longer imports the tests package).  Synthetic domains: the I3RC phase-1 recipes restated from
I3RC-Examples/i3rcStepCloud.f95:27-75 and Example-Drivers/planeParallel.f95:299-379 (float32 arithmetic
in the reference's operator order).  Pure numpy; no oracle or product code imported here."""
import numpy as np

f32 = np.float32


def powi(a, n):
    """real(4)**integer as a Fortran compiler lowers it (square and multiply in float32)."""
    a = f32(a)
    r = f32(1.0)
    while True:
        if n & 1:
            r = f32(r * a)
        n //= 2
        if n == 0:
            break
        a = f32(a * a)
    return r


def hg_coefficients(g, n):
    return np.array([powi(g, l) for l in range(1, n + 1)], dtype=np.float32)


def plane_parallel(optical_depth=1.0, ssa=1.0, nx=1, ny=1, nlayers=1, domain_size=500.0, thickness=250.0):
    xe = (f32(domain_size) / f32(nx)) * np.arange(0, nx + 1, dtype=np.float32)
    ye = (f32(domain_size) / f32(ny)) * np.arange(0, ny + 1, dtype=np.float32)
    ze = (f32(thickness) / f32(nlayers)) * np.arange(0, nlayers + 1, dtype=np.float32)
    ext = np.full((nlayers, ny, nx), f32(optical_depth) / f32(thickness), np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=np.full_like(ext, f32(ssa)), pf=np.ones(ext.shape, np.int32))


def step_cloud(ssa=1.0, nlayers=32, ncolumns=32):
    """32 x 1 x nlayers step cloud: optical depth 2 (columns 1-16) and 18 (17-32), 500 m wide, 250 m thick."""
    dx = f32(500.0) / f32(ncolumns)
    dz = f32(250.0) / f32(nlayers)
    xe = dx * np.arange(0, ncolumns + 1, dtype=np.float32)
    ye = np.array([0.0, 500.0], np.float32)
    ze = dz * np.arange(0, nlayers + 1, dtype=np.float32)
    col = np.concatenate([np.full(ncolumns // 2, 2, np.float32), np.full(ncolumns // 2, 18, np.float32)]) / f32(250.0)
    ext = np.ascontiguousarray(np.broadcast_to(col[None, None, :], (nlayers, 1, ncolumns)), dtype=np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=np.full_like(ext, f32(ssa)), pf=np.ones(ext.shape, np.int32))


def irregular_domain(seed=3, nx=7, ny=5, nz=9, ssa=0.95, z0=0.0):
    """Small irregularly spaced domain with empty cells: exercises findIndex paths and zero-extinction steps.
    z0 = 100 gives the THIN ELEVATED domain of the reference's start-of-photon quirk: photons start at
    z0 + (1 - spacing(1)) (zMax - z0), which rounds to zMax itself, findZIndex answers nz + 1, the first tracer step
    is 0 and every photon is dropped (ray tracing); with z0 = 0 the photons enter the domain."""
    rng = np.random.default_rng(seed)
    xe = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, nx))]).astype(np.float32)
    ye = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, ny))]).astype(np.float32)
    ze = (np.concatenate([[0.0], np.cumsum(rng.uniform(5, 30, nz))]) + z0).astype(np.float32)
    ext = rng.uniform(0.0, 0.08, (nz, ny, nx)).astype(np.float32)
    ext[rng.random(ext.shape) < 0.3] = 0.0
    pf = np.where(ext > 0, 1, 0).astype(np.int32)
    s = np.where(ext > 0, f32(ssa), f32(0.0)).astype(np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=s, pf=pf)


def column_clouds(seed=11, nx=9, ny=6, nz=12, ssa=0.97):
    """Irregularly spaced domain in which every column holds ONE run of layers with ONE extinction value -- the form the kernels keep
    as column records (include/i3rc_hip.h, i3rc_hip_select_grid_place): runs that touch the top, the bottom, both, runs of one
    layer, and clear columns."""
    rng = np.random.default_rng(seed)
    xe = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, nx))]).astype(np.float32)
    ye = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, ny))]).astype(np.float32)
    ze = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 30, nz))]).astype(np.float32)
    ext = np.zeros((nz, ny, nx), np.float32)
    for j in range(ny):
        for i in range(nx):
            kind = rng.integers(0, 6)
            if kind == 0:
                continue                                   # a clear column
            lo, hi = sorted(rng.integers(0, nz, 2))
            if kind == 1: lo = 0                           # from the surface up
            if kind == 2: hi = nz - 1                      # up to the top
            if kind == 3: lo, hi = 0, nz - 1               # the whole column
            if kind == 4: hi = lo                          # one layer
            ext[lo:hi + 1, j, i] = f32(rng.uniform(0.002, 0.1))
    pf = np.where(ext > 0, 1, 0).astype(np.int32)
    s = np.where(ext > 0, f32(ssa), f32(0.0)).astype(np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=s, pf=pf)


# ---- I3RC phase-1 fields (input data: tools/data/i3rc_phase1_inputs.npz, made by tools/data/make_i3rc_inputs.py) ---------------------------
import os as _os

_GOLDEN = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "data", "i3rc_phase1_inputs.npz")


def _inputs():
    return np.load(_GOLDEN)


def radar_cloud(ssa=1.0):
    """640 x 1 x 54 MMCR cloud, I3RC-Examples/i3rcRadarCloud.f95:28-31,107-125: file row j -> layer nz+1-j,
    optical depth per cell / deltaZ; deltaX = 50, deltaZ = 45."""
    tau = _inputs()["mmcr_tau"]                      # [54 rows, 640]
    nz, nx = tau.shape
    ext = (tau[::-1, :] / f32(45.0)).astype(np.float32)[:, None, :]
    xe = f32(50.0) * np.arange(0, nx + 1, dtype=np.float32)
    ye = np.array([0.0, f32(50.0) * f32(nx)], np.float32)
    ze = f32(45.0) * np.arange(0, nz + 1, dtype=np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=np.ascontiguousarray(ext), ssa=np.full(ext.shape, f32(ssa), np.float32),
                pf=np.ones(ext.shape, np.int32))


def radar_cloud_64(ssa=1.0):
    """BASELINE.json's labelled 64 x 64 x 54 synthetic: ext3(i,j,k) = ext2(10*mod(i+j-2,64)+1, k),
    deltaX = deltaY = 500, deltaZ = 45."""
    r = radar_cloud(ssa)
    e2 = r["ext"][:, 0, :]
    i = np.arange(1, 65)[None, :]
    j = np.arange(1, 65)[:, None]
    col = 10 * np.mod(i + j - 2, 64)                 # 0-based column of the radar field
    ext = np.ascontiguousarray(e2[:, col], dtype=np.float32)   # [54, 64(j), 64(i)]
    xe = f32(500.0) * np.arange(0, 65, dtype=np.float32)
    return dict(xe=xe, ye=xe.copy(), ze=r["ze"], ext=ext, ssa=np.full(ext.shape, f32(ssa), np.float32),
                pf=np.ones(ext.shape, np.int32))


def landsat_cloud(ssa=1.0, nlayers=119):
    """128 x 128 x 119 Landsat scene, I3RC-Examples/i3rcLandsatCloud.f95:27-35,92-108: cloud fills the lowest
    nint(dz/20) layers of each column with ext = tau / (nint(dz/20)*20); z0 = 200; deltaXY = 30, deltaZ = 20.
    nlayers=36 gives BASELINE.json's labelled 128 x 128 x 36 synthetic."""
    inp = _inputs()
    tau = inp["landsat_tau"]                         # [y, x]
    thick = (inp["landsat_dz_km"] * f32(1000.0)).astype(np.float32)
    ny, nx = tau.shape
    if nlayers == 119:
        dzl = f32(20.0)
        nfill = np.floor(thick / dzl + f32(0.5)).astype(np.int64)   # Fortran nint(): halves away from zero (np.rint: to even)
        with np.errstate(all="ignore"):
            e = np.where(tau > np.finfo(np.float32).tiny, tau / (nfill.astype(np.float32) * dzl), f32(0.0)).astype(np.float32)
        ze = dzl * np.arange(0, nlayers + 1, dtype=np.float32) + f32(200.0)
    else:
        dzl = f32(2380.0) / f32(nlayers)
        nfill = np.where(tau > 0, np.maximum(1, np.floor(thick / dzl + f32(0.5)).astype(np.int64)), 0)
        with np.errstate(all="ignore"):
            e = np.where(tau > 0, tau / (nfill.astype(np.float32) * dzl), f32(0.0)).astype(np.float32)
        ze = (dzl * np.arange(0, nlayers + 1, dtype=np.float32) + f32(200.0)).astype(np.float32)
    k = np.arange(nlayers)[:, None, None]
    ext = np.where(k < nfill[None], e[None], f32(0.0)).astype(np.float32)
    s = np.where(ext > 0, f32(ssa), f32(0.0)).astype(np.float32)
    pf = np.where(ext > 0, 1, 0).astype(np.int32)
    xe = f32(30.0) * np.arange(0, nx + 1, dtype=np.float32)
    ye = f32(30.0) * np.arange(0, ny + 1, dtype=np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=np.ascontiguousarray(ext), ssa=s, pf=pf)


def landsat_tiled(tiles=2, ssa=1.0):
    """The Landsat scene repeated tiles x tiles times in x and y (256 x 256 x 119 = 31 MB of extinction for tiles = 2): a
    field well beyond the 4 MB of an XCD's L2, the size of a production LES cloud field.  Periodic boundaries make every
    tile's photons statistically those of the single scene: fluxes and radiances have the single scene's means."""
    d = landsat_cloud(ssa=ssa)
    nx, ny = d["ext"].shape[2], d["ext"].shape[1]
    rep = lambda a: np.ascontiguousarray(np.tile(a, (1, tiles, tiles)))
    return dict(xe=f32(30.0) * np.arange(0, nx * tiles + 1, dtype=np.float32), ye=f32(30.0) * np.arange(0, ny * tiles + 1, dtype=np.float32),
                ze=d["ze"], ext=rep(d["ext"]), ssa=rep(d["ssa"]), pf=rep(d["pf"]))


def c1_phase_function():
    """Deirmendjian C1 tabulated phase function (1801 angle/value pairs), i3rcRadarCloud.f95:70-76."""
    inp = _inputs()
    ang = (inp["c1_angle_deg"] * np.arccos(f32(-1.0)) / f32(180.0)).astype(np.float32)
    return ang, inp["c1_value"].astype(np.float32)


def two_component(seed=5, nx=6, ny=4, nz=8):
    """Two components (cloud in the middle layers + a horizontally uniform background), two table entries."""
    rng = np.random.default_rng(seed)
    xe = f32(25.0) * np.arange(0, nx + 1, dtype=np.float32)
    ye = f32(40.0) * np.arange(0, ny + 1, dtype=np.float32)
    ze = f32(30.0) * np.arange(0, nz + 1, dtype=np.float32)
    cloud = np.zeros((nz, ny, nx), np.float32)
    cloud[2:6] = rng.uniform(0.0, 0.06, (4, ny, nx)).astype(np.float32)
    cloud[rng.random(cloud.shape) < 0.2] = 0
    pf1 = np.where(cloud > 0, rng.integers(1, 3, cloud.shape), 0).astype(np.int32)
    ssa1 = np.where(cloud > 0, f32(0.98), f32(0.0)).astype(np.float32)
    gas = np.broadcast_to(np.linspace(0.004, 0.001, nz, dtype=np.float32)[:, None, None], (nz, ny, nx)).copy()
    return dict(xe=xe, ye=ye, ze=ze, ext=[cloud, gas], ssa=[ssa1, np.full_like(gas, f32(0.6))],
                pf=[pf1, np.ones(gas.shape, np.int32)])
