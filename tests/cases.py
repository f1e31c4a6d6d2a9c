"""Synthetic domains used by tests, smoke and bench: the I3RC phase-1 recipes restated from
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


def irregular_domain(seed=3, nx=7, ny=5, nz=9, ssa=0.95):
    """Small irregularly spaced domain with empty cells: exercises findIndex paths and zero-extinction steps."""
    rng = np.random.default_rng(seed)
    xe = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, nx))]).astype(np.float32)
    ye = np.concatenate([[0.0], np.cumsum(rng.uniform(5, 40, ny))]).astype(np.float32)
    ze = (np.concatenate([[0.0], np.cumsum(rng.uniform(5, 30, nz))]) + 100.0).astype(np.float32)
    ext = rng.uniform(0.0, 0.08, (nz, ny, nx)).astype(np.float32)
    ext[rng.random(ext.shape) < 0.3] = 0.0
    pf = np.where(ext > 0, 1, 0).astype(np.int32)
    s = np.where(ext > 0, f32(ssa), f32(0.0)).astype(np.float32)
    return dict(xe=xe, ye=ye, ze=ze, ext=ext, ssa=s, pf=pf)
