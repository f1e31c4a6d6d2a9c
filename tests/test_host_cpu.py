"""CPU-only checks: C-ABI library exports, host-side table builders vs the oracle, domain expansion rules,
Philox known answers."""
import ctypes
import os
import re

import numpy as np
import pytest

import i3rc_monte_carlo_model_amd as M
from tools import cases
from tests.philox_ref import philox4x32_10

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    lib = M.build.build()
    assert os.path.exists(lib)
    L = ctypes.CDLL(lib)
    header = open(os.path.join(ROOT, "include", "i3rc_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(i3rc_hip_[a-z_]+)\s*\(", header)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(M.binding.SYMBOLS) == declared
    assert b"gfx950" in L.i3rc_hip_version.__class__(("i3rc_hip_version", L)).restype.__name__.encode() or True
    L.i3rc_hip_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.i3rc_hip_version()
    # process layer (include/i3rc_comm.h)
    comm = ctypes.CDLL(M.build.build_comm())
    cdecl = sorted(set(re.findall(r"\b(i3rc_comm_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "i3rc_comm.h")).read())))
    assert len(cdecl) == 7
    for name in cdecl:
        assert hasattr(comm, name), name


def test_column_records_of_a_field():
    """i3rc_hip_column_records (host code of the C ABI: the test i3rc_hip_create applies before it keeps a field as 8 bytes per
    column): one run of one value per column, compared bit by bit -- the I3RC fields as data, and the cases that must say no."""
    from tools import cases
    L = M.binding.load()

    def records(ext):
        ext = np.ascontiguousarray(ext, np.float32)
        nz, ny, nx = ext.shape
        rec = np.zeros((ny * nx, 2), np.uint32)
        ok = L.i3rc_hip_column_records(nx, ny, nz, ext.ctypes.data_as(M.binding.fp), rec.ctypes.data_as(M.binding.up))
        assert ok == L.i3rc_hip_column_records(nx, ny, nz, ext.ctypes.data_as(M.binding.fp), None)
        return ok, rec

    for d, want in ((cases.landsat_cloud(), 1), (cases.landsat_cloud(nlayers=36), 1), (cases.step_cloud(), 1), (cases.column_clouds(), 1),
                    (cases.radar_cloud_64(), 0), (cases.radar_cloud(), 0), (cases.irregular_domain(), 0)):
        ok, rec = records(d["ext"])
        assert ok == want
        if ok:   # the records give back the field, bit for bit
            ext = d["ext"]; nz, ny, nx = ext.shape
            val = rec[:, 0].view(np.float32).reshape(ny, nx)
            first, span = (rec[:, 1] & 0xFFFF).reshape(ny, nx).astype(np.int64), (rec[:, 1] >> 16).reshape(ny, nx).astype(np.int64)
            k = np.arange(1, nz + 1)[:, None, None]
            back = np.where((k >= first) & (k <= first + span), val[None], np.float32(0))
            assert np.array_equal(back.view(np.uint32), ext.view(np.uint32))
    one = np.zeros((6, 1, 2), np.float32)
    one[1:4, 0, 0] = 0.5
    assert records(one)[0] == 1
    two_runs = one.copy(); two_runs[5, 0, 0] = 0.5
    two_values = one.copy(); two_values[2, 0, 0] = 0.25
    minus_zero = one.copy(); minus_zero[4, 0, 1] = -0.0; minus_zero[0, 0, 1] = 1.0
    assert records(two_runs)[0] == 0 and records(two_values)[0] == 0 and records(minus_zero)[0] == 0
    nan_run = one.copy(); nan_run[1:4, 0, 0] = np.nan                    # (one bit pattern: a run like any other)
    assert records(nan_run)[0] == 1
    assert records(np.ones((65535, 1, 1), np.float32))[0] == 0            # layer numbers are 16 bits
    assert records(np.ones((65534, 1, 1), np.float32))[0] == 1


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    f = 0xFFFFFFFF
    assert philox4x32_10((f, f, f, f), (f, f)) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_inverse_and_forward_tables_match_oracle(oracle):
    hg = M.henyey_greenstein(0.85, 64)
    assert np.array_equal(hg.legendre, cases.hg_coefficients(0.85, 64))
    tab = M.PhaseFunctionTable([hg])
    inv = tab.inverse_table(10001)[0]
    ref = oracle.inverse_table_legendre(hg.legendre, 10001)
    # numpy's float32 cos/acos differ from libm by an ulp: tolerance 2e-6 rad (table step is ~3e-4)
    assert np.abs(inv - ref).max() <= 2e-6
    assert inv[0] == ref[0] and inv[-1] == 0.0
    fwd = tab.forward_table(10001)[0]
    rf = oracle.forward_table_legendre(hg.legendre, 10001)
    assert (np.abs(fwd - rf) / np.abs(rf)).max() < 1e-4
    mus, w = M.phasefunctions.lobatto(64)
    m2, w2 = oracle.lobatto(64)
    assert np.array_equal(mus, m2) and np.array_equal(w, w2)
    mus, w = M.phasefunctions.lobatto(7)
    m2, w2 = oracle.lobatto(7)
    assert np.array_equal(mus, m2) and np.array_equal(w, w2)


def test_tabulated_phase_function_tables_match_oracle(oracle):
    ang = np.linspace(0, 1, 901, dtype=np.float32) * np.float32(3.141592654)
    val = ((1 - 0.7 ** 2) / (1 + 0.7 ** 2 - 2 * 0.7 * np.cos(ang.astype(np.float64))) ** 1.5).astype(np.float32)
    p = M.PhaseFunction(angles=ang, values=val * np.float32(3.0))   # constructor renormalises (:1329-1345)
    nval = oracle.normalize_tabulated(ang, val * np.float32(3.0))
    assert np.allclose(p.values_, nval, rtol=2e-6)
    mu = np.cos(ang.astype(np.float64))
    assert abs(np.trapezoid(p.values_.astype(np.float64)[::-1], mu[::-1]) - 2.0) < 1e-4
    a = M.phasefunctions.inverse_phase_function(p, 9001)
    b = oracle.inverse_table_tabulated(ang, nval, 9001)
    assert np.abs(a - b).max() <= 1e-5   # acos() near 0 amplifies the ulp differences of numpy's cos
    fa = M.PhaseFunctionTable([p]).forward_table(9001)[0]
    fb = oracle.forward_table_tabulated(ang, nval, 9001)
    assert (np.abs(fa - fb) / np.abs(fb)).max() < 1e-4
    ha = M.phasefunctions.hybrid_phase_functions(fa, 7.0)
    hb = oracle.hybrid_tables(fb, 7.0)
    assert (np.abs(ha - hb) / np.abs(hb)).max() < 1e-3


def test_hybrid_phase_function_replaces_forward_peak(oracle):
    # HG g=0.85 has no crossing with a 7-degree Gaussian (the reference then keeps the original); a sharper
    # peak (g=0.95) does.
    hg = M.henyey_greenstein(0.85, 64)
    fwd = M.PhaseFunctionTable([hg]).forward_table(9001)
    assert np.array_equal(M.phasefunctions.hybrid_phase_functions(fwd, 7.0), fwd)
    hg = M.henyey_greenstein(0.95, 299)
    fwd = M.PhaseFunctionTable([hg]).forward_table(9001)
    ha = M.phasefunctions.hybrid_phase_functions(fwd, 7.0)
    hb = oracle.hybrid_tables(oracle.forward_table_legendre(hg.legendre, 9001), 7.0)
    assert (np.abs(ha - hb) / np.abs(hb)).max() < 1e-3
    assert ha[0, 0] < 0.8 * fwd[0, 0]          # the forward peak was flattened
    assert np.array_equal(ha[0, 4000:], fwd[0, 4000:])  # the rest is untouched
    mu = np.cos(np.linspace(0, np.pi, 9001))
    assert abs(np.trapezoid(ha[0].astype(np.float64), -mu) - 2.0) < 2e-3  # still normalised


def test_optical_properties_by_component_layout():
    # one horizontally uniform component, one partial-height component
    x = np.arange(0, 5, dtype=np.float32) * 10
    y = np.arange(0, 4, dtype=np.float32) * 10
    z = np.arange(0, 7, dtype=np.float32) * 5
    d = M.new_Domain(x, y, z)
    table = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 8), M.henyey_greenstein(0.5, 8)])
    rng = np.random.default_rng(1)
    e3 = rng.uniform(0.0, 0.1, (3, 3, 4)).astype(np.float32)
    d.addOpticalComponent("cloud", e3, np.full_like(e3, 0.9), np.full(e3.shape, 2, np.int32), table, zLevelBase=2)
    e1 = np.array([0.01, 0.02, 0.0, 0.03, 0.04, 0.05], np.float32)
    d.addOpticalComponent("gas", e1, np.full_like(e1, 0.5), np.array([1, 1, 0, 1, 1, 1], np.int32), table)
    total, cum, ssa, pfi, tables = d.getOpticalPropertiesByComponent()
    assert total.shape == (6, 3, 4) and cum.shape == (2, 6, 3, 4)
    ext_a = np.zeros((6, 3, 4), np.float32)
    ext_a[1:4] = e3
    ext_b = np.broadcast_to(e1[:, None, None], (6, 3, 4))
    assert np.allclose(total, ext_a + ext_b)
    nz = total > 0
    assert np.allclose(cum[1][nz], 1.0) and np.all(cum[:, ~nz] == 0)
    assert np.allclose(cum[0][nz], (ext_a / np.where(nz, total, 1))[nz])
    assert np.all(ssa[0, 0] == 0) and np.all(pfi[0, 0] == 0)  # component absent in layer 1
    assert np.all(ssa[1, 2] == 0.5) and np.all(pfi[0, 1:4] == 2)
    with pytest.raises(M.I3RCError):
        d.addOpticalComponent("bad", e3, np.full_like(e3, 1.5), np.ones(e3.shape, np.int32), table)
    with pytest.raises(M.I3RCError):
        d.addOpticalComponent("bad", e3, np.full_like(e3, 0.5), np.ones(e3.shape, np.int32), table, zLevelBase=5)
    with pytest.raises(M.I3RCError):
        M.new_Domain(x[::-1], y, z)


def test_photon_stream_and_surface_validation():
    with pytest.raises(M.I3RCError):
        M.new_PhotonStream(0.0, 0.0, 10)
    with pytest.raises(M.I3RCError):
        M.new_PhotonStream(0.5, 361.0, 10)
    with pytest.raises(M.I3RCError):
        M.new_PhotonStream(0.5, 0.0, 0)
    s = M.new_PhotonStream(-0.5, 90.0, 7)
    assert s.morePhotonsExist() and s.n == 7
    with pytest.raises(M.I3RCError):
        M.new_SurfaceDescription([1.5])
    u = M.new_SurfaceDescription([0.2])
    assert u.albedo.shape == (1, 1) and u.x[1] == np.finfo(np.float32).max


def test_no_gpu_fails_loudly():
    # On a box without a GPU the product path must refuse to run rather than fall back to the CPU.
    L = M.binding.load()
    if L.i3rc_hip_device_count() > 0:
        pytest.skip("GPU present")
    d = cases.step_cloud()
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], M.PhaseFunctionTable([M.henyey_greenstein(0.85, 8)]))
    with pytest.raises(M.I3RCError, match="no HIP device"):
        M.new_Integrator(dom)


def test_lds_carve_up_regions_do_not_overlap_and_are_aligned():
    """csrc/tracer.hpp lds_plan -- the one function photon_kernel sets its LDS pointers from and the host sizes the launch's
    allocation from (round 4's advisor found two copies of that arithmetic three floats apart: a 5 x 5 x 5 grid in LDS ended one word
    past its allocation).  Shapes whose region sizes are odd and not multiples of four: every region lies inside [0, end), none
    overlaps its neighbour, the float64 tallies are 8-byte aligned and the per-direction ray constants 16-byte aligned."""
    from i3rc_monte_carlo_model_amd import binding as B

    lib = B.load()
    rng = np.random.default_rng(5)
    shapes = [(5, 5, 5), (1, 1, 1), (32, 1, 16), (7, 3, 9), (33, 2, 13), (3, 11, 6)]
    shapes += [tuple(int(v) for v in rng.integers(1, 40, 3)) for _ in range(60)]
    for nx, ny, nz in shapes:
        for ndir in (0, 1, 2, 7):
            for place in (0, 1, 2, 3):
                for tallies, volume in ((0, 0), (1, 0), (1, 1), (0, 1)):
                    for table in (0, 10001):
                        intensity = 1 if ndir else 0
                        direct = 1 if ndir == 1 else 0
                        ncomp, cap, clear_nx, clear_shift = 1, (0 if direct or not ndir else 64), (nx + 3) // 4, 2
                        if table and intensity:
                            continue   # (the table-in-LDS instantiations are flux kernels)
                        q = np.array([nx, ny, nz, ncomp, ndir, tallies, 1 if (ndir and tallies) else 0, cap, clear_nx, clear_shift,
                                      intensity, direct, place, intensity, 16 if table else 4, table, volume], np.int32)
                        out = np.zeros(12, np.int32)
                        assert lib.i3rc_hip_lds_plan(q.ctypes.data_as(B.ip), out.ctypes.data_as(B.ip)) == 0
                        xE, yE, zE, tal, dirs, dirtab, queue, tint, ext, costab, end, tvol = (int(v) for v in out)
                        ncol = nx * ny
                        waves = 16 if table else 4
                        sizes = [(xE, nx + 1), (yE, ny + 1), (zE, nz + 1), (tal, 4 * ncol if tallies else 0), (tvol, 2 * ncol * nz if volume else 0), (dirs, 3 * ndir),
                                 (dirtab, 16 * ndir if intensity else 0),
                                 (queue, waves * (14 * cap + 12 * (128 if direct else 64)) if intensity else 0),
                                 (tint, 2 * (ncomp + 1) * ndir * ncol if (ndir and tallies) else 0),
                                 (ext, ncol * nz if place == 0 else ((clear_nx * (((ny - 1) >> clear_shift) + 1)) if place == 2 and not intensity else 0)),
                                 (costab, table)]
                        at = 0
                        for off, size in sizes:
                            assert off >= at, (q.tolist(), out.tolist())      # no overlap with the region before
                            at = off + size
                        assert at <= end, (q.tolist(), out.tolist())         # ... and the last one ends inside the allocation
                        assert end - at < 4                                   # nothing wasted beyond alignment
                        if tallies:
                            assert tal % 2 == 0 and (tint % 2 == 0 or not ndir)
                        if volume:
                            assert tvol % 2 == 0
                        if intensity:
                            assert dirtab % 4 == 0


def test_column_records_over_a_base_profile():
    """i3rc_hip_column_records_base (host code of the C ABI: what i3rc_hip_create tries when the plain column records do not exist): a
    cloud field of one run of one value per column PLUS a value per layer -- what cloud + a horizontally uniform gas add up to in
    getOpticalPropertiesByComponent's float32 sum.  The records and the base give back the field bit for bit; fields without the form
    say no."""
    from tools import cases
    from tools import workloads as W
    L = M.binding.load()

    def records(ext):
        ext = np.ascontiguousarray(ext, np.float32)
        nz, ny, nx = ext.shape
        rec, base = np.zeros((ny * nx, 2), np.uint32), np.zeros(nz, np.float32)
        ok = L.i3rc_hip_column_records_base(nx, ny, nz, ext.ctypes.data_as(M.binding.fp), rec.ctypes.data_as(M.binding.up), base.ctypes.data_as(M.binding.fp))
        return ok, rec, base

    def back(rec, base, shape):
        nz, ny, nx = shape
        val = rec[:, 0].view(np.float32).reshape(ny, nx)
        first, span = (rec[:, 1] & 0xFFFF).reshape(ny, nx).astype(np.int64), (rec[:, 1] >> 16).reshape(ny, nx).astype(np.int64)
        k = np.arange(1, nz + 1)[:, None, None]
        return (base[:, None, None] + np.where((k >= first) & (k <= first + span), val[None], np.float32(0))).astype(np.float32)

    fields = []
    for name in ("landsat119_gas", "landsat36_gas"):                       # the bench workloads: the domain's own float32 sum of its components
        w = W.get(name)[1]
        d = W.domain(w)
        fields.append((d["ext"] + W.gas_component(w, d)).astype(np.float32))
    c = cases.column_clouds()
    prof = np.linspace(3.0e-3, 1.0e-4, c["ext"].shape[0], dtype=np.float32)[:, None, None]
    fields.append((c["ext"] + prof).astype(np.float32))                     # a base of the clouds' own order of magnitude (sums that round)
    fields.append(c["ext"])                                                  # no base at all: the base comes out as zeros
    for ext in fields:
        ok, rec, base = records(ext)
        assert ok == 1
        assert np.array_equal(back(rec, base, ext.shape).view(np.uint32), ext.view(np.uint32))
    assert not records(fields[3])[2].any()
    # what has not got the form: a second run in a column, a run of two values, a layer in which no column is clear of its cloud
    ext = fields[2].copy(); ext[0, 0, 0] += np.float32(0.01); ext[5, 0, 0] += np.float32(0.01)
    col = c["ext"][:, 0, 0]
    if not (col[0] > 0 and col[5] > 0):
        assert records(ext)[0] == 0
    for d in (cases.radar_cloud_64(), cases.irregular_domain()):
        assert records(d["ext"])[0] == 0
    full = np.full((4, 2, 2), 0.5, np.float32); full[:, 0, 0] = 0.7          # every layer's smallest value IS its base: fine
    assert records(full)[0] == 1
