"""The Fortran-95 shell (i3rc-monte-carlo-model_amd/fortran): same module API as the reference, over the C ABI.

CPU: the shell builds with amdflang; its host-side numerics reproduce the pinned reference numbers; the minimal
netCDF classic module round-trips a domain and is readable by an independent reader (scipy); where the reference
tree is present, its own drivers compile and link UNCHANGED against the shell (drop-in boundary).
GPU: the shell's integrator and the reference's unchanged drivers run on the device."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "i3rc-monte-carlo-model_amd", "fortran")
BUILD = os.path.join(FDIR, "build")
HAVE_FLANG = os.path.exists("/opt/rocm/bin/amdflang")
HAVE_REF = os.path.isdir("/root/reference/Example-Drivers")


def _run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)


@pytest.fixture(scope="module")
def shell_built():
    if not HAVE_FLANG:
        pytest.skip("amdflang not available")
    import i3rc_monte_carlo_model_amd as M

    M.build.build()  # the shell links against csrc/libi3rc_hip.so
    r = _run(["make", "-C", FDIR, "-s", "all"])
    assert r.returncode == 0, r.stdout + r.stderr
    return BUILD


def _need(exe):
    """A prebuilt piece of the shell the GPU tests run.  The GPU box has the same image (amdflang is there): a tree
    without fortran/build is a tree whose build step was skipped, and that is a failure, not a reason to skip."""
    if not os.path.exists(exe):
        if HAVE_FLANG:
            import i3rc_monte_carlo_model_amd as M

            M.build.build()
            r = _run(["make", "-C", FDIR, "-s", "all"])
            assert r.returncode == 0 and os.path.exists(exe), f"{exe} missing and `make -C fortran all` failed:\n" + r.stdout + r.stderr
        else:
            pytest.skip("no Fortran compiler on this machine and no prebuilt shell")
    return exe


def _fields(out, key):
    for line in out.splitlines():
        if line.startswith(key):
            return line[len(key):].split()
    raise AssertionError(f"{key} not in output:\n{out}")


def test_shell_host_numerics_match_reference_pins(shell_built):
    r = _run([os.path.join(shell_built, "shellSelfTest"), "cpu"])
    assert r.returncode == 0 and "cpu checks done" in r.stdout, r.stdout + r.stderr
    # MT19937 known answers
    assert _fields(r.stdout, "mt_scalar") == ["0.543404937", "0.671155632", "0.278369397", "0.412046403", "0.424517602"]
    assert _fields(r.stdout, "mt_vector") == ["168774779", "197189863", "1009846582", "-287080014", "-590412790"]
    # inverse table HG g=.85, 64 moments, 10001 steps; forward P(0), P(pi)
    assert _fields(r.stdout, "inverse") == ["3.141593", "3.045351", "0.497639", "0.252000", "0.134744", "0.002211", "0.000000"]
    fwd = [float(v) for v in _fields(r.stdout, "forward")]
    assert f"{fwd[0]:.4f}" == "82.1977" and f"{fwd[2]:.7f}" == "0.0456438"
    # domain: 3-D + horizontally uniform partial-height component, written and read back through netCDF classic
    assert _fields(r.stdout, "domain") == ["3", "2", "4", "2", "cloud", "gas"]
    rt = _fields(r.stdout, "roundtrip")
    assert float(rt[0]) == 0.0 and float(rt[1]) == 0.0 and rt[2] == "F"
    lay = [float(v) for v in _fields(r.stdout, "layers")]
    assert lay == [0.01, 0.021, pytest.approx(0.02 / 0.021, abs=1e-6)]


def test_netcdf_classic_files_are_readable_by_an_independent_reader(shell_built, tmp_path):
    from scipy.io import netcdf_file

    dom = str(tmp_path / "step.dom")
    r = _run([os.path.join(shell_built, "makeStepCloudDomain"), dom, "16", "0.99"])
    assert r.returncode == 0, r.stdout + r.stderr
    f = netcdf_file(dom, "r", mmap=False)
    assert f.dimensions["x-Edges"] == 33 and f.dimensions["z-Grid"] == 16 and f.numberOfComponents == 1
    assert f.Component1_Name == b"cloud" and f.Component1_zLevelBase == 1
    ext = f.variables["Component1_Extinction"].data
    assert ext.shape == (16, 1, 32)   # file order is slowest first: (z, y, x)
    assert np.allclose(ext[:, 0, :16], 2 / 250) and np.allclose(ext[:, 0, 16:], 18 / 250)
    assert np.allclose(f.variables["Component1_SingleScatteringAlbedo"].data, 0.99)
    assert f.variables["Component1_PhaseFunctionIndex"].data.dtype.kind == "i"
    assert f.Component1_phaseFunctionStorageType == b"LegendreCoefficients"
    coef = f.variables["Component1_legendreCoefficients"].data
    assert coef.shape == (64,) and abs(coef[0] - 0.85) < 1e-7 and abs(coef[1] - 0.7225) < 1e-6
    assert np.allclose(f.variables["x-Edges"].data, np.arange(33) * 15.625)


def _write_i3rc_data_files(directory):
    """The I3RC phase-1 input data (tools/data/i3rc_phase1_inputs.npz) as text files in the case definition's own
    formats, for the Fortran case generators."""
    inp = np.load(os.path.join(ROOT, "tools", "data", "i3rc_phase1_inputs.npz"))
    with open(os.path.join(directory, "mmcr_tau_32km_020898"), "w") as f:
        for row in inp["mmcr_tau"]:
            f.write("".join("%8.3f" % v for v in row) + "\n")
    for name, key, fmt in (("scene43.tau.128x128", "landsat_tau", "%7.2f"), ("scene43.dz.128x128", "landsat_dz_km", "%7.3f")):
        with open(os.path.join(directory, name), "w") as f:
            for row in inp[key]:
                f.write("".join(fmt % v for v in row) + "\n")
    with open(os.path.join(directory, "C.1_PF"), "w") as f:
        for a, v in zip(inp["c1_angle_deg"], inp["c1_value"]):
            f.write("%7.1f   %.7E\n" % (a, v))
    with open(os.path.join(directory, "C.1_leg_coef"), "w") as f:
        for v in inp["c1_legendre"]:
            f.write("   %.7g\n" % v)
    return inp


def test_i3rc_case_generators_write_the_case_definitions(shell_built, tmp_path):
    # Fortran tools: radar and Landsat domains from the I3RC data files, compared with the
    # numpy recipes the GPU parity tests use (tools/cases.py) -- two independent statements of the same recipe
    from scipy.io import netcdf_file
    from tools import cases

    inp = _write_i3rc_data_files(str(tmp_path))
    dom = str(tmp_path / "radar.dom")
    r = _run([os.path.join(shell_built, "makeRadarCloudDomain"), str(tmp_path), dom, "0.99", "hg"])
    assert r.returncode == 0 and "mean column optical depth" in r.stdout, r.stdout + r.stderr
    assert os.path.exists(dom), (r.stdout, os.listdir(str(tmp_path)))
    f = netcdf_file(dom, "r", mmap=False)
    want = cases.radar_cloud(ssa=0.99)
    assert np.array_equal(f.variables["Component1_Extinction"].data, want["ext"])
    assert np.array_equal(f.variables["x-Edges"].data, want["xe"]) and np.array_equal(f.variables["z-Edges"].data, want["ze"])
    assert np.allclose(f.variables["Component1_SingleScatteringAlbedo"].data, 0.99)
    coef = f.variables["Component1_legendreCoefficients"].data
    assert coef.shape == (299,) and abs(coef[0] - 0.85) < 1e-7
    tau_mean = float(r.stdout.split("optical depth")[1].split(",")[0])
    assert abs(tau_mean - inp["mmcr_tau"].sum() / 640) < 2e-3          # mean column optical depth 19.5
    # the tabulated C1 phase function travels as angle / value pairs, the expanded one as Legendre coefficients
    r = _run([os.path.join(shell_built, "makeRadarCloudDomain"), str(tmp_path), dom, "1.0", "c1"])
    assert r.returncode == 0, r.stdout + r.stderr
    f = netcdf_file(dom, "r", mmap=False)
    assert f.Component1_phaseFunctionStorageType == b"Angle-Value"
    ang, val = cases.c1_phase_function()
    assert np.allclose(f.variables["Component1_scatteringAngle"].data, ang, rtol=1e-6)
    stored = f.variables["Component1_phaseFunctionValues"].data.ravel()
    assert stored.shape == (1801,) and np.allclose(stored / stored[0], val / val[0], rtol=2e-6)   # normalised on construction
    r = _run([os.path.join(shell_built, "makeRadarCloudDomain"), str(tmp_path), dom, "1.0", "c1legendre"])
    assert r.returncode == 0, r.stdout + r.stderr
    f = netcdf_file(dom, "r", mmap=False)
    coef = f.variables["Component1_legendreCoefficients"].data
    assert coef.shape == (299,) and np.allclose(coef[:3], inp["c1_legendre"][1:4] / np.array([3, 5, 7]), rtol=1e-6)
    # Landsat scene: the reference's 119 layers and the labelled 36-layer re-binning
    for nl in (119, 36):
        dom = str(tmp_path / f"landsat{nl}.dom")
        r = _run([os.path.join(shell_built, "makeLandsatCloudDomain"), str(tmp_path), dom, "1.0", str(nl)])
        assert r.returncode == 0, r.stdout + r.stderr
        f = netcdf_file(dom, "r", mmap=False)
        want = cases.landsat_cloud(nlayers=nl)
        ext = f.variables["Component1_Extinction"].data
        assert ext.shape == (nl, 128, 128)
        assert np.array_equal(ext > 0, want["ext"] > 0) and np.allclose(ext, want["ext"], rtol=3e-7, atol=0)
        assert np.allclose(f.variables["z-Edges"].data, want["ze"], rtol=1e-6)
        pfi = f.variables["Component1_PhaseFunctionIndex"].data
        assert np.array_equal(pfi == 1, ext > 0) and np.array_equal(pfi == 0, ext == 0)
        tau_mean = float(r.stdout.split("optical depth")[1].split(",")[0])
        assert abs(tau_mean - inp["landsat_tau"].mean()) < 2e-3       # mean column optical depth 10.06


def test_optical_properties_to_domain_importer(shell_built, tmp_path):
    # SHDOM-like ASCII property file -> domain file (Tools/OpticalPropertiesToDomain.readme:27-65)
    from scipy.io import netcdf_file

    g1, g2 = 0.85, 0.6
    chi1 = [(2 * l + 1) * g1**l for l in range(1, 9)]
    chi2 = [(2 * l + 1) * g2**l for l in range(1, 5)]
    prp = tmp_path / "field.prp"
    lines = ["Tabulated phase function property file", "2 1 3", "100. 250.  0. 50. 150. 400.", "2",
             "8 " + " ".join("%.7f" % c for c in chi1[:5]), " ".join("%.7f" % c for c in chi1[5:]),   # entry continues on the next line
             "4 " + " ".join("%.7f" % c for c in chi2)]
    cells = [(1, 1, 1, 280.0, 0.01, 1.0, 1), (2, 1, 1, 280.0, 0.02, 0.9, 2), (2, 1, 3, 270.0, 0.005, 1.0, 1)]
    lines += ["%d %d %d %.1f %.5f %.3f %d" % c for c in cells]
    prp.write_text("\n".join(lines) + "\n")
    out = tmp_path / "field.dom"
    nml = tmp_path / "convert.nml"
    nml.write_text(f"&fileNames\n  PropFileName = '{prp}',\n  outputFileName = '{out}'\n/\n")
    r = _run([os.path.join(shell_built, "opticalPropertiesToDomain"), str(nml)])
    assert r.returncode == 0 and "3 cells listed, 2 phase functions" in r.stdout, r.stdout + r.stderr
    f = netcdf_file(str(out), "r", mmap=False)
    ext = f.variables["Component1_Extinction"].data            # (z, y, x)
    assert ext.shape == (3, 1, 2)
    assert np.allclose(ext[0, 0], [0.01, 0.02]) and ext[1].max() == 0 and np.allclose(ext[2, 0], [0, 0.005])
    assert np.allclose(f.variables["Component1_SingleScatteringAlbedo"].data[0, 0], [1.0, 0.9])
    assert list(f.variables["Component1_PhaseFunctionIndex"].data[0, 0]) == [1, 2]
    assert np.allclose(f.variables["z-Edges"].data, [0, 50, 150, 400]) and np.allclose(f.variables["x-Edges"].data, [0, 100, 200])
    start, length = f.variables["Component1_start"].data, f.variables["Component1_length"].data
    coef = f.variables["Component1_legendreCoefficients"].data
    assert list(length) == [8, 4] and list(start) == [1, 9]
    assert np.allclose(coef[:8], [g1**l for l in range(1, 9)], rtol=1e-5) and np.allclose(coef[8:], [g2**l for l in range(1, 5)], rtol=1e-5)
    # a bad index is refused
    prp.write_text("\n".join(lines[:7] + ["3 1 1 280. 0.01 1.0 1"]) + "\n")
    assert _run([os.path.join(shell_built, "opticalPropertiesToDomain"), str(nml)]).returncode != 0


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present (GPU box)")
def test_reference_drivers_link_unchanged(shell_built):
    # drop-in boundary: the reference's own driver sources, compiled in place, link against the shell
    r = _run(["make", "-C", FDIR, "linkcheck"])
    assert r.returncode == 0 and "compiled and linked unchanged" in r.stdout, r.stdout + r.stderr
    for exe in ("monteCarloDriver_ref", "planeParallel_ref"):
        assert os.access(os.path.join(shell_built, exe), os.X_OK)
    # nothing of the reference's source is kept in the repository
    tracked = _run(["git", "-C", ROOT, "ls-files"]).stdout.split()
    assert not any(os.path.basename(t) in ("monteCarloDriver.f95", "planeParallel.f95") for t in tracked)


def _domain_file_arrays(path):
    """every variable and global attribute of a domain file (scipy's netCDF reader: independent of the shell's)"""
    from scipy.io import netcdf_file

    f = netcdf_file(path, "r", mmap=False)
    return {k: np.array(v.data) for k, v in f.variables.items()}, {k: getattr(f, k) for k in f._attributes}


def _same_variables(a, b, what):
    va, vb = _domain_file_arrays(a)[0], _domain_file_arrays(b)[0]
    assert sorted(va) == sorted(vb), (what, sorted(set(va) ^ set(vb)))
    for k in va:
        assert va[k].dtype == vb[k].dtype and va[k].shape == vb[k].shape, (what, k, va[k].shape, vb[k].shape)
        assert np.array_equal(va[k].view(np.uint8), vb[k].view(np.uint8)), (what, k, int((va[k] != vb[k]).sum()))
    return va


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present (GPU box)")
def test_reference_case_generators_and_tools_run_unchanged_on_the_shell(shell_built, tmp_path):
    """SURVEY.md 8f row 4, pinned by the reference's own programs: I3RC-Examples/i3rcStepCloud.f95, i3rcLandsatCloud.f95 and
    Tools/OpticalPropertiesToDomain.f95, compiled unchanged and in place against the shell (`make linkcheck`), write domain files whose
    every variable -- edges, extinction, single-scattering albedo, table index, Legendre coefficients, table keys -- equals BIT FOR
    BIT what the shell's own tools (fortran/tools/*.f95, written from the case definitions and the importer's readme) write; the
    files differ in two text attributes (the component's name and description).  The reference's inputs are read where they lie or
    from the fixture's restatement of them; every output goes to the test's directory."""
    r = _run(["make", "-C", FDIR, "linkcheck"])
    assert r.returncode == 0 and "case generators and Tools compiled and linked unchanged" in r.stdout, r.stdout + r.stderr
    data = tmp_path / "Data"          # (the generators read and write ./Data/: a directory of the test's own)
    data.mkdir()
    _write_i3rc_data_files(str(data))
    for exe in ("i3rcStepCloud_ref", "i3rcLandsatCloud_ref"):
        r = _run([os.path.join(shell_built, exe)], cwd=str(tmp_path))
        assert r.returncode == 0, exe + r.stdout + r.stderr
    for ref_name, tool, args in (("StepCloud_NonAbsorbing.opt", "makeStepCloudDomain", ["32", "1.0"]),
                                 ("StepCloud_Absorbing.opt", "makeStepCloudDomain", ["32", "0.99"]),
                                 ("LandsatCloud_NonAbsorbing.opt", "makeLandsatCloudDomain", [str(data), "1.0", "119"]),
                                 ("LandsatCloud_Absorbing.opt", "makeLandsatCloudDomain", [str(data), "0.99", "119"])):
        mine = str(tmp_path / ("shell_" + ref_name))
        cmd = [os.path.join(shell_built, tool)] + ([mine] + args if tool == "makeStepCloudDomain" else [args[0], mine] + args[1:])
        r = _run(cmd)
        assert r.returncode == 0, r.stdout + r.stderr
        v = _same_variables(str(data / ref_name), mine, ref_name)
        assert v["Component1_Extinction"].shape == ((32, 1, 32) if "Step" in ref_name else (119, 128, 128))
    # the importer, on the reference's own example: an LES stratocumulus field of 73 728 cells, 27 phase functions
    prp = "/root/reference/Tools/Examples/les_stcu_w213.prp"
    for who, exe in (("ref", "OpticalPropertiesToDomain_ref"), ("shell", "opticalPropertiesToDomain")):
        nml = tmp_path / f"{who}.nml"
        nml.write_text(f"&fileNames\n PropFileName = '{prp}',\n outputFileName = '{tmp_path}/{who}_les.dom'\n/\n")
        r = _run([os.path.join(shell_built, exe), str(nml)], cwd=str(tmp_path))
        assert r.returncode == 0, who + r.stdout + r.stderr
    v = _same_variables(str(tmp_path / "ref_les.dom"), str(tmp_path / "shell_les.dom"), "les_stcu_w213")
    assert v["Component1_Extinction"].shape == (18, 64, 64) and v["Component1_length"].shape == (27,)


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present (GPU box)")
def test_reference_tool_chain_domains_are_the_committed_fixtures(shell_built):
    """tests/golden/tools_*.dom.gz -- MakeMieTable -> PhysicalPropertiesToDomain, the reference's programs unchanged on the shell, on
    the reference's example inputs (tests/golden/make_tool_domains.py) -- made again here: the same bytes."""
    import gzip
    import importlib.util
    import tempfile

    r = _run(["make", "-C", FDIR, "linkcheck"])
    assert r.returncode == 0, r.stdout + r.stderr
    spec = importlib.util.spec_from_file_location("make_tool_domains", os.path.join(ROOT, "tests", "golden", "make_tool_domains.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with tempfile.TemporaryDirectory() as tmp:
        files = mod.make(tmp)
    for name, data in files.items():
        assert gzip.decompress(open(os.path.join(ROOT, "tests", "golden", mod.OUT[name]), "rb").read()) == data, name


def _tool_chain_domain(name, directory):
    """one of tests/golden/tools_*.dom.gz unpacked into `directory`"""
    import gzip

    path = os.path.join(str(directory), name + ".dom")
    with open(path, "wb") as f:
        f.write(gzip.decompress(open(os.path.join(ROOT, "tests", "golden", f"tools_{name}.dom.gz"), "rb").read()))
    return path


def test_python_mirror_reads_domain_files(shell_built, tmp_path):
    """read_Domain of the Python mirror (host.py; Code/opticalProperties.f95:708-871): a file the shell's generator wrote gives the
    arrays of the numpy statement of the case; the tool-chain fixtures give their components -- three-dimensional and horizontally
    uniform ones, level bases, tables of 35 entries with up to 1381 Legendre coefficients."""
    import i3rc_monte_carlo_model_amd as M
    from tools import cases

    dom = str(tmp_path / "step.dom")
    assert _run([os.path.join(shell_built, "makeStepCloudDomain"), dom, "16", "0.99"]).returncode == 0
    d, want = M.read_Domain(dom), cases.step_cloud(ssa=0.99, nlayers=16)
    c = d.components[0]
    assert d.shape == (16, 1, 32) and len(d.components) == 1 and c["zbase"] == 1 and not c["uniform"]
    assert np.array_equal(d.x, want["xe"]) and np.array_equal(d.z, want["ze"]) and np.array_equal(c["ext"], want["ext"])
    assert np.array_equal(c["ssa"], want["ssa"]) and np.array_equal(c["pfi"], want["pf"])
    assert c["table"].n_entries == 1 and np.array_equal(c["table"].entries[0].legendre, M.henyey_greenstein(0.85, 64).legendre)
    les = M.read_Domain(_tool_chain_domain("les_stcu_rayleigh", tmp_path))
    assert les.shape == (18, 64, 64) and [c["name"] for c in les.components] == ["Particle type 1", "Rayleigh scattering"]
    cloud, gas = les.components
    assert cloud["ext"].shape == (16, 64, 64) and cloud["zbase"] == 2 and cloud["table"].n_entries == 35
    assert max(e.legendre.size for e in cloud["table"].entries) == 1381
    assert gas["uniform"] and gas["ext"].shape == (18, 1, 1) and gas["zbase"] == 1 and np.all(gas["ssa"] == 1)
    assert np.allclose(gas["table"].entries[0].legendre, [0.0, 0.1])                      # Rayleigh: 1 + P2 / 2
    total, cum, ssa, pfi, tables = les.getOpticalPropertiesByComponent()
    assert total.shape == (18, 64, 64) and np.all(total > 0) and np.all(cum[1] == 1) and len(tables) == 2
    mix = M.read_Domain(_tool_chain_domain("mixture", tmp_path))
    assert mix.shape == (11, 1, 1) and [c["ext"].shape[0] for c in mix.components] == [10, 10, 11]
    assert np.all(mix.components[2]["ssa"] == 0)                                           # molecular absorption
    with pytest.raises(M.I3RCError):
        M.read_Domain(str(tmp_path / "no such file"))


def _spawn_ranks(cmd, world, port, extra_env=None, cwd=None):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), I3RC_COMM_BACKEND="shm")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_multiple_processes_module_two_and_three_ranks(shell_built):
    # N > 1 path of module MultipleProcesses (the stand-in for multipleProcesses_mpi.f95) on CPU: shared-memory backend
    exe = os.path.join(shell_built, "commSelfTest")
    for world, port in ((2, 29631), (3, 29632)):
        rcs, outs = _spawn_ranks([exe], world, port)
        assert rcs == [0] * world, outs
        for r, o in enumerate(outs):
            assert f"rank {r} of {world} sums ok master={'T' if r == 0 else 'F'}" in o, o
    r = _run([exe])   # a single process needs no launcher
    assert r.returncode == 0 and "rank 0 of 1 sums ok master=T" in r.stdout


# ---- GPU --------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_shell_integrator_on_gpu():
    exe = _need(os.path.join(BUILD, "shellSelfTest"))
    r = _run([exe, "gpu"], cwd=ROOT)
    assert r.returncode == 0 and "gpu checks done" in r.stdout, r.stdout + r.stderr
    up, down, more = _fields(r.stdout, "slab")
    # planeParallel.nml values: reference 0.16420 / 0.83580 (4e4 photons, +-0.0036); here 8e5 photons
    assert abs(float(up) - 0.1642) < 0.004 and abs(float(down) - 0.8358) < 0.004 and more == "F"
    assert abs(float(up) + float(down) - 1.0) < 1e-4
    s = _fields(r.stdout, "surface")
    assert float(s[0]) > float(up) and 0 < float(s[2]) < 1 and 0 < float(s[3]) < 1 and s[4] == "F"
    # photon sources with explicit start positions / directions (RandomAzimuth, Spotlight, Flux, internal detector)
    src = [float(v) for v in _fields(r.stdout, "sources")]
    ra, spot, flux, internal = src[0:2], src[2:4], src[4:6], src[6:8]
    for f_up, f_dn in (ra, spot):      # a horizontally uniform slab: same fluxes as the Directional stream (4e5 photons)
        assert abs(f_up - float(up)) < 0.004 and abs(f_dn - float(down)) < 0.004
    for f_up, f_dn in (ra, spot, flux, internal):
        assert abs(f_up + f_dn - 1.0) < 2e-4
    assert float(up) * 0.5 < flux[0] < 0.5          # isotropic-flux illumination: between overhead and grazing sun
    assert 0.5 < internal[0] < 1.0                   # upward-looking detector in mid-slab: most photons leave through the top
    # the batch loop through the shell's own entry points: streamed (announced, taken over batch by batch) and with its moments
    # gathered on the device -- the same twelve batches, so the same sums; reportResults before a batch is selected and
    # selectBatchResults after a change of the tally layout in mid-loop are refused
    st = _fields(r.stdout, "streamed")
    assert st[0] == "T" and st[-1] == "T", st
    s1, m1, s2, m2, mean1, mean2 = (float(v) for v in st[1:7])
    assert abs(s1 - m1) < 2e-6 * s1 and abs(s2 - m2) < 4e-6 * s2 and abs(mean1 - m1) < 2e-6 * m1 and abs(mean2 - m2) < 4e-6 * m2
    assert 12 * 0.2 < s1 < 12 * 0.5                  # (a slab of optical depth 1 over a surface of albedo 0.2, sun at 60 degrees)


@pytest.mark.gpu
def test_multiple_processes_module_over_rccl_one_rank():
    # the RCCL backend of module MultipleProcesses on the real thing, as far as one GPU goes: a launcher environment of
    # one rank -> TCP rendezvous with itself, ncclCommInitRank, every sumAcrossProcesses one ncclAllReduce on the GPU
    exe = _need(os.path.join(BUILD, "commSelfTest"))
    rcs, outs = _spawn_ranks([exe], 1, 29641, extra_env=dict(I3RC_COMM_BACKEND="rccl"))
    assert rcs == [0] and "rank 0 of 1 sums ok master=T" in outs[0], outs


def same_to_the_printed_digits(a, b, what):
    """two result files of the drivers, line by line: equal to the digits their formats print (see the caller's note)"""
    assert len(a) == len(b), what
    values = off = 0
    for x, y in zip(a, b):
        if x == y or "Property_File" in x:
            continue
        fx, fy = x.split(), y.split()
        assert len(fx) == len(fy), (what, x, y)
        assert not x.lstrip().startswith("!") or "Average" in x, (what, x, y)   # (header lines agree as text)
        for u, v in zip(fx, fy):
            if u != v:
                assert abs(float(u) - float(v)) <= 1.0001e-4, (what, x, y)
                off += 1
    values = sum(len(x.split()) for x in a)
    assert off <= max(1, values // 100), (what, off, values)


@pytest.mark.gpu
def test_reference_drivers_run_unchanged_on_gpu(tmp_path):
    pp = os.path.join(BUILD, "planeParallel_ref")
    mc = os.path.join(BUILD, "monteCarloDriver_ref")
    mk = os.path.join(BUILD, "makeStepCloudDomain")
    if not (os.path.exists(pp) and os.path.exists(mc) and os.path.exists(mk)):
        pytest.skip("reference drivers were not built in this tree (needs /root/reference at build time)")
    r = _run([pp, os.path.join(FDIR, "examples", "planeParallel.nml")], cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    nums = [float(v) for v in r.stdout.strip().splitlines()[-1].split()]
    # columns: tau omega g theta0 Fup Fdn errUp errDn Fabs errAbs  (reference result 0.16420 / 0.83580 +- 0.0036)
    assert nums[:4] == [1.0, 1.0, 0.85, 60.0]
    assert abs(nums[4] - 0.1642) < 4 * 0.0036 and abs(nums[5] - 0.8358) < 4 * 0.0036 and abs(nums[4] + nums[5] - 1) < 1e-3
    # monteCarloDriver: domain file -> read_Domain -> 10 batches -> ASCII + netCDF output
    out = tmp_path
    dom = str(out / "stepCloud.dom")
    assert _run([mk, dom, "32", "1.0"]).returncode == 0
    nml = open(os.path.join(FDIR, "examples", "stepCloud.nml")).read().replace("gpurun_out/fortran/", str(out) + "/")
    nml = nml.replace('outputAbsVolumeFile = ""', f'outputAbsVolumeFile = "{out}/stepCloud_absvol.txt"')
    nml = nml.replace("reportVolumeAbsorption = .false.", "reportVolumeAbsorption = .true.")
    nml_path = str(out / "stepCloud.nml")
    open(nml_path, "w").write(nml)
    r = _run([mc, nml_path], cwd=ROOT)
    assert r.returncode == 0 and "Wrote ASCII results" in r.stdout and "Wrote netcdf results" in r.stdout, r.stdout + r.stderr
    flux = open(str(out / "stepCloud_flux.txt")).read()
    m = re.search(r"Average:\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", flux)
    fup, eup, fdn, edn = map(float, m.groups())
    # Fup 0.3254, Fdn 0.6746 at 1e6 photons
    assert abs(fup - 0.3254) < 4 * max(eup, 4e-4) and abs(fdn - 0.6746) < 4 * max(edn, 4e-4)
    # the shell's own driver (same namelists, same ASCII formats) on the same run deck
    own = os.path.join(BUILD, "i3rcDriver")
    if os.path.exists(own):
        nml2 = nml.replace("stepCloud_flux.txt", "own_flux.txt").replace("stepCloud_rad.txt", "own_rad.txt")
        nml2 = nml2.replace("stepCloud_absprof.txt", "own_absprof.txt").replace("stepCloud_results.nc", "own_results.nc")
        nml2 = nml2.replace("stepCloud_absvol.txt", "own_absvol.txt")
        open(str(out / "own.nml"), "w").write(nml2)
        r2 = _run([own, str(out / "own.nml")], cwd=ROOT)
        assert r2.returncode == 0 and "Wrote ASCII results" in r2.stdout and "Wrote netCDF results" in r2.stdout, r2.stdout + r2.stderr
        own_flux = open(str(out / "own_flux.txt")).read()
        # Same seeds, same kernels, the same statistics: the files agree to the digits their formats print.  The shell's driver
        # gathers its batch moments on the device in float64 (computeRadiativeTransferBatchMoments), the reference's driver adds
        # real(4) values one after the other: where a value lies within 1e-7 of a rounding boundary of F9.4 the last printed digit
        # may differ by one -- allowed for at most one value in a hundred, by one unit.
        same_to_the_printed_digits(own_flux.splitlines()[9:], flux.splitlines()[9:], "flux")
        for name in ("rad", "absprof", "absvol"):   # whole files, headers included (Property_File differs by name only)
            a = open(str(out / f"own_{name}.txt")).read().splitlines()
            b = open(str(out / f"stepCloud_{name}.txt")).read().splitlines()
            same_to_the_printed_digits(a, b, name)
        # ... and the shell's driver with the batch-by-batch loop (I3RC_DRIVER_MOMENTS=0) gives what its device moments give
        nml3 = nml2.replace("own_", "loop_")
        open(str(out / "loop.nml"), "w").write(nml3)
        r3 = subprocess.run([own, str(out / "loop.nml")], cwd=ROOT, env=dict(os.environ, I3RC_DRIVER_MOMENTS="0"), capture_output=True, text=True, timeout=600)
        assert r3.returncode == 0 and "Wrote ASCII results" in r3.stdout, r3.stdout + r3.stderr
        for name in ("flux", "rad", "absprof", "absvol"):
            a = open(str(out / f"own_{name}.txt")).read().splitlines()
            b = open(str(out / f"loop_{name}.txt")).read().splitlines()
            same_to_the_printed_digits(a[9:] if name == "flux" else a, b[9:] if name == "flux" else b, "loop " + name)
        from scipy.io import netcdf_file as _nc

        fo = _nc(str(out / "own_results.nc"), "r", mmap=False)
        fr = _nc(str(out / "stepCloud_results.nc"), "r", mmap=False)
        assert set(fo.variables) == set(fr.variables) and set(fo.dimensions) == set(fr.dimensions)
        for k in fr.variables:
            assert fo.variables[k].dimensions == fr.variables[k].dimensions
            # two drivers on the same deck: the reference's adds real(4) values batch by batch, the shell's gathers float64 moments
            # (standard errors are square roots of small differences of squares: they amplify a last-bit change of a mean)
            tol = 5e-3 if k.endswith("_StdErr") else 2e-5
            assert np.allclose(fo.variables[k].data, fr.variables[k].data, rtol=tol, atol=1e-6), k
        assert set(fo._attributes) == set(fr._attributes)
        for k in fr._attributes:
            if not k.startswith("Cpu_time") and k != "Domain_filename":
                assert np.all(fo._attributes[k] == fr._attributes[k]), k
        # ... and as two processes (batches split over ranks, moments summed across processes): same 10 batches
        nml3 = nml2.replace("own_flux.txt", "own2_flux.txt").replace("own_rad.txt", "own2_rad.txt").replace("own_absprof.txt", "own2_absprof.txt")
        open(str(out / "own2.nml"), "w").write(nml3)
        rcs, outs = _spawn_ranks([own, str(out / "own2.nml")], 2, 29641, cwd=ROOT)
        assert rcs == [0, 0], outs
        assert "batches on each of" in outs[0]
        # (round 5: the two processes' device moments are summed in float64, packed into ONE all-reduce -- sumBatchMomentsAcrossProcesses --
        # instead of ten real(4) reduces: the files of two processes equal those of one to the printed digits, every line of them)
        two = open(str(out / "own2_flux.txt")).read().splitlines()
        same_to_the_printed_digits(two[9:], own_flux.splitlines()[9:], "two processes: flux")
        for name in ("rad", "absprof"):
            a = open(str(out / f"own2_{name}.txt")).read().splitlines()
            b = open(str(out / f"own_{name}.txt")).read().splitlines()
            same_to_the_printed_digits(a, b, "two processes: " + name)
    from scipy.io import netcdf_file

    f = netcdf_file(str(out / "stepCloud_results.nc"), "r", mmap=False)
    assert f.variables["fluxUp"].data.shape == (1, 32) and f.variables["intensity"].data.shape == (1, 1, 32)
    assert abs(f.variables["fluxUp"].data.mean() - fup) < 1e-3


@pytest.mark.gpu
def test_generated_case_domains_run_through_the_driver_on_gpu(tmp_path):
    # case generator -> domain file -> read_Domain -> i3rcDriver on the GPU, against the Python host path on the numpy
    # statement of the same case (Landsat scene re-binned to 36 layers: clear cells carry phase function index 0)
    import i3rc_monte_carlo_model_amd as M
    from tools import cases

    gen, drv = _need(os.path.join(BUILD, "makeLandsatCloudDomain")), _need(os.path.join(BUILD, "i3rcDriver"))
    _write_i3rc_data_files(str(tmp_path))
    dom = str(tmp_path / "landsat36.dom")
    assert _run([gen, str(tmp_path), dom, "0.99", "36"]).returncode == 0 and os.path.exists(dom)
    nml = tmp_path / "landsat.nml"
    nml.write_text(f"""&radiativeTransfer
  solarFlux = 1., solarMu = 0.5, solarAzimuth = 30., surfaceAlbedo = 0.2 /
&monteCarlo
  numPhotonsPerBatch = 200000, numBatches = 10, iseed = 7, nPhaseIntervals = 10001 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .true. /
&output
  reportVolumeAbsorption = .false., reportAbsorptionProfile = .true. /
&fileNames
  domainFileName = "{dom}", outputFluxFile = "{tmp_path}/flux.txt", outputAbsProfFile = "{tmp_path}/prof.txt" /
""")
    r = _run([drv, str(nml)], cwd=ROOT)
    assert r.returncode == 0 and "Wrote ASCII results" in r.stdout, r.stdout + r.stderr
    m = re.search(r"Average:\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", open(str(tmp_path / "flux.txt")).read())
    fup, eup, fdn, edn, fab, eab = map(float, m.groups())
    d = cases.landsat_cloud(ssa=0.99, nlayers=36)
    dm = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dm.addOpticalComponent("cloud", d["ext"], d["ssa"], np.maximum(d["pf"], 1), M.PhaseFunctionTable([M.henyey_greenstein(0.85, 299)]))
    g = M.new_Integrator(dm)
    g.specifyParameters(surfaceAlbedo=0.2, minInverseTableSize=10001)
    ups, dns, abss = [], [], []
    for b in range(1, 11):   # the driver's batches: seed (iseed, batch), so the two runs trace the same photons
        res = g.computeRadiativeTransfer(M.new_RandomNumberSequence((7, b)), M.new_PhotonStream(0.5, 30.0, 200000))
        ups.append(res["fluxUp"].mean()); dns.append(res["fluxDown"].mean()); abss.append(res["fluxAbsorbed"].mean())
    # solarFlux = 1 and the same photons: the means agree to the 4 decimals the flux file prints
    assert abs(fup - np.mean(ups)) < 2e-4 and abs(fdn - np.mean(dns)) < 2e-4 and abs(fab - np.mean(abss)) < 2e-4
    assert abs(eup - np.std(ups, ddof=1) / np.sqrt(10)) < 2e-4


@pytest.mark.gpu
def test_reference_tool_chain_domain_through_the_drivers_on_gpu(tmp_path):
    """What a user of the reference's tools hands to the drivers: the LES stratocumulus field + Rayleigh scattering that the
    reference's MakeMieTable and PhysicalPropertiesToDomain wrote (tests/golden/tools_les_stcu_rayleigh.dom.gz: two components, a Mie
    table of 35 entries with up to 1381 Legendre coefficients, irregular layers) -- read_Domain, tables, ten batches with two radiance
    directions over a Lambertian surface -- through the shell's driver, through the reference's unchanged monteCarloDriver where it
    was built, and through the Python mirror on the same seeds (each side reads the file and makes its own tables)."""
    import i3rc_monte_carlo_model_amd as M

    drv = _need(os.path.join(BUILD, "i3rcDriver"))
    dom = _tool_chain_domain("les_stcu_rayleigh", tmp_path)
    deck = """&radiativeTransfer
  solarFlux = 1., solarMu = 0.5, solarAzimuth = 20., surfaceAlbedo = 0.06, intensityMus = 1., 0.6, intensityPhis = 0., 135. /
&monteCarlo
  numPhotonsPerBatch = 100000, numBatches = 10, iseed = 21, nPhaseIntervals = 10001 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .true., useRussianRouletteForIntensity = .true., zetaMin = 0.3 /
&output
  reportVolumeAbsorption = .false., reportAbsorptionProfile = .true. /
&fileNames
  domainFileName = "%s", outputFluxFile = "%s/WHO_flux.txt", outputRadFile = "%s/WHO_rad.txt", outputAbsProfFile = "%s/WHO_prof.txt" /
""" % (dom, tmp_path, tmp_path, tmp_path)
    (tmp_path / "own.nml").write_text(deck.replace("WHO", "own"))
    r = _run([drv, str(tmp_path / "own.nml")], cwd=ROOT)
    assert r.returncode == 0 and "Wrote ASCII results" in r.stdout, r.stdout + r.stderr
    flux = open(str(tmp_path / "own_flux.txt")).read()
    m = re.search(r"Average:\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", flux)
    fup, eup, fdn, edn, fab, eab = map(float, m.groups())
    assert 0.2 < fup < 0.9 and abs(fup + fab + fdn * (1 - 0.06) - 1.0) < 2e-3       # what comes in goes out or is absorbed (surface included)
    mc = os.path.join(BUILD, "monteCarloDriver_ref")
    if os.path.exists(mc):   # the reference's own driver, unchanged, on the same deck: the same photons, the same files
        (tmp_path / "ref.nml").write_text(deck.replace("WHO", "ref"))
        r = _run([mc, str(tmp_path / "ref.nml")], cwd=ROOT)
        assert r.returncode == 0 and "Wrote ASCII results" in r.stdout, r.stdout + r.stderr
        for name in ("flux", "rad", "prof"):
            a = open(str(tmp_path / f"own_{name}.txt")).read().splitlines()
            b = open(str(tmp_path / f"ref_{name}.txt")).read().splitlines()
            same_to_the_printed_digits(a[9:] if name == "flux" else a, b[9:] if name == "flux" else b, "tool-chain domain: " + name)
    # the Python mirror on the file: its own read_Domain, its own tables, the driver's seeds
    g = M.new_Integrator(M.read_Domain(dom))
    g.specifyParameters(surfaceAlbedo=0.06, minInverseTableSize=10001, intensityMus=[1.0, 0.6], intensityPhis=[0.0, 135.0],
                        useRussianRouletteForIntensity=True, zetaMin=0.3)
    res = [g.computeRadiativeTransfer(M.new_RandomNumberSequence((21, b)), M.new_PhotonStream(0.5, 20.0, 100000)) for b in range(1, 11)]
    assert "wide" in g.kernel_name(), g.kernel_name()     # several components, radiance: the widened class's kernels
    ups, dns, abss = ([float(x[k].mean()) for x in res] for k in ("fluxUp", "fluxDown", "fluxAbsorbed"))
    assert abs(fup - np.mean(ups)) < 2e-4 and abs(fdn - np.mean(dns)) < 2e-4 and abs(fab - np.mean(abss)) < 2e-4
    assert abs(eup - np.std(ups, ddof=1) / np.sqrt(10)) < 2e-4
    pixels = np.array([[float(v) for v in line.split()] for line in open(str(tmp_path / "own_rad.txt")).read().splitlines()
                       if not line.lstrip().startswith("!")])                     # x, y, mean, standard error: 64 x 64 pixels per direction
    assert pixels.shape == (2 * 64 * 64, 4)
    for k in range(2):   # (the file prints four decimals per pixel: their mean over 4096 pixels is good to 1e-5)
        want = np.mean([float(x["intensity"][k].mean()) for x in res])
        assert abs(pixels[k * 4096:(k + 1) * 4096, 2].mean() - want) < 2e-4, (k, pixels[k * 4096:(k + 1) * 4096, 2].mean(), want)
    g.finalize_Integrator()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["stepcloud_mu1", "stepcloud_mu05_absorbing", "radar640_nadir", "landsat36_flux", "example_mixture"])
def test_the_shell_driver_on_gpu_against_the_whole_reference_column_by_column(tmp_path, case):
    """BASELINE.json's parity rule -- per column and domain mean |gpu - ref| <= 3 sqrt(se_gpu^2 + se_ref^2) -- against THE REFERENCE ITSELF:
    tests/golden/ref_driver_*.nc are the result files of the reference's own driver on the reference's own integrator and modules, all
    unmodified (oracle/_ref/ref_driver, tests/golden/make_ref_driver.py: the step cloud of the reference's generator, 200 batches of 1e5
    photons; sun at the zenith, conservative; sun at 60 degrees, omega = 0.99, albedo 0.2; and the radar cloud 640 x 1 x 54 of BASELINE.json configs[2] with its nadir radiance, 40 batches of
    5e4; the Landsat scene in 36 layers of configs[3], 100 batches of 1e5, 16 384 columns; and THE REFERENCE'S OWN EXAMPLE as shipped -- Example-Drivers/monteCarloDriver.nml on the
    three-component column of Tools/Examples, three radiance directions, 200 batches of 1e4 where the deck has 4), and the shell's driver runs the same decks on the device: fluxUp, fluxDown, fluxAbsorbed, the nadir radiance per column (Student-t allowance for 32 columns at 398 degrees of
    freedom), the absorbed profile per layer and every domain mean."""
    import importlib.util

    from scipy import stats
    from scipy.io import netcdf_file

    spec = importlib.util.spec_from_file_location("make_ref_driver", os.path.join(ROOT, "tests", "golden", "make_ref_driver.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    drv = _need(os.path.join(BUILD, "i3rcDriver"))
    _need(os.path.join(BUILD, "makeStepCloudDomain")), _need(os.path.join(BUILD, "makeRadarCloudDomain")), _need(os.path.join(BUILD, "makeLandsatCloudDomain"))
    dom, out = str(tmp_path / "case.dom"), str(tmp_path / "results.nc")
    _write_i3rc_data_files(str(tmp_path))
    mod.make_domain(case, dom, str(tmp_path))
    (tmp_path / "deck.nml").write_text(mod.deck(case, dom, out))
    r = _run([drv, str(tmp_path / "deck.nml")], cwd=ROOT)
    assert r.returncode == 0 and "Wrote netCDF results" in r.stdout, r.stdout + r.stderr
    g, ref = netcdf_file(out, "r", mmap=False), netcdf_file(os.path.join(ROOT, "tests", "golden", f"ref_driver_{case}.nc"), "r", mmap=False)
    nb = mod.CASES[case].get("batches", 200)
    assert int(ref.Number_of_batches) == int(g.Number_of_batches) == nb and int(ref.Total_number_of_photons) == int(g.Total_number_of_photons)
    dof = 2 * nb - 2
    for key in ("fluxUp", "fluxDown", "fluxAbsorbed", "intensity", "absorptionProfile"):
        if key not in ref.variables:
            assert key == "intensity" and key not in g.variables     # (the Landsat deck asks for no radiance)
            continue
        mg, sg = g.variables[key].data.astype(np.float64).ravel(), g.variables[key + "_StdErr"].data.astype(np.float64).ravel()
        mr, sr = ref.variables[key].data.astype(np.float64).ravel(), ref.variables[key + "_StdErr"].data.astype(np.float64).ravel()
        if not mr.any():
            assert not mg.any(), key          # (nothing absorbs in the conservative case: zeros on both sides)
            continue
        z = np.abs(mg - mr) / (np.sqrt(sg ** 2 + sr ** 2) + 1e-7)
        expected = 2 * stats.t.sf(3.0, dof) * z.size
        assert (z > 3.0).sum() <= int(np.ceil(expected + 3 * np.sqrt(expected) + 1)), (key, int((z > 3.0).sum()), float(z.max()))
        assert z.max() < stats.t.isf(0.5e-3 / z.size, dof), (key, float(z.max()))
        # the domain mean: its standard error from the columns' (columns of a batch are not independent: an upper bound, as the drivers' own)
        se = np.sqrt((sg ** 2).sum() + (sr ** 2).sum()) / z.size
        assert abs(mg.mean() - mr.mean()) <= 3 * se + 1e-6, (key, mg.mean(), mr.mean(), se)


def test_photon_stream_constructors_equal_the_oracles_bit_for_bit(shell_built, oracle):
    """the shell's six photon sources (fortran/monteCarloIllumination.f95, written from the
    reference's interface) against the oracle's restatement of Code/monteCarloIllumination.f95:62-424
    (oracle/integrator.c, oracle/illumination.c) -- two independent implementations, same MT19937 seed: every
    position, cosine and azimuth identical in every bit, including the reference's quirks (Internal_Intensity keeps
    its azimuth in degrees, the finite detector is not centred)."""
    n = 2000
    r = _run([os.path.join(shell_built, "dumpPhotonStreams"), str(n)])
    assert r.returncode == 0, r.stdout + r.stderr
    got = {}
    for line in r.stdout.splitlines():
        f = line.split()
        got.setdefault(f[0], []).append([int(h, 16) for h in f[2:7]])
    O = oracle
    rng = lambda: O.RandomNumberSequence([10, 1])
    want = {
        "directional": O.photons_directional(rng(), 0.6, 135.0, n),
        "randomAzimuth": O.photons_random_azimuth(rng(), 0.6, n),
        "flux": O.photons_flux(rng(), n),
        "spotlight": O.photons_spotlight(0.6, 135.0, 0.25, 0.75, n),
        "internalFluxUp": O.photons_internal_flux(rng(), 0.4, 0.5, 0.3, True, n),
        "internalFluxDownFinite": O.photons_internal_flux(rng(), 0.4, 0.5, 0.3, False, n, delta_x=0.1, delta_y=0.2),
        "internalIntensity": O.photons_internal_intensity(rng(), 0.4, 0.5, 0.3, -0.7, 200.0, n, delta_x=0.1),
    }
    assert sorted(got) == sorted(want)
    for name, arrays in want.items():
        bits = np.stack([a.view(np.uint32) for a in arrays], axis=1)
        mine = np.array(got[name], dtype=np.uint32)
        assert mine.shape == (n, 5), (name, mine.shape)
        assert np.array_equal(mine, bits), (name, np.argwhere(mine != bits)[:5])
    # ... and both against the REFERENCE'S OWN constructors: the same dump program compiled against the reference's unmodified
    # Code/monteCarloIllumination.f95 (oracle/_ref/ref_streams; tests/golden/make_ref_photon_streams.py), 500 photons a stream
    # (a stream draws its deviates array by array: its first 500 photons depend on how many it has)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_photon_streams.npz"))
    assert sorted(ref.files) == sorted(want)
    r = _run([os.path.join(shell_built, "dumpPhotonStreams"), str(len(ref["flux"]))])
    assert r.returncode == 0, r.stdout + r.stderr
    shell = {}
    for line in r.stdout.splitlines():
        f = line.split()
        shell.setdefault(f[0], []).append([int(h, 16) for h in f[2:7]])
    for name in want:
        assert np.array_equal(ref[name], np.array(shell[name], dtype=np.uint32)), name


def _gpu_count():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: two ranks of module MultipleProcesses over RCCL (one-GPU boxes run the one-rank case)")
def test_multiple_processes_module_over_rccl_two_ranks():
    # sumAcrossProcesses (scalar, 1-D ... 4-D) = ncclAllReduce between two rank processes, one GPU each
    exe = _need(os.path.join(BUILD, "commSelfTest"))
    rcs, outs = _spawn_ranks([exe], 2, 29651, extra_env=dict(I3RC_COMM_BACKEND="rccl"))
    assert rcs == [0, 0], outs
    assert "rank 0 of 2 sums ok master=T" in outs[0] and "rank 1 of 2 sums ok master=F" in outs[1], outs
