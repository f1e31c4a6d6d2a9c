"""The benchmark workloads: BASELINE.json's configurations as synthetic inputs, shared by bench.py,
tools/cpu_baseline.py, tools/run_case.py and tools/config_bench.py.  Pure description: nothing here touches the GPU
or the oracle; `make_integrator` builds the product-side problem, `make_oracle` the checker-side one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIRS7 = dict(intensityMus=[1.0, 0.5, 0.5, 0.8, 0.8, 0.3, 0.3], intensityPhis=[0.0, 0.0, 180.0, 90.0, 270.0, 45.0, 225.0])
NADIR = dict(intensityMus=[1.0], intensityPhis=[0.0])
RRI = dict(useRussianRouletteForIntensity=True, zetaMin=0.3)   # the driver's defaults (monteCarloDriver.f95:75-76)

# name -> description of one workload.  `photons` = photons per GPU per step at the size BASELINE.json quotes
# (configs 3 / 4: 1e9 photons over 8 GPUs = 1.25e8 per GPU; `photons_node` = the whole batch, what `bench.py --gpus N` shards in its
# default (strong) mode); `cpu_photons` = photons per CPU-baseline batch so that the
# bounded CPU sample is about 10-30 s on 16 cores.
WORKLOADS = {
    "step16": dict(label="i3rcStepCloud 32x1x16 (HG g=0.85 64 moments, omega=1, mu0=1, albedo 0), flux up/down",
                   baseline_config=1, domain=("step_cloud", dict(nlayers=16)), moments=64, mu0=1.0, params={},
                   photons=100_000_000, cpu_photons=200_000),
    "step32": dict(label="i3rcStepCloud 32x1x32 (the reference generator's shape; HG g=0.85 64 moments, omega=1, mu0=1), flux",
                   baseline_config=1, domain=("step_cloud", dict(nlayers=32)), moments=64, mu0=1.0, params={},
                   photons=100_000_000, cpu_photons=200_000),
    "radar64_nadir": dict(label="i3rcRadarCloud 64x64x54 MMCR domain (labelled synthetic, HG g=0.85 299 moments), flux + "
                                "local-estimate nadir radiance (roulette, zetaMin 0.3)",
                          baseline_config=2, domain=("radar_cloud_64", {}), moments=299, mu0=1.0, params=dict(**NADIR, **RRI),
                          photons=100_000_000, cpu_photons=40_000),
    "radar640": dict(label="i3rcRadarCloud 640x1x54 (reference-exact field), flux", baseline_config=2,
                     domain=("radar_cloud", {}), moments=299, mu0=1.0, params={}, photons=100_000_000, cpu_photons=60_000),
    "radar640_nadir": dict(label="i3rcRadarCloud 640x1x54 (reference-exact field), flux + nadir radiance (roulette, zetaMin 0.3)",
                           baseline_config=2, domain=("radar_cloud", {}), moments=299, mu0=1.0, params=dict(**NADIR, **RRI),
                           photons=100_000_000, cpu_photons=40_000),
    "landsat36": dict(label="i3rcLandsatCloud 128x128x36 (labelled synthetic, HG g=0.85 299 moments, mu0=1), flux",
                      baseline_config=3, domain=("landsat_cloud", dict(nlayers=36)), moments=299, mu0=1.0, params={},
                      photons=125_000_000, photons_node=1_000_000_000, cpu_photons=60_000),
    # config 3 with omega = 0.99 in the cloudy cells (parity test: the value the cells with extinction share, volume absorption)
    "landsat36_absorbing": dict(label="i3rcLandsatCloud 128x128x36 (labelled synthetic), omega = 0.99, mu0=1, flux + absorption", baseline_config=3,
                                domain=("landsat_cloud", dict(nlayers=36, ssa=0.99)), moments=299, mu0=1.0, params={},
                                photons=125_000_000, photons_node=1_000_000_000, cpu_photons=60_000),
    # the I3RC cases' ABSORBING versions (the reference's generators write both: I3RC-Examples/i3rcStepCloud.f95:97, i3rcLandsatCloud.f95:138):
    # omega = 0.99 -- every scattering tallies absorbed flux and volume absorption (:642-649)
    "step16_absorbing": dict(label="i3rcStepCloud 32x1x16, omega = 0.99, mu0=1, flux + absorption", baseline_config=1,
                             domain=("step_cloud", dict(nlayers=16, ssa=0.99)), moments=64, mu0=1.0, params={}, photons=100_000_000, cpu_photons=200_000),
    "landsat119_absorbing": dict(label="i3rcLandsatCloud 128x128x119, omega = 0.99, mu0=1, flux + absorption", baseline_config=3,
                                 domain=("landsat_cloud", dict(ssa=0.99)), moments=299, mu0=1.0, params={}, photons=125_000_000, cpu_photons=40_000),
    "landsat119": dict(label="i3rcLandsatCloud 128x128x119 (reference-exact field, mu0=1), flux", baseline_config=3,
                       domain=("landsat_cloud", {}), moments=299, mu0=1.0, params={}, photons=125_000_000, photons_node=1_000_000_000, cpu_photons=40_000),
    "landsat119_7dir": dict(label="i3rcLandsatCloud 128x128x119 + 7 radiance directions + Lambertian surface 0.2 "
                                  "(surfaceProperties object), mu0=0.5, roulette zetaMin 0.3",
                            baseline_config=4, domain=("landsat_cloud", {}), moments=299, mu0=0.5,
                            params=dict(**DIRS7, **RRI), surface=0.2, photons=125_000_000, photons_node=1_000_000_000, cpu_photons=6_000),
    "landsat36_7dir": dict(label="i3rcLandsatCloud 128x128x36 + 7 radiance directions + Lambertian surface 0.2, mu0=0.5",
                           baseline_config=4, domain=("landsat_cloud", dict(nlayers=36)), moments=299, mu0=0.5,
                           params=dict(**DIRS7, **RRI), surface=0.2, photons=125_000_000, photons_node=1_000_000_000, cpu_photons=8_000),
    # beyond BASELINE.json: a field of production size (31 MB of extinction: 8 times an XCD's L2), the Landsat scene tiled 2 x 2
    "landsat_tiled": dict(label="Landsat scene tiled 2x2: 256x256x119 (31 MB field), mu0=1, flux", baseline_config=3,
                          domain=("landsat_tiled", {}), moments=299, mu0=1.0, params={}, photons=125_000_000, cpu_photons=40_000),
    "landsat_tiled_7dir": dict(label="Landsat scene tiled 2x2: 256x256x119 + 7 radiance directions + Lambertian surface 0.2, mu0=0.5",
                               baseline_config=4, domain=("landsat_tiled", {}), moments=299, mu0=0.5,
                               params=dict(**DIRS7, **RRI), surface=0.2, photons=125_000_000, cpu_photons=6_000),
    # ---- beyond the common class (round 5): what Tools/PhysicalPropertiesToDomain.f95 produces is cloud + aerosol + gas -- several
    # components, every cell optically active --; irregular x / y spacing and a gridded surface send a run to the GENERAL kernels
    "landsat119_gas": dict(label="i3rcLandsatCloud 128x128x119 + a horizontally uniform Rayleigh-like gas (two components, two "
                                 "phase-function tables, every cell active: optical depth 0.04), mu0=1, flux", baseline_config=3,
                           domain=("landsat_cloud", {}), gas=(2.0e-5, 1.5e-5), moments=299, mu0=1.0, params={}, photons=125_000_000, cpu_photons=20_000),
    "landsat119_gas_7dir": dict(label="i3rcLandsatCloud 128x128x119 + gas (two components) + 7 radiance directions + Lambertian surface 0.2, mu0=0.5",
                                baseline_config=4, domain=("landsat_cloud", {}), gas=(2.0e-5, 1.5e-5), moments=299, mu0=0.5,
                                params=dict(**DIRS7, **RRI), surface=0.2, photons=125_000_000, cpu_photons=4_000),
    "landsat36_gas": dict(label="i3rcLandsatCloud 128x128x36 + gas (two components), mu0=1, flux", baseline_config=3,
                          domain=("landsat_cloud", dict(nlayers=36)), gas=(2.0e-5, 1.5e-5), moments=299, mu0=1.0, params={}, photons=125_000_000, cpu_photons=40_000),
    # ... with absorption in both components (parity test: omega and the table entry read per cell AND component)
    "landsat36_gas_absorbing": dict(label="i3rcLandsatCloud 128x128x36, omega = 0.99, + gas with omega = 0.9 (two components), mu0=1, flux + absorption",
                                    baseline_config=3, domain=("landsat_cloud", dict(nlayers=36, ssa=0.99)), gas=(2.0e-5, 1.5e-5), gas_ssa=0.9, moments=299,
                                    mu0=1.0, params={}, photons=125_000_000, cpu_photons=40_000),
    # ... three components, what PhysicalPropertiesToDomain's own example makes (droplets + aerosol + gas): the Landsat scene, an aerosol layer
    # (HG g = 0.7, omega = 0.92, optical depth 0.15 in the lowest third of the domain) and the gas
    "landsat36_aerosol_gas": dict(label="i3rcLandsatCloud 128x128x36 (omega 0.99) + aerosol layer (omega 0.92) + gas (omega 0.9): three components, mu0=1, flux + absorption",
                                  baseline_config=3, domain=("landsat_cloud", dict(nlayers=36, ssa=0.99)), gas=(2.0e-5, 1.5e-5), gas_ssa=0.9, aerosol=0.15, moments=299,
                                  mu0=1.0, params={}, photons=125_000_000, cpu_photons=40_000),
    "landsat119_irregular_7dir": dict(label="i3rcLandsatCloud 128x128x119 on an irregular x / y grid (cell widths 30 m +- 20 %) + 7 radiance "
                                            "directions + Lambertian surface 0.2, mu0=0.5", baseline_config=4, domain=("landsat_cloud", {}), irregular=0.2,
                                      moments=299, mu0=0.5, params=dict(**DIRS7, **RRI), surface=0.2, photons=125_000_000, cpu_photons=6_000),
    "landsat119_brdfgrid_7dir": dict(label="i3rcLandsatCloud 128x128x119 + 7 radiance directions + a gridded Lambertian surface (8 x 8 cells, "
                                           "reflectance 0.05 ... 0.35), mu0=0.5", baseline_config=4, domain=("landsat_cloud", {}), surface_grid=8,
                                     moments=299, mu0=0.5, params=dict(**DIRS7, **RRI), photons=125_000_000, cpu_photons=6_000),
    # ---- a domain as the reference's own tool chain writes it (tests/golden/make_tool_domains.py: MakeMieTable -> PhysicalPropertiesToDomain on
    # Tools/Examples, the reference's programs unchanged on the shell): an LES stratocumulus field of 64 x 64 x 16 cloudy cells in 18 irregular
    # layers, a Mie table of 35 effective radii (up to 1381 Legendre coefficients, the table entry chosen per cell), + Rayleigh scattering
    "les_stcu_rayleigh": dict(label="LES stratocumulus 64x64x18 from the reference's tool chain (Mie table of 35 entries + Rayleigh gas: two components), "
                                    "mu0=0.5, Lambertian surface 0.06, flux", baseline_config=3, domain_file="tests/golden/tools_les_stcu_rayleigh.dom.gz",
                              mu0=0.5, params={}, albedo=0.06, photons=100_000_000, cpu_photons=100_000),
    "les_stcu_rayleigh_2dir": dict(label="LES stratocumulus 64x64x18 from the reference's tool chain (Mie table + Rayleigh gas) + 2 radiance directions, "
                                         "mu0=0.5, Lambertian surface 0.06", baseline_config=4, domain_file="tests/golden/tools_les_stcu_rayleigh.dom.gz",
                                   mu0=0.5, params=dict(intensityMus=[1.0, 0.6], intensityPhis=[0.0, 135.0], **RRI), albedo=0.06,
                                   photons=50_000_000, cpu_photons=50_000),
}
# Work per photon of the REFERENCE'S ALGORITHM on the four bench workloads (S tracer iterations incl. local-estimate rays,
# K scatterings, E boundary tallies): the figures SURVEY.md 8(d)'s byte formula is evaluated with.  Recorded from runs in
# which the kernel still traced every ray (profiles/r02m_*_bench.json; equal to the oracle's counters within Monte Carlo
# noise, tests/test_gpu_baseline_configs.py); bench.py prefers the oracle's live counters when its CPU-baseline leg ran.
REFERENCE_WORK = {
    "step16": dict(S=45.791, K=17.002, E=1.0000),
    "radar64_nadir": dict(S=198.07, K=44.342, E=0.9998),
    "landsat36": dict(S=150.02, K=15.787, E=0.9999),
    "landsat119_7dir": dict(S=3206.2, K=23.228, E=1.1236),
}

ALIASES = {"step_cloud": "step16", "landsat7": "landsat119_7dir", "radar_nadir": "radar640_nadir", "radar": "radar640",
           "landsat": "landsat119"}


def get(name):
    name = ALIASES.get(name, name)
    if name not in WORKLOADS:
        raise SystemExit(f"unknown workload {name!r}; one of {sorted(WORKLOADS)}")
    return name, WORKLOADS[name]


def domain(w):
    import numpy as np

    from tools import cases

    if "domain_file" in w:
        return domain_from_file(w)[1]
    fn, kw = w["domain"]
    d = getattr(cases, fn)(**kw)
    if w.get("irregular"):   # irregular x / y spacing: cell widths drawn once (fixed seed) within +- the given fraction of the regular width
        rng = np.random.default_rng(119)
        for key in ("xe", "ye"):
            e = d[key].astype(np.float64)
            widths = np.diff(e) * rng.uniform(1.0 - w["irregular"], 1.0 + w["irregular"], len(e) - 1)
            d[key] = np.concatenate([[e[0]], e[0] + np.cumsum(widths)]).astype(np.float32)
    return d


def domain_from_file(w):
    """A workload whose domain is a FILE (gzipped netCDF classic, the reference's domain-file schema): (Domain, arrays) -- the Python
    mirror's read_Domain, and every component's arrays on the whole grid [component][z][y][x] as the oracle takes them."""
    import gzip
    import os
    import tempfile

    import numpy as np

    import i3rc_monte_carlo_model_amd as M

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), w["domain_file"])
    with tempfile.NamedTemporaryFile(suffix=".dom") as tmp:
        tmp.write(gzip.decompress(open(path, "rb").read()))
        tmp.flush()
        dom = M.read_Domain(tmp.name)
    nz, ny, nx = dom.shape

    def full(c, key, dtype):
        a = np.zeros((nz, ny, nx), dtype)
        z0 = c["zbase"] - 1
        a[z0:z0 + c[key].shape[0]] = np.broadcast_to(c[key], (c[key].shape[0], ny, nx))
        return a
    d = dict(xe=dom.x, ye=dom.y, ze=dom.z, ext=np.stack([full(c, "ext", np.float32) for c in dom.components]),
             ssa=np.stack([full(c, "ssa", np.float32) for c in dom.components]), pf=np.stack([full(c, "pfi", np.int32) for c in dom.components]))
    return dom, d


AEROSOL_G, AEROSOL_SSA, AEROSOL_MOMENTS = 0.7, 0.92, 32


def aerosol_component(w, d):
    """extinction of the horizontally uniform aerosol layer of an `aerosol=<optical depth>` workload: the lowest third of the layers, [nz][ny][nx]"""
    import numpy as np

    nz = d["ext"].shape[0]
    top = max(1, nz // 3)
    depth = float(d["ze"][top] - d["ze"][0])
    profile = np.zeros(nz, np.float32)
    profile[:top] = np.float32(w["aerosol"] / depth)
    return np.broadcast_to(profile[:, None, None], d["ext"].shape).copy()


def gas_component(w, d):
    """extinction of the horizontally uniform gas of a `gas=(bottom, top)` workload, [nz][ny][nx]"""
    import numpy as np

    nz = d["ext"].shape[0]
    profile = np.linspace(w["gas"][0], w["gas"][1], nz, dtype=np.float32)
    return np.broadcast_to(profile[:, None, None], d["ext"].shape).copy()


GAS_LEGENDRE = (0.0, 0.1)   # the Rayleigh-like phase function's Legendre coefficients (P1, P2 without the factor 2l + 1)


def surface_grid(w, d):
    """a gridded Lambertian surface over the domain: n x n cells, reflectance 0.05 ... 0.35 in a fixed pattern"""
    import numpy as np

    n = w["surface_grid"]
    xs = np.linspace(d["xe"][0], d["xe"][-1], n + 1).astype(np.float32)
    ys = np.linspace(d["ye"][0], d["ye"][-1], n + 1).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    refl = (0.05 + 0.30 * (((3 * i + 5 * j) % 7) / 6.0)).astype(np.float32)   # [ny][nx]
    return xs, ys, refl


def n_dir(w):
    return len(w["params"].get("intensityMus", ()))


def make_integrator(w, device=0):
    """The product-side problem (Python mirror of the reference's module API over the C ABI)."""
    import numpy as np

    import i3rc_monte_carlo_model_amd as M

    if "domain_file" in w:
        dom, d = domain_from_file(w)
    else:
        d = domain(w)
        table = M.PhaseFunctionTable([M.henyey_greenstein(0.85, w["moments"])])
        dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
        dom.addOpticalComponent("cloud", d["ext"], d["ssa"], d["pf"], table)
    if "aerosol" in w:
        aer = aerosol_component(w, d)
        dom.addOpticalComponent("aerosol", aer, np.where(aer > 0, np.float32(AEROSOL_SSA), np.float32(0)).astype(np.float32), (aer > 0).astype(np.int32),
                                M.PhaseFunctionTable([M.henyey_greenstein(AEROSOL_G, AEROSOL_MOMENTS)]))
    if "gas" in w:
        gas = gas_component(w, d)
        dom.addOpticalComponent("gas", gas, np.full_like(gas, np.float32(w.get("gas_ssa", 1.0))), np.ones(gas.shape, np.int32),
                                M.PhaseFunctionTable([M.PhaseFunction(legendre=np.array(GAS_LEGENDRE, np.float32))]))
    g = M.new_Integrator(dom, device=device)
    kw = dict(w["params"])
    if "surface_grid" in w:
        xs, ys, refl = surface_grid(w, d)
        kw["surfaceBDRF"] = M.new_SurfaceDescription(refl, xs, ys)
    elif "surface" in w:
        kw["surfaceBDRF"] = M.new_SurfaceDescription([w["surface"]])
    else:
        kw["surfaceAlbedo"] = w.get("albedo", 0.0)
    g.specifyParameters(minInverseTableSize=10001, minForwardTableSize=10001, useRayTracing=True, useRussianRoulette=True, **kw)
    return g, d


def make_oracle(w):
    """The same problem for the CPU oracle (test infrastructure; bench.py's cpu_baseline leg only)."""
    import numpy as np

    from oracle import pyoracle as O

    nd = n_dir(w)
    if "domain_file" in w:   # (the tables of a file's components: the Python mirror's restatement of the reference's table routines)
        dom, d = domain_from_file(w)
        inv = [c["table"].inverse_table(10001) for c in dom.components]
        fwd = [c["table"].forward_table(10001) for c in dom.components] if nd else None
    else:
        d = domain(w)
        coef = O.hg_coefficients(0.85, w["moments"])
        inv = [O.inverse_table_legendre(coef, 10001)]
        fwd = [O.forward_table_legendre(coef, 10001)] if nd else None
    ext, ssa, pf = d["ext"], d["ssa"], d["pf"]
    if "aerosol" in w or "gas" in w:
        ext, ssa, pf = [ext], [ssa], [pf]
        if "aerosol" in w:
            aer = aerosol_component(w, d)
            acoef = O.hg_coefficients(AEROSOL_G, AEROSOL_MOMENTS)
            ext.append(aer), ssa.append(np.where(aer > 0, np.float32(AEROSOL_SSA), np.float32(0)).astype(np.float32)), pf.append((aer > 0).astype(np.int32))
            inv.append(O.inverse_table_legendre(acoef, 10001))
            if nd:
                fwd.append(O.forward_table_legendre(acoef, 10001))
        if "gas" in w:
            gas = gas_component(w, d)
            gcoef = np.array(GAS_LEGENDRE, np.float32)
            ext.append(gas), ssa.append(np.full_like(gas, np.float32(w.get("gas_ssa", 1.0)))), pf.append(np.ones(gas.shape, np.int32))
            inv.append(O.inverse_table_legendre(gcoef, 10001))
            if nd:
                fwd.append(O.forward_table_legendre(gcoef, 10001))
        ext, ssa, pf = np.stack(ext), np.stack(ssa), np.stack(pf)
    o = O.Integrator(d["xe"], d["ye"], d["ze"], ext, ssa, pf, inv, fwd, fwd)
    kw = {}
    if nd:
        kw.update(intensityMus=w["params"]["intensityMus"], intensityPhis=w["params"]["intensityPhis"],
                  useRRForIntensity=int(bool(w["params"].get("useRussianRouletteForIntensity"))),
                  zetaMin=w["params"].get("zetaMin", 0.3))
    if "surface_grid" in w:
        xs, ys, refl = surface_grid(w, d)
        kw.update(surfaceBDRF=(xs, ys, refl))
    elif "surface" in w:   # new_SurfaceDescription((/ albedo /)): one cell with edges (0, huge), Code/surfaceProperties.f95:98-117
        huge = np.finfo(np.float32).max
        kw.update(surfaceBDRF=(np.array([0.0, huge], np.float32), np.array([0.0, huge], np.float32),
                               np.array([[w["surface"]]], np.float32)))
    if "albedo" in w:
        kw.update(surfaceAlbedo=w["albedo"])
    if kw:
        o.specify(**kw)
    return o, d
