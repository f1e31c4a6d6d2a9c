"""Makes tests/golden/ref_driver_*.nc: the result files of THE WHOLE REFERENCE -- Example-Drivers/monteCarloDriver.f95 on the reference's
own integrator and modules, all unmodified (oracle/Makefile, target _ref_loop: oracle/_ref/ref_driver; what that build is:
oracle/ref_loop.f95's header) -- for BASELINE.json's CPU-reference configuration, the I3RC step cloud as the reference's generator makes it
(32 x 1 x 32): sun at the zenith, conservative, with the nadir radiance; and sun at 60 degrees, omega = 0.99, over a surface of albedo 0.2.
Two hundred batches of 1e5 photons each (2e7 photons: eight minutes of the reference on one core) (the driver's batch / seed scheme, monteCarloDriver.f95:264-326), written by the driver's own
writeResults_netcdf (:609-854).  The GPU tests run the shell's driver on the same decks and compare column by column.
Also BASELINE.json configs[2]'s reference-exact field, the radar cloud 640 x 1 x 54 with its nadir radiance (40 batches of 5e4 photons).
Only runs where /root/reference exists.   usage: python3 tests/golden/make_ref_driver.py [--all]"""
import os
import shutil
import sys
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(ROOT, "i3rc-monte-carlo-model_amd", "fortran", "build")

DECK = """&radiativeTransfer
  solarFlux = 1., solarMu = {mu0}, solarAzimuth = 0., surfaceAlbedo = {albedo}, intensityMus = 1., intensityPhis = 0. /
&monteCarlo
  numPhotonsPerBatch = 100000, numBatches = 200, iseed = 10, nPhaseIntervals = 10001 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .true., useRussianRouletteForIntensity = .true., zetaMin = 0.3,
  useHybridPhaseFunsForIntenCalcs = .false., hybridPhaseFunWidth = 7., numOrdersOrigPhaseFunIntenCalcs = 0,
  limitIntensityContributions = .false., maxIntensityContribution = 77. /
&output
  reportVolumeAbsorption = .false., reportAbsorptionProfile = .true. /
&fileNames
  domainFileName = "{dom}", outputNetcdfFile = "{out}" /
"""
CASES = {"stepcloud_mu1": dict(ssa="1.0", mu0="1.", albedo="0."), "stepcloud_mu05_absorbing": dict(ssa="0.99", mu0="0.5", albedo="0.2"),
         # BASELINE.json configs[2]'s reference-exact field: the radar cloud 640 x 1 x 54, flux + nadir radiance, 40 batches of 5e4 photons
         "radar640_nadir": dict(ssa="1.0", mu0="1.", albedo="0.", radar=True, batches=40, photons=50000),
         # BASELINE.json configs[3]: the Landsat scene re-binned to 36 layers, flux, sun at the zenith: 100 batches of 1e5 photons
         "landsat36_flux": dict(ssa="1.0", mu0="1.", albedo="0.", landsat=36, batches=100, photons=100000, no_radiance=True),
         # THE REFERENCE'S OWN EXAMPLE as it ships it (Example-Drivers/monteCarloDriver.nml): the three-component column its tool chain makes
         # (Tools/Examples -> mixture.dom: tests/golden/tools_mixture.dom.gz), sun at 60 degrees, three radiance directions; 200 batches of
         # 1e4 photons where the shipped deck runs 4
         "example_mixture": dict(mu0="0.5", albedo="0.", fixture="tools_mixture.dom.gz", batches=200, photons=10000,
                                 directions=("1., .5, .5", "0., 0., 180."))}


def deck(case, dom, out):
    c = CASES[case]
    text = DECK.format(mu0=c["mu0"], albedo=c["albedo"], dom=dom, out=out)
    if c.get("no_radiance"):
        text = text.replace(", intensityMus = 1., intensityPhis = 0.", "")
    if c.get("directions"):
        text = text.replace("intensityMus = 1., intensityPhis = 0.", "intensityMus = %s, intensityPhis = %s" % c["directions"])
    return text.replace("numPhotonsPerBatch = 100000, numBatches = 200", f"numPhotonsPerBatch = {c.get('photons', 100000)}, numBatches = {c.get('batches', 200)}")


def make_domain(case, dom, data_dir):
    """the case's domain file through the shell's generators (whose files are the reference generators' bit for bit: tests/test_fortran_shell.py)"""
    c = CASES[case]
    if c.get("fixture"):
        import gzip
        with open(dom, "wb") as f:
            f.write(gzip.decompress(open(os.path.join(HERE, c["fixture"]), "rb").read()))
        return
    if c.get("radar"):
        cmd = [os.path.join(BUILD, "makeRadarCloudDomain"), data_dir, dom, c["ssa"], "hg"]
    elif c.get("landsat"):
        cmd = [os.path.join(BUILD, "makeLandsatCloudDomain"), data_dir, dom, c["ssa"], str(c["landsat"])]
    else:
        cmd = [os.path.join(BUILD, "makeStepCloudDomain"), dom, "32", c["ssa"]]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    if not os.path.isdir("/root/reference/Example-Drivers"):
        raise SystemExit("needs /root/reference")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref_loop"])
    for case, c in CASES.items():
        with tempfile.TemporaryDirectory() as tmp:
            dom, out, nml = os.path.join(tmp, "step.dom"), os.path.join(tmp, "results.nc"), os.path.join(tmp, "deck.nml")
            if os.path.exists(os.path.join(HERE, f"ref_driver_{case}.nc")) and "--all" not in sys.argv:
                continue                      # (the step cloud's two take eight minutes: --all makes them again)
            make_domain(case, dom, "/root/reference/I3RC-Examples/Data")
            open(nml, "w").write(deck(case, dom, out))
            r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_driver"), nml], capture_output=True, text=True, cwd=tmp)
            assert r.returncode == 0 and "Wrote netcdf results" in r.stdout, r.stdout + r.stderr
            shutil.copy(out, os.path.join(HERE, f"ref_driver_{case}.nc"))
            print(case, r.stdout.strip().splitlines()[-3:])
