"""Makes tests/golden/tools_les_stcu_rayleigh.dom.gz and tests/golden/tools_mixture.dom.gz: two domain files as the
REFERENCE'S OWN TOOL CHAIN writes them from the reference's own example inputs (Tools/Examples), with the reference's
programs compiled unchanged and in place against this tree's Fortran shell (`make linkcheck` in
i3rc-monte-carlo-model_amd/fortran: build/MakeMieTable_ref, build/PhysicalPropertiesToDomain_ref -- never copied,
never tracked):

  MakeMieTable (Tools/MakeMieTable.f95 + mieindsub.f + RefractiveIndex-IceAndWater.f)
      mie_table_cloud.nml    -> cloud_w0.67_mie.phasetab   water droplets at 0.675 um, 35 effective radii, up to 1381 Legendre terms
      mie_table_aerosol.nml  -> dust_w0.67_mie.phasetab    absorbing aerosol, 35 effective radii
  PhysicalPropertiesToDomain (Tools/PhysicalPropertiesToDomain.f95)
      cloud_to_domain.nml, with RayleighWavelength = 0.675 and the table above
                             -> an LES stratocumulus field of 64 x 64 x 16 cloudy cells (i3rc_les_stcu.lwc) in a domain of 18 layers
                                + Rayleigh scattering in every layer: TWO components
      cloudAndDust_to_domain.nml, with the two tables above
                             -> one column of 11 layers: droplets + dust + molecular absorption: THREE components

The files are inputs of GPU tests (tests/test_fortran_shell.py, tests/test_gpu_baseline_configs.py): what a user of the reference's tools
would hand to the drivers.  They are data; the programs that made them stay where they are.  Only runs where /root/reference exists.
usage: python3 tests/golden/make_tool_domains.py [--check]     (--check: make them again and compare with the committed files)"""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.path.join(ROOT, "i3rc-monte-carlo-model_amd", "fortran", "build")
EX = "/root/reference/Tools/Examples"
OUT = {"i3rc_les_stcu.dom": "tools_les_stcu_rayleigh.dom.gz", "mixture.dom": "tools_mixture.dom.gz"}


def run(tool, nml, cwd):
    r = subprocess.run([os.path.join(BUILD, tool), nml], cwd=cwd, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(f"{tool} {nml}: rc {r.returncode}\n{r.stdout}\n{r.stderr}")
    return r.stdout


def make(directory):
    """The tool chain in `directory`; the reference's inputs are read where they lie (absolute paths in the namelists),
    every output goes to `directory`."""
    def edited(name, subs):
        text = open(os.path.join(EX, name)).read()
        for a, b in subs:
            assert a in text, (name, a)
            text = text.replace(a, b)
        open(os.path.join(directory, name), "w").write(text)
        return name

    run("MakeMieTable_ref", edited("mie_table_cloud.nml", []), directory)
    run("MakeMieTable_ref", edited("mie_table_aerosol.nml", []), directory)
    run("PhysicalPropertiesToDomain_ref",
        edited("cloud_to_domain.nml", [("cloud_w2.13_mie", "cloud_w0.67_mie"), ("i3rc_les_stcu.lwc", EX + "/i3rc_les_stcu.lwc"),
                                       ("RayleighWavelength=0.0", "RayleighWavelength=0.675")]), directory)
    run("PhysicalPropertiesToDomain_ref",
        edited("cloudAndDust_to_domain.nml", [("cloud_w2.13_mie", "cloud_w0.67_mie"), ("dust_w2.13_mie", "dust_w0.67_mie"),
                                              ("cloud_dust.part", EX + "/cloud_dust.part"), ("molec_abs_w213.dat", EX + "/molec_abs_w213.dat")]),
        directory)
    return {k: open(os.path.join(directory, k), "rb").read() for k in OUT}


if __name__ == "__main__":
    if not os.path.isdir(EX) or not os.path.exists(os.path.join(BUILD, "PhysicalPropertiesToDomain_ref")):
        raise SystemExit("needs /root/reference and `make linkcheck` (i3rc-monte-carlo-model_amd/fortran)")
    with tempfile.TemporaryDirectory() as tmp:
        files = make(tmp)
    for name, data in files.items():
        path = os.path.join(HERE, OUT[name])
        if "--check" in sys.argv:
            same = gzip.decompress(open(path, "rb").read()) == data
            print(f"{OUT[name]}: {'identical' if same else 'DIFFERENT'} ({len(data)} bytes)")
            if not same:
                raise SystemExit(1)
        else:
            with open(path, "wb") as f:
                f.write(gzip.compress(data, 9, mtime=0))
            print(f"wrote {path} ({len(data)} bytes, {os.path.getsize(path)} compressed)")
