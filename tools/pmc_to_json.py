"""rocprofv3 --pmc passes of tools/pmc_profile.sh -> a per-kernel summary and the figures bench.py quotes.
  python3 tools/pmc_to_json.py --summarise <dir> <workload>   reads <dir>/p*/**/*counter_collection.csv, writes <dir>/summary.txt
  python3 tools/pmc_to_json.py --collect profiles/r02_pmc.json profiles/r02_<workload>_pmc_summary.txt ...
        summary files (first line: "# workload <name> photons <n>") -> JSON {workload: {valu_instr_per_photon, ...}}
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KB, and on gfx950 FETCH_SIZE counts the 128-byte
fabric requests at 64 bytes each (/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import re
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _commit():
    import subprocess

    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        return None


COMMIT = _commit()   # (HEAD where the summaries were turned into the JSON: the build container; None on the GPU box, which has no .git)


def summarise(out, workload):
    vals, photons = {}, 0
    for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "photon_kernel" in r["Kernel_Name"]:
                # the 1-photon table warm-up launch is a dispatch of the same kernel: its counts are negligible and included
                vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
                vals["_VGPR"] = r["VGPR_Count"]; vals["_SGPR"] = r["SGPR_Count"]; vals["_LDS"] = r["LDS_Block_Size"]
                vals["_kernel"] = re.sub(r"^void i3rc::", "", r["Kernel_Name"]).split("(")[0]
    for f in glob.glob(os.path.join(out, "p*.out")):
        t = open(f).read()
        m = re.search(r"ms, (\d+) photons\)", t)
        if m:
            photons = int(m.group(1))
        # what the profiled launch did, in the kernel's own words (tools/run_case.py's line): its name and its work per photon --
        # bench.py holds its own run against these, so that counters of another kernel or another workload do not pass as this one's
        m = re.search(r"kernel (photon_kernel<[^>]*>) S=([\d.]+) K=([\d.]+)", t)
        if m:
            vals["_launch"] = m.group(1); vals["_S"] = m.group(2); vals["_K"] = m.group(3)
    with open(os.path.join(out, "summary.txt"), "w") as o:
        o.write(f"# workload {workload} photons {photons}\n")
        for k in sorted(vals):
            o.write(f"{k} {vals[k]}\n")
    print(open(os.path.join(out, "summary.txt")).read())


def collect(dst, files):
    res = {}
    for f in files:
        lines = open(f).read().splitlines()
        m = re.match(r"# workload (\S+) photons (\d+)", lines[0])
        name, n = m.group(1), int(m.group(2))
        v = {}
        for ln in lines[1:]:
            k, x = ln.split(" ", 1)
            try:
                v[k] = float(x)
            except ValueError:
                v[k] = x
        e = {"photons": n, "kernel": v.get("_kernel"), "summary": os.path.relpath(f, ROOT) if os.path.isabs(f) else f}
        if "_launch" in v:
            e["launch"] = v["_launch"]; e["work_per_photon"] = {"S": float(v["_S"]), "K": float(v["_K"])}
        e["collected_at_commit"] = COMMIT
        if "SQ_INSTS_VALU" in v:
            e["valu_instr_per_photon"] = v["SQ_INSTS_VALU"] / n
            e["salu_instr_per_photon"] = v.get("SQ_INSTS_SALU", 0) / n
            # everything the one scalar unit of a compute unit issues: scalar ALU, branches, scalar memory
            e["scalar_instr_per_photon"] = (v.get("SQ_INSTS_SALU", 0) + v.get("SQ_INSTS_BRANCH", 0) + v.get("SQ_INSTS_SMEM", 0)) / n
        if "SQ_THREAD_CYCLES_VALU" in v and "SQ_INSTS_VALU" in v:
            # thread-cycles per VALU wave-instruction / 64: the share of the 64 lanes that were switched on
            # (SQ_THREAD_CYCLES_VALU counts active lanes x 4 cycles... calibrated: a full wave gives 64 per instruction
            # in the units the r01 summaries used: SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU))
            e["lane_occupancy"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"])
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["fetch_KB"] = v["FETCH_SIZE"]; e["write_KB"] = v["WRITE_SIZE"]
            e["hbm_bytes_per_photon"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / n
        for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"):
            if k in v:
                e[k] = v[k]
        res[name] = e
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "--collect":
        collect(sys.argv[2], sys.argv[3:])
