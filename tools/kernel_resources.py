"""Register / scratch / occupancy table of every gfx950 kernel of the library, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; no GPU needed).  tests/test_build_isa.py asserts on the same data.
  python3 tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def demangle_photon_kernel(name):
    m = re.match(r"_ZN4i3rc13photon_kernelINS_(\d+)(\w+?)ELb(\d)ELb(\d)ELi(\d)E(?:Lb(\d)E)?(?:Lb(\d)E)?(?:Lb(\d)E)?EE", name)
    if not m:
        return re.sub(r"^_ZN4i3rc\d+", "", name)[:40]
    rng = m.group(2)[:int(m.group(1))]
    place = ["GRID_LDS", "GRID_GLOBAL", "GRID_BRICKS", "GRID_COLUMNS", "GRID_COLBASE"][int(m.group(5))]
    tbl = ", table in LDS" if m.group(6) == "1" else ""   # (the fifth template argument: the round-3 experiment, kernels.hpp TBL)
    if m.group(7) == "1": tbl += ", one direction"          # (the sixth: radiance kernels without an event ring, kernels.hpp DIRECT)
    if m.group(8) == "1": tbl += ", wide"     # (the seventh: the common class with several components, kernels.hpp MULTI)
    return f"photon_kernel<{rng}, {'true' if m.group(3) == '1' else 'false'}, {'true' if m.group(4) == '1' else 'false'}, {place}{tbl}>"


def resources(extra=()):
    import i3rc_monte_carlo_model_amd as M

    flags = [f for f in M.build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [M.build.hipcc()] + flags + list(extra) + ["--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-I",
                                                       os.path.join(ROOT, "include"), "-o", os.devnull,
                                                       os.path.join(M.build.CSRC, "i3rc_hip.hip")]
    err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s*\[-Rpass-analysis", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip()}
            cur["name"] = demangle_photon_kernel(cur["mangled"])
            out.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            try:
                cur[k.strip()] = int(v.strip())
            except ValueError:
                cur[k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    rows = resources(sys.argv[1:])
    keys = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
    print(f"{'kernel':62s} sgpr vgpr agpr scratch sSpill vSpill occ lds")
    for r in rows:
        print(f"{r['name']:62s} " + " ".join(f"{r.get(k, '-'):>5}" for k in keys))
