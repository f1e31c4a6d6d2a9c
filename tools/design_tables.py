"""Rewrites the value cells of DESIGN.md's evidence tables (section 6 / 7) from the files under profiles/ that the rows cite, so that
a new evidence round (tools/evidence_round.sh <tag> counters | loop | collect) reaches the document without retyping:
    python3 tools/design_tables.py [tag]         (default r05; prints the rows it changed)
A row is found by the beginning of its first cell; rows this script does not know are left alone (tests/test_evidence.py checks all)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", f"{TAG}_{name}")
J = lambda w: json.load(open(P(f"{w}_bench.json")))
pmc = json.load(open(P("pmc.json")))
text = lambda name: open(P(name)).read()


def e(x, digits=4):          # 3.492e9 style
    m, ex = ("%.*e" % (digits - 1, x)).split("e")
    return f"{m}e{int(ex)}"


def kernel_avg_ns(w):
    for line in open(P(f"{w}_kernel_stats.csv")):
        if "photon_kernel" in line:
            return float(line.rsplit('",', 1)[1].split(",")[2])
    raise SystemExit(f"no photon kernel in the kernel stats of {w}")


def grab(name, pattern, group=1, cast=float, which=0):
    m = re.findall(pattern, text(name))
    if not m:
        raise SystemExit(f"{name}: nothing matches {pattern!r}")
    v = m[which]
    return cast(v[group - 1] if isinstance(v, tuple) else v)


rows = {}   # beginning of the first cell -> list of value cells


def bench_rows():
    s, r, l3, l7 = J("step16"), J("radar64_nadir"), J("landsat36"), J("landsat119_7dir")
    rows["step cloud 32×1×16, 1e8 photons, flux (configs[1], the headline): photons/s"] = [e(s["value"])]
    rows["… kernel time per launch, HIP events, ms"] = [["%.2f" % s["roofline"]["kernel_ms_avg"]], ["%.1f" % r["roofline"]["kernel_ms_avg"]]]
    rows["… rocprofv3 kernel average over 6 calls, ns"] = [[e(kernel_avg_ns("step16"))], [e(kernel_avg_ns("radar64_nadir"))]]
    rows["… `roofline.frac` (kernel's own work = the reference's)"] = ["%.3f" % s["roofline"]["frac"]]
    rows["… `step16`: vector instructions per photon; lane occupancy"] = ["%.1f; %.3f" % (pmc["step16"]["valu_instr_per_photon"], pmc["step16"]["lane_occupancy"])]
    rows["… `step16`: HBM bytes per photon (2·FETCH + WRITE)"] = ["%.3f" % pmc["step16"]["hbm_bytes_per_photon"]]
    rows["… CPU port on the box's 16 cores, photons/s"] = [e(s["cpu_baseline"]["value"])]
    rows["radar-64 64×64×54 + nadir radiance, 1e8 photons (configs[2]): photons/s"] = [e(r["value"])]
    rows["… `roofline.frac`; with the reference algorithm's work |"] = [["%.3f; %.3f" % (r["roofline"]["frac"], r["roofline"]["reference_equivalent"]["frac"])],
                                                                     ["%.3f; %.3f" % (l7["roofline"]["frac"], l7["roofline"]["reference_equivalent"]["frac"])]]
    rows["… `radar64_nadir`: vector instructions per photon; scalar; lane occupancy"] = ["%.1f; %.1f; %.3f" % (pmc["radar64_nadir"]["valu_instr_per_photon"], pmc["radar64_nadir"]["salu_instr_per_photon"], pmc["radar64_nadir"]["lane_occupancy"])]
    rows["Landsat 128×128×36 flux, 1.25e8 photons (configs[3] per GPU"] = [e(l3["value"])]
    rows["… `roofline.frac`; vector instructions per photon; lane occupancy"] = ["%.3f; %.1f; %.3f" % (l3["roofline"]["frac"], l3["roofline"]["issue"]["valu_instr_per_photon"], l3["roofline"]["issue"]["lane_occupancy"])]
    rows["Landsat 128×128×119 + 7 directions + surface, 1.25e8 photons (configs[4] per GPU"] = [e(l7["value"])]
    rows["… `landsat119_7dir`: vector instructions per photon; lane occupancy; HBM bytes per photon"] = ["%.0f; %.3f; %.1f" % (pmc["landsat119_7dir"]["valu_instr_per_photon"], pmc["landsat119_7dir"]["lane_occupancy"], pmc["landsat119_7dir"]["hbm_bytes_per_photon"])]


def config_rows():
    c = lambda label: e(grab("config_bench.txt", re.escape(label) + r": ([\d.e+]+) photons/s"))
    rows["step cloud 32×1×32 (the reference generator's shape)"] = [c("step cloud 32x1x32 flux")]
    rows["radar 640×1×54 flux, 1e8: photons/s"] = [c("radar 640x1x54 flux")]
    rows["radar 640×1×54 + nadir, 5e7: photons/s"] = [c("radar 640x1x54 flux + nadir radiance (RR, zeta 0.3)")]
    rows["radar-64 + nadir, 5e7: photons/s"] = [c("radar-64 64x64x54 flux + nadir radiance")]
    rows["Landsat 128×128×119 flux, 1e8"] = [c("landsat 128x128x119 flux mu0=1")]
    rows["Landsat 128×128×36 flux, 1e8: photons/s"] = [c("landsat 128x128x36 flux mu0=1")]
    rows["Landsat-119 + 7 directions + surface, 2e7: photons/s"] = [c("landsat 128x128x119 + 7 radiances + Lambertian 0.2 (BRDF object), mu0=.5")]


def phase_rows():
    t = text("phase_profile.txt")
    blocks = dict((m.group(1), m.group(2)) for m in re.finditer(r"^(\w+): .*?\n((?:  .*\n)+)", t, re.M))
    names = {"event": "event", "photon steps": "step(photon)", "ray steps": "step(ray)", "service": "service", "expand": "expand"}
    def ph(w, kind):
        m = re.search(re.escape(names[kind]) + r"\s+([\d.]+) % of wave time\s+([\d.]+) phases/photon\s+\d+ cycles/phase\s+([\d.]+) lanes/phase", blocks[w])
        return "%s; %s; %s" % m.groups()
    def logic(w):
        return re.search(r"\(loop logic\)\s+([\d.]+) %", blocks[w]).group(1)
    for w, label in (("step16", "step16"), ("radar64_nadir", "radar64_nadir"), ("landsat36", "landsat36"), ("landsat119_7dir", "landsat119_7dir")):
        for kind in names:
            if names[kind] in blocks[w]:
                for key in (f"{label}: {kind}", f"{label} (column records): {kind}"):
                    rows[key] = [ph(w, kind)]
        rows[f"{label}: loop logic"] = [logic(w)]


def loop_rows():
    ft = text("fused_timing.txt")
    def fused(w, n):
        a = re.search(rf"^{w} \d+ x {n} photons .*fused=1 .*?: ([\d.]+) ms per batch", ft, re.M).group(1)
        b = re.search(rf"^{w} \d+ x {n} photons .*fused=0 .*?: ([\d.]+) ms per batch", ft, re.M).group(1)
        return [a, b]
    rows["step cloud 32×1×16 (1000 batches)"] = fused("step16", r"1e\+06")
    rows["step cloud 32×1×16, 10⁵-photon batches"] = fused("step16", r"1e\+05")
    rows["step cloud 32×1×32 | "] = fused("step32", r"1e\+06")
    rows["radar 640×1×54 | "] = fused("radar640", r"1e\+06")
    rows["Landsat 128×128×36 | "] = fused("landsat36", r"1e\+06")
    rows["Landsat 128×128×119 | "] = fused("landsat119", r"1e\+06")
    rows["**radar-64 + nadir radiance** (round 3"] = fused("radar64_nadir", r"1e\+06")
    rows["**Landsat-119 + 7 directions + surface** | "] = fused("landsat119_7dir", r"1e\+06")
    mt = text("moments_timing.txt")
    def mom(w):
        m = re.search(rf"^{w} .*?batch moments on the device ([\d.]+) ms per batch .*?per-batch blocks to the host ([\d.]+) ms per batch", mt, re.M)
        return [m.group(1), m.group(2)]
    rows["Landsat 128×128×36 (5.1 MB of tallies per batch"] = mom("landsat36")
    rows["Landsat 128×128×119 (15.6 MB per batch)"] = mom("landsat119")
    rows["step cloud 32×1×16 (5 KB per batch"] = mom("step16")
    rows["radar-64 + nadir | "] = [mom("radar64_nadir"), None]
    rows["Landsat-119 + 7 directions + surface | "] = mom("landsat119_7dir")
    lt = text("lookahead_timing.txt")
    def look(w):
        a = re.search(rf"^{w} .*look-ahead 3: .*steady state ([\d.]+) ms per batch", lt, re.M).group(1)
        b = re.search(rf"^{w} .*look-ahead 0: .*steady state ([\d.]+) ms per batch", lt, re.M).group(1)
        return [a, b]
    rows["step cloud 32×1×16 | "] = look("step16")
    rows["step cloud 32×1×32 | 0"] = None
    rows["Landsat 128×128×36 (guessed loops"] = look("landsat36")
    rows["radar-64 + nadir | "][1] = look("radar64_nadir")
    rows["__step32_look"] = look("step32")
    st = text("strong_scaling_proxy.txt")
    val = lambda k: e(float(re.search(rf"{k} step\(s\) in flight: ([\d.e+]+) photons/s", st).group(1)))
    rows["one step in flight"] = [val(1)]
    rows["two | "] = [val(2)]
    rows["three (the default for N > 1)"] = [val(3)]
    rows["the whole batch of 1e8 photons, one step in flight"] = [e(float(re.search(r"whole batch of 1e8 photons x 5 steps, 1 step in flight: ([\d.e+]+) photons/s", st).group(1)))]


def general_kernel_rows():
    """profiles/<tag>_general_kernels.txt: one launch each (the last of three) of the general-class workloads on the kernels of HEAD, on the
    general kernels and on the bricks"""
    lines = [l for l in text("general_kernels.txt").split("\n") if "photons/s" in l and "kernel photon_kernel" in l]
    def rate(workload, *needles, absent=()):
        for l in lines:
            if l.startswith(workload + ":") and all(n in l for n in needles) and not any(a in l for a in absent):
                return e(float(l.split(":")[1].split("photons/s")[0]))
        raise SystemExit(f"general_kernels.txt: no line for {workload} {needles}")
    rows["Landsat-119 + gas, flux, 5e7 photons: general flux kernel on records over a base profile (HEAD)"] = [rate("landsat119_gas", "GRID_COLBASE")]
    rows["… on the bricks (the round-4 place of such a field)"] = [rate("landsat119_gas", "GRID_BRICKS")]
    rows["Landsat-119 + gas + 7 directions, 1e7: widened-class kernel, records over a base profile (HEAD)"] = [rate("landsat119_gas_7dir", "wide")]
    rows["… general radiance kernel, the same records"] = [rate("landsat119_gas_7dir", "true, true, GRID_COLBASE")]
    rows["… general radiance kernel on the bricks (round 4)"] = [rate("landsat119_gas_7dir", "GRID_BRICKS")]
    rows["Landsat-119 + 7 directions, irregular x / y grid, 1e7: widened-class kernel (HEAD)"] = [rate("landsat119_irregular_7dir", "wide")]
    rows["… general radiance kernel (round 4)"] = [[rate("landsat119_irregular_7dir", "true, true")], [rate("landsat119_brdfgrid_7dir", "true, true")]]
    rows["Landsat-119 + 7 directions, gridded surface, 1e7: widened-class kernel (HEAD)"] = [rate("landsat119_brdfgrid_7dir", "wide")]
    rows["for comparison, the common class on the same scene: Landsat-119 flux, 5e7"] = [rate("landsat119", "table in LDS")]
    rows["… Landsat-119 + 7 directions + uniform surface, 1e7"] = [rate("landsat119_7dir", "GRID_COLUMNS>")]


def fused_wide_rows():
    """profiles/<tag>_fused_wide.txt: the batch loops of the workloads beyond the BASELINE four, fused launches against one launch per batch"""
    ft = text("fused_wide.txt")
    def pair(w):
        a = re.search(rf"^{w} \d+ x 1e\+06 photons .*fused=1 .*?: ([\d.]+) ms per batch", ft, re.M).group(1)
        b = re.search(rf"^{w} \d+ x 1e\+06 photons .*fused=0 .*?: ([\d.]+) ms per batch", ft, re.M).group(1)
        return [a, b]
    for label, w in (("Landsat 128×128×36 + gas (two components), flux", "landsat36_gas"), ("Landsat 128×128×119 + gas, flux", "landsat119_gas"),
                     ("Landsat-119 + gas + 7 directions + surface", "landsat119_gas_7dir"), ("Landsat-119 + 7 directions, irregular x / y grid", "landsat119_irregular_7dir"),
                     ("Landsat-119 + 7 directions, gridded surface", "landsat119_brdfgrid_7dir"), ("Landsat-36 + aerosol layer + gas (three components), flux", "landsat36_aerosol_gas"),
                     ("LES stratocumulus + Rayleigh gas (the tool chain's domain), flux", "les_stcu_rayleigh"), ("step cloud 32×1×16, ω = 0.99", "step16_absorbing"),
                     ("Landsat 128×128×36, ω = 0.99", "landsat36_absorbing")):
        rows[label + " | "] = pair(w)


bench_rows(); config_rows(); phase_rows(); loop_rows(); general_kernel_rows(); fused_wide_rows()
lines = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")
# the Fortran drivers end to end: two runs in the file; the row quotes both and carries the smaller one as its value
dt = text("driver_timing.txt")
own = [int(x) for x in re.findall(r"i3rcDriver 1000 batches x 1e6 photons: (\d+) ms wall", dt)]
ref = [int(x) for x in re.findall(r"reference driver \(unchanged\) 1000 batches: (\d+) ms wall", dt)]
steady = re.findall(r"i3rcDriver ([\d.]+) ms per batch .*?unchanged reference driver ([\d.]+) ms per batch", dt)
drv = {
    "| unchanged reference driver, 1000 batches, wall ms": "| unchanged reference driver, 1000 batches, wall ms (the two runs: %s) | %d | `profiles/%s_driver_timing.txt` |" % (", ".join(map(str, ref)), min(ref), TAG),
    "| `i3rcDriver` (device moments), 1000 batches, wall ms": "| `i3rcDriver` (device moments), 1000 batches, wall ms (%s) | %d | `profiles/%s_driver_timing.txt` |" % (", ".join(map(str, own)), min(own), TAG),
    "| steady state of the batch loop (1000 − 10 batches): unchanged reference driver": "| steady state of the batch loop (1000 − 10 batches): unchanged reference driver, ms per batch (%s) | %s | `profiles/%s_driver_timing.txt` |" % (", ".join(x[1] for x in steady), min(x[1] for x in steady), TAG),
    "| … `i3rcDriver`, ms per batch": "| … `i3rcDriver`, ms per batch (%s; process start and file output are in these figures, which scatter by 0.05 from run to run) | %s | `profiles/%s_driver_timing.txt` |" % (", ".join(x[0] for x in steady), min(x[0] for x in steady), TAG),
}
for i, line in enumerate(lines):
    for k, v in drv.items():
        if line.startswith(k) and line != v:
            print(line, "\n ->", v); lines[i] = v
seen = {}
changed = 0
for i, line in enumerate(lines):
    if not line.startswith("| ") or "`profiles/" not in line:
        continue
    cells = [c.strip() for c in line.strip().strip("|").split("|")]
    for key, vals in rows.items():
        if vals is None or key.startswith("__"):
            continue
        k = key.rstrip()
        probe = "| " + k if not k.endswith("|") else "| " + k
        if not (line.startswith("| " + key.rstrip(" |")) and (not key.endswith("| ") or cells[0] == key.rstrip(" |").strip())):
            continue
        n = seen.get(key, 0); seen[key] = n + 1
        v = vals
        if v and isinstance(v[0], list):          # the same label occurs several times: one entry per occurrence
            if n >= len(v) or v[n] is None:
                continue
            v = v[n]
        if cells[0].startswith("step cloud 32×1×32") and "lookahead" in cells[-1]:
            v = rows["__step32_look"]
        if len(v) != len(cells) - 2:
            continue
        new = "| " + " | ".join([cells[0]] + list(v) + [cells[-1]]) + " |"
        if new != line:
            print(line, "\n ->", new); lines[i] = new; changed += 1
        break
# Rows of the per-workload tables beyond the BASELINE four (general class, absorbing cases, tool-chain domain): found by the FILE they
# cite and the kind of their first cell -- "...: photons/s" of <w>_bench.json, "… kernel time per launch, HIP events, ms; `roofline.frac`",
# "… rocprofv3 kernel average, ns" of <w>_kernel_stats.csv, "… `<w>`: vector instructions per photon; lane occupancy; HBM bytes per photon"
BASE4 = ("step16", "radar64_nadir", "landsat36", "landsat119_7dir")
for i, line in enumerate(lines):
    if not line.startswith("| ") or f"`profiles/{TAG}_" not in line:
        continue
    cells = [c.strip() for c in line.strip().strip("|").split("|")]
    if len(cells) != 3:
        continue
    m = re.match(rf"`profiles/{TAG}_(\w+?)_(bench\.json|kernel_stats\.csv)`", cells[2])
    new = None
    if m and m.group(1) not in BASE4 and os.path.exists(P(f"{m.group(1)}_bench.json")):
        w = m.group(1)
        if m.group(2) == "bench.json" and cells[0].endswith("photons/s"):
            new = e(J(w)["value"])
        elif m.group(2) == "bench.json" and cells[0].startswith("… kernel time per launch, HIP events, ms; `roofline.frac`"):
            new = "%.1f; %.3f" % (J(w)["roofline"]["kernel_ms_avg"], J(w)["roofline"]["frac"])
        elif m.group(2) == "kernel_stats.csv" and cells[0].startswith("… rocprofv3 kernel average, ns"):
            new = e(kernel_avg_ns(w))
    mp = re.match(r"… `(\w+)`: vector instructions per photon; lane occupancy; HBM bytes per photon$", cells[0])
    if mp and mp.group(1) not in BASE4 and mp.group(1) in pmc:
        q = pmc[mp.group(1)]
        new = "%.0f; %.3f; %.1f" % (q["valu_instr_per_photon"], q["lane_occupancy"], q["hbm_bytes_per_photon"])
    if new is not None and new != cells[1]:
        out = "| " + " | ".join([cells[0], new, cells[2]]) + " |"
        print(line, "\n ->", out); lines[i] = out; changed += 1
open(os.path.join(ROOT, "DESIGN.md"), "w").write("\n".join(lines))
print(changed, "rows changed")
