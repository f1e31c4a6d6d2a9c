"""Build-time audit of the gfx950 code (no GPU needed: hipcc cross-compiles).

ROCm 7.2's LLVM lowers the negation of a wave-uniform run-time flag that is tested in divergent code to a LANE MASK
computed by VALU under the current exec mask (v_cndmask_b32 v, 0, 1, flag ; v_cmp_ne_u32 mask, 1, v): lanes that
are inactive there get a 0 bit.  That is harmless while every use sits under a subset of that exec mask.  It is
wrong when the pair is emitted inside an inner loop (whose exec mask shrinks as lanes leave it) and the mask is used
after the loop: this happened to the nested local estimate when its two shadow-ray legs were two loops and the place
of the extinction grid a run-time flag -- lanes that had left the first loop early took the LDS branch in the second
one and read zeros (tests/test_gpu_features.py replay cases on the radar / Landsat fields).  The kernels are written
so that the pattern does not arise in inner loops (grid place as a template parameter, one loop for all legs); this
test keeps it that way."""
import os
import re
import subprocess

import pytest

import i3rc_monte_carlo_model_amd as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    return _device_asm(tmp_path_factory.mktemp("isa"))


def _device_asm(tmp_path):
    out = tmp_path / "i3rc_hip.s"
    flags = [f for f in M.build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [M.build.hipcc()] + flags + ["--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"), "-o", str(out),
                                       os.path.join(M.build.CSRC, "i3rc_hip.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


def test_no_exec_dependent_flag_masks_in_inner_loops(device_asm):
    lines = device_asm
    kernel, depth, found, kernels = None, 0, [], 0
    select = re.compile(r"v_cndmask_b32_e64 (v\d+), 0, 1, s\[\d+:\d+\]")
    for i, line in enumerate(lines):
        m = re.match(r"^(_ZN4i3rc\w+):", line)
        if m:
            kernel, depth = m.group(1), 0
            kernels += 1
        m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", line)
        if m:   # a basic block: its loop depth is in the comment on the label (".LBBn_m: ; in Loop: Header=BBn_k
            # Depth=d"; a loop header lists its parents first and itself last, over several comment lines)
            depths, j = [int(x) for x in re.findall(r"Depth=(\d+)", line)], i + 1
            while j < len(lines) and lines[j].lstrip().startswith(";"):
                depths += [int(x) for x in re.findall(r"Depth=(\d+)", lines[j])]
                j += 1
            depth = max(depths) if depths else 0
        m = select.search(line)
        if m and kernel:
            for nxt in lines[i + 1:i + 3]:
                if re.search(r"v_cmp_ne_u32_e64 s\[\d+:\d+\], 1, " + m.group(1) + r"\b", nxt) and depth >= 2:
                    found.append((kernel[:70], i + 1, depth))
    assert kernels >= 20, kernels          # the scan saw the kernels at all
    assert not found, found


def test_production_kernels_keep_their_state_in_registers(device_asm):
    """No production instantiation of photon_kernel may spill vector registers or use scratch memory (the code object's
    own resource figures, hipcc -Rpass-analysis=kernel-resource-usage): round 1's general radiance kernel carried
    26-29 spilled vector registers and 116 bytes of scratch per lane.  Scalar-register spills (held in vector-register
    lanes, no memory traffic) are bounded: the specialised kernels -- every BASELINE configuration runs one of them --
    may have a handful.  (The specialised flux kernels on LDS / linear grids are compiled for eight waves per SIMD, whose
    scalar-register budget costs them ~10 spills and still gains 4 %: kernels.hpp, I3RC_FLUX_WAVES; the bricked one none.)"""
    import sys

    sys.path.insert(0, ROOT)
    from tools.kernel_resources import resources

    everything = resources()

    def scratch_traffic(mangled):
        body, inside = [], False
        for line in device_asm:
            if line.startswith(mangled + ":"):
                inside = True
            elif inside and line.startswith(".Lfunc_end"):
                break
            elif inside:
                body.append(line.split(";")[0])
        assert len(body) > 1000, mangled
        return any("scratch_" in b for b in body)
    # every PhiloxStream instantiation, the table-in-LDS ones among them: BENCH's headline (step cloud) and Landsat-36 run those
    rows = [r for r in everything if r["name"].startswith("photon_kernel<PhiloxStream")]
    # (round 4: ... and the radiance kernels without an event ring, "one direction")
    # (... and one of each for fields with column records, GRID_COLUMNS)
    # (round 5: ... and the common class with several components, "wide": flux, ring, one direction x four places)
    # (... the radiance kernels for several components -- ring and one direction -- x five places, and for column records over a base
    # profile, GRID_COLBASE, the three general kernels on top of those)
    assert len(rows) == 28 + 10 + 3 and sum("table in LDS" in r["name"] for r in rows) == 4 and sum("one direction" in r["name"] for r in rows) == 8 + 5 + 1 and \
        sum("wide" in r["name"] for r in rows) == 10 and sum("GRID_COLBASE" in r["name"] for r in rows) == 5, [r["name"] for r in rows]
    fused = [r for r in everything if r["name"].startswith("photon_kernel<PhiloxBatchStream")]   # the fused multi-batch kernels: seven flux, eight radiance (round 4)
    # (round 5: ... and fifteen for the widened class -- flux, ring, one direction x five places)
    assert len(fused) == 30 and sum(r["name"].startswith("photon_kernel<PhiloxBatchStream, true") for r in fused) == 18 and sum("wide" in r["name"] for r in fused) == 15, [r["name"] for r in fused]
    for r in rows + fused:
        if r in fused and "table in LDS" in r["name"]:
            # the fused kernels want 66 vector registers; their table-in-LDS instantiations (1024 threads, two workgroups per CU: eight
            # waves per SIMD, 64 registers) keep two of them in scratch -- measured faster all the same (kernels.hpp, i3rc_hip.hip)
            assert r["VGPRs Spill"] <= 2 and r["ScratchSize [bytes/lane]"] <= 16, (r["name"], r["VGPRs Spill"], r["ScratchSize [bytes/lane]"])
            continue
        # (a private segment that no instruction of the kernel touches -- 20 bytes in the general radiance kernels and three fused ones since the
        # LDS carve-up grew a field: slots the register allocator reserved and then did not need -- is no scratch TRAFFIC: the assembly says)
        assert r["VGPRs Spill"] == 0 and (r["ScratchSize [bytes/lane]"] == 0 or not scratch_traffic(r["mangled"])), (r["name"], r["VGPRs Spill"], r["ScratchSize [bytes/lane]"])
        if ", false, GRID" in r["name"]:                       # specialised (GENERAL = false)
            limit = 16
            # (the bricked flux kernel reads its clear-air map through two more scalar values: a couple of spills)
            if r["name"].startswith("photon_kernel<PhiloxStream, false"): limit = 4 if "GRID_BRICKS" in r["name"] else 12
            # (1024-thread workgroups with the inverse table's cosines in LDS: the table's LDS address and length are two more scalar values)
            if r["name"].startswith("photon_kernel<PhiloxStream, false") and "table in LDS" in r["name"] and "GRID_BRICKS" not in r["name"]: limit = 14
            # (column records are read through a buffer descriptor: four scalar registers where a pointer is two)
            if r["name"].startswith("photon_kernel<PhiloxStream, false") and "GRID_COLUMNS" in r["name"]: limit = 15
            # (the bricked radiance kernels -- off the BASELINE path since the Landsat scene is read from column records -- carry the brick
            # geometry in scalar registers on top of everything else; the uniform test for direction cosines of zero, round 4, took three more)
            if r["name"].startswith("photon_kernel<PhiloxStream, true") and "GRID_BRICKS" in r["name"]: limit = 20
            if r["name"].startswith("photon_kernel<PhiloxBatchStream"): limit = 12 if "table in LDS" in r["name"] else (6 if r["name"].startswith("photon_kernel<PhiloxBatchStream, true") else 4)
            # (the widened class -- several components, irregular x / y, gridded surface behind run-time switches -- at four waves per SIMD)
            if "wide" in r["name"]: limit = 26 if "GRID_BRICKS" in r["name"] else 24   # (the bricked form, with the brick geometry on top: one more since the volume tallies got their LDS flag)
            assert r["SGPRs Spill"] <= limit, (r["name"], r["SGPRs Spill"])
    # the replay build (test infrastructure on the device) must not use scratch either
    for r in everything:
        if r["name"].startswith("photon_kernel<ReplayStream"):
            assert r["VGPRs Spill"] == 0 and r["ScratchSize [bytes/lane]"] == 0, r["name"]
