#!/usr/bin/env python3
"""bench.py -- photons/s of the I3RC photon-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N --steps K --warmup W]        N=1 directly;
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   one rank per GPU (RCCL)

A "step" is one pass of the hot path (computeRadiativeTransfer :262-398 -> computeRT :400-707) over one batch
of synthetic input: the I3RC step cloud (BASELINE.json configs[1]: 32x1x16, HG g=0.85, omega=1, mu0=1, flux
up/down) with --photons photons per GPU per step (default 1e8 = the quoted photon count).  Weak scaling: every
rank traces its own --photons photons (disjoint Philox counter ranges of the same (seed, batch) key), then
the packed float64 tally buffer is summed across ranks with ONE all-reduce (RCCL over xGMI) -- the exchange
that replaces Code/multipleProcesses_mpi.f95:57-131.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_photon(c, n_dir=0, absorbing=False):
    """bytes/photon = 16 S + 20 K + 16 K [omega<1] + 8 E + 24 K D, from the kernel's own
    work counters (S cell steps incl. shadow rays, K scatterings, E boundary tallies, D directions)."""
    n = c["photons"]
    S = (c["cellSteps"] + c["shadowSteps"]) / n
    K = c["scatterings"] / n
    E = (c["exitsTop"] + c["surfaceHits"]) / n
    return 16 * S + 20 * K + (16 * K if absorbing else 0) + 8 * E + 24 * K * n_dir, dict(S=S, K=K, E=E, D=n_dir)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--photons", type=int, default=100_000_000, help="photons per GPU per step")
    ap.add_argument("--nlayers", type=int, default=16, help="16 = BASELINE.json label, 32 = reference generator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    n_gpus = world

    # CPU baseline first (rank 0, N=1 only), in a child process, before this process touches the GPU.
    cpu_baseline = None
    if n_gpus == 1 and not a.no_cpu_baseline:
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--nlayers", str(a.nlayers)],
                                 capture_output=True, text=True, timeout=600, check=True).stdout.strip().splitlines()[-1]
            cpu_baseline = json.loads(out)
            cpu_baseline.pop("meanFluxUp", None)
        except Exception as e:  # the baseline is reported, not required
            cpu_baseline = {"value": None, "unit": "photons/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    import numpy as np
    import torch

    import i3rc_monte_carlo_model_amd as M
    from i3rc_monte_carlo_model_amd import binding as B
    from i3rc_monte_carlo_model_amd.multigpu import all_reduce_tallies, max_over_ranks
    from tests import cases

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the integrator has no CPU fallback")
    # Rehearsal mode for a one-GPU box (never used by the driver): I3RC_BENCH_REHEARSAL=1 lets all ranks share GPU 0
    # and reduces the tallies through gloo on host copies, so the N>1 control flow can be exercised without RCCL.
    rehearsal = os.environ.get("I3RC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if n_gpus > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- problem: resident on the device before timing -------------------------------------------------
    d = cases.step_cloud(nlayers=a.nlayers)
    table = M.PhaseFunctionTable([M.henyey_greenstein(0.85, 64)])
    dom = M.new_Domain(d["xe"], d["ye"], d["ze"])
    dom.addOpticalComponent("cloud: non-absorbing", d["ext"], d["ssa"], d["pf"], table)
    integ = M.new_Integrator(dom, device=local_rank)
    integ.specifyParameters(surfaceAlbedo=0.0, minInverseTableSize=10001, useRayTracing=True, useRussianRoulette=True)
    lay = integ.layout()
    tally = torch.zeros(lay.total, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    lib = B.load()
    assert lib.i3rc_hip_bind_tally_buffer(integ._h, tally.data_ptr(), tally.numel() * 8) == 0
    assert lib.i3rc_hip_set_stream(integ._h, stream.cuda_stream) == 0

    iseed = 10

    def step(batch):
        # computeRadiativeTransfer zeroes its tallies per call (:296-309); rank r owns photons [r*n, (r+1)*n)
        tally.zero_()
        integ.launch(M.new_RandomNumberSequence((iseed, batch)), M.new_PhotonStream(1.0, 0.0, a.photons),
                     firstPhoton=rank * a.photons, zero=False)
        if rehearsal and dist is not None:
            host = tally.cpu()
            all_reduce_tallies(host, dist)
            tally.copy_(host)
        else:
            all_reduce_tallies(tally, dist)  # the single exchange step: sum of tallies over GPUs (RCCL)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        step(1000 + w)
    sync()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(1 + k)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, device="cpu" if rehearsal else "cuda")

    # ---- kernel durations of the K timed launches (HIP events recorded on the launch stream, read now) ------
    kernel_ms = integ.kernel_ms_history(min(a.steps, 64))
    # ---- results of the last step (already all-reduced across ranks) ------------------------------------
    raw = tally.cpu().numpy()
    res = integ.finish(raw)
    counters = res["counters"]
    # the last step's tallies must be those of exactly one step: every photon leaves through the top, reaches the
    # (black) surface or is dropped by the tracer -- a wrong count or a lost / doubled tally shows here
    closure = float(res["fluxUp"].mean() + res["fluxDown"].mean()) + counters["dropped"] / counters["photons"]
    if counters["photons"] != float(a.photons) * n_gpus or abs(closure - 1.0) > 1e-4:
        raise SystemExit(f"bench: inconsistent results (photons {counters['photons']:.0f}, energy closure {closure:.6f})")
    local = {k: v / n_gpus for k, v in counters.items()}  # identical work per rank (weak scaling)
    avg_ms = float(np.mean(kernel_ms))
    bpp, skd = algorithmic_bytes_per_photon(local)
    achieved = bpp * a.photons / (avg_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"step_cloud_{a.nlayers}"
            if key in tj and tj[key].get("photons"):
                traffic = tj[key]["hbm_bytes_per_launch"] * (a.photons / tj[key]["photons"])
        except Exception:
            traffic = None

    if rank == 0:
        total_photons = float(a.photons) * n_gpus * a.steps
        value = total_photons / elapsed
        line = {
            "metric": "photons/s (whole node) + achieved HBM GB/s, I3RC step-cloud 1e8 photons",
            "value": value,
            "unit": "photons/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"i3rcStepCloud 32x1x{a.nlayers} (HG g=0.85 64 moments, omega=1, mu0=1, albedo 0), "
                                   f"{a.photons:.3g} photons per GPU per step, flux up/down",
                       "photons_per_gpu_per_step": a.photons, "parallelism": f"photon batches sharded over {n_gpus} GPU(s), "
                       "one RCCL all-reduce of the float64 tally buffer per step" if n_gpus > 1 else "single GPU",
                       "rng": "Philox4x32-10 per photon, key (iseed=10, batch)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "photon_kernel<PhiloxStream, false, false, GRID_LDS>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_photon": bpp, "per_photon": skd,
                         "note": "working set is LDS/L2 resident: the path is latency/issue bound, not HBM bound (DESIGN.md)"},
            "cpu_baseline": cpu_baseline,
            "result_check": {"meanFluxUp": float(res["fluxUp"].mean()), "meanFluxDown": float(res["fluxDown"].mean()),
                             "dropped_fraction": counters["dropped"] / counters["photons"]},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    integ.finalize_Integrator()


if __name__ == "__main__":
    main()
