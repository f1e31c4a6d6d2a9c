#!/usr/bin/env python3
"""bench.py -- photons/s of the I3RC photon-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--scaling weak|strong]

`--gpus N` measures N GPUs BY ITSELF: called without a launcher (RANK unset) this process never touches the GPU; it
starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, one per
GPU), relays rank 0's JSON line and returns the ranks' exit code.  N = 1 goes through the same path.  Under an
external launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) the process is a rank
already; WORLD_SIZE must then equal --gpus, else the run aborts (exit 2).

A "step" is one pass of the hot path (computeRadiativeTransfer :262-398 -> computeRT :400-707) over one batch of
synthetic input.  Default workload = BASELINE.json configs[1], the I3RC step cloud (32x1x16, HG g=0.85, omega=1, mu0=1,
flux up/down) with 1e8 photons per GPU per step; `--config` selects the other BASELINE configurations
(tools/workloads.py: radar64_nadir, landsat36, landsat119_7dir, ...).  Weak scaling (default): every rank traces its
own --photons photons (disjoint Philox counter ranges of the same (seed, batch) key).  `--scaling strong`: ONE batch of
--photons photons is sharded over the ranks by photon range (multigpu.shard_photons).  Either way the packed float64
tally buffer is then summed across ranks with ONE all-reduce (RCCL over xGMI) -- the exchange that replaces the ten
MPI_REDUCE calls of Code/multipleProcesses_mpi.f95:57-131 (Example-Drivers/monteCarloDriver.f95:333-352).
Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (DESIGN.md section 6 explains every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Vector-instruction issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD once >= 2 waves
# share it (guide: "v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4"), at the 2.4 GHz peak clock.  The kernels
# run 5 waves per SIMD.  tools/microbench/issue_rate.hip measures the actual cost per instruction kind on the box
# (profiles/*_issue_rate_microbench.txt: 2.5 cycles for the simplest class, 3.6-4.9 for most, 8.2 for transcendentals).
N_SIMD = 256 * 4
ISSUE_PEAK = N_SIMD * 2.4e9 / 2.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="step16", help="workload (tools/workloads.py); default = BASELINE.json configs[1]")
    ap.add_argument("--photons", type=int, default=0, help="photons per GPU per step (weak) / per step (strong); 0 = the workload's")
    ap.add_argument("--nlayers", type=int, default=16, help="step cloud: 16 = BASELINE.json label, 32 = reference generator (= --config step32)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="default: N = 1 -> the same thing; N > 1 -> strong (ONE batch of --photons photons per step sharded over the GPUs: the "
                         "metric's case) as the headline, the weak figure measured after it and reported beside it")
    ap.add_argument("--overlap", choices=("auto", "0", "1", "2", "3", "4"), default="auto",
                    help="k: k + 1 steps in flight (as many handles / streams / tally buffers); auto = three steps in flight for N > 1, else one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contact-timeout", type=float, default=float(os.environ.get("I3RC_BENCH_CONTACT_TIMEOUT", "300")),
                    help="N > 1: seconds every rank has for init_process_group and a first all-reduce before the run is called off (naming the rank)")
    ap.add_argument("--collective-timeout", type=float, default=float(os.environ.get("I3RC_BENCH_COLLECTIVE_TIMEOUT", "600")),
                    help="N > 1: the process group's timeout, which bounds every collective of the run (torch's default for NCCL: 600 s)")
    ap.add_argument("--process-group", action="store_true", default=os.environ.get("I3RC_BENCH_PROCESS_GROUP") == "1",
                    help="N = 1: initialise torch.distributed all the same (a one-rank world over RCCL) and run the N > 1 code paths through it")
    ap.add_argument("--contact-only", action="store_true", help="N > 1: stop after the first-contact check (process group up, one all-reduce on every rank)")
    ap.add_argument("--scale-photons", type=float, default=1.0,
                    help="test knob: scales the workload's photon counts (per GPU and per node), so that the default N > 1 mode -- configs 3 / 4 "
                         "shard the node's batch of 1e9 photons -- can be rehearsed at a small size")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if a.config == "step16" and a.nlayers == 32:
        a.config = "step32"
    return a


def run_cpu_baseline(a):
    """The oracle (kind "port") on the host cores, in a child process that never touches the GPU."""
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--config", a.config],
                             capture_output=True, text=True, timeout=900, check=True).stdout.strip().splitlines()[-1]
        cb = json.loads(out)
        cb.pop("meanFluxUp", None)
        return cb
    except Exception as e:  # the baseline is reported, not required
        return {"value": None, "unit": "photons/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


def wait_for_ranks(procs, poll_s=0.2, grace_s=10.0, contact=None):
    """Wait for all rank processes; when one exits non-zero the others are terminated (then killed) instead of being
    left in a rendezvous or a barrier.  Returns rank 0's stdout / stderr and every exit code.
    contact = (status directory, seconds): every rank must have reported its first contact (process group up, one all-reduce:
    worker's first_contact) within that time, else the run is called off and the silent ranks are named."""
    import threading

    captured = {}
    t_start = time.time()

    def drain():   # rank 0's pipes are read while we poll, so that a long line never blocks the child
        captured["out"], captured["err"] = procs[0].communicate()

    t = threading.Thread(target=drain, daemon=True)
    t.start()
    failed = False
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            failed = True
            break
        if all(c == 0 for c in codes):
            break
        if contact is not None and time.time() - t_start > contact[1]:
            missing = [r for r in range(len(procs)) if not os.path.exists(os.path.join(contact[0], f"rank{r}.contact"))]
            if missing:
                sys.stderr.write(f"bench.py: no first contact from rank(s) {missing} within {contact[1]:.0f} s (process group / first all-reduce): calling the run off\n")
                failed = True
                break
            contact = None
        time.sleep(poll_s)
    if failed:
        t_fail = time.time()   # (ranks that fail for the same reason -- no GPU, a bad argument -- say so themselves within a moment)
        while time.time() - t_fail < 3.0 and any(p.poll() is None for p in procs):
            time.sleep(poll_s)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    t.join(timeout=grace_s)
    return captured.get("out", ""), captured.get("err", ""), [p.returncode for p in procs]


# ------------------------------------------------------------------------------------------------------------------
# launcher: no GPU call, no torch import in this process
# ------------------------------------------------------------------------------------------------------------------
def launcher(a):
    cpu_baseline = None
    if a.gpus == 1 and not a.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(a)   # before any rank exists: the host cores are free
    # Ranks rendezvous on a port found by bind-and-close: somebody else may take it before rank 0 binds it (EADDRINUSE in
    # its stderr) -- then the whole set is started again on a fresh port.  All children are polled: the first one that
    # fails takes the others with it (they would sit in the rendezvous or in a barrier until its timeout).
    import tempfile

    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        status_dir = tempfile.mkdtemp(prefix="i3rc_bench_")   # every rank reports its first contact here (see first_contact)
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), I3RC_BENCH_SPAWNED="1", I3RC_BENCH_STATUS_DIR=status_dir)
            if cpu_baseline is not None and r == 0:   # rank 0 writes the line: it quotes the oracle's work counters in its roofline
                env["I3RC_BENCH_CPU_BASELINE"] = json.dumps(cpu_baseline)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None, text=True))
        # (the deadline allows for a fresh box's first `import torch`: up to two minutes)
        out0, err0, codes = wait_for_ranks(procs, contact=(status_dir, a.contact_timeout + 150.0) if a.gpus > 1 else None)
        sys.stderr.write(err0 or "")
        import shutil

        shutil.rmtree(status_dir, ignore_errors=True)
        if any(codes) and attempt < 2 and ("EADDRINUSE" in (err0 or "") or "Address already in use" in (err0 or "")):
            sys.stderr.write(f"bench.py: port {port} was taken before the ranks met; starting them again on another one\n")
            continue
        break
    if any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}" + "".join(f"; rank {r} failed with code {c}" for r, c in enumerate(codes) if c not in (0, None, -15)) + "\n")
        sys.stdout.write(out0 or "")
        return max(abs(c) for c in codes) or 1
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = json.loads(ln)
        else:
            print(ln)
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    line["launch"] = f"bench.py spawned {a.gpus} rank process(es) itself (one per GPU, 127.0.0.1:{port})"
    print(json.dumps(line))
    return 0


# ------------------------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes_per_photon(c, n_dir=0, absorbing=False):
    """bytes/photon = 16 S + 20 K + 16 K [omega<1] + 8 E + 24 K D, from the kernel's own
    work counters (S cell steps incl. shadow rays, K scatterings, E boundary tallies, D directions)."""
    n = c["photons"]
    S = (c["cellSteps"] + c["shadowSteps"]) / n
    K = c["scatterings"] / n
    E = (c["exitsTop"] + c["surfaceHits"]) / n
    return 16 * S + 20 * K + (16 * K if absorbing else 0) + 8 * E + 24 * K * n_dir, dict(S=S, K=K, E=E, D=n_dir)


def load_pmc(config):
    """Counter figures of this workload's kernel from the committed rocprofv3 PMC passes (profiles/*_pmc.json, written
    by tools/pmc_to_json.py from the summaries next to it): HBM bytes and vector instructions per photon."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith("_pmc.json"):
            try:
                j = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            if config in j:
                best = dict(j[config], source="profiles/" + f)
    return best


def load_calibration(config):
    """Port-against-reference calibration (tools/cpu_calibration.py, run in the build container): the newest
    profiles/*_cpu_calibration.json; the ratio of the nearest shape and the range over all cases."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith("_cpu_calibration.json"):
            try:
                best = (f, json.load(open(os.path.join(pdir, f))))
            except Exception:
                continue
    if best is None:
        return None
    f, j = best
    nearest = {"step16": "step32", "landsat36": "landsat119", "radar64_nadir": "radar640_nadir", "landsat36_7dir": "landsat119_7dir"}.get(config, config)
    rows = [c for c in j["cases"] if c["workload"] == nearest]
    return {"source": "profiles/" + f, "what": j["what"], "host": j["host"], "nearest_case": rows[0] if rows else None,
            "ratio_port_over_reference": (rows[0]["ratio_port_over_reference"] if rows else j["ratio_geomean"]),
            "ratio_range": [j["ratio_min"], j["ratio_max"]], "note": j["note"]}


def first_contact(a, dist, torch, rank, world, local_rank, rehearsal, pg_seconds):
    """First-contact check of an N > 1 run: one small all-reduce (every rank contributes rank + 1) and a barrier, under a watchdog --
    a rank whose collective does not come back within --contact-timeout says so on stderr and leaves with exit code 5 instead
    of sitting in RCCL until somebody kills the job.  Reports to the launcher's status directory."""
    import threading

    done = threading.Event()

    def watchdog():
        if not done.wait(a.contact_timeout):
            sys.stderr.write(f"bench.py: rank {rank} (cuda:{local_rank}): first all-reduce did not complete within {a.contact_timeout:.0f} s "
                             f"(backend {dist.get_backend()}, world {world}): giving up\n")
            sys.stderr.flush()
            os._exit(5)

    threading.Thread(target=watchdog, daemon=True).start()
    t0 = time.perf_counter()
    probe = torch.full((4,), float(rank + 1), dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(probe)
    if not rehearsal:
        torch.cuda.synchronize()
    want = world * (world + 1) / 2.0
    ok = bool((probe.cpu() == want).all())
    dist.barrier()
    done.set()
    sys.stderr.write(f"[bench] rank {rank}/{world} cuda:{local_rank}: process group ({dist.get_backend()}) up in {pg_seconds:.1f} s, "
                     f"first all-reduce {'ok' if ok else 'WRONG: ' + str(probe.tolist())} in {time.perf_counter() - t0:.2f} s\n")
    d = os.environ.get("I3RC_BENCH_STATUS_DIR")
    if d and os.path.isdir(d):
        open(os.path.join(d, f"rank{rank}.contact"), "w").write("ok" if ok else "wrong sum")
    return 0 if ok else 5


def worker(a):
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a number for the wrong GPU count\n")
        return 2
    n_gpus = world
    # launched as a rank by somebody else (torchrun) at N = 1: the CPU baseline still runs first, in a child process,
    # before this process touches the GPU
    cpu_baseline = None
    if os.environ.get("I3RC_BENCH_CPU_BASELINE"):
        cpu_baseline = json.loads(os.environ["I3RC_BENCH_CPU_BASELINE"])   # measured by the launcher before any rank existed
    elif n_gpus == 1 and not a.no_cpu_baseline and not os.environ.get("I3RC_BENCH_SPAWNED"):
        cpu_baseline = run_cpu_baseline(a)

    import numpy as np
    import torch

    from i3rc_monte_carlo_model_amd import binding as B
    from i3rc_monte_carlo_model_amd.multigpu import all_reduce_tallies, max_over_ranks, shard_photons
    import i3rc_monte_carlo_model_amd as M
    from tools import workloads as W

    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs a GPU: the integrator has no CPU fallback\n")
        return 3
    # Rehearsal mode for a one-GPU box (never used by the driver): I3RC_BENCH_REHEARSAL=1 lets all ranks share GPU 0
    # and reduces the tallies through gloo on host copies, so the N>1 control flow can be exercised without RCCL
    # (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("I3RC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} visible\n")
        return 3
    torch.cuda.set_device(local_rank)
    dist = None
    # --process-group (or I3RC_BENCH_PROCESS_GROUP=1) at N = 1: bring the process group up all the same -- a world of ONE rank over RCCL --
    # so that everything an N > 1 run does beside tracing (init_process_group on the device, the first-contact check, the all-reduce of
    # every step, the gathers of the report) executes on the real backend of a one-GPU box: the two-rank RCCL tests need two GPUs
    if n_gpus > 1 or a.process_group:
        import torch.distributed as dist

        import datetime

        t_pg = time.perf_counter()
        # (the process group's timeout bounds EVERY collective of the run, the timed steps' all-reduces included: --collective-timeout,
        # torch's own default of ten minutes unless asked otherwise; first contact has its own, shorter watchdog: --contact-timeout)
        limit = datetime.timedelta(seconds=max(a.collective_timeout, a.contact_timeout))
        if rehearsal:
            dist.init_process_group(backend="gloo", timeout=limit)
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        assert dist.get_world_size() == a.gpus
        rc = first_contact(a, dist, torch, rank, n_gpus, local_rank, rehearsal, time.perf_counter() - t_pg)
        if rc or a.contact_only:
            if not rc and rank == 0:
                print(json.dumps({"contact": "ok", "n_gpus": n_gpus, "backend": dist.get_backend(), "timeout_s": a.contact_timeout}), flush=True)
            dist.destroy_process_group()
            return rc

    # ---- problem: resident on the device before timing -------------------------------------------------
    name, w = W.get(a.config)
    nd = W.n_dir(w)
    lib = B.load()
    # Steps in flight.  One (N = 1): zero, trace, (all-reduce), one after the other on torch's current stream.  Three (N > 1;
    # --overlap 1: two): as many handles, each with a stream and a tally buffer of its own, take the steps in turn, so that the tail
    # of step k -- its last photons keep a few wavefronts busy for a millisecond, a quarter of a 1.25e7-photon shard -- and
    # its all-reduce overlap the trace of step k + 1.  Every step still zeroes, traces and reduces its own buffer.
    overlap = (n_gpus > 1) if a.overlap == "auto" else (a.overlap != "0")
    lanes = []
    for k in range(1 if not overlap else (3 if a.overlap == "auto" else int(a.overlap) + 1)):   # (auto, N > 1: three -- the shard of an 8-GPU run on one GPU: 0.76 / 0.88 / 0.92 of the whole batch's rate with 1 / 2 / 3)
        integ, d = W.make_integrator(w, device=local_rank)
        lay = integ.layout()
        tally = torch.zeros(lay.total, dtype=torch.float64, device="cuda")
        stream = torch.cuda.Stream() if overlap else torch.cuda.current_stream()
        assert lib.i3rc_hip_bind_tally_buffer(integ._h, tally.data_ptr(), tally.numel() * 8) == 0
        assert lib.i3rc_hip_set_stream(integ._h, stream.cuda_stream) == 0
        # the phase-function tables are built and uploaded before the timed region (the reference builds them in its 1-photon
        # warm-up, monteCarloDriver.f95:233-253); no kernel launch here, so that every photon_kernel dispatch a profiler sees
        # is a full step
        integ._ensure_tables()
        lanes.append(dict(integ=integ, tally=tally, stream=stream))
    integ = lanes[0]["integ"]
    absorbing = bool(np.any(d["ssa"][d["ext"] > 0] < 1.0))
    torch.cuda.synchronize()
    iseed = 10

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling, steps, warmup):
        """`steps` timed steps of the workload's batch: weak = every rank its own --photons photons, strong = one batch of
        --photons photons sharded over the ranks by photon range."""
        per_step = a.photons or int(round(w["photons"] * a.scale_photons))
        if scaling == "strong" and n_gpus > 1 and not a.photons:
            per_step = int(round(w.get("photons_node", w["photons"]) * a.scale_photons))   # configs 3 / 4: the batch BASELINE.json quotes for the whole node (1e9 photons)
        if scaling == "strong":
            first, mine = shard_photons(per_step, n_gpus, rank)
            total_per_step = per_step
        else:
            first, mine = rank * per_step, per_step
            total_per_step = per_step * n_gpus

        reduce_events = []   # (N > 1: an event on either side of every timed step's all-reduce, on the step's own stream)

        def step(k, batch, timed=False):
            ln = lanes[k % len(lanes)]
            with torch.cuda.stream(ln["stream"]):
                ln["tally"].zero_()   # computeRadiativeTransfer zeroes its tallies per call (:296-309)
                ln["integ"].launch(M.new_RandomNumberSequence((iseed, batch)), M.new_PhotonStream(w["mu0"], 0.0, mine),
                                   firstPhoton=first, zero=False)
                if timed and dist is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ln["stream"])
                if rehearsal and dist is not None:
                    host = ln["tally"].cpu()
                    all_reduce_tallies(host, dist)
                    ln["tally"].copy_(host)
                else:
                    all_reduce_tallies(ln["tally"], dist)  # the single exchange step: sum of tallies over GPUs (RCCL)
                if timed and dist is not None:
                    e1.record(ln["stream"])
                    reduce_events.append((e0, e1))
            return ln

        for k in range(warmup):
            step(k, 1000 + k)
        sync()
        before = [int(ln["integ"].timed_launches()) for ln in lanes]
        t0 = time.perf_counter()
        last = None
        for k in range(steps):
            last = step(k, 1 + k, timed=True)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0    # this rank's own work, before it waits for the others
        sync()
        elapsed = time.perf_counter() - t0
        elapsed = max_over_ranks(elapsed, dist, device="cpu" if rehearsal else "cuda")
        reduce_ms = [e0.elapsed_time(e1) for e0, e1 in reduce_events]
        # kernel durations of the timed launches (HIP events recorded on the launch streams, read now)
        kernel_ms, n_launches = [], 0
        for ln, b0 in zip(lanes, before):
            nl = int(ln["integ"].timed_launches()) - b0   # a step is one launch unless a batch exceeds the launch limit
            n_launches += nl
            if nl:
                kernel_ms += [float(x) for x in ln["integ"].kernel_ms_history(min(nl, 64))]
        # N > 1: what every rank did on its own, gathered on rank 0 -- so that the first run on a real 8-GPU node explains its own
        # efficiency: a slow rank (max / min of the ranks' own times), the all-reduce (from the end of a rank's trace to the end of the
        # collective: it includes the wait for the slowest rank's trace of that step), the kernels themselves
        ranks = None
        if dist is not None:
            mine_stats = {"rank": rank, "device": f"cuda:{local_rank}", "photons_per_step": mine, "own_elapsed_s": own_elapsed,
                          "kernel_ms_per_step": (float(np.mean(kernel_ms)) * n_launches / steps) if kernel_ms else None,
                          "allreduce_ms_per_step": (float(np.mean(reduce_ms)) if reduce_ms else None),
                          "allreduce_ms_max": (float(np.max(reduce_ms)) if reduce_ms else None)}
            gathered = [None] * n_gpus
            dist.all_gather_object(gathered, mine_stats)
            own = [g["own_elapsed_s"] for g in gathered]
            ranks = {"per_rank": gathered, "own_elapsed_s_max": max(own), "own_elapsed_s_min": min(own), "slowest_rank": int(np.argmax(own)),
                     "imbalance": max(own) / max(min(own), 1e-12),
                     "note": "own_elapsed_s: a rank's wall time for its timed steps up to its own stream synchronisation, before the closing barrier; "
                             "allreduce_ms: HIP events on the step's stream around the collective (with steps in flight it overlaps the next step's trace)"}
        return dict(scaling=scaling, elapsed=elapsed, steps=steps, mine=mine, per_step=per_step, total_per_step=total_per_step,
                    kernel_ms=kernel_ms, launches_per_step=n_launches / steps, last=last, ranks=ranks)

    # N = 1: weak and strong are the same run.  N > 1 without --scaling: the metric's case -- ONE batch of the workload's
    # photons per step, sharded over the GPUs (strong) -- is the headline; the weak figure (every GPU the whole batch) is
    # measured after it and reported beside it.
    headline = a.scaling or ("strong" if n_gpus > 1 else "weak")
    m = measure(headline, a.steps, a.warmup)
    weak = None
    if a.scaling is None and n_gpus > 1:
        mw = measure("weak", a.steps, 1)
        weak = {"value": float(mw["total_per_step"]) * mw["steps"] / mw["elapsed"], "unit": "photons/s", "ms_per_step": mw["elapsed"] / mw["steps"] * 1e3,
                "photons_per_step": mw["total_per_step"], "photons_per_gpu_per_step": mw["mine"], "steps": mw["steps"],
                "note": "weak scaling measured after the headline run: every GPU traces the workload's whole batch per step"}
        m["last"] = None   # (the lanes now hold the weak run's tallies: the result check below uses what it left)
        check = mw
    else:
        check = m
    elapsed, mine, per_step, total_per_step = m["elapsed"], m["mine"], m["per_step"], m["total_per_step"]
    kernel_ms, launches_per_step = m["kernel_ms"], m["launches_per_step"]
    # ---- results of the last step (already all-reduced across ranks) ------------------------------------
    raw = check["last"]["tally"].cpu().numpy()
    res = integ.finish(raw)
    counters = res["counters"]
    # the last step's tallies must be those of exactly one step: every photon leaves through the top, ends at the
    # surface / by roulette or is dropped by the tracer -- a wrong count or a lost / doubled tally shows here
    if counters["photons"] != float(check["total_per_step"]):
        sys.stderr.write(f"bench: inconsistent results (photons {counters['photons']:.0f}, expected {check['total_per_step']})\n")
        return 4
    albedo = w.get("surface", w.get("albedo", 0.0))
    if albedo == 0.0 and "surface_grid" not in w and not absorbing:
        closure = float(res["fluxUp"].mean() + res["fluxDown"].mean()) + counters["dropped"] / counters["photons"]
        if abs(closure - 1.0) > 1e-4:
            sys.stderr.write(f"bench: energy closure {closure:.6f}\n")
            return 4
    devices = [f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"]
    if dist is not None:
        gathered = [None] * n_gpus
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    if rank == 0:
        share = check["mine"] / float(check["total_per_step"])   # this rank's share of the (all-reduced) work counters
        local = {k: v * share / check["mine"] * mine for k, v in counters.items()}   # ... scaled to a headline step's photons
        avg_ms = float(np.mean(kernel_ms)) * launches_per_step     # kernel time per step on this rank
        kbpp, kskd = algorithmic_bytes_per_photon(local, nd, absorbing)        # what this kernel actually did
        # roofline.achieved: SURVEY.md 8(d)'s byte formula evaluated with THE KERNEL'S OWN work counters.  With the local
        # estimate's roulette the kernel leaves out rays that are known to contribute nothing, so it touches fewer bytes than
        # the reference's loop would on the same input; that figure -- the formula with the work per photon of the REFERENCE'S
        # ALGORITHM, counted by the CPU restatement -- is reported beside it as reference_equivalent.  Flux-only kernels do
        # exactly the reference's work: there the two are the same.
        ref, ref_src = None, None
        if cpu_baseline and cpu_baseline.get("oracle_per_photon"):
            ref, ref_src = cpu_baseline["oracle_per_photon"], "CPU restatement (oracle) on this workload, counted in this run's cpu_baseline leg"
        elif name in W.REFERENCE_WORK:
            ref, ref_src = W.REFERENCE_WORK[name], "CPU-restatement-equivalent counters recorded in tools/workloads.py (REFERENCE_WORK)"
        if ref is None or nd == 0:
            ref, ref_src = kskd, "the kernel's own counters (flux only: identical to the reference's algorithm)"
        rbpp = 16 * ref["S"] + 20 * ref["K"] + (16 * ref["K"] if absorbing else 0) + 8 * ref["E"] + 24 * ref["K"] * nd
        achieved = kbpp * mine / (avg_ms * 1e-3) / 1e9
        ref_achieved = rbpp * mine / (avg_ms * 1e-3) / 1e9
        reference_equivalent = {"achieved": ref_achieved, "frac": ref_achieved / HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_photon": rbpp,
                                "per_photon": dict(S=ref["S"], K=ref["K"], E=ref["E"], D=nd), "per_photon_source": ref_src,
                                "note": "what the REFERENCE'S loop would touch on this input in the kernel's time (it traces every local-estimate ray "
                                        "before it plays the ray's roulette); the kernel's own bytes are roofline.achieved"}
        kernel_work = {"per_photon": kskd, "bytes_per_photon": kbpp, "GBps": achieved, "frac": achieved / HBM_PEAK_GBS,
                       "rays_skipped_per_photon": local.get("raysSkipped", 0.0) / mine,
                       "note": "the kernel's own counters (= roofline.achieved): local-estimate rays whose roulette is lost before the trace are not traced"}
        pmc = load_pmc(name)
        traffic = measured = issue = None
        if pmc:
            traffic = pmc["hbm_bytes_per_photon"] * mine
            measured = traffic / (avg_ms * 1e-3) / 1e9
            ipp = pmc["valu_instr_per_photon"]
            rate = ipp * mine / (avg_ms * 1e-3)
            # The counters are a profiled run's, read from a committed file -- not counted in this run.  What this run DID count itself is its
            # work per photon (the kernel's own counters: per_photon above) and its kernel's name: both are held against the profiled
            # launch's, so that a stale or foreign counter file shows in the line instead of passing as this run's.
            prof = pmc.get("work_per_photon")
            stale = []
            if pmc.get("launch") and pmc["launch"] != integ.kernel_name():
                stale.append(f"the counters are of {pmc['launch']}, this run's kernel is {integ.kernel_name()}")
            if prof:
                for key in ("S", "K"):
                    if abs(prof[key] - kskd[key]) > 0.02 * max(kskd[key], 1e-9):
                        stale.append(f"work per photon {key}: counters' run {prof[key]:.1f}, this run {kskd[key]:.1f}")
            counters_from = {"file": pmc["source"], "photons": pmc.get("photons"), "collected_at_commit": pmc.get("collected_at_commit"),
                             "launch": pmc.get("launch"), "work_per_photon": prof, "this_run_work_per_photon": {"S": kskd["S"], "K": kskd["K"]},
                             "matches_this_run": (not stale) if (prof or pmc.get("launch")) else None, "mismatch": stale or None}
            issue = {"valu_instr_per_photon": ipp, "lane_occupancy": pmc["lane_occupancy"], "counters_from": counters_from,
                     "achieved_instr_per_s": rate, "peak": ISSUE_PEAK, "frac": rate / ISSUE_PEAK,
                     "unit": "wave64 VALU instructions/s (whole chip)",
                     "cycles_per_valu_instr_per_simd": N_SIMD * 2.4e9 / rate,
                     "assumption": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (>= 2 waves per SIMD; "
                                   "guide constants table); counters from " + pmc["source"] + f" ({pmc.get('photons', 0):.0f} photons)",
                     "measured_issue_costs": "tools/microbench/issue_rate.hip at 5 waves per SIMD (profiles/*_issue_rate_microbench.txt): "
                                             "v_add_f32 / v_mul_f32 / v_add_u32 / v_and_b32 2.5-2.7 cycles, v_fma_f32 3.6, v_cmp / v_min / "
                                             "v_max3 / v_bfe / v_lshl_add / integer multiplies / f64 / packed f32 4.2-4.9, "
                                             "v_rcp / v_sqrt / v_log 8.2: the 2-cycle figure holds for the simplest class only, "
                                             "this kernel's mix is issued at its own pace",
                     "useful_lane_frac": rate / ISSUE_PEAK * pmc["lane_occupancy"]}
            if "scalar_instr_per_photon" in pmc:   # the scalar unit (one per compute unit) is nearly as busy as the vector units
                spp = pmc["scalar_instr_per_photon"]
                issue["scalar"] = {"instr_per_photon": spp, "per_cycle_per_cu": spp * mine / (avg_ms * 1e-3) / (N_SIMD / 4 * 2.4e9),
                                   "measured_peak_per_cycle_per_cu": 4.0 / 4.91,
                                   "note": "SALU + branch + scalar-memory instructions; peak: s_add_u32 issues every 4.91 cycles per SIMD "
                                           "(tools/microbench/issue_rate.hip), one scalar unit per compute unit of four SIMDs"}
        if cpu_baseline is not None:
            cpu_baseline["calibration"] = load_calibration(name)
        total_photons = float(total_per_step) * a.steps
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
            "scaling": headline,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{w['label']}; {mine:.4g} photons per GPU per step" +
                                   (f" ({per_step:.4g} per step sharded over {n_gpus} GPUs)" if headline == "strong" else ""),
                       "name": name, "baseline_config_index": w["baseline_config"],
                       "photons_per_gpu_per_step": mine, "photons_per_step": total_per_step,
                       "parallelism": (f"photon batches sharded over {n_gpus} GPUs, one RCCL all-reduce of the float64 tally "
                                       f"buffer ({lay.total * 8} bytes) per step" if n_gpus > 1 else "single GPU"),
                       "steps_in_flight": len(lanes),
                       "overlap": (f"{len(lanes)} steps in flight: as many handles, each with a stream and a tally buffer of its own, take the steps in turn -- "
                                   "the tail and the all-reduce of step k overlap the trace of step k + 1" if overlap else
                                   "none: zero, trace, all-reduce one after the other on one stream"),
                       "rng": "Philox4x32-10 per photon, key (iseed=10, batch)",
                       **({"scale_photons": a.scale_photons} if a.scale_photons != 1.0 else {})},
            "world_size": dist.get_world_size() if dist is not None else 1,
            "backend": (dist.get_backend() if dist is not None else None),
            "devices": devices,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "achieved_is": "ALGORITHMIC bytes (SURVEY.md 8d formula x the S/K/E/D per photon THIS KERNEL counted, see per_photon) / kernel "
                                        "time: NOT bytes that crossed the HBM interface (measured_hbm_GBps); the same formula with the reference "
                                        "algorithm's work per photon is reference_equivalent",
                         "measured_hbm_GBps": measured,
                         "measured_hbm_frac": (measured / HBM_PEAK_GBS if measured is not None else None),
                         "kernel": integ.kernel_name(), "kernel_ms_avg": avg_ms, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_photon": kbpp, "per_photon": kskd, "per_photon_source": "the kernel's own work counters (this run's last step)",
                         "reference_equivalent": reference_equivalent,
                         "kernel_work": kernel_work,
                         "issue": issue,
                         "note": "working set is LDS/L2 resident: the path is vector-issue bound, not HBM bound -- see `issue` (DESIGN.md section 5)"},
            "cpu_baseline": cpu_baseline,
            "result_check": {"meanFluxUp": float(res["fluxUp"].mean()), "meanFluxDown": float(res["fluxDown"].mean()),
                             "dropped_fraction": counters["dropped"] / counters["photons"]},
        }
        if weak is not None:
            line["weak"] = weak
        if m.get("ranks") is not None:
            line["ranks"] = m["ranks"]
        if nd:
            line["result_check"]["meanIntensity"] = [float(x) for x in res["intensity"].mean(axis=(1, 2))]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for ln in lanes:
        ln["integ"].finalize_Integrator()
    return 0


def main():
    a = parse_args()
    if "RANK" in os.environ:
        return worker(a)
    return launcher(a)


if __name__ == "__main__":
    sys.exit(main())
