"""bench.py end to end on the GPU box: the self-spawning launcher (one rank process per GPU), the N = 2 control flow
(rehearsal mode: both ranks share GPU 0, tallies reduced through gloo -- RCCL refuses two ranks on one device) and
the other BASELINE workloads at a small size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=280):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_1_goes_through_the_spawn_path():
    j = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--photons", "2000000", "--no-cpu-baseline"])
    assert j["n_gpus"] == 1 and j["world_size"] == 1 and "spawned 1 rank" in j["launch"]
    assert j["config"]["photons_per_step"] == 2000000 and j["value"] > 1e7
    assert j["roofline"]["kernel"] == "photon_kernel<PhiloxStream, false, false, GRID_LDS, table in LDS>"
    assert abs(j["result_check"]["meanFluxUp"] - 0.3253) < 2e-3
    assert len(j["devices"]) == 1 and "ranks" not in j
    # (round 5) the instruction counters are a profiled run's, read from a committed file: the line says which, and whether that run's
    # kernel and work per photon are this run's
    cf = j["roofline"]["issue"]["counters_from"]
    assert cf["file"].startswith("profiles/") and cf["photons"] > 0 and abs(cf["this_run_work_per_photon"]["S"] - j["roofline"]["per_photon"]["S"]) < 1e-9
    assert cf["matches_this_run"] in (True, None), cf


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_end_to_end_in_rehearsal_mode(scaling):
    n = 1000001
    j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--photons", str(n), "--scaling", scaling],
               env=dict(I3RC_BENCH_REHEARSAL="1"))
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["backend"] == "gloo" and len(j["devices"]) == 2
    assert j["scaling"] == scaling
    # the all-reduced photon count is checked inside bench.py against exactly this figure
    assert j["config"]["photons_per_step"] == (2 * n if scaling == "weak" else n)
    assert j["config"]["photons_per_gpu_per_step"] == (n if scaling == "weak" else n // 2 + 1)
    assert abs(j["result_check"]["meanFluxUp"] - 0.3253) < 3e-3
    # (round 5) an N > 1 line says what every rank did on its own: kernel time, all-reduce time, own wall time -- and who was slowest
    rk = j["ranks"]
    assert [r["rank"] for r in rk["per_rank"]] == [0, 1] and rk["slowest_rank"] in (0, 1)
    for r in rk["per_rank"]:
        assert r["kernel_ms_per_step"] > 0 and r["allreduce_ms_per_step"] > 0 and 0 < r["own_elapsed_s"] <= j["ms_per_step"] * j["steps"] / 1e3 * 1.001
    assert rk["own_elapsed_s_min"] <= rk["own_elapsed_s_max"] and rk["imbalance"] >= 1.0


def test_two_ranks_default_is_the_metrics_case_with_the_weak_figure_beside_it():
    # no --scaling: N > 1 reports ONE batch of --photons photons per step sharded over the ranks (strong) as the headline,
    # three steps in flight, and the weak figure beside it
    n = 1000001
    j = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--photons", str(n)], env=dict(I3RC_BENCH_REHEARSAL="1"))
    assert j["scaling"] == "strong" and j["config"]["photons_per_step"] == n and j["config"]["steps_in_flight"] == 3
    assert j["weak"]["photons_per_step"] == 2 * n and j["weak"]["value"] > 1e6
    assert abs(j["result_check"]["meanFluxUp"] - 0.3253) < 3e-3


@pytest.mark.parametrize("config,scale", [("landsat36", 4e-4), ("landsat119_7dir", 4e-5)])
def test_configs_3_and_4_shard_the_nodes_batch_with_three_steps_in_flight(config, scale):
    # The default N > 1 mode on BASELINE.json's configs 3 / 4: ONE batch of the node's photons (1e9, here scaled down) per step
    # sharded over the ranks, three steps in flight (--overlap 2: three handles / streams / tally buffers whose all-reduces are
    # issued in step order), the weak figure beside it; every run starts with the first-contact check (process group up, one
    # all-reduce on every rank, under a watchdog).  Rehearsal mode: both ranks on GPU 0, tallies through gloo.
    j = _bench(["--gpus", "2", "--config", config, "--steps", "4", "--warmup", "1", "--overlap", "2", "--scale-photons", str(scale), "--contact-timeout", "120"],
               env=dict(I3RC_BENCH_REHEARSAL="1"))
    node = int(round(1_000_000_000 * scale))
    assert j["scaling"] == "strong" and j["config"]["photons_per_step"] == node and j["config"]["photons_per_gpu_per_step"] == node // 2
    assert j["config"]["steps_in_flight"] == 3 and j["config"]["scale_photons"] == scale and j["world_size"] == 2
    assert j["weak"]["photons_per_gpu_per_step"] == int(round(125_000_000 * scale)) and j["weak"]["photons_per_step"] == 2 * j["weak"]["photons_per_gpu_per_step"]
    assert 0.2 < j["result_check"]["meanFluxUp"] < 0.7
    if config == "landsat119_7dir":
        assert len(j["result_check"]["meanIntensity"]) == 7 and all(0.01 < x < 1.0 for x in j["result_check"]["meanIntensity"])


def test_first_contact_only():
    # --contact-only: the ranks bring the process group up, do one all-reduce each and leave; the launcher relays rank 0's line
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["I3RC_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--contact-only", "--contact-timeout", "120"], env=e, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["contact"] == "ok" and j["n_gpus"] == 2 and j["backend"] == "gloo"
    assert r.stderr.count("first all-reduce ok") == 2, r.stderr


def test_steps_in_flight_on_one_gpu():
    # --overlap 1: two handles / streams / tally buffers take the steps in turn; every step is still checked in full
    j = _bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--photons", "3000000", "--overlap", "1", "--no-cpu-baseline"])
    assert j["config"]["steps_in_flight"] == 2 and j["scaling"] == "weak" and "weak" not in j
    assert j["roofline"]["launches_per_step"] == 1.0
    assert abs(j["result_check"]["meanFluxUp"] - 0.3253) < 2e-3
    # the roofline's headline is the kernel's own work; the reference algorithm's figure sits beside it (equal for flux runs)
    assert j["roofline"]["achieved"] == j["roofline"]["kernel_work"]["GBps"]
    assert abs(j["roofline"]["reference_equivalent"]["achieved"] - j["roofline"]["achieved"]) < 1e-6 * j["roofline"]["achieved"]


@pytest.mark.parametrize("config,kernel", [("radar64_nadir", "photon_kernel<PhiloxStream, true, false, GRID_GLOBAL, one direction>"),
                                           ("landsat36", "photon_kernel<PhiloxStream, false, false, GRID_COLUMNS, table in LDS>"),
                                           ("landsat119_7dir", "photon_kernel<PhiloxStream, true, false, GRID_COLUMNS>")])
def test_other_baseline_workloads(config, kernel):
    n = {"radar64_nadir": 2000000, "landsat36": 4000000, "landsat119_7dir": 300000}[config]
    j = _bench(["--config", config, "--steps", "1", "--warmup", "0", "--photons", str(n), "--no-cpu-baseline"])
    assert j["config"]["name"] == config and j["roofline"]["kernel"] == kernel
    assert j["roofline"]["per_photon"]["S"] > 100
    if config != "landsat36":   # the kernel skips rays whose roulette is lost: its own bytes are below the reference algorithm's
        assert j["roofline"]["achieved"] < 0.9 * j["roofline"]["reference_equivalent"]["achieved"]
    if config != "landsat36":
        assert all(0.01 < x < 1.0 for x in j["result_check"]["meanIntensity"])


def _gpu_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: the real RCCL all-reduce between two ranks (one-GPU boxes run the gloo rehearsal above)")
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_over_rccl(scaling):
    # the exchange step on the real thing: two rank processes, one GPU each, ONE ncclAllReduce of the packed float64 tally
    # buffer per step (what replaces Code/multipleProcesses_mpi.f95:57-131); bench.py checks the all-reduced photon count
    n = 4000001
    j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--photons", str(n), "--scaling", scaling])
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["backend"] == "nccl" and len(j["devices"]) == 2
    assert "cuda:0" in j["devices"][0] and "cuda:1" in j["devices"][1]
    assert j["config"]["photons_per_step"] == (2 * n if scaling == "weak" else n)
    assert abs(j["result_check"]["meanFluxUp"] - 0.3253) < 2e-3


def test_one_rank_world_over_rccl():
    # --process-group at N = 1: a world of ONE rank over RCCL -- init_process_group on the device, the first-contact check, one
    # all-reduce (ncclAllReduce) of the packed float64 tally buffer per step, the gathers of the report: every line an N > 1 run
    # executes beside tracing, on the real backend, on a one-GPU box (two ranks over RCCL need two GPUs: test_two_ranks_over_rccl)
    j = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--photons", "2000000", "--no-cpu-baseline", "--process-group"])
    assert j["backend"] == "nccl" and j["world_size"] == 1 and j["n_gpus"] == 1
    assert j["config"]["photons_per_step"] == 2000000 and abs(j["result_check"]["meanFluxUp"] - 0.3253) < 2e-3
    r = j["ranks"]["per_rank"]
    assert len(r) == 1 and r[0]["rank"] == 0 and r[0]["kernel_ms_per_step"] > 0 and r[0]["allreduce_ms_per_step"] >= 0
