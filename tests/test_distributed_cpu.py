"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + single all-reduce + normalisation
logic that bench.py and the drivers use on GPUs (there with backend nccl = RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import i3rc_monte_carlo_model_amd as M
from i3rc_monte_carlo_model_amd.multigpu import all_reduce_tallies, max_over_ranks, shard_photons


def test_shards_partition_the_batch():
    for n in (1, 7, 100, 10 ** 8 + 3):
        for w in (1, 2, 3, 8):
            parts = [shard_photons(n, w, r) for r in range(w)]
            assert parts[0][0] == 0
            for (f0, n0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + n0 == f1
            assert parts[-1][0] + parts[-1][1] == n
            assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_photons(10, 2, 2)


def _fake_trace(first, n, ncol):
    """Deterministic stand-in for a rank's raw tallies: photon i adds to column i % ncol, weight from i."""
    idx = np.arange(first, first + n, dtype=np.int64)
    t = np.zeros(3 * ncol + 16, np.float64)
    np.add.at(t, idx % ncol, 1.0)                       # fluxUp
    np.add.at(t, ncol + (idx * 7) % ncol, 0.5)          # fluxDown
    t[3 * ncol + 0] = n                                 # photons counter
    return t


def _worker(rank, world, port, n_total, ncol, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = shard_photons(n_total, world, rank)
    t = torch.from_numpy(_fake_trace(first, n, ncol))
    all_reduce_tallies(t, dist)
    m = max_over_ranks(float(rank + 1), dist)
    if rank == 0:
        np.save(out, np.concatenate([t.numpy(), [m]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_single_rank(tmp_path):
    n_total, ncol = 100003, 32
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(2, port, n_total, ncol, out), nprocs=2, join=True)
    got = np.load(out)
    want = _fake_trace(0, n_total, ncol)
    assert np.array_equal(got[:-1], want)      # sums of counts are exact in float64
    assert got[-1] == 2.0                      # max over ranks
    assert got[3 * ncol] == n_total


# ---- bench.py's own launcher (no GPU here: the ranks must fail loudly, and the launcher must say so) -------------
def _bench(args, env=None, timeout=300):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    # as a rank of somebody else's launcher with the wrong world size: exit 2, no result line
    r = _bench(["--gpus", "8", "--no-cpu-baseline"], env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
    r = _bench(["--gpus", "1", "--no-cpu-baseline"], env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"))
    assert r.returncode == 2


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour of the launcher")
def test_bench_launcher_starts_n_ranks_and_reports_their_failure_without_a_gpu():
    # --gpus 2 without a launcher: two rank processes are started (each says it needs a GPU), exit code non-zero
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr
    assert "rank exit codes [3, 3]" in r.stderr


# ---- eight ranks: a batch sharded by photon range, one all-reduce, the reference's normalisation -----------------------
def _philox_vec(photon, key):
    """Philox4x32-10 block (photon, 0, 0, 0) of every photon of an int64 array: words 0 and 1 are a photon's start
    position (photon_kernel, part C), whichever rank traces it."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    c0, c1 = (photon.astype(np.uint64) & mask), (photon.astype(np.uint64) >> np.uint64(32))
    c2, c3 = np.zeros_like(c0), np.zeros_like(c0)
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & mask, p0 & mask
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1


def _beer_lambert_shard(first, n, tau, key):
    """The closed-form physics of a black (omega = 0) 1-D-column domain under an overhead sun, photon by photon from the
    photon's OWN Philox stream: it starts over column floor(u0 nx) and either reaches the surface (fluxDown, counted when
    u1 < exp(-tau)) or is absorbed on the way (fluxAbsorbed).  Packed like the tally buffer: up | down | absorbed | counters."""
    ncol = len(tau)
    idx = np.arange(first, first + n, dtype=np.int64)
    w0, w1 = _philox_vec(idx, key)
    u0 = (w0.astype(np.float64) / 4294967295.0).astype(np.float32)
    u1 = (w1.astype(np.float64) / 4294967295.0).astype(np.float32)
    col = np.minimum((u0 * np.float32(ncol)).astype(np.int64), ncol - 1)
    through = u1 < np.exp(-tau[col])
    t = np.zeros(3 * ncol + 16, np.float64)
    np.add.at(t, ncol + col[through], 1.0)
    np.add.at(t, 2 * ncol + col[~through], 1.0)
    t[3 * ncol] = n
    return t


def _normalise(t, ncol):
    """computeRadiativeTransfer :353-356 on a regular grid: tallies / (photons / columns)."""
    return t[:3 * ncol].reshape(3, ncol) / (t[3 * ncol] / ncol)


def _worker8(rank, world, port, n_total, tau, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = shard_photons(n_total, world, rank)
    t = torch.from_numpy(_beer_lambert_shard(first, n, tau, (10, 1)))
    all_reduce_tallies(t, dist)
    if rank == world - 1:     # (any rank holds the sum: an all-reduce, not a reduce to rank 0)
        np.save(out, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_reduce_normalise(tmp_path):
    """World size 8 (the node BASELINE.json's configs 3 and 4 are quoted on): one batch sharded over the ranks by
    multigpu.shard_photons, each rank's photons traced from their own Philox streams, ONE all-reduce of the packed float64
    buffer, the reference's normalisation -- equal to the unsharded batch in every bit, and to Beer-Lambert within 4 sigma."""
    n_total, ncol = 400003, 16
    tau = np.linspace(0.1, 3.0, ncol).astype(np.float32)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "reduced8.npy")
    mp.spawn(_worker8, args=(8, port, n_total, tau, out), nprocs=8, join=True)
    got = np.load(out)
    want = _beer_lambert_shard(0, n_total, tau, (10, 1))
    assert np.array_equal(got, want)
    assert got[3 * ncol] == n_total
    flux = _normalise(got, ncol)
    assert np.all(flux[0] == 0.0)
    per_col = n_total / ncol
    assert np.all(np.abs(flux[1] - np.exp(-tau)) < 4 * np.sqrt(np.exp(-tau) * (1 - np.exp(-tau)) / per_col) + 4 / per_col)
    assert np.allclose(flux[1] + flux[2], flux[1:].sum(axis=0))
    assert abs((flux[1] + flux[2]).mean() - 1.0) < 1e-12


# ---- first contact of an N > 1 bench run (bench.first_contact): gloo here, RCCL on the node -------------------------------------
_CONTACT = r'''
import argparse, os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist, datetime
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
a = argparse.Namespace(contact_timeout=float(os.environ["T"]))
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=30))
if os.environ.get("HANG") == str(rank):
    time.sleep(60)          # a rank that joins the group and never reaches its first collective
rc = bench.first_contact(a, dist, torch, rank, world, rank, True, 0.0)
# the steps of a run with three in flight: every rank issues the lanes' all-reduces in the same order (lane k %% 3), each on its own buffer
if rc == 0:
    lanes = [torch.zeros(5, dtype=torch.float64) for _ in range(3)]
    for k in range(7):
        t = lanes[k %% 3]; t.zero_(); t += (rank + 1) * (k + 1)
        dist.all_reduce(t)
        assert float(t[0]) == (k + 1) * world * (world + 1) / 2, (k, t)
    dist.barrier()
    dist.destroy_process_group()
sys.exit(rc)
'''


def _contact_ranks(world, timeout_s, hang=None, status_dir=None):
    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), T=str(timeout_s))
        if hang is not None:
            env["HANG"] = str(hang)
        if status_dir:
            env["I3RC_BENCH_STATUS_DIR"] = status_dir
        procs.append(subprocess.Popen([sys.executable, "-c", _CONTACT % root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


def test_first_contact_check_passes_and_names_a_rank_that_hangs(tmp_path):
    # all ranks there: every one reports "first all-reduce ok", leaves its mark in the status directory, and the three lanes'
    # all-reduces -- issued in the order a run with three steps in flight issues them -- give the right sums
    d = tmp_path / "status"
    d.mkdir()
    procs = _contact_ranks(3, 30, status_dir=str(d))
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], outs
    assert all("first all-reduce ok" in e for _, e in outs)
    assert sorted(os.listdir(d)) == ["rank0.contact", "rank1.contact", "rank2.contact"]
    # rank 1 never reaches its first collective: the others do not wait for ever -- they say which rank they are, what they were
    # waiting for, and leave with exit code 5 within the stated time
    import time
    t0 = time.time()
    procs = _contact_ranks(2, 4, hang=1)
    out0, err0 = procs[0].communicate(timeout=60)
    assert procs[0].returncode == 5 and "rank 0" in err0 and "first all-reduce did not complete within 4 s" in err0, err0
    assert time.time() - t0 < 40
    procs[1].kill(); procs[1].wait()


def test_launcher_calls_a_run_off_when_a_rank_makes_no_first_contact(tmp_path):
    # bench.wait_for_ranks with a status directory: a set of ranks of which one never reports is terminated and the silent rank named
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    d = tmp_path / "status"
    d.mkdir()
    good = "import os, time; open(os.path.join(%r, 'rank0.contact'), 'w').write('ok'); time.sleep(60)" % str(d)
    procs = [subprocess.Popen([sys.executable, "-c", good], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True),
             subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"], stdout=subprocess.DEVNULL)]
    import contextlib, io
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        out, e, codes = bench.wait_for_ranks(procs, contact=(str(d), 2.0), grace_s=5.0)
    assert "no first contact from rank(s) [1] within 2 s" in err.getvalue(), err.getvalue()
    assert all(c not in (0, None) for c in codes), codes
