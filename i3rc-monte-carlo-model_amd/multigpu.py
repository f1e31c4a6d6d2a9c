"""Multi-GPU layer: one process per GPU, photons of a batch sharded over ranks, ONE all-reduce of the packed
float64 tally buffer (RCCL over xGMI on GPUs; gloo in CPU tests).  Replaces the ten MPI_REDUCE calls of
Code/multipleProcesses_mpi.f95:57-131 as used by Example-Drivers/monteCarloDriver.f95:333-352.

Because every photon owns a Philox stream keyed by (seed, batch) with its global index as counter, the union
of the shards is the same set of photon paths whatever the number of ranks: results do not depend on the GPU
count (the property the reference gets from per-batch seeds, monteCarloDriver.readme:45-48)."""


def shard_photons(n_total, world_size, rank):
    """Contiguous photon-index range [first, first + n) of `rank`; sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    base, rem = divmod(int(n_total), world_size)
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def all_reduce_tallies(tally, dist=None):
    """Sum the packed tally buffer (torch tensor, float64) over all ranks in place; no-op for one process."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tally)
    return tally


def max_over_ranks(value, dist=None, device="cpu"):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
