"""Multi-GPU layer: fake experiments shard embarrassingly over the GPUs of a node.

The reference runs its ensemble as a sequential host loop (src/sxmc.cpp:59-145); iterations are
independent, so experiment k goes to rank k mod G, every rank holds a full replica of the MC sample
tables, and the data path needs NO collective.  The only exchange is the gather of the
per-experiment intervals (interval.h:22-27: point_estimate, lower, upper, coverage per parameter)
at the end -- one small RCCL all_gather over xGMI (backend "nccl" is RCCL on ROCm; "gloo" on CPU
for the tests).  torch.distributed is plumbing here, as is torch itself.
"""
import os

import numpy as np

INTERVAL_FIELDS = 4     # point_estimate, lower, upper, coverage (interval.h:22-27)


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Join the job torch.distributed.run started (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env).
    Returns (rank, local_rank, world).  A single-process run needs no process group."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                import torch
                # SXMC_DIST_BACKEND=gloo rehearses a multi-rank run on a box with fewer GPUs than ranks
                backend = os.environ.get("SXMC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                import torch
                kw["device_id"] = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shutdown():
    _, _, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def experiments_of_rank(nexperiments, rank, world):
    """Experiment k -> rank k mod world (round-robin keeps ranks within one experiment of each other)."""
    return list(range(rank, nexperiments, world))


def experiment_seed(base_seed, k):
    """Per-experiment seed: the reference draws all experiments from one sequential gRandom stream
    (sxmc.cpp:190-191), which cannot be sharded; a hash of (base, k) replaces it."""
    x = (int(base_seed) * 0x9E3779B97F4A7C15 + (int(k) + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 29
    return x & 0x7FFFFFFFFFFFFFFF


def _device_for_collectives():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    _, _, world = env_world()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value):
    """MAX of a python float over the ranks (the bench contract's timing rule)."""
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    _, _, world = env_world()
    if world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device_for_collectives())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_intervals(local, nexperiments, nparameters):
    """local: float32 [len(experiments_of_rank), nparameters, 4] in the rank's experiment order.
    Returns float32 [nexperiments, nparameters, 4] in experiment order on every rank (all_gather of
    equal-size padded blocks: ~61 KB at 256 experiments x 15 parameters, latency-bound)."""
    rank, _, world = env_world()
    local = np.ascontiguousarray(local, dtype=np.float32).reshape(-1, nparameters, INTERVAL_FIELDS)
    assert local.shape[0] == len(experiments_of_rank(nexperiments, rank, world))
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    per = (nexperiments + world - 1) // world
    pad = np.full((per, nparameters, INTERVAL_FIELDS), np.nan, dtype=np.float32)
    pad[: local.shape[0]] = local
    dev = _device_for_collectives()
    mine = torch.from_numpy(pad).to(dev)
    out = torch.empty((world,) + pad.shape, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1)) if dist.get_backend() == "nccl" else \
        dist.all_gather(list(out.unbind(0)), mine)
    blocks = out.cpu().numpy()
    full = np.empty((nexperiments, nparameters, INTERVAL_FIELDS), dtype=np.float32)
    for r in range(world):
        ks = experiments_of_rank(nexperiments, r, world)
        full[ks] = blocks[r, : len(ks)]
    return full


def median(values):
    """utils.h:76-90 `median`: middle of the sorted list, mean of the two middle ones for even sizes."""
    v = sorted(float(x) for x in values)
    half = len(v) // 2
    return 1.0 * (v[half - 1] + v[half]) / 2 if len(v) % 2 == 0 else v[half]
