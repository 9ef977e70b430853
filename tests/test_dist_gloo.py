"""CPU, world_size 2 over gloo: the multi-GPU layer (sxmc_amd/dist.py) -- experiment sharding, the
max-over-ranks timing rule and the gather of per-experiment intervals.  On the GPU node the same
code runs over RCCL (backend "nccl")."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from sxmc_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharding_covers_every_experiment_once():
    for nexp in (1, 7, 256):
        for world in (1, 2, 4, 8):
            seen = sorted(k for r in range(world) for k in dist.experiments_of_rank(nexp, r, world))
            assert seen == list(range(nexp))
            sizes = [len(dist.experiments_of_rank(nexp, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_experiment_seeds_are_distinct_and_stable():
    seeds = [dist.experiment_seed(3, k) for k in range(256)]
    assert len(set(seeds)) == 256
    assert dist.experiment_seed(3, 5) == seeds[5] and dist.experiment_seed(4, 5) != seeds[5]


def test_median_matches_reference_rule():
    assert dist.median([3, 1, 2]) == 2            # utils.h:76-90
    assert dist.median([4, 1, 3, 2]) == 2.5


def test_single_process_paths():
    loc = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    assert np.array_equal(dist.gather_intervals(loc, 2, 3), loc)
    assert dist.max_over_ranks(1.5) == 1.5


WORKER = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %r)
    from sxmc_amd import dist
    rank, local_rank, world = dist.init(backend="gloo")
    assert world == 2
    nexp, P = 5, 3
    mine = dist.experiments_of_rank(nexp, rank, world)
    local = np.zeros((len(mine), P, 4), np.float32)
    for i, k in enumerate(mine):
        local[i] = 100 * k + np.arange(P * 4).reshape(P, 4)
    dist.barrier()
    full = dist.gather_intervals(local, nexp, P)
    for k in range(nexp):
        assert np.array_equal(full[k], 100 * k + np.arange(P * 4).reshape(P, 4)), (rank, k)
    t = dist.max_over_ranks(1.0 + rank)
    assert t == 2.0
    assert dist.sum_over_ranks(1.0 + rank) == 3.0
    # value = steps of all ranks / max time (bench.py's weak-scaling rule)
    upper = [float(full[k, 0, 2]) for k in range(nexp)]
    assert dist.median(upper) == 202.0
    dist.shutdown()
    print("rank", rank, "ok")
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert "rank %d ok" % rank in out


WORKER8 = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, %r)
    from sxmc_amd import dist
    rank, local_rank, world = dist.init(backend="gloo")
    assert world == 8
    nexp, P = 256, 15                      # BASELINE config 4: 256 experiments, 15 parameters, 8 ranks
    mine = dist.experiments_of_rank(nexp, rank, world)
    assert len(mine) == 32 and mine[0] == rank and mine[1] == rank + 8
    local = np.zeros((len(mine), P, 4), np.float32)
    for i, k in enumerate(mine):
        local[i, :, 0] = k                                    # the experiment's index rides in the payload
        local[i, :, 2] = 1000.0 - k + np.arange(P)            # "upper limit"
        local[i, :, 3] = 0.9
    dist.barrier()
    full = dist.gather_intervals(local, nexp, P)
    assert full.shape == (256, 15, 4) and not np.isnan(full).any()
    assert np.array_equal(full[:, 0, 0], np.arange(nexp, dtype=np.float32))       # every experiment once, in order
    assert dist.median(full[:, 0, 2]) == 1000.0 - 127.5                           # utils.h:76-90, even count
    assert dist.max_over_ranks(float(rank)) == 7.0 and dist.sum_over_ranks(1.0) == 8.0
    # an uneven count: the last ranks hold one experiment fewer, their padded rows never reach the result
    nexp2 = 250
    mine2 = dist.experiments_of_rank(nexp2, rank, world)
    loc2 = np.full((len(mine2), P, 4), 0.0, np.float32)
    for i, k in enumerate(mine2):
        loc2[i, :, 0] = k
    full2 = dist.gather_intervals(loc2, nexp2, P)
    assert full2.shape == (250, 15, 4) and np.array_equal(full2[:, 3, 0], np.arange(nexp2, dtype=np.float32))
    dist.shutdown()
    print("rank", rank, "ok")
""")


def run_ranks(script, world, timeout=300, extra_env=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    return procs, outs


def test_eight_ranks_gather_256_experiments_over_gloo(tmp_path):
    """The shape of the driver's 8-GPU run (BASELINE config 4), rehearsed on the CPU: experiment k on rank k mod 8, one
    all_gather of padded blocks, [256, 15, 4] back in experiment order on every rank, the median of the limits."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % ROOT)
    procs, outs = run_ranks(script, 8)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert "rank %d ok" % rank in out


WORKER_NO_ID = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, %r)
    import torch.distributed as td
    from sxmc_amd import capi, dist
    rank, local_rank, world = dist.init(backend="gloo")
    real_call = capi.call
    def failing_call(name, *a):
        if name == "sxmc_comm_unique_id":
            raise capi.SxmcError(capi.ERR_HIP, "injected: no id")
        return real_call(name, *a)
    capi.call = failing_call
    # rank 0 cannot make the id: EVERY rank must come out of RcclComm() with the same error, and the process group
    # must still be in step afterwards (round 3: rank 0 skipped the broadcast, the others waited in it)
    try:
        dist.RcclComm(init_timeout=20)
        raise SystemExit("RcclComm() did not fail")
    except capi.SxmcError as exc:
        assert "could not make an RCCL id" in str(exc) and "injected: no id" in str(exc), str(exc)
    t = torch.ones(1, dtype=torch.float64)
    td.all_reduce(t)
    assert t.item() == world
    dist.shutdown()
    print("rank", rank, "ok")
""")


def test_rccl_id_failure_on_rank0_reaches_every_rank(tmp_path):
    script = tmp_path / "worker_no_id.py"
    script.write_text(WORKER_NO_ID % ROOT)
    procs, outs = run_ranks(script, 3, timeout=240)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out[-3000:]
        assert "rank %d ok" % rank in out
