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
