"""CPU: `python bench.py --gpus N` started bare.  The parent never touches the GPU: it starts N
fresh worker processes with the environment torch.distributed.run would give them, relays rank 0's line, and ends
the job when a rank fails.  The workers here talk over gloo; on the GPU node the same code runs over RCCL."""
import json
import os
import subprocess
import sys
import textwrap
import time

from sxmc_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from sxmc_amd import dist
    rank, local_rank, world = dist.init(backend="gloo")
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["SXMC_LAUNCHED_BY"] == "bench.py"
    rec, comm = dist.collective_record(0, {"name": "none", "pci_bus_id": "0000:00:00.0"})
    assert comm is None                        # no RCCL communicator over gloo
    total = dist.sum_over_ranks(rank + 1.0)
    print("noise on stdout of rank", rank) if rank else None
    if rank == 0:
        print(json.dumps({"world": world, "sum": total, "collective": rec}))
    dist.shutdown()
""")

FAILING = textwrap.dedent("""
    import os, sys, time
    if os.environ["RANK"] == "1":
        sys.exit(3)
    time.sleep(120)
""")


def test_spawned_ranks_form_a_job_and_rank0_line_is_relayed(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    with open(tmp_path / "out.txt", "w+") as out, open(tmp_path / "err.txt", "w+") as err:
        rc = dist.spawn_ranks(3, [sys.executable, str(script)], out=out, err=err)
        out.seek(0)
        err.seek(0)
        lines, errtxt = out.read().strip().splitlines(), err.read()
    assert rc == 0, errtxt[-3000:]
    lines = [x for x in lines if not x.startswith("[Gloo]")]     # (gloo announces itself on stdout)
    assert len(lines) == 1                        # only rank 0 writes to the job's stdout
    rec = json.loads(lines[0])
    assert rec["world"] == 3 and rec["sum"] == 6.0
    c = rec["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 3 and c["rccl_nranks"] is None
    assert c["allreduce_of_ones"] == 3.0 and c["launched_by"] == "bench.py"
    assert [d["rank"] for d in c["devices"]] == [0, 1, 2] and len({d["pid"] for d in c["devices"]}) == 3
    assert c["distinct_cards"] == 1 and "rehearsal" in c["note"]     # three ranks on one (pretend) card: said so
    assert "noise on stdout of rank 1" in errtxt  # the other ranks' stdout goes to stderr


def test_a_failing_rank_ends_the_job(tmp_path):
    script = tmp_path / "f.py"
    script.write_text(FAILING)
    t0 = time.time()
    with open(tmp_path / "err.txt", "w+") as err:
        rc = dist.spawn_ranks(2, [sys.executable, str(script)], out=err, err=err, grace_seconds=5)
        err.seek(0)
        text = err.read()
    assert rc == 3 and time.time() - t0 < 30      # rank 0 (asleep for two minutes) was ended
    assert "rank 1 exited with 3" in text


def test_bare_bench_starts_workers_and_fails_loudly_without_a_gpu():
    """In this container there is no GPU: the bare multi-rank bench must get as far as its workers -- each of
    which refuses to run without an MI355X -- and exit non-zero; it must not hang or fall back to anything."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the multi-rank bench is exercised by tests/test_gpu_bench_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "bench launcher: rank" in r.stderr
    assert r.stdout.strip() == ""
