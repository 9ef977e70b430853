"""GPU box (one card): `python bench.py --gpus 2` started BARE, rehearsed over gloo with both ranks on the one card
(SXMC_DIST_BACKEND=gloo).  What the driver's multi-GPU run does, minus RCCL between processes: the launcher starts
two fresh workers, each walks its chain and its share of the fake experiments, rank 0 proves parity against the
oracle, the line carries `collective` (and says it was a rehearsal), and the C++ one-process runner follows."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bare_two_rank_bench_over_gloo(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SXMC_DIST_BACKEND"] = "gloo"
    env["SXMC_BENCH_FULL"] = str(tmp_path / "full.json")      # the full record (the stdout line is the compact one)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--prewarm", "20", "--scale", "0.02", "--events", "5000", "--experiments", "4", "--exp-steps", "300",
                        "--also", "cpp_multi_gpu"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 8192
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["collective"]["backend"] == "gloo" and line["collective"]["rehearsal"] is True
    assert line["parity"]["ok"] and line["roofline"]["launches_timed"] >= 100 and line["cpu_baseline"] is None
    assert line["experiments"]["gathered_shape"] == [4, 15, 4] and line["also"]["cpp_multi_gpu"]["ranks"] == 2
    rec = json.load(open(tmp_path / "full.json"))
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    c = rec["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["rccl_nranks"] is None and c["launched_by"] == "bench.py"
    assert c["distinct_cards"] == 1 and "rehearsal" in c["note"]
    assert [d["rank"] for d in c["devices"]] == [0, 1] and all(d["pci_bus_id"] for d in c["devices"])
    assert rec["parity"]["ok"] and rec["parity"]["bins_and_norms_bit_exact"]      # rank 0, also at N > 1
    assert rec["cpu_baseline"] is None                                            # timed at N = 1 only
    assert rec["roofline"]["launches_timed"] >= 100 and rec["roofline"]["sample"].startswith("post-timed")
    assert rec["experiments"]["count"] == 4 and rec["experiments"]["gathered_shape"] == [4, 15, 4]
    cpp = rec["also"]["cpp_multi_gpu"]
    assert cpp["ranks"] == 2 and cpp["exchange"].startswith("host staging") and cpp["experiments"] == 4


@pytest.mark.gpu
def test_six_rank_rehearsal_of_the_node_run(tmp_path):
    """The driver's 8-GPU run (BASELINE config 4: 256 experiments sharded k mod N, one gather of [256, 15, 4]) rehearsed
    as far as one card allows: SIX ranks share it over gloo -- the pool's process guard ends a run with more than six
    processes on a card, so N = 8 itself is rehearsed on the CPU (tests/test_dist_gloo.py: 8 ranks, 256 experiments)
    and in one process (tests/cpp: eight logical ranks of sxmc::ensemble_multi_gpu).  Every rank walks its 42-43
    experiments; the line must say n_gpus 6, carry the whole gather and stay short."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SXMC_DIST_BACKEND"] = "gloo"
    env["SXMC_BENCH_FULL"] = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "6", "--steps", "20", "--warmup", "5",
                        "--prewarm", "20", "--scale", "0.002", "--events", "3000", "--experiments", "256",
                        "--exp-steps", "50", "--also", "none"], capture_output=True, text=True, env=env, timeout=1100)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8192, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 6 and line["scaling"] == "weak" and line["value"] > 0 and line["parity"]["ok"]
    c = line["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 6 and c["rehearsal"] is True and c["allreduce_of_ones"] == 6.0
    ex = line["experiments"]
    assert ex["count"] == 256 and ex["gathered_shape"] == [256, 15, 4] and ex["gather_complete"] is True
    assert ex["median_upper_limit_source0"] is not None and ex["median_upper_limit_source0"] > 0
    assert ex["experiments_per_sec"] > 0
    rec = json.load(open(tmp_path / "full.json"))
    assert [d["rank"] for d in rec["collective"]["devices"]] == list(range(6))
    assert len({d["pid"] for d in rec["collective"]["devices"]}) == 6 and rec["collective"]["distinct_cards"] == 1
