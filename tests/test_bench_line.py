"""CPU: the ONE line bench.py prints must stay short enough for the driver's bounded tail of stdout (round 3's 23 KB
line was recorded as unparsed) and must be strict JSON carrying the contract's keys, `roofline` and `cpu_baseline`.
The canned record is a real full record of the default run (profiles/r03_final_default_bench.json), also stretched
to the N = 8 shape with every optional object present."""
import copy
import json
import os

from sxmc_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    with open(os.path.join(ROOT, "profiles", "r03_final_default_bench.json")) as f:
        return json.load(f)


def strict(line):
    def no_constants(name):
        raise ValueError("not strict JSON: " + name)
    return json.loads(line, parse_constant=no_constants)


def test_default_run_line_is_short_strict_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 20000                      # what round 3 printed
    line = benchline.dumps_line(full)
    assert "\n" not in line and len(line.encode()) < benchline.MAX_LINE_BYTES
    rec = strict(line)
    for k in CONTRACT:
        assert k in rec, k
    assert rec["value"] == round(full["value"], 3) or abs(rec["value"] - full["value"]) < 1e-3 * full["value"]
    assert rec["config"]["workload"].startswith("C3") and "model" not in rec["config"]
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "fill_ordered_kernel"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and rf["traffic"] > rf["algorithmic_bytes_per_launch"]
    assert rf["launches_timed"] == 100 and 0 < rf["whole_step_frac"] < rf["frac"]
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and len(cpu["sample"]) <= 96
    assert cpu["all_cores"]["cores"] >= 1
    assert rec["parity"] == {"ok": True, "bins_and_norms_bit_exact": True, "lut_bit_exact": True,
                             "nll_rel_diff": rec["parity"]["nll_rel_diff"], "samples_checked": 100000000}
    assert rec["experiments"]["form"] == "lockstep" and rec["experiments"]["experiments_per_sec"] > 0
    for name in ("c3_lookahead", "c3_lut_materialized", "c2", "c2_float_columns", "c5", "cpp_host"):
        sub = rec["also"][name]
        assert sub["value"] > 0 and not any(isinstance(v, str) and len(v) > 64 for v in sub.values())
    assert rec["also"]["c2"]["parity_ok"] is True and rec["also"]["c5"]["kernel"] == "fill_sparse_kernel"


def test_eight_rank_line_with_everything_present_still_fits():
    full = canned()
    full["n_gpus"] = 8
    full["cpu_baseline"] = None
    full["intervals_gathered"] = [8, 15, 4]
    full["collective"] = {
        "backend": "nccl", "library": "RCCL (torch.distributed 'nccl' on ROCm) + librccl through the C ABI (sxmc_comm_*)",
        "world_size": 8, "rccl_nranks": 8, "rccl_device_of_rank0": 0, "allreduce_of_ones": 8.0, "distinct_cards": 8,
        "devices": [{"rank": r, "local_rank": r, "device_index": r, "name": "AMD Instinct MI355X (gfx950:sramecc+:xnack-)",
                     "pci_bus_id": "0000:%02x:00.0" % (5 + 16 * r), "pid": 1000 + r, "host": "node-with-a-long-hostname"}
                    for r in range(8)],
        "note": None, "launched_by": "torch.distributed.run", "intervals_through_c_abi_match_torch": True,
        "experiment_intervals_gathered_by": "sxmc_comm_allgather_f32 (librccl through the C ABI)"}
    full["experiments"]["count"] = 64
    full["experiments"]["gathered_shape"] = [64, 15, 4]
    # every sub-record the bench knows, each as long as a measured leg's
    leg = copy.deepcopy(full["also"]["c3_lut_materialized"])
    for name in ("c3_dropin", "c3_1e5_walk", "bench_pdfz", "bench_pdfz_group", "cpp_multi_gpu", "c3_step_end"):
        full["also"][name] = copy.deepcopy(leg)
    full["also"]["cpp_multi_gpu"].update({"ranks": 8, "rccl_nranks": 8, "exchange": "ncclAllGather (RCCL)",
                                          "experiments_per_sec": 3.1, "setup_locks": [{"device": d} for d in range(8)]})
    line = benchline.dumps_line(full)
    assert len(line.encode()) < benchline.MAX_LINE_BYTES
    rec = strict(line)
    assert rec["n_gpus"] == 8 and rec["cpu_baseline"] is None
    c = rec["collective"]
    assert c == {"backend": "nccl", "world_size": 8, "rccl_nranks": 8, "allreduce_of_ones": 8.0, "distinct_cards": 8,
                 "launched_by": "torch.distributed.run", "intervals_through_c_abi_match_torch": True,
                 "experiment_intervals_gathered_by": "librccl (C ABI)", "rehearsal": False}
    assert "devices" not in c and rec["experiments"]["gathered_shape"] == [64, 15, 4]
    assert len(rec["also"]) == 12


def test_nan_and_failed_legs_do_not_break_the_line():
    full = canned()
    full["roofline"]["traffic"] = None
    full["parity"]["nll_rel_diff"] = float("nan")
    full["also"]["cpp_host"] = {"failed": "tests/cpp/bench_cpp exited with 1", "stderr": "x" * 600}
    rec = strict(benchline.dumps_line(full))
    assert rec["roofline"]["traffic"] is None and rec["parity"]["nll_rel_diff"] is None
    assert rec["also"]["cpp_host"] == {"failed": "tests/cpp/bench_cpp exited with 1"}
