"""The ONE line bench.py prints on stdout, kept short.

The driver reads the last line of bench.py's stdout from a bounded tail of the output; round 3's line (23 KB: every
sub-record with its own config, roofline, provenance and prose) no longer fitted and was recorded as unparsed.  The
full record now goes to a side file (`bench_full.json`, and to stderr); the line on stdout keeps the contract's keys
and, per object, only scalars -- no prose.  `compact()` is a pure function from the full record to that line's object
(no torch, no GPU), so that a CPU test can hold it to its size: tests/test_bench_line.py.
"""
import json
import math

MAX_LINE_BYTES = 8192     # what the test asserts; the line of a default run is ~4 KB


def _num(x, digits=6):
    """Numbers at `digits` significant digits (a float that is an integer stays one); everything else unchanged."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if math.isnan(x) or math.isinf(x):
            return None                      # (strict JSON has no NaN / Infinity)
        if x == 0:
            return 0.0
        r = float("%.*g" % (digits, x))
        return int(r) if r.is_integer() and abs(r) < 1e15 and abs(x) >= 1e5 else r
    return x


def _pick(d, keys, digits=6):
    """The scalar entries `keys` of `d` (short strings included, prose and containers left out)."""
    if not isinstance(d, dict):
        return None
    out = {}
    for k in keys:
        if k not in d or isinstance(d[k], (dict, list)):
            continue
        if isinstance(d[k], str) and len(d[k]) > 48:
            continue
        out[k] = _num(d[k], digits)
    return out


def _kernel_name(rf):
    """First word of the roofline's kernel description (the symbol rocprofv3 lists)."""
    k = (rf or {}).get("kernel") or ""
    return k.split(" (")[0][:64]


def compact_roofline(rf):
    if not isinstance(rf, dict):
        return None
    out = {"bound": rf.get("bound"), "kernel": _kernel_name(rf), "unit": rf.get("unit")}
    for k in ("achieved", "peak", "frac", "frac_survey_8d", "traffic", "algorithmic_bytes_per_launch",
              "survey_bytes_per_launch", "avg_launch_ms", "launches_timed", "whole_step_frac", "evaluations_per_launch"):
        if k in rf:
            out[k] = _num(rf[k])
    for k in ("traffic_source", "streams", "fill_form", "ordered_form", "bound_note"):
        if isinstance(rf.get(k), str):
            out[k] = rf[k][:160 if k == "bound_note" else 96]
        elif isinstance(rf.get(k), dict):
            out[k] = _pick(rf[k], ("evals_per_sec", "fill_kernel_us", "frac"))
    if isinstance(rf.get("systematics_at_timed_steps"), list):
        out["systematics_at_timed_steps"] = [_num(x, 3) for x in rf["systematics_at_timed_steps"][:8]]
    if isinstance(rf.get("f64_stream"), dict):      # the conservative, pure-f64 figure (also.c3_float_stream)
        out["f64_stream"] = _pick(rf["f64_stream"], ("evals_per_sec", "fill_kernel_us", "frac"))
    if isinstance(rf.get("sample"), str) and len(rf["sample"]) <= 48:
        out["sample"] = rf["sample"]
    prov = rf.get("traffic_provenance")
    if isinstance(prov, dict):
        out["traffic_stale"] = bool(prov.get("stale"))
    return out


def compact_cpu(cpu):
    if not isinstance(cpu, dict):
        return None
    out = _pick(cpu, ("value", "unit", "cores", "kind"))
    # the contract's `sample`: what was timed, in a few words (the full sentence is in the side file)
    s = cpu.get("sample_short") or cpu.get("sample") or ""
    out["sample"] = s if len(s) <= 96 else s[:93] + "..."
    ac = cpu.get("all_cores")
    if isinstance(ac, dict):
        out["all_cores"] = _pick(ac, ("value", "cores"))
    return out


def compact_parity(par):
    if not isinstance(par, dict):
        return None
    return _pick(par, ("ok", "bins_and_norms_bit_exact", "lut_bit_exact", "nll_rel_diff", "samples_checked"), digits=3)


def compact_config(cfg):
    if not isinstance(cfg, dict):
        return None
    out = {"workload": cfg.get("workload")}
    for k in ("nsamples_total", "nsignals", "nobservables", "nevents", "nparameters", "step_form", "steps_per_graph",
              "launches_per_step", "lut_materialized", "autotuned_lanes_per_cu", "scale", "accepted_fraction_rank0",
              "samples_per_sec", "device", "compute_units"):
        if k in cfg:
            out[k] = _num(cfg[k])
    if cfg.get("nbins") is not None:
        out["nbins"] = "x".join(str(b) for b in cfg["nbins"])
    plan = cfg.get("launch_plan") or []
    if plan:
        # "launch 0: members=12 nobs=1 ... table=ordered threads=768 grid=256 ..." -> the words that name the kernel form
        words = [w for w in plan[0].split() if w.split("=")[0] in ("hist", "program", "table", "threads", "grid")]
        out["launch_plan"] = " ".join(words)
    la = cfg.get("lookahead")
    if isinstance(la, dict):
        out["steps_per_pass"] = _num(la.get("steps_per_pass"), 4)
    return out


def compact_experiments(ex):
    if not isinstance(ex, dict):
        return None
    out = _pick(ex, ("count", "steps_each", "form", "experiments_per_sec", "steps_per_sec_inside",
                     "median_upper_limit_source0", "gather_complete"))
    if ex.get("gathered_shape") is not None:
        out["gathered_shape"] = ex["gathered_shape"]
    ls = ex.get("lockstep")
    if isinstance(ls, dict):
        out["lockstep"] = _pick(ls, ("chains_per_fill", "sets_per_gpu", "experiments_per_sec", "steps_per_sec_inside",
                                     "steps_per_sec_while_stepping_rank0", "intervals_identical_to_separate_fills",
                                     "skipped"))
    sp = ex.get("separate_fills")
    if isinstance(sp, dict):
        out["separate_fills"] = _pick(sp, ("experiments_per_sec", "steps_per_sec_inside"))
    return out


def compact_collective(col):
    if not isinstance(col, dict):
        return None
    out = _pick(col, ("backend", "world_size", "rccl_nranks", "allreduce_of_ones", "distinct_cards", "launched_by",
                      "intervals_through_c_abi_match_torch"))
    by = col.get("experiment_intervals_gathered_by")
    if by:
        out["experiment_intervals_gathered_by"] = "librccl (C ABI)" if "sxmc_comm" in by else "torch.distributed"
    out["rehearsal"] = col.get("backend") != "nccl"       # ranks sharing cards over gloo (SXMC_DIST_BACKEND)
    if col.get("backend") == "nccl" and col.get("rccl_nranks") is None:
        out["c_abi_communicator"] = "failed: intervals gathered by torch.distributed"
    return out


def compact_also(rec):
    """A sub-record in a handful of scalars: value, the fill kernel's time and fraction, parity."""
    if not isinstance(rec, dict):
        return None
    if "failed" in rec:
        return {"failed": str(rec["failed"])[:120]}
    out = {}
    for k in ("value", "unit", "steps", "ms_per_step", "fill_kernel_us", "frac", "seconds", "steps_per_sec", "accepted",
              "rows_kept", "lookahead", "samples_per_sec", "vs_published", "experiments_per_sec",
              "experiments_per_sec_after_setup", "steps_per_sec_inside", "ranks", "rccl_nranks", "exchange",
              "chain_identical_to_group_path", "ratio_to_lut_materialized", "launches_per_step", "budget_seconds",
              "within_budget", "nsteps", "locking", "experiments", "steps_each", "within_5_percent_of_prediction",
              "eight_gpu_projection_experiments_per_sec", "sync_interval"):
        if k in rec and not isinstance(rec[k], (dict, list)):
            v = rec[k]
            out[k] = (v if len(v) <= 48 else v[:45] + "...") if isinstance(v, str) else _num(v)
    for k in ("gathered_shape", "predicted_experiments_per_sec"):
        if isinstance(rec.get(k), list) and len(rec[k]) <= 4:
            out[k] = [_num(x) for x in rec[k]]
    if rec.get("per_device_locks_failed"):
        out["per_device_locks_failed"] = str(rec["per_device_locks_failed"].get("failed"))[:100]
    rf = rec.get("roofline")
    if isinstance(rf, dict):
        out["whole_step_frac"] = _num(rf.get("whole_step_frac"))
        out["kernel"] = _kernel_name(rf)
    par = rec.get("parity")
    if isinstance(par, dict):
        out["parity_ok"] = bool(par.get("ok"))
    cpu = rec.get("cpu_baseline")
    if isinstance(cpu, dict):
        out["cpu_evals_per_sec"] = _num(cpu.get("value"))
    for sub in ("ensemble", "ensemble_lockstep", "sequential", "lookahead_walk"):
        if isinstance(rec.get(sub), dict):
            out[sub] = {k: _num(v) for k, v in rec[sub].items()
                        if k in ("experiments", "steps_each", "chains_per_fill", "sets", "seconds", "experiments_per_sec",
                                 "steps_per_sec_inside", "steps_per_sec", "steps", "accepted", "rows_kept", "passes",
                                 "within_budget")}
    return out


def compact(result):
    """The object of the one stdout line, from the full record bench.py assembled."""
    r = result
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "dtype_filter", "data"):
        if k in r:
            out[k] = _num(r[k], 7)
    out["config"] = compact_config(r.get("config"))
    out["roofline"] = compact_roofline(r.get("roofline"))
    out["cpu_baseline"] = compact_cpu(r.get("cpu_baseline"))
    out["parity"] = compact_parity(r.get("parity"))
    out["collective"] = compact_collective(r.get("collective"))
    if r.get("intervals_gathered") is not None:
        out["intervals_gathered"] = r["intervals_gathered"]
    out["experiments"] = compact_experiments(r.get("experiments"))
    also = r.get("also")
    out["also"] = {k: compact_also(v) for k, v in also.items()} if isinstance(also, dict) else None
    if r.get("full_record"):
        out["full_record"] = r["full_record"]
    return out


def dumps_line(result):
    """The line itself: strict JSON, no spaces wasted; raises if it would not fit."""
    line = json.dumps(compact(result), allow_nan=False, separators=(", ", ": "))
    if len(line.encode()) >= MAX_LINE_BYTES:
        # (cannot happen with the keys above; if a future key makes it happen, drop the sub-records, keep the contract)
        slim = compact(result)
        slim["also"] = {k: {"value": (v or {}).get("value")} for k, v in (slim.get("also") or {}).items()}
        line = json.dumps(slim, allow_nan=False, separators=(", ", ": "))
    return line
