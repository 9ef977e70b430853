#!/usr/bin/env python3
"""bench.py -- NLL evaluations per second of the sxmc hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one full NLL evaluation
at a new parameter vector = zero + histogram fill of ALL signals' MC samples with the systematics
applied per sample + PDF lookup at the data events + event log-sum + reduction + nll_total, followed
by the fused Metropolis accept/reject and next proposal (so the parameters really change every
step), i.e. one MCMC step of the reference (mcmc.cpp:261-348).

Default workload (N=1): BASELINE.json configs[2] = "C3": 1e8 samples, 3 observables, 12 signals,
shift+scale+resolution_scale floated, 20^3 bins, 1e5 data events.  Inputs are synthetic, generated on
the GPU and resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: fake experiments shard over the GPUs (sxmc.cpp:59 is an independent-iteration loop); every rank
holds a replica of the MC tables and walks its own chain with its own seed -- no data-path
collective; the per-rank intervals are gathered with one RCCL all_gather at the end.  value = steps of
all ranks / max-over-ranks time ("weak" scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def make_c3_on_gpu(torch, dev, scale, seed, nevents):
    """C3 tables generated with torch on the GPU (same distributions as workloads.config3)."""
    from sxmc_amd import workloads
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    counts = workloads.split_counts(int(1e8 * scale), 12)
    tensors, signals = [], []
    per = max(1, nevents // 12)
    ev_rows = []
    for j, n in enumerate(counts):
        e_true = torch.empty(n, device=dev).normal_(2.0 + 0.5 * j, 1.2, generator=g)
        e = e_true + torch.empty(n, device=dev).normal_(0.0, 0.3, generator=g)
        r = 6.0 * torch.empty(n, device=dev).uniform_(0.0, 1.0, generator=g) ** (1.0 / 3.0)
        c = torch.empty(n, device=dev).uniform_(-1.0, 1.0, generator=g)
        tab = torch.stack([e, r, c, e_true, torch.zeros(n, device=dev)], dim=1).contiguous()
        del e_true, e, r, c
        tensors.append(tab)
        ev_rows.append(tab[:per, :3].cpu().numpy())
        signals.append(workloads.Signal(_Shape(n, 5), 5, nexpected=80.0 + 5 * j, source_id=j))
    ev = np.concatenate(ev_rows, axis=0)[:nevents]
    events = np.zeros((ev.shape[0], 4), dtype=np.float32)
    events[:, :3] = ev
    w = workloads.Workload("C3", 3, [0.0, 0.0, -1.0], [10.0, 6.0, 1.0], [20, 20, 20], signals,
                           workloads.C3_SYSTS, workloads.C3_SIGMAS, events,
                           "S=12, N=%d, D=3, 20^3 bins, shift+scale+resolution_scale" % sum(counts))
    return w, tensors


def make_c5_on_gpu(torch, dev, scale, seed, nevents, nbins=(200, 200, 200, 4, 4)):
    """C5 tables on the GPU (same distributions as workloads.config5): S=20, D=5, F=7."""
    from sxmc_amd import workloads
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    counts = workloads.split_counts(int(1e9 * scale), 20)
    tensors, signals, ev_rows = [], [], []
    per = max(1, nevents // 20)
    for j, n in enumerate(counts):
        def u(lo, hi):
            return torch.empty(n, device=dev).uniform_(lo, hi, generator=g)
        e_true = torch.empty(n, device=dev).normal_(2.0 + 0.3 * j, 1.2, generator=g)
        e = e_true + torch.empty(n, device=dev).normal_(0.0, 0.3, generator=g)
        cols = [e, 6.0 * u(0.0, 1.0) ** (1.0 / 3.0), u(-1.0, 1.0), u(0.0, 1.0), u(0.0, 1.0), e_true,
                torch.zeros(n, device=dev)]
        tab = torch.stack(cols, dim=1).contiguous()
        del cols, e, e_true
        tensors.append(tab)
        ev_rows.append(tab[:per, :5].cpu().numpy())
        signals.append(workloads.Signal(_Shape(n, 7), 7, nexpected=50.0 + j, source_id=j))
    ev = np.concatenate(ev_rows, axis=0)[:nevents]
    events = np.zeros((ev.shape[0], 6), dtype=np.float32)
    events[:, :5] = ev
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])]
    w = workloads.Workload("C5", 5, [0.0, 0.0, -1.0, 0.0, 0.0], [10.0, 6.0, 1.0, 1.0, 1.0], list(nbins), signals,
                           systs, workloads.C3_SIGMAS, events,
                           "S=20, N=%d, D=5, bins %s" % (sum(counts), "x".join(str(b) for b in nbins)))
    return w, tensors


class _Shape:
    """Stands in for a host table when the samples live only on the GPU."""

    def __init__(self, n, f):
        self.shape = (n, f)


def make_workload(args, torch, dev, seed):
    from sxmc_amd import workloads
    name = args.workload.lower()
    if name == "c3":
        return make_c3_on_gpu(torch, dev, args.scale, seed, args.events)
    if name == "c5":
        nb = tuple(int(x) for x in args.c5_bins.split(","))
        return make_c5_on_gpu(torch, dev, args.scale, seed, args.events, nb)
    makers = {"c1": workloads.config1, "c2": workloads.config2, "c5": workloads.config5,
              "bench_pdfz": workloads.bench_pdfz, "bench_pdfz_group": workloads.bench_pdfz_group}
    w = makers[name](args.scale, seed=seed) if name == "c1" else makers[name](args.scale, seed=seed,
                                                                             nevents=args.events)
    tensors = [torch.from_numpy(s.samples).to(dev) for s in w.signals]
    return w, tensors


def cpu_baseline(w, host_tables, vector, nevals, nthreads):
    """The oracle (CPU restatement of the reference loop) timed on this box's host cores, same
    workload, same inputs.  Returns (seconds per evaluation, bins, norms, nll)."""
    from oracle import oracle
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    ne = w.events.shape[0]
    rbs = [oracle.set_eval_points(geom, w.events, s.dataset) for s in w.signals]
    best = None
    for _ in range(nevals):
        t0 = time.perf_counter()
        lut = np.zeros((w.nsignals, ne), np.float32)
        norms = np.zeros(w.nsignals, np.uint32)
        all_bins = []
        for j, s in enumerate(w.signals):
            if time.perf_counter() - t0 > 30:     # keep a long CPU leg visibly alive
                print("cpu_baseline: signal %d/%d, %.0f s" % (j, w.nsignals, time.perf_counter() - t0),
                      file=sys.stderr, flush=True)
            bins, norm = oracle.bin_samples(geom, host_tables[j], s.nfields, w.systematics,
                                            vector[w.nsources:], nthreads=nthreads)
            oracle.eval_pdf(rbs[j], bins, norm, geom.bin_volume, out=lut[j])
            norms[j] = norm
            all_bins.append(bins)
        val, _ = oracle.full_nll(lut, vector, ne, w.nsignals, w.nsources, w.parameter_means(),
                                 w.parameter_sigmas(), [s.nexpected for s in w.signals],
                                 [s.n_mc for s in w.signals], [s.source_id for s in w.signals], norms)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, all_bins, norms, val


def chain_intervals(chain, nparameters):
    """Per-parameter (point_estimate, lower, upper, coverage) from a chain: the payload of the
    RCCL gather (interval.h:22-27).  Central 90% of the samples (projection-style)."""
    out = np.zeros((nparameters, 4), dtype=np.float32)
    for i in range(nparameters):
        col = chain[:, i]
        out[i] = (np.mean(col), np.quantile(col, 0.05), np.quantile(col, 0.95), 0.9)
    return out


def value_per_gpu_hint(args, elapsed):
    return args.steps / elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c3", help="c3 (default, the metric's config), c1, c2, c5, bench_pdfz, bench_pdfz_group")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the sample counts (testing only)")
    ap.add_argument("--events", type=int, default=100000)
    ap.add_argument("--c5-bins", default="200,200,200,4,4", help="bins per observable for --workload c5")
    ap.add_argument("--form", default="graph", choices=["step", "fused", "graph", "reference", "pdfz"],
                    help="fused: zero, fill, lookup+event sum, step end = 4 launches; graph (default): the same "
                         "launches replayed from a HIP graph of --graph-steps recorded steps; step: the last two "
                         "merged (measured slower: every workgroup pays a release + ticket); reference: the "
                         "reference's own sequence with the lut re-read; pdfz: only EvalAsync + EvalFinished of all evaluators "
                         "per step, the loop of the reference's bench_sxmc (bench_sxmc.cpp:90-96, 193-200)")
    ap.add_argument("--lut-output", action="store_true",
                    help="keep the step's intermediates readable between steps: lookup table written, histograms and "
                         "normalisations left in place (4 launches per step).  Default: the walk reads none of them, so "
                         "the event sum runs over the distinct tuples of event bins weighted by multiplicity and the step "
                         "end also clears histograms and normalisations for the next step (3 launches per step)")
    ap.add_argument("--no-autotune", action="store_true",
                    help="keep the analytic launch shape instead of timing a few lane counts per CU at set-up")
    ap.add_argument("--prewarm", type=int, default=300,
                    help="untimed steps before the --warmup steps (GPU clocks and graph replay settle)")
    ap.add_argument("--graph-steps", type=int, default=10, help="steps recorded per HIP graph (--form graph)")
    ap.add_argument("--launch", default="0,0", help="bin_threads,bin_blocks_per_cu (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-evals", type=int, default=2)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--experiments", type=int, default=-1,
                    help="fake experiments for the ensemble leg (fake data + MCMC + intervals), sharded k mod N; "
                         "-1 = three per rank, 0 = skip")
    ap.add_argument("--exp-steps", type=int, default=2000, help="MCMC steps per fake experiment in the ensemble leg")
    ap.add_argument("--exp-concurrent", type=int, default=3,
                    help="fake experiments in flight per GPU in the ensemble leg (one stream each, shared MC tables)")
    ap.add_argument("--partition", type=int, default=0, help="0 auto, 1 sliced, 2 interleaved")
    ap.add_argument("--no-sparse", action="store_true", help="fill HBM-resident histograms densely (global atomics)")
    ap.add_argument("--no-prebin", action="store_true", help="bin every observable in the kernel (no pre-binned column)")
    ap.add_argument("--nsyst", type=int, default=-1, help="keep only the first K systematics (measurement only)")
    ap.add_argument("--debug-mode", type=int, default=0,
                    help="roofline measurement hook (wrong results): 1 stream only, 2 compute only, 4 no histogram")
    args = ap.parse_args()

    import torch

    from sxmc_amd import capi, dist
    from sxmc_amd.mcmc import MCMC

    rank, local_rank, world = dist.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count()   # one GPU per rank on the real node
    torch.cuda.set_device(device_index)
    capi.call("sxmc_set_device", device_index)
    dev = torch.device("cuda", device_index)
    dist.init()
    info = capi.device_info(device_index)

    # ---- inputs: same MC tables on every rank (replica), own data events + chain seed per rank
    w, tensors = make_workload(args, torch, dev, args.seed)
    if args.nsyst >= 0:
        w.systematics = w.systematics[:args.nsyst]
        w.description += " [first %d systematics only]" % args.nsyst
    exp_seed = dist.experiment_seed(args.seed, rank)
    rng = np.random.default_rng(exp_seed)
    w.events = w.events[rng.permutation(w.events.shape[0])]

    want_cpu = (not args.no_cpu_baseline) and rank == 0 and world == 1
    host_tables = [t.cpu().numpy() for t in tensors] if want_cpu else None

    fused = {"step": "step", "fused": True, "graph": True, "reference": False, "pdfz": True}[args.form]
    m = MCMC(w, seed=exp_seed & 0xFFFFFFFF, fused=fused, samples_on_device=tensors,
             stream=capi.new_stream() if args.form == "graph" else None, lut_output=args.lut_output,
             consume=not args.lut_output)
    del tensors
    torch.cuda.empty_cache()
    threads, bpc = (int(x) for x in args.launch.split(","))
    m.group.SetLaunchConfig(threads, bpc)
    m.group.SetPartition(args.partition)
    m.group.SetPrebinning(not args.no_prebin)
    m.group.SetSparse(not args.no_sparse)
    if args.debug_mode:
        want_cpu = False
    m.setup(sync_interval=max(args.steps, args.warmup + 1, 1))

    m.group.SetDebugMode(args.debug_mode)
    # EvalHist::Optimize's role (pdfz.cpp:622-727): a few trial launches pick the lane count per CU for this box
    tuned_threads = 0
    if not args.no_autotune and threads == 0 and bpc == 0:
        tuned_threads = m.group.Optimize(m.stream)
        capi.synchronize()

    def one_step():
        if args.form == "pdfz":      # EvalAsync on all, EvalFinished on all (bench_sxmc.cpp:193-200)
            m.group.EvalAsync(True, None)
            m.group.EvalFinished()
        else:
            m.step()

    # form "graph": the fused sequence replayed from a HIP graph of --graph-steps recorded steps.  Replayed
    # launches carry no events, so the last steps of every run are launched one by one with HIP events
    # around the fill kernel: the roofline figures come from those, inside the same timed region.
    gs = args.graph_steps if args.form == "graph" else 0
    def eager_share(n):          # about a tenth of the steps, and whatever does not fill a whole graph
        return n if gs <= 0 or n < 2 * gs else n - ((n - n // 10) // gs) * gs

    graph_state = {"steps_per_graph": gs, "fallback": None}

    def run_steps(n):
        ne = eager_share(n)
        if n > ne and graph_state["steps_per_graph"] > 0:
            try:
                m.steps(n - ne, graph_state["steps_per_graph"])
            except capi.SxmcError as exc:
                # recording refused (nothing was launched): same launches one by one, said so in the output
                graph_state["steps_per_graph"], graph_state["fallback"] = 0, str(exc)
                m._graph = None
                ne = n
        elif n > ne:
            ne = n
        for _ in range(ne):
            one_step()

    if args.form == "graph":
        m.step()                     # brings the launch plan up to date; recording cannot
        m.flush()
    # untimed: clocks and graph replay settle over the first few hundred steps, whatever --warmup says
    for lo in range(0, args.prewarm, 100):
        run_steps(min(100, args.prewarm - lo, args.steps))
        m.flush()
    run_steps(args.warmup)
    m.flush()

    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    m.group.Profile(True, args.steps)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0)
    fill_ms_total, nfill = m.group.ProfileRead()
    m.group.Profile(False, 0)

    chain, accepted = m.flush()
    if chain.shape[0] == 0:
        chain = np.zeros((1, w.nparameters + 1), np.float32)
    intervals = dist.gather_intervals(chain_intervals(chain, w.nparameters)[None], world, w.nparameters)

    # ---- ensemble leg (sxmc.cpp:59-145): whole fake experiments, experiment k on rank k mod N, the MC
    # tables stay resident; one RCCL all_gather of the per-experiment intervals at the end.  Outside the
    # timed region of the headline metric; reported beside it.
    experiments = None
    nexp = 3 * world if args.experiments < 0 else args.experiments
    # (fake data sets are drawn from 1-3 D histograms only, as in the reference: pdfz.cpp:499-501)
    if nexp > 0 and not args.debug_mode and args.form != "pdfz" and w.nobs <= 3:
        from sxmc_amd import ensemble
        for s_ in w.signals:
            s_.nexpected_saved = s_.nexpected
        mine = dist.experiments_of_rank(nexp, rank, world)
        local = np.zeros((len(mine), w.nparameters, 4), np.float32)
        # chains for concurrent experiments: own non-blocking stream, own per-chain state, ONE copy of the tables
        nconc = max(1, min(args.exp_concurrent, len(mine)))
        form = {"step": "step", "fused": True, "graph": True, "reference": False, "pdfz": True}[args.form]
        exp_graph = args.graph_steps if args.form in ("fused", "graph") and graph_state["fallback"] is None else 0
        pool = [MCMC(w, seed=1, fused=form, stream=capi.new_stream(), share_with=m, lut_output=args.lut_output,
                     consume=not args.lut_output) for _ in range(nconc)]
        for c in pool:
            if tuned_threads not in (0, 512):
                c.group.SetLaunchConfig(tuned_threads, 1)       # what the trial launches chose for this box
            else:
                c.group.SetLaunchConfig(threads, bpc)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for lo in range(0, len(mine), nconc):
            batch = mine[lo:lo + nconc]
            res = ensemble.run_experiments_concurrently(
                w, [dist.experiment_seed(args.seed, k) for k in batch], args.exp_steps, pool[:len(batch)],
                burnin_fraction=0.1, sync_interval=args.exp_steps, graph_steps=exp_graph)
            for i, r in enumerate(res):
                local[lo + i] = r[0]
        torch.cuda.synchronize()
        dist.barrier()
        exp_elapsed = dist.max_over_ranks(time.perf_counter() - t1)
        allint = dist.gather_intervals(local, nexp, w.nparameters)
        experiments = {
            "count": nexp, "steps_each": args.exp_steps, "seconds": exp_elapsed, "concurrent_per_gpu": nconc,
            "steps_per_graph": exp_graph,
            "experiments_per_sec": nexp / exp_elapsed,
            "steps_per_sec_inside": nexp * args.exp_steps / exp_elapsed,
            "median_upper_limit_source0": dist.median(allint[:, 0, 2]),
            "gathered_shape": [int(x) for x in allint.shape],
            "note": "fake data set + MCMC walk with burn-in re-tuning + contour intervals per experiment; "
                    "projected to 1e5-step chains: %.4f experiments/s" % (value_per_gpu_hint(args, elapsed) * world / 1e5),
        }

    total_steps = args.steps * world
    value = total_steps / elapsed
    ab = m.group.AlgorithmicBytes()
    fill_bytes = ab["fill_read"] + ab["hist"]
    fill_ms = fill_ms_total / max(nfill, 1)
    achieved = fill_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
    # SURVEY.md 8(d) counts 4 bytes for every column the computation needs (observables + referenced truth
    # fields); `fill_bytes` above is smaller when untouched observables are streamed as a pre-binned column.
    # `achieved` uses the smaller figure (what the kernel must move); the survey's figure is reported beside it.
    extra_fields = {s["true_obs"] for s in w.systematics if s.get("true_obs", -1) >= w.nobs}
    survey_bytes = 4.0 * (w.nobs + len(extra_fields)) * w.nsamples_total + ab["hist"]

    # HBM bytes of the dominant kernel from the PMC counters: collected in separate rocprofv3 --pmc
    # passes (tools/profile_on_gpu.sh), corrected as MI355X_MICROARCH.md prescribes, kept per workload
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f).get(w.name + ("_no_prebin" if args.no_prebin else ""))
        if t and args.scale == 1.0 and args.nsyst < 0:
            traffic = t["bytes_per_launch"]
    except (OSError, ValueError):
        pass

    result = {
        # BASELINE.json's metric string; `value` is its first half (NLL evaluations per second, whole job), the
        # second half (experiments per second) is the "experiments" object below
        "metric": "NLL evals/sec (10^8 samples, 3 obs, 12 signals) + experiments/sec at 1/2/4/8 GPUs",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: %s" % (w.name, w.description),
            "nsamples_total": int(w.nsamples_total), "nsignals": w.nsignals, "nobservables": w.nobs,
            "nbins": w.nbins, "nevents": int(w.events.shape[0]), "nparameters": w.nparameters,
            "step_form": args.form, "steps_per_graph": graph_state["steps_per_graph"], "graph_fallback": graph_state["fallback"],
            "prewarm_steps": args.prewarm,
            "lut_materialized": bool(args.lut_output or args.form in ("reference", "pdfz")),
            "launches_per_step": 3 if (args.form in ("pdfz", "step") or m.consume) else 4,
            "steps_launched_one_by_one_with_events": eager_share(args.steps) if graph_state["steps_per_graph"] else args.steps, "debug_mode": args.debug_mode, "autotuned_lanes_per_cu": tuned_threads, "partition": args.partition, "prebinning": not args.no_prebin, "launch": args.launch, "scale": args.scale,
            "sharding": "experiment-per-rank replicas, no data-path collective; RCCL all_gather of intervals at end",
            "samples_per_sec": value * w.nsamples_total,
            "experiments_per_sec_at_1e5_steps": value / 1e5,
            "accepted_fraction_rank0": accepted / max(args.steps, 1),
            "device": info["name"], "compute_units": info["compute_units"],
        },
        "roofline": {
            "bound": "hbm", "kernel": "fill_kernel (histogram fill, all signals batched)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": fill_bytes, "avg_launch_ms": fill_ms, "launches_timed": nfill,
            "survey_bytes_per_launch": survey_bytes,
            "achieved_at_survey_bytes": survey_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0,
            "whole_step_algorithmic_bytes": fill_bytes + ab["event"],
            "whole_step_frac": (fill_bytes + ab["event"]) * value / world / 1e9 / HBM_PEAK_GBS,
        },
        "cpu_baseline": None,
        "intervals_gathered": [int(x) for x in intervals.shape],
        "experiments": experiments,
    }

    if want_cpu:
        # same inputs, same parameter vector: time the oracle and assert parity in the same run
        vector = m.proposed_vector.get()
        m.group.EvalAsync(False, m.stream)      # histograms (dense, as CreateHistogram would)
        capi.synchronize()
        gpu_bins = [p.GetBins() for p in m.pdfs]
        m.group.EvalAsync(True, m.stream)       # lookup + NLL
        m.nll(m.proposed_vector, m.proposed_nll)
        capi.synchronize()
        gpu_nll = float(m.proposed_nll.get()[0])
        gpu_norms = m.normalizations.get()
        sec1, bins, norms, cpu_nll = cpu_baseline(w, host_tables, vector, args.cpu_evals, 1)
        # all-core variant: private histograms per thread, so cap the threads where the histogram is huge
        total_bins = int(np.prod(w.nbins))
        ncores = max(1, min(os.cpu_count() or 1, int(4e9 // (4 * total_bins))))
        secn, bins_n, norms_n, _ = cpu_baseline(w, host_tables, vector, 1, ncores)
        exact = all(np.array_equal(a, b) for a, b in zip(gpu_bins, bins)) and np.array_equal(gpu_norms, norms)
        rel = abs(gpu_nll - cpu_nll) / abs(cpu_nll)
        result["cpu_baseline"] = {
            "value": 1.0 / sec1, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d full NLL evaluations of the same workload (all %d samples, %d events), best of %d, "
                      "oracle/libsxmc_oracle.so single thread (the reference's CPU mode is a serial loop)"
                      % (args.cpu_evals, w.nsamples_total, w.events.shape[0], args.cpu_evals),
            "all_cores": {"value": 1.0 / secn, "cores": ncores,
                          "note": "same oracle, pthreads over sample chunks with private histograms"},
        }
        result["parity"] = {"bins_and_norms_bit_exact": bool(exact), "nll_gpu": gpu_nll, "nll_cpu": cpu_nll,
                            "nll_rel_diff": rel, "nll_tolerance": 1e-6}
        if not exact or not rel <= 1e-6:
            print(json.dumps(result))
            raise SystemExit("PARITY FAILURE: GPU result differs from the CPU oracle")

    if rank == 0:
        print(json.dumps(result))
    dist.shutdown()


if __name__ == "__main__":
    main()
