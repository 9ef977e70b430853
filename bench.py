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
ROOFLINE_LAUNCHES = 100    # event-timed fill launches behind `roofline` (after the timed region, whatever --steps is)


def source_fingerprints():
    """sha256 (16 hex digits) of the sources that decide what a fill launch reads and writes: the kernels and the
    host planner.  profiles/traffic.json keeps the pair its PMC passes were taken with (tools/update_traffic.py); a
    line whose sources differ says `traffic_stale` instead of passing an old measurement off as this code's."""
    import hashlib
    csrc = os.path.join(ROOT, "sxmc_amd", "csrc")

    def digest(names):
        h = hashlib.sha256()
        for n in names:
            path = os.path.join(csrc, n)
            if os.path.exists(path):
                h.update(n.encode() + b"\0" + open(path, "rb").read())
        return h.hexdigest()[:16]
    return {"kernels": digest(["fill_kernels.inc.h", "pdfz_kernels.hip", "sxmc_device_types.h", "layout_kernels.hip"]),
            "planner": digest(["sxmc_launch_plan.cpp", "sxmc_host.h", "sxmc_plan.h"])}


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


def oracle_threads(w, tables, which, cores=None):
    """Threads for the all-cores leg of the CPU baseline: (signals evaluated at the same time, threads per signal).
    A thread must have enough samples to outweigh its own creation and its private histogram (round 3 started 256
    pthreads per signal whatever its size: at config 2 -- 1.67e6 samples per signal -- that was SLOWER than one
    thread): at least 2e5 samples each (~3 ms of the serial loop), the private histograms of all threads together
    below 4 GB, never more threads than cores."""
    cores = cores or os.cpu_count() or 1
    nbins = int(np.prod(w.nbins))
    by_memory = max(1, int(4e9 // (4 * nbins)))
    total = sum(tables[j].shape[0] for j in which)
    budget = max(1, min(cores, by_memory, int(total // 2e5)))
    together = max(1, min(len(which), budget))
    return together, max(1, budget // together)


def oracle_eval(w, tables, vector, which, nthreads, together=1):
    """One oracle evaluation (pdfz.cpp:349-436) of the signals `which` at `vector`.  tables[j]: host
    float32 [n, F] of signal j.  nthreads: pthreads per signal inside the oracle (sample chunks with private
    histograms, summed); together: signals evaluated at the same time (host threads around the ctypes calls, which
    release the GIL).  Returns (seconds, {j: bins}, {j: norm}, {j: lut row})."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    ne = w.events.shape[0]
    t0 = time.perf_counter()
    bins_of, norm_of, lut_of = {}, {}, {}

    def one(j):
        s = w.signals[j]
        if time.perf_counter() - t0 > 30:     # keep a long CPU leg visibly alive
            print("oracle: signal %d, %.0f s" % (j, time.perf_counter() - t0), file=sys.stderr, flush=True)
        rb = oracle.set_eval_points(geom, w.events, s.dataset)
        bins, norm = oracle.bin_samples(geom, tables[j], s.nfields, w.systematics, vector[w.nsources:],
                                        nthreads=nthreads)
        row = np.zeros(ne, np.float32)
        oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=row)
        return j, bins, norm, row

    if together > 1:
        with ThreadPoolExecutor(together) as pool:
            results = list(pool.map(one, which))
    else:
        results = [one(j) for j in which]
    for j, bins, norm, row in results:
        bins_of[j], norm_of[j], lut_of[j] = bins, norm, row
    return time.perf_counter() - t0, bins_of, norm_of, lut_of


def oracle_nll(w, vector, lut, norms):
    """MCMC::nll on the CPU (nll_kernels.cpp:89-188) over a lookup table [S, E] and the norms."""
    from oracle import oracle
    ne = w.events.shape[0]
    val, _ = oracle.full_nll(lut, vector, ne, w.nsignals, w.nsources, w.parameter_means(), w.parameter_sigmas(),
                             [s.nexpected for s in w.signals], [s.n_mc for s in w.signals],
                             [s.source_id for s in w.signals], norms)
    return val


def chain_intervals(chain, nparameters):
    """Per-parameter (point_estimate, lower, upper, coverage) from a chain: the payload of the
    RCCL gather (interval.h:22-27).  Central 90% of the samples (projection-style)."""
    out = np.zeros((nparameters, 4), dtype=np.float32)
    for i in range(nparameters):
        col = chain[:, i]
        out[i] = (np.mean(col), np.quantile(col, 0.05), np.quantile(col, 0.95), 0.9)
    return out


FORMS = {"step": "step", "fused": True, "graph": True, "reference": False, "pdfz": True, "dropin": "dropin"}
K40_SAMPLES_PER_SEC = 2.99546e9     # /root/reference/README.md:322, `bench_sxmc pdfz` on an Nvidia Tesla K40 (the table's best)


class Leg:
    """One measured workload: tables resident in HBM, a chain walking it, the timed region, the roofline
    figures of its fill kernel and the parity check against the oracle."""

    def __init__(self, args, torch, dev, name, form, lut_output, seed, exp_seed, scale=None, events=None,
                 keep_host="none", lookahead=False, overrides=None):
        from sxmc_amd import capi
        from sxmc_amd.mcmc import MCMC
        if overrides:                      # (a sub-record measured with other switches than the headline's)
            args = argparse.Namespace(**dict(vars(args), **overrides))
        self.args, self.torch, self.name, self.form, self.lut_output = args, torch, name, form, lut_output
        a = argparse.Namespace(**vars(args))
        a.workload = name
        if scale is not None:
            a.scale = scale
        if events is not None:
            a.events = events
        self.scale = a.scale
        w, tensors = make_workload(a, torch, dev, seed)
        if args.extra_ctscale and w.name == "C3":
            w.systematics = list(w.systematics) + [dict(type="ctscale", obs=2, pars=[3])]
            w.syst_sigmas = list(w.syst_sigmas) + [0.01]
            w.description += " + ctscale(c)"
        if args.nsyst >= 0:
            w.systematics = w.systematics[:args.nsyst]
            w.description += " [first %d systematics only]" % args.nsyst
        rng = np.random.default_rng(exp_seed)
        w.events = w.events[rng.permutation(w.events.shape[0])]
        self.w = w
        # host copies for the oracle: every signal, or the first and the last (the largest workloads)
        if keep_host == "all":
            self.host_signals = list(range(w.nsignals))
        elif keep_host == "ends":
            self.host_signals = sorted({0, w.nsignals - 1})
        else:
            self.host_signals = []
        self.host_tables = {j: tensors[j].cpu().numpy() for j in self.host_signals}
        self.m = MCMC(w, seed=exp_seed & 0xFFFFFFFF, fused=FORMS[form], samples_on_device=tensors,
                      stream=capi.new_stream() if form == "graph" else None, lut_output=lut_output or form == "dropin",
                      consume=not lut_output and form != "dropin")
        del tensors
        torch.cuda.empty_cache()
        threads, bpc = (int(x) for x in args.launch.split(","))
        m = self.m
        m.group.SetLaunchConfig(threads, bpc)
        m.group.SetPartition(args.partition)
        m.group.SetPrebinning(not args.no_prebin)
        m.group.SetBucketing(not args.no_bucket)
        m.group.SetOrdering(not args.no_order)
        if args.no_codes:
            m.group.SetCodes(False)
        if args.no_boxes:
            m.group.SetBoxes(False)
        m.group.SetTailKernel(not args.no_tail)
        m.group.SetRuntimeKernels(not args.no_rtc)
        m.group.SetSparse(not args.no_sparse)
        self.graph_state = {"steps_per_graph": args.graph_steps if form == "graph" else 0, "fallback": None}
        self.tuned_threads = 0
        # the look-ahead walk: two evaluations per pass over the tables (needs histograms in LDS, event classes)
        self.lookahead = bool(lookahead) and form == "graph" and not lut_output and name.lower() in ("c3", "c1")

    def setup(self, steps, warmup):
        from sxmc_amd import capi
        args, m = self.args, self.m
        # (the jump buffer holds every step between two flushes: the longest run here, the post-timed sample included)
        m.setup(sync_interval=max(steps, warmup + 1, min(args.prewarm, 100) + 1, 2 * ROOFLINE_LAUNCHES + 8, 1))
        if args.debug_mode:               # (measurement build only: SXMC_HIP_LIB=.../libsxmc_hip_measure.so)
            m.group.SetDebugMode(args.debug_mode)
        if self.form == "pdfz":
            # bench_sxmc evaluates at params = 0 (bench_sxmc.cpp:66, 166): the evaluators stay bound to the proposal
            # vector, which is put back to the means (the systematics' are 0) and never stepped
            m.proposed_vector.set(self.w.parameter_means().astype(np.float64))
        # EvalHist::Optimize's role (pdfz.cpp:622-727): a few trial launches pick the lane count per CU for this box
        if not args.no_autotune and args.launch == "0,0":
            self.tuned_threads = m.group.Optimize(m.stream)
            capi.synchronize()
        if self.form == "graph":
            m.step()                     # brings the launch plan up to date; recording cannot
            m.flush()
        self.la = None
        if self.lookahead and not m.group.LookaheadSupported():
            self.lookahead = False       # (config 1: its step ends in the one-workgroup form; it walks sequentially)
        if self.lookahead:
            from sxmc_amd.mcmc import LookaheadWalk
            lt, lb = (int(x) for x in args.launch.split(","))
            self.la = LookaheadWalk(m, threads=lt if lt else 768, blocks_per_cu=lb if lt else 1)
            self.la.bind()
        # untimed: clocks and graph replay settle over the first few hundred steps, whatever --warmup says
        for lo in range(0, args.prewarm, 100):
            self.run_steps(min(100, args.prewarm - lo, max(steps, 1)))
            m.flush()
        self.run_steps(warmup)
        m.flush()

    def one_step(self, through_group=False):
        m = self.m
        if self.form == "pdfz" and through_group:
            # the same launches through the explicit group call: what the roofline sample profiles (the fill's events
            # hang on m.group; the evaluations the library batches behind the per-evaluator calls run the same plan)
            m.group.EvalAsync(True, None)
            m.group.EvalFinished()
        elif self.form == "pdfz":    # EvalAsync on all, EvalFinished on all (bench_sxmc.cpp:90-96, 193-200): the
            for p in m.pdfs:         # evaluators' OWN calls -- the library defers them and launches one batch
                p.EvalAsync()
            for p in m.pdfs:
                p.EvalFinished()
        else:
            m.step()

    # form "graph": the fused sequence replayed from a HIP graph of --graph-steps recorded steps.  Replayed
    # launches carry no events: the roofline figures come from steps launched one by one AFTER the timed region.
    def eager_share(self, n):          # whatever does not fill a whole graph
        # (round 3 also launched a tenth of the timed steps one by one, with events around the fill, for the roofline
        #  figures; those now come from launches made AFTER the timed region, and the timed region is the walk as a
        #  driver walks it -- graph replays: at --steps 20 half of the 3 ms timed were the slower eager launches)
        gs = self.graph_state["steps_per_graph"]
        return n if gs <= 0 else n % gs

    def run_steps(self, n):
        from sxmc_amd import capi
        m, gstate = self.m, self.graph_state
        if self.la is not None:
            # look-ahead walk: passes (two evaluations each) until the chain has advanced by exactly n steps
            self.la.steps(n, graph_passes=gstate["steps_per_graph"])
            return
        ne = self.eager_share(n)
        if n > ne and gstate["steps_per_graph"] > 0:
            try:
                m.steps(n - ne, gstate["steps_per_graph"])
            except capi.SxmcError as exc:
                # recording refused (nothing was launched): same launches one by one, said so in the output
                gstate["steps_per_graph"], gstate["fallback"] = 0, str(exc)
                m._graph = None
                ne = n
        elif n > ne:
            ne = n
        for _ in range(ne):
            self.one_step()

    def timed(self, steps, collective=True):
        """Exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
        from sxmc_amd import dist
        m, torch = self.m, self.torch
        m.group.Profile(True, steps)
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run_steps(steps)
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if collective:
            elapsed = dist.max_over_ranks(elapsed)
        self.fill_ms_region, self.nfill_region = m.group.ProfileRead()
        m.group.Profile(False, 0)
        self._launches = m.group.LastStepLaunches() if (m.consume and m.tail and self.form != "pdfz") else None
        self.elapsed, self.steps = elapsed, steps
        # ---- the roofline sample, decoupled from --steps: AFTER the timed region (ms_per_step comes from the timed
        # steps alone) the walk goes on for ROOFLINE_LAUNCHES more steps launched one by one, each fill bracketed by
        # HIP events on the stream it is launched on, from the state the chain has reached
        n = max(ROOFLINE_LAUNCHES, 0)
        self.fill_ms_total, self.nfill = self.fill_ms_region, self.nfill_region
        self.roofline_sample = "inside the timed region: the %d steps launched one by one" % self.nfill_region
        self.region_chain = None
        if n > 0:
            self.region_chain = m.flush()          # (rows kept, accepted) of the timed steps: what the line reports
            m.group.Profile(True, 2 * n + 8)
            if self.la is not None:
                self.la.steps(2 * n, graph_passes=0)      # (one or two steps per pass: at least n passes)
            else:
                for _ in range(n):
                    self.one_step(through_group=True)
            torch.cuda.synchronize()
            post_ms, post_n = m.group.ProfileRead()
            m.group.Profile(False, 0)
            m.flush()
            if post_n >= self.nfill_region:
                self.fill_ms_total, self.nfill = post_ms, post_n
                self.roofline_sample = "post-timed, %d launches" % post_n
        self.event_bracket_ms = self.empty_bracket_ms()
        return elapsed

    def empty_bracket_ms(self, n=50):
        """What two hipEventRecord calls measure with NOTHING between them, on the walk's stream (mean of n).  The fill's
        own durations do NOT contain it: profiled fills are launched through hipExtLaunchKernelGGL /
        hipExtModuleLaunchKernel with a start and a stop event, which carry the dispatch's own begin and end timestamps
        -- the duration rocprofv3 --kernel-trace reports.  (Round 2 recorded an event before and after the launch: that
        also timed the packets in between, ~2.5 us per launch, a fifth of config 2's fill.)  Reported so that the
        difference is on record."""
        import ctypes as C

        from sxmc_amd import capi
        e0, e1 = C.c_void_p(0), C.c_void_p(0)
        capi.call("sxmc_event_create", C.byref(e0))
        capi.call("sxmc_event_create", C.byref(e1))
        total, ms = 0.0, C.c_float(0)
        st = capi.ptr(self.m.stream)
        for _ in range(n):
            capi.call("sxmc_event_record", e0, st)
            capi.call("sxmc_event_record", e1, st)
            capi.call("sxmc_event_synchronize", e1)
            capi.call("sxmc_event_elapsed_ms", e0, e1, C.byref(ms))
            total += ms.value
        capi.call("sxmc_event_destroy", e0)
        capi.call("sxmc_event_destroy", e1)
        return total / n

    def launches_per_step(self):
        if getattr(self, "_launches", None):
            return self._launches                      # what the library actually launched for the last timed step
        return 3 if (self.form in ("pdfz", "step") or self.m.consume) else 4

    def roofline(self, world=1):
        args, w, m = self.args, self.w, self.m
        ab = m.group.AlgorithmicBytes()
        fill_bytes = ab["fill_read"] + ab["hist"]
        fill_ms = self.fill_ms_total / max(self.nfill, 1)
        achieved = fill_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
        value = self.steps * world / self.elapsed
        # SURVEY.md 8(d) counts 4 bytes for every column the computation needs (observables + referenced truth
        # fields); `fill_bytes` is smaller when untouched observables are not streamed as float columns.
        # `achieved` uses the smaller figure (what the kernel must move); the survey's figure is reported beside it.
        extra_fields = {s["true_obs"] for s in w.systematics if s.get("true_obs", -1) >= w.nobs}
        survey_bytes = 4.0 * (w.nobs + len(extra_fields)) * w.nsamples_total + ab["hist"]
        # HBM bytes of the dominant kernel from the PMC counters: collected in separate rocprofv3 --pmc
        # passes (tools/profile_on_gpu.sh), corrected as MI355X_MICROARCH.md prescribes, kept per workload
        traffic, traffic_note = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                t = json.load(f).get(w.name + ("_no_prebin" if args.no_prebin else "") + ("_no_bucket" if args.no_bucket else "") +
                                       ("_no_order" if args.no_order and not args.no_bucket else "") +
                                       ("_no_codes" if args.no_codes and not (args.no_order or args.no_bucket) else "") +
                                       ("_boxed" if "boxed+codes(now)" in m.group.LaunchInfo() or "table=boxed+codes " in m.group.LaunchInfo() else "") +
                                       ("_lookahead" if self.la is not None else ""))
            if args.extra_ctscale or args.no_sparse:
                t = None
            if t and self.scale == 1.0 and args.nsyst < 0:
                traffic = t["bytes_per_launch"]
                now, then = source_fingerprints(), t.get("profiled_sources")
                stale = [k for k in now if not then or then.get(k) != now[k]]
                traffic_note = {"from": t.get("source"), "profiled_sources": then, "current_sources": now,
                                "stale": bool(stale),
                                "warning": ("profiles/traffic.json was measured before the last change to the %s source(s): "
                                            "`traffic` is that older build's figure until the PMC passes are repeated "
                                            "(tools/profile_on_gpu.sh + tools/update_traffic.py)" % " and ".join(stale))
                                if stale else None}
        except (OSError, ValueError):
            pass
        info = m.group.LaunchInfo()
        # (a plan with two forms names the one its launches take now: "boxed+codes(now)|ordered+codes")
        info = info.replace("boxed+codes|ordered+codes(now)", "ordered+codes").replace("boxed+codes(now)|ordered+codes",
                                                                                      "boxed+codes")
        kname = ("fill_boxed_kernel" if "table=boxed" in info else
                 "fill_ordered_kernel" if "table=ordered" in info and "+runs" not in info
                 else "fill_sparse_kernel" if "+runs" in info and not args.no_sparse else "fill_kernel")
        neval = 2 if self.la is not None else 1
        if self.la is not None:
            kname = "sx_rtc_fill = " + kname.replace("_kernel", "_body") + " for two parameter vectors (look-ahead pass)"
        return {
            "bound": "hbm", "kernel": kname + " (histogram fill, all signals batched; rocprofv3 lists run-time "
                                              "compiled kernels as sx_rtc_fill)",
            "launch_plan": info.strip().split("\n"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # `frac` is on the bytes THIS kernel must move (codes: 4 B/sample at config 3).  SURVEY.md 8(d) counts 4 bytes for
            # every column the computation needs (16 B/sample there): on those bytes the same launch time is a multiple of
            # the peak -- the kernel does not stream them, it streams a 16-bit code per field and reads float values only
            # for the samples its error bound cannot decide
            "frac_survey_8d": (survey_bytes / (fill_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if fill_ms > 0 else 0.0,
            "traffic": traffic, "traffic_provenance": traffic_note,
            # (PMC counters cannot be read inside this run: separate rocprofv3 --pmc passes, tools/profile_on_gpu.sh)
            "traffic_source": ("profiles/traffic.json: " + str(traffic_note.get("from"))) if traffic_note else None,
            # (the boxed fill is not bound by its stream: say so where `bound` is read)
            "bound_note": ("boxed fill: vector issue, LDS additions and the float-column drain, not the stream "
                           "(the stream alone takes 41 of its 63-65 us: profiles/r05_boxed_parts.log)"
                           if "boxed+codes(now)" in m.group.LaunchInfo() or "table=boxed+codes " in m.group.LaunchInfo() else None),
            # (... and where the chain was: the systematics' parameters of the evaluations around the timed region -- the
            # boxed fill's time depends on the resolution parameter)
            "systematics_at_timed_steps": [float(x) for x in m.proposed_vector.get()[w.nsources:]] if w.systematics else None,
            # (a plan with two forms of the fill: which one the timed launches took, and who chose)
            "fill_form": ("boxed: chosen by the walk from the first proposal's parameters (sxmc_group_adapt_fill_form)"
                          if "boxed+codes(now)" in m.group.LaunchInfo() else
                          "ordered: chosen by the walk from its parameters" if "ordered+codes(now)" in m.group.LaunchInfo()
                          else None),
            "streams": ("u16 codes of one observable (f32 filter), f64 interval arithmetic on the other's granule boxes, "
                        "f64 exact fallback" if "boxed+codes" in info else
                        "u16 codes (f32 filter) + f64 exact fallback" if "ordered+codes" in info else "f32 columns (f64 arithmetic)"),
            "algorithmic_bytes_per_launch": fill_bytes, "bytes_per_sample": ab["fill_read"] / max(w.nsamples_total, 1),
            "avg_launch_ms": fill_ms, "launches_timed": self.nfill, "sample": self.roofline_sample,
            "timing": "HIP events stamped by the dispatch itself (hipExtLaunchKernelGGL start/stop events) on the stream the "
                      "fill is launched on: the kernel's own duration, as rocprofv3 --kernel-trace reports it",
            # two hipEventRecord calls with nothing between them (NOT part of avg_launch_ms, see empty_bracket_ms)
            "empty_event_bracket_ms": self.event_bracket_ms,
            "in_timed_region": {"launches": self.nfill_region,
                                "avg_launch_ms": self.fill_ms_region / max(self.nfill_region, 1)},
            # a look-ahead pass fills the histograms of TWO evaluations from one pass over the tables: `achieved` and
            # `frac` count the bytes the launch must stream once; per evaluation it is half of that
            "evaluations_per_launch": neval,
            "achieved_counting_every_evaluation": neval * achieved,
            "survey_bytes_per_launch": survey_bytes,
            "achieved_at_survey_bytes": survey_bytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0,
            "whole_step_algorithmic_bytes": fill_bytes + ab["event"],
            # (bytes the step must move x steps per second; a look-ahead pass streams them once for the 1-2 steps it
            #  decides, so there it is bytes x PASSES per second)
            "whole_step_frac": (fill_bytes + neval * ab["event"]) * value / world / 1e9 / HBM_PEAK_GBS /
                               (self.la.steps_seen / self.la.passes_seen if self.la is not None and self.la.passes_seen
                                else 1.0),
        }

    def config(self):
        w, args, m = self.w, self.args, self.m
        la = None
        if self.la is not None and self.la.passes_seen:
            la = {"passes": self.la.passes_seen, "steps": self.la.steps_seen,
                  "steps_per_pass": self.la.steps_seen / self.la.passes_seen,
                  "note": "value counts chain steps (each one NLL evaluation the walk uses); every pass evaluates the "
                          "likelihood twice -- at the step's proposal and at the vector the next step proposes after "
                          "a rejection -- and decides one or two steps; the chain is the sequential one bit for bit "
                          "(tests/test_gpu_nll.py: test_lookahead_walk_is_the_sequential_chain)"}
        cfg = self._config_plain()
        cfg["lookahead"] = la
        return cfg

    def _config_plain(self):
        w, args, m = self.w, self.args, self.m
        return {
            "workload": "%s: %s" % (w.name, w.description),
            "nsamples_total": int(w.nsamples_total), "nsignals": w.nsignals, "nobservables": w.nobs,
            "nbins": w.nbins, "nevents": int(w.events.shape[0]), "nparameters": w.nparameters,
            "step_form": self.form, "steps_per_graph": self.graph_state["steps_per_graph"],
            "graph_fallback": self.graph_state["fallback"],
            "lut_materialized": bool(self.lut_output or self.form in ("reference", "pdfz")),
            "launches_per_step": self.launches_per_step(),
            "steps_launched_one_by_one_with_events":
                self.eager_share(self.steps) if self.graph_state["steps_per_graph"] else self.steps,
            "autotuned_lanes_per_cu": self.tuned_threads, "scale": self.scale,
            "launch_plan": m.group.LaunchInfo().strip().split("\n"),
        }

    def parity(self, time_evals=0):
        """GPU against the oracle at the chain's current proposal, on the signals whose tables were kept on
        the host: every bin count and norm (dense evaluation, what CreateHistogram reads), the lookup-table
        rows the walk's own evaluation form produces (sparse event-bin counters where histograms exceed LDS)
        bit for bit, and the NLL.  When only some signals went through the oracle, the oracle's NLL chain runs
        on their oracle rows plus the GPU's rows of the others.  time_evals > 0 also times the oracle
        (single thread, best of time_evals) -> cpu_baseline.  Raises SystemExit on a mismatch."""
        from sxmc_amd import capi
        m, w = self.m, self.w
        which = self.host_signals
        if not which:
            return None, None
        m.flush()                               # (room in the jump buffer for the step below)
        vector = m.proposed_vector.get()
        m.group.EvalAsync(False, m.stream)      # histograms (dense, as CreateHistogram would)
        capi.synchronize()
        together, per_signal = oracle_threads(w, self.host_tables, which)
        ncores = together * per_signal
        secn, bins, norms, rows = oracle_eval(w, self.host_tables, vector, which, per_signal, together)
        exact_bins = True
        for j in which:
            exact_bins = exact_bins and np.array_equal(m.pdfs[j].GetBins(), bins[j])
            bins[j] = None
        m.group.EvalAsync(True, m.stream)       # lookup (+ NLL below) in the evaluation form of the walk
        m.nll(m.proposed_vector, m.proposed_nll)
        capi.synchronize()
        gpu_nll = float(m.proposed_nll.get()[0])
        gpu_norms = m.normalizations.get()
        gpu_lut = m.lut.get().reshape(w.nsignals, -1)
        exact_norms = all(int(gpu_norms[j]) == int(norms[j]) for j in which)
        exact_lut = all(np.array_equal(gpu_lut[j].view(np.uint32), rows[j].view(np.uint32)) for j in which)
        lut = gpu_lut.copy()
        for j in which:
            lut[j] = rows[j]
        cpu_nll = oracle_nll(w, vector, lut, gpu_norms)
        rel = abs(gpu_nll - cpu_nll) / abs(cpu_nll)
        # the walk's own step form (event classes / fused lookup) at the same vector
        m.step(debug_mode=True)
        chain, _ = m.flush()
        step_nll = float(chain[-1, -1])
        step_rel = abs(step_nll - np.float32(cpu_nll)) / abs(cpu_nll)
        par = {"bins_and_norms_bit_exact": bool(exact_bins and exact_norms), "lut_bit_exact": bool(exact_lut),
               "signals_checked": [int(j) for j in which], "signals_total": w.nsignals,
               "samples_checked": int(sum(self.host_tables[j].shape[0] for j in which)),
               "nll_gpu": gpu_nll, "nll_cpu": cpu_nll, "nll_rel_diff": rel, "nll_tolerance": 1e-6,
               "nll_of_walk_step_form_float32": step_nll, "nll_of_walk_step_form_rel_diff": step_rel,
               "nll_oracle_inputs": "oracle lookup rows for every signal" if len(which) == w.nsignals else
                                    "oracle lookup rows for the checked signals, GPU rows for the others"}
        cpu = None
        if time_evals > 0:
            nchecked = par["samples_checked"]
            best = min(oracle_eval(w, self.host_tables, vector, which, 1)[0] for _ in range(time_evals))
            frac = nchecked / float(w.nsamples_total)
            if ncores > 1:      # (the parity evaluation above was the first all-cores run: best of two)
                secn = min(secn, oracle_eval(w, self.host_tables, vector, which, per_signal, together)[0])
            cpu = {"value": frac / best, "unit": "evals/s", "cores": 1, "kind": "port",
                   "sample": "%d oracle evaluations (zero + fill + lookup) of %d of the %d signals = %d of the %d samples, "
                             "%d events, best of %d, oracle/libsxmc_oracle.so single thread (the reference's CPU mode "
                             "is a serial loop)%s"
                             % (time_evals, len(which), w.nsignals, nchecked, w.nsamples_total, w.events.shape[0],
                                time_evals, "" if frac == 1.0 else "; value scaled to the whole workload by sample count"),
                   "sample_short": "%d evals x %.3g samples (%d/%d signals), best, 1 thread"
                                   % (time_evals, nchecked, len(which), w.nsignals),
                   "all_cores": {"value": frac / secn, "cores": ncores, "host_cores": os.cpu_count() or 1,
                                 "signals_at_once": together, "threads_per_signal": per_signal,
                                 "note": "same oracle: %d signals at a time x %d pthreads over sample chunks with private "
                                         "histograms (>= 2e5 samples per thread, private histograms below 4 GB in all, "
                                         "at most one thread per host core)" % (together, per_signal)}}
        ok = par["bins_and_norms_bit_exact"] and exact_lut and rel <= 1e-6 and step_rel <= 2e-6
        par["ok"] = bool(ok)
        return par, cpu

    def close(self):
        from sxmc_amd import capi
        m = self.m
        capi.synchronize()
        if getattr(self, "la", None) is not None:
            self.la.close()              # (the multigroup goes before its groups)
            self.la = None
        if m._graph is not None:
            m._graph.close()
        for p in m.pdfs:
            p.close()
        m.group.close()
        self.host_tables = {}
        self.m = None
        import gc
        gc.collect()
        self.torch.cuda.empty_cache()


def also_record(args, torch, dev, name, form, lut_output, steps, warmup, exp_seed, keep_host, share=None,
                lookahead=False, overrides=None):
    """A sub-record of the default run: another workload (or another step form of the headline workload)
    measured the same way -- evals/s, the fill kernel's time and roofline fraction, parity."""
    t0 = time.perf_counter()
    leg = Leg(args, torch, dev, name, form, lut_output, args.seed, exp_seed, scale=1.0, keep_host=keep_host,
              lookahead=lookahead, overrides=overrides)
    leg.setup(steps, warmup)
    elapsed = leg.timed(steps, collective=False)
    rf = leg.roofline()
    par, cpu = leg.parity(time_evals=1)
    rec = {"value": steps / elapsed, "unit": "steps/s (each one NLL evaluation the walk uses)" if lookahead else "evals/s",
           "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * elapsed / steps, "fill_kernel_us": 1e3 * rf["avg_launch_ms"], "frac": rf["frac"],
           "config": leg.config(), "roofline": rf, "parity": par, "cpu_baseline": cpu,
           "leg_seconds": None}
    leg.close()
    rec["leg_seconds"] = time.perf_counter() - t0
    if par is not None and not par["ok"]:
        print(json.dumps({name: rec}), file=sys.stderr)
        raise SystemExit("PARITY FAILURE (%s): GPU result differs from the CPU oracle" % name)
    return rec


def run_bench_cpp(argv, timeout):
    """tests/cpp/bench_cpp as a child process (its own HIP context; no Python in it).  Returns (records, failure):
    the JSON lines it printed, and None or a description of what went wrong -- a bench_cpp that crashes, faults or
    exits non-zero is a FAILURE of this bench, not a skipped leg."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bench_cpp")
    if not os.path.exists(exe):
        return [], {"failed": "tests/cpp/bench_cpp is not built (__graft_entry__.build() builds it)"}
    try:
        r = subprocess.run([exe] + [str(a) for a in argv], capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as exc:
        return [], {"failed": "tests/cpp/bench_cpp did not finish within %d s" % timeout,
                    "stderr": (exc.stderr or b"")[-600:].decode(errors="replace") if isinstance(exc.stderr, bytes)
                    else (exc.stderr or "")[-600:]}
    lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
    if r.returncode != 0:
        return lines, {"failed": "tests/cpp/bench_cpp exited with %d" % r.returncode, "stderr": r.stderr[-600:]}
    return lines, None


def cpp_host_records(args, want, lut_materialized_value=None, nsteps=4000):
    """BASELINE config 3 at full size walked entirely by the C++ host layer (tests/cpp/bench_cpp: sxmc::build_pdfz +
    sxmc::MCMC over the C ABI, no Python in that process): the north star's "host code stays C++".  ONE child process
    builds the tables once and walks, as asked for in `want`:
      cpp_host     -- the walk sxmc::MCMC chooses by itself (lookahead_auto: one evaluation per step where the plan
                      streams codes, the look-ahead pass where it streams float columns), 4 000 steps incl. set-up, both
                      burn-in re-tunings and the flushes, then 16 whole fake experiments, 8 in flight with a fill each;
      c3_dropin    -- mcmc.cpp:264-271 + 314-348 AS WRITTEN (S x EvalAsync, S x EvalFinished, nll_event_chunks,
                      finish_nll_jump_pick_combo; lookup table materialised; legacy default stream; no graph, no group
                      call): what an unchanged caller gets, the library batching the S evaluations behind the API;
      c3_1e5_walk  -- BASELINE config 3 as written: 1e5 steps, burn-in fraction 0.1 (re-tuning at 1e4 and 2e4 steps),
                      sync_interval 10 000 (mcmc.cpp:261-378), one evaluation per step and as the look-ahead walk;
                      BASELINE.md section 4's bar: 60 % of the HBM roofline = 1e5 steps in 34 s.
    Returns {name: record}; a bench_cpp that fails gives every wanted name a {"failed": ...} record."""
    t0 = time.perf_counter()
    walks = []
    if "cpp_host" in want:
        walks.append("auto=%d" % nsteps)
    if "c3_dropin" in want:
        walks.append("reference=3000")
    if "c3_1e5_walk" in want:
        walks += ["sequential=100000", "lookahead=100000"]
    argv = ["--scale", "1.0", "--graph-steps", args.graph_steps, "--burnin", "0.1", "--sync-interval", "10000"]
    if walks:
        argv += ["--walks", ",".join(walks)]
    if "cpp_host" in want:
        # (a fill per chain, 8 experiments in flight: over codes the lockstep passes no longer pay)
        argv += ["--experiments", 16, "--exp-steps", 2000, "--chains", 1, "--sets", 8]
    if "c4_per_gpu" in want:
        # BASELINE config 4's per-GPU share AS WRITTEN: 8 whole fake experiments of config 3's 1e5 steps each, all eight in
        # flight on this GPU with a fill each (sxmc::ensemble_concurrent; sxmc.cpp:59-145 with fit.nsteps = 1e5)
        argv += ["--c4", "8x100000"]
    if not walks:
        argv += ["--no-walk"]
    lines, failure = run_bench_cpp(argv, 560)
    if failure:
        return {name: dict(failure) for name in want}
    by_walk = {}
    for ln in lines:
        if "walk" in ln:
            by_walk[(ln["walk"], ln["steps"])] = ln
    out = {}
    if "cpp_host" in want:
        rec = dict(by_walk[("auto", nsteps)])
        ens = [ln for ln in lines if "ensemble_" in ln.get("driver", "") and ln.get("leg") != "c4_per_gpu"]
        if ens:
            rec["ensemble"] = ens[0]              # 16 whole fake experiments, 8 in flight with a fill each: two per lane
        rec["value"], rec["unit"] = rec["steps_per_sec"], "evals/s"
        rec["note"] = "whole walk of %d steps including set-up, re-tuning and flushes; host = C++ only" % nsteps
        out["cpp_host"] = rec
    if "c3_dropin" in want:
        rec = dict(by_walk[("reference", 3000)])
        rec["value"], rec["unit"] = rec["steps_per_sec"], "evals/s"
        rec["launches_per_step"] = 5     # zero, fill (all signals), eval_pdf, nll_event_chunks, finish_nll_jump_pick_combo
        rec["deferred_evaluations_per_launch"] = rec["deferred_evaluations"] / max(rec["deferred_launches"], 1)
        if lut_materialized_value:
            rec["ratio_to_lut_materialized"] = rec["value"] / lut_materialized_value
        rec["note"] = ("the reference's own call sequence (mcmc.cpp:264-271, 314-348), unchanged: whole walk of 3000 steps "
                       "incl. set-up and re-tunings; the S = 12 EvalAsync calls of a step leave the library as ONE launch "
                       "sequence (sxmc_hist_eval_async defers), EvalFinished waits lazily (sxmc_set_lazy_finish)")
        out["c3_dropin"] = rec
    if "c3_1e5_walk" in want:
        seq, la = by_walk[("sequential", 100000)], by_walk[("lookahead", 100000)]
        budget = 34.0     # >= 60 % of the HBM roofline <=> 1e5 steps in <= 34 s
        out["c3_1e5_walk"] = {
            "value": seq["steps_per_sec"], "unit": "evals/s", "nsteps": 100000, "seconds": seq["seconds"],
            "steps_per_sec": seq["steps_per_sec"], "accepted": seq["accepted"], "rows_kept": seq["rows_kept"],
            "budget_seconds": budget, "within_budget": bool(seq["seconds"] <= budget),
            "sequential": dict(seq, within_budget=bool(seq["seconds"] <= budget)),
            "lookahead_walk": dict(la, within_budget=bool(la["seconds"] <= budget)),
            "note": "BASELINE config 3 as written: sxmc::MCMC (C++ host) walks 1e5 steps, burn-in fraction 0.1 (widths "
                    "re-tuned from the chain at 1e4 and 2e4 steps, rows before 2e4 dropped), sync_interval 10 000 "
                    "(mcmc.cpp:261-378); `seconds` includes the walk's set-up; top level = one evaluation per step"}
    if "c4_per_gpu" in want:
        c4 = [ln for ln in lines if ln.get("leg") == "c4_per_gpu"]
        if not c4:
            out["c4_per_gpu"] = {"failed": "bench_cpp printed no c4_per_gpu record"}
        else:
            rec = dict(c4[0])
            # 0.124-0.133 was written before this was first measured (round 4) for the ordered form
            # alone; with the boxed form where the chains' resolution parameters allow it, one box measured 0.143
            lo, hi = 0.124, 0.150
            v = rec["experiments_per_sec"]
            rec.update({"value": v, "unit": "experiments/s (1e5 steps each, one GPU)",
                        "predicted_experiments_per_sec": [lo, hi],
                        "within_5_percent_of_prediction": bool(0.95 * lo <= v <= 1.05 * hi),
                        "eight_gpu_projection_experiments_per_sec": 8 * v,
                        "note": "BASELINE config 4 = nexperiments = 256 x config 3 (1e5 steps each) over 8 GPUs: this is ONE "
                                "GPU's share run as written -- 8 experiments in flight on the card, each a fake data set "
                                "drawn on the device, a 1e5-step walk with burn-in re-tuning and flushes every 1e4 steps, "
                                "contour intervals; 256 experiments on 8 such GPUs = 32 per GPU = 4 rounds of this"})
            out["c4_per_gpu"] = rec
    share = (time.perf_counter() - t0) / max(len(out), 1)
    for rec in out.values():
        rec["leg_seconds"] = share
    return out


def pdfz_record(args, torch, dev, name, exp_seed):
    """The reference's own benchmark loops (bench/bench_sxmc.cpp): `pdfz` = ONE evaluator of 1e7 N(0,1) samples, 1-D,
    1000 bins on [-3, 3), one shift systematic at p = 0, 1e5 evaluation points; 1 warm-up + 100 x (EvalAsync,
    EvalFinished) (:34-102) -- and `pdfz_group` = 29 evaluators of 1e3 ... 3e6 samples, per repetition EvalAsync on all,
    then EvalFinished on all (:105-225).  The evaluators' own calls, nothing else: what the timed loop of bench_sxmc
    does.  samples/s as it prints them, beside README.md:317-322's table (best entry: 2.995e9 on a Tesla K40)."""
    t0 = time.perf_counter()
    leg = Leg(args, torch, dev, name, "pdfz", False, 7 if name == "bench_pdfz" else 8, exp_seed, scale=1.0,
              keep_host="all", overrides={"prewarm": 1})
    leg.setup(100, 1)
    elapsed = leg.timed(100, collective=False)
    rf = leg.roofline()
    par, cpu = leg.parity(time_evals=1)
    n = int(leg.w.nsamples_total)
    rec = {"value": 100 / elapsed, "unit": "evals/s", "steps": 100, "warmup": 1, "ms_per_step": 1e3 * elapsed / 100,
           "samples_per_sec": n * 100 / elapsed, "vs_published": n * 100 / elapsed / K40_SAMPLES_PER_SEC,
           "published": {"samples_per_sec": K40_SAMPLES_PER_SEC, "device": "Nvidia Tesla K40",
                         "source": "reference README.md:317-322 (bench_sxmc pdfz)"},
           "fill_kernel_us": 1e3 * rf["avg_launch_ms"], "frac": rf["frac"], "config": leg.config(), "roofline": rf,
           "parity": par, "cpu_baseline": cpu,
           "note": "timed loop = the evaluators' own EvalAsync / EvalFinished calls (the library batches them); the "
                   "histogram's one observable is only shifted, so the table is kept sorted by it and a 256-sample granule "
                   "whose end values fall into one bin is counted without reading its samples: the fill streams 12 bytes "
                   "per granule (`frac` is on those bytes; the kernel is launch-bound)"}
    leg.close()
    rec["leg_seconds"] = time.perf_counter() - t0
    if par is not None and not par["ok"]:
        print(json.dumps({name: rec}), file=sys.stderr)
        raise SystemExit("PARITY FAILURE (%s): GPU result differs from the CPU oracle" % name)
    return rec


def cpp_multi_gpu_record(args, ngpus, collective):
    """BASELINE config 4's shape from the C++ host layer, ONE process: sxmc::ensemble_multi_gpu -- a host thread per
    GPU with its own replica of the evaluators, experiment k on device k mod G, eight in flight per card with a fill
    each (over codes the lockstep passes no longer pay), ONE RCCL all-gather
    of the intervals (sxmc_comm_allgather_f32) -- over the full-size C3 tables.  Run by rank 0 after the Python
    ranks have finished and released their cards.  In a rehearsal (ranks sharing a card over gloo) the device threads
    run on the same cards as the ranks did and their blocks meet through host memory: RCCL refuses two ranks on one
    card, and the record says "host staging (rehearsal)"."""
    t0 = time.perf_counter()
    nexp = 16 * ngpus if args.experiments < 0 else max(args.experiments, ngpus)   # (two experiments per lane)
    devices = ",".join(str(d["device_index"]) for d in collective["devices"])
    extra = [] if collective["backend"] == "nccl" else ["--host-staging"]
    argv = ["--scale", args.scale, "--no-walk", "--graph-steps", args.graph_steps, "--experiments", nexp,
            "--exp-steps", args.exp_steps, "--chains", 1, "--sets", 8, "--device-list", devices] + extra
    # Measured with one set-up lock PER CARD (graph recording stays exclusive for the process: MultiGpuOptions::
    # PER_DEVICE) -- with ONE lock for the process, the library's default until a run on more than one card is on
    # record, the cards' set-ups take turns and the runner's rate says more about the lock than about the cards.  Should
    # the relaxed mode fail on a real node, the leg is repeated with the process-wide lock and the record carries both.
    lines, failure = run_bench_cpp(argv + ["--per-device-locks"], 300)
    relaxed_failure = None
    if failure:
        relaxed_failure = failure
        lines, failure = run_bench_cpp(argv, 300)
    if failure:
        failure["per_device_locks"] = relaxed_failure
        return failure
    rec = lines[-1]
    if relaxed_failure:
        rec["per_device_locks_failed"] = relaxed_failure
    rec["leg_seconds"] = time.perf_counter() - t0
    return rec


def emit(result, json_out, path=None):
    """The job's ONE stdout line: the compact object of sxmc_amd/benchline.py (a few KB: the contract's keys, one
    roofline and one cpu_baseline object, every sub-record as a handful of scalars).  The FULL record -- every
    sub-record's config, launch plan, provenance and notes -- goes to `bench_full.json` beside this file (the line
    names it) and to stderr.  Round 3 printed the full record as the line: 23 KB, which the driver's bounded tail of
    stdout no longer held."""
    from sxmc_amd import benchline
    path = path or os.environ.get("SXMC_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
            f.write("\n")
        result = dict(result, full_record=os.path.relpath(path, ROOT))
    except OSError as exc:
        print("bench.py: could not write %s: %s" % (path, exc), file=sys.stderr)
    print("bench.py full record: " + json.dumps(result), file=sys.stderr, flush=True)
    print(benchline.dumps_line(result), file=json_out, flush=True)


def parse_args(argv=None):
    """The command line (argv None: sys.argv) -- also what tools/ use to build the same legs as the bench."""
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
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (timing AND parity)")
    ap.add_argument("--cpu-evals", type=int, default=2)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--experiments", type=int, default=-1,
                    help="fake experiments for the ensemble leg (fake data + MCMC + intervals), sharded k mod N; "
                         "-1 = eight per rank, 0 = skip")
    ap.add_argument("--exp-steps", type=int, default=2000, help="MCMC steps per fake experiment in the ensemble leg")
    ap.add_argument("--exp-lockstep", type=int, default=2,
                    help="ensemble leg, second pass: chains per lockstep set (one fill pass per step for the set); 0 = skip."
                         "  2 (default): with the ordered fill two chains leave LDS for two replicas of each histogram")
    ap.add_argument("--exp-sets", type=int, default=4, help="lockstep sets in flight per GPU (one stream each)")
    ap.add_argument("--exp-concurrent", type=int, default=4,
                    help="fake experiments in flight per GPU in the ensemble leg (one stream each, shared MC tables)")
    ap.add_argument("--also", default="auto",
                    help="sub-records measured after the headline: comma list of c3_float_stream, c3_lookahead, c3_lut_materialized, c2, c5, "
                         "c2_float_columns, bench_pdfz, bench_pdfz_group, cpp_host, c3_dropin, c3_1e5_walk, c4_per_gpu, cpp_multi_gpu; auto = the single-GPU ones when the headline is the full-size C3 on one "
                         "GPU, cpp_multi_gpu (sxmc::ensemble_multi_gpu over the same cards) at N > 1; none = skip")
    ap.add_argument("--also-steps", type=int, default=200, help="timed steps of each sub-record (C5: a quarter)")
    ap.add_argument("--partition", type=int, default=0, help="0 auto, 1 sliced, 2 interleaved")
    ap.add_argument("--no-sparse", action="store_true", help="fill HBM-resident histograms densely (global atomics)")
    ap.add_argument("--no-prebin", action="store_true", help="bin every observable in the kernel (no pre-binned column)")
    ap.add_argument("--no-tail", action="store_true",
                    help="step end as its own kernels (lookup + event sum, then step end + clearing: 3 launches per step) "
                         "instead of one workgroup doing all of it in one launch")
    ap.add_argument("--no-rtc", action="store_true",
                    help="programs of systematics outside the library's table run the run-time decoded kernel instead "
                         "of one specialised through hiprtc")
    ap.add_argument("--extra-ctscale", action="store_true",
                    help="C3 with a fourth systematic, a cos-theta scale on c: a program that is not in the table")
    ap.add_argument("--lookahead", action="store_true",
                    help="the look-ahead walk for the headline leg: every pass over the tables evaluates the step's proposal "
                         "AND the vector the next step proposes after a rejection; one or two steps per pass, the same "
                         "chain bit for bit (+25 %% steps/s at config 3).  The default run carries it as the sub-record "
                         "c3_lookahead; the headline walks with one evaluation per step, whose fill kernel is the "
                         "HBM-bound one the roofline figures are about")
    ap.add_argument("--no-order", action="store_true",
                    help="bucketed tables without the ordered observable (every written observable is streamed)")
    ap.add_argument("--no-codes", action="store_true",
                    help="ordered tables streamed as float columns (8 B/sample at config 3) instead of 16-bit codes with an "
                         "exact recheck of the samples near a bin edge (4 B/sample); the sub-record c3_float_stream")
    ap.add_argument("--no-boxes", action="store_true",
                    help="the ordered form with codes (4 B/sample at config 3: r ordered, e and e_true streamed as codes) "
                         "instead of the boxed form (2 B/sample: e boxed, r streamed as codes); the sub-record "
                         "c3_ordered_codes")
    ap.add_argument("--no-bucket", action="store_true",
                    help="stream the table in the caller's row order (no copy grouped by the untouched observables' bins)")
    ap.add_argument("--nsyst", type=int, default=-1, help="keep only the first K systematics (measurement only)")
    ap.add_argument("--debug-mode", type=int, default=0,
                    help="kernel measurement hooks (WRONG RESULTS; needs SXMC_HIP_LIB=sxmc_amd/csrc/libsxmc_hip_measure.so): "
                         "1 stream only, 2 compute only, 4 no histogram, 16 no drain, 32 no LDS additions")
    return ap.parse_args(argv)


def main():
    args = parse_args()

    from sxmc_amd import dist

    rank, local_rank, world = dist.env_world()
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started bare (`python bench.py --gpus N`): this process, which has made no HIP call and has not asked
        # torch about the GPU (it does not even import torch), starts N fresh workers of the same command line
        # -- one rank per GPU, what torch.distributed.run would start -- and relays rank 0's JSON line
        raise SystemExit(dist.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    # the job's stdout carries ONE line, the JSON record: whatever libraries print on fd 1 meanwhile (gloo and RCCL
    # announce themselves there) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    from sxmc_amd import capi
    from sxmc_amd.mcmc import MCMC

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ncards = torch.cuda.device_count()
    backend = os.environ.get("SXMC_DIST_BACKEND") or "nccl"
    if world > ncards and backend == "nccl":
        raise SystemExit("--gpus %d on a box with %d GPU(s): RCCL needs one card per rank (a rehearsal of the "
                         "multi-rank path over gloo, ranks sharing a card, is SXMC_DIST_BACKEND=gloo; its line says so)"
                         % (world, ncards))
    device_index = local_rank % ncards   # one GPU per rank on the real node
    torch.cuda.set_device(device_index)
    capi.call("sxmc_set_device", device_index)
    dev = torch.device("cuda", device_index)
    dist.init()
    info = capi.device_info(device_index)
    # what the collectives of this job really run on (backend, RCCL's own rank count, every rank's card)
    collective, rccl = dist.collective_record(device_index, info)

    # ---- inputs: same MC tables on every rank (replica), own data events + chain seed per rank
    exp_seed = dist.experiment_seed(args.seed, rank)
    want_cpu = (not args.no_cpu_baseline) and rank == 0 and not args.debug_mode   # parity: rank 0 at every N
    big = args.workload.lower() == "c5" and args.scale >= 0.2
    leg = Leg(args, torch, dev, args.workload, args.form, args.lut_output, args.seed, exp_seed,
              keep_host=("ends" if big else "all") if want_cpu else "none",
              lookahead=args.lookahead)
    w, m = leg.w, leg.m
    leg.setup(args.steps, args.warmup)
    tuned_threads = leg.tuned_threads
    graph_state = leg.graph_state
    threads, bpc = (int(x) for x in args.launch.split(","))

    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    elapsed = leg.timed(args.steps)

    chain, accepted = leg.region_chain if leg.region_chain is not None else m.flush()
    if chain.shape[0] == 0:
        chain = np.zeros((1, w.nparameters + 1), np.float32)
    mine_iv = chain_intervals(chain, w.nparameters)[None]
    intervals = dist.gather_intervals(mine_iv, world, w.nparameters)
    if rccl is not None:
        # the same exchange on librccl through the C ABI (what sxmc::ensemble_multi_gpu calls); NaN-free payload
        via_abi = dist.gather_intervals(mine_iv, world, w.nparameters, comm=rccl)
        collective["intervals_through_c_abi_match_torch"] = bool(np.array_equal(via_abi, intervals))

    # ---- ensemble leg (sxmc.cpp:59-145): whole fake experiments, experiment k on rank k mod N, the MC
    # tables stay resident; one RCCL all_gather of the per-experiment intervals at the end.  Outside the
    # timed region of the headline metric; reported beside it.
    experiments = None
    nexp = 8 * world if args.experiments < 0 else args.experiments
    # (fake data sets are drawn from 1-3 D histograms only, as in the reference: pdfz.cpp:499-501)
    if nexp > 0 and not args.debug_mode and args.form != "pdfz" and w.nobs <= 3:
        from sxmc_amd import ensemble
        mine = dist.experiments_of_rank(nexp, rank, world)
        local = np.zeros((len(mine), w.nparameters, 4), np.float32)
        # chains for concurrent experiments: own non-blocking stream, own per-chain state, ONE copy of the tables
        nconc = max(1, min(args.exp_concurrent, len(mine)))
        form = FORMS[args.form]
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
        separate = {"concurrent_per_gpu": nconc, "seconds": exp_elapsed, "experiments_per_sec": nexp / exp_elapsed,
                    "steps_per_sec_inside": nexp * args.exp_steps / exp_elapsed,
                    "note": "one fill per chain per step, chains on their own streams (round 1's form)"}
        # ---- the same experiments with the chains in LOCKSTEP sets: one fill pass per step for the chains of a set
        lockstep = None
        L = args.exp_lockstep
        # (decided from what EVERY rank holds -- the smallest share -- so that all ranks take the same path through the
        # barriers below, whatever --experiments is)
        if L >= 2 and form is True and nexp // world >= L:
            from sxmc_amd.mcmc import LockstepChains
            nsets = max(1, args.exp_sets)
            sets = []
            for _ in range(nsets):
                st = capi.new_stream()
                cs = [MCMC(w, seed=1, fused=True, stream=st, share_with=m, lut_output=False, consume=True)
                      for _ in range(L)]
                for c in cs:
                    c.group.SetLaunchConfig(*((tuned_threads, 1) if tuned_threads not in (0, 512) else (threads, bpc)))
                sets.append(LockstepChains(cs))
            per_round = nsets * L
            local2 = np.zeros_like(local)
            try:
                dist.barrier()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                done = 0
                ls_timing = {}
                for lo in range(0, len(mine) - len(mine) % per_round, per_round):
                    batch = mine[lo:lo + per_round]
                    res = ensemble.run_experiments_in_lockstep(
                        w, [dist.experiment_seed(args.seed, k) for k in batch], args.exp_steps, sets,
                        burnin_fraction=0.1, sync_interval=args.exp_steps, graph_steps=exp_graph, timing=ls_timing)
                    for i, r in enumerate(res):
                        local2[lo + i] = r[0]
                    done += len(batch)
                torch.cuda.synchronize()
                dist.barrier()
                ls_elapsed = dist.max_over_ranks(time.perf_counter() - t2)
                ndone = int(dist.sum_over_ranks(done))
                same = bool(np.array_equal(local2[:done], local[:done]))
                lockstep = {"chains_per_fill": L, "sets_per_gpu": nsets, "count": ndone, "seconds": ls_elapsed,
                            "experiments_per_sec": ndone / ls_elapsed,
                            "steps_per_sec_inside": ndone * args.exp_steps / ls_elapsed,
                            # the stepping alone (set-up of the experiments -- fake data, evaluation points, first
                            # evaluation -- and their intervals left out): what an experiment of 1e5 steps consists of
                            "stepping_seconds_rank0": ls_timing.get("stepping_seconds"),
                            "steps_per_sec_while_stepping_rank0": (done * args.exp_steps / ls_timing["stepping_seconds"]
                                                                   if ls_timing.get("stepping_seconds") else None),
                            "intervals_identical_to_separate_fills": same,
                            "note": "sxmc_multigroup_step_async: the chains of a set share ONE pass over the tables per "
                                    "step; sets on their own streams"}
            except capi.SxmcError as exc:
                lockstep = {"skipped": str(exc)}
            for st in sets:
                capi.synchronize()
                st.close()
                for c in st.chains:
                    for p in c.pdfs:
                        p.close()
                    c.group.close()
        best = lockstep if lockstep and "experiments_per_sec" in lockstep and \
            lockstep["experiments_per_sec"] > separate["experiments_per_sec"] else separate
        allint = dist.gather_intervals(local, nexp, w.nparameters, comm=rccl)
        if collective is not None:
            collective["experiment_intervals_gathered_by"] = ("sxmc_comm_allgather_f32 (librccl through the C ABI)"
                                                              if rccl is not None else "torch.distributed all_gather")
        experiments = {
            "count": nexp, "steps_each": args.exp_steps, "seconds": exp_elapsed, "concurrent_per_gpu": nconc,
            "steps_per_graph": exp_graph, "separate_fills": separate, "lockstep": lockstep,
            # the headline of this object: the faster of the two forms (both are listed above)
            "form": "lockstep" if best is lockstep else "separate_fills",
            "experiments_per_sec": best["experiments_per_sec"],
            "steps_per_sec_inside": best["steps_per_sec_inside"],
            "median_upper_limit_source0": dist.median(allint[:, 0, 2]),
            "gathered_shape": [int(x) for x in allint.shape],
            # every experiment's block arrived from its rank (the blocks are padded with NaN rows, which must not remain)
            "gather_complete": bool(not np.isnan(allint).any()),
            "note": "fake data set + MCMC walk with burn-in re-tuning + contour intervals per experiment; "
                    "at 1e5 steps per experiment (BASELINE config 3/4) that is %.4f experiments/s on this job"
                    % (best["steps_per_sec_inside"] / 1e5)
                    + ("" if not (best is lockstep and lockstep.get("steps_per_sec_while_stepping_rank0")) else
                       " counting the set-up of these short experiments, %.4f per GPU by the stepping rate alone"
                       % (lockstep["steps_per_sec_while_stepping_rank0"] / 1e5)),
        }
        for c in pool:
            capi.synchronize()
            if c._graph is not None:
                c._graph.close()
            for p in c.pdfs:
                p.close()
            c.group.close()
        del pool

    value = args.steps * world / elapsed
    cfg = leg.config()
    cfg.update({
        "prewarm_steps": args.prewarm, "debug_mode": args.debug_mode, "partition": args.partition,
        "prebinning": not args.no_prebin, "bucketing": not args.no_bucket, "ordering": not (args.no_order or args.no_bucket),
        "codes": not (args.no_codes or args.no_order or args.no_bucket),
        "launch": args.launch,
        "sharding": "experiment-per-rank replicas, no data-path collective; RCCL all_gather of intervals at end",
        "samples_per_sec": value * w.nsamples_total,
        "experiments_per_sec_at_1e5_steps": value / 1e5,
        "accepted_fraction_rank0": accepted / max(args.steps, 1),
        "device": info["name"], "compute_units": info["compute_units"],
    })
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
        # the arithmetic the path computes in.  Every bin index is the reference's f64 result (bit-exact against the oracle);
        # where the plan streams CODES, 99.9 % of the samples get there through a filter -- f32 multiply-adds on 16-bit
        # codes plus an error bound -- and only the ambiguous ones through the f64 operations themselves
        "dtype": ("f64 exact; f32-on-u16-codes filter decides the unambiguous samples" if "+codes" in m.group.LaunchInfo()
                  else "f64"),
        "dtype_filter": "f32-on-u16-codes" if "+codes" in m.group.LaunchInfo() else None,
        "data": "synthetic",
        "config": cfg,
        "roofline": leg.roofline(world),
        "cpu_baseline": None,
        "parity": None,
        "collective": collective,
        "intervals_gathered": [int(x) for x in intervals.shape],
        "experiments": experiments,
        "also": None,
    }

    if want_cpu:
        # same inputs, same parameter vector: parity asserted for every workload, the oracle timed beside it
        # (the oracle is TIMED at N = 1 only; at N > 1 rank 0 still proves parity with one oracle evaluation)
        par, cpu = leg.parity(time_evals=args.cpu_evals if world == 1 else 0)
        result["parity"], result["cpu_baseline"] = par, cpu
        if not par["ok"]:
            emit(result, json_out)
            raise SystemExit("PARITY FAILURE: GPU result differs from the CPU oracle")

    # ---- sub-records: the other single-GPU configurations and the lookup-table-materialising step form,
    # measured in the same run so that their numbers carry the driver's clock too
    also = args.also
    if also == "auto":
        full_c3 = args.workload.lower() == "c3" and args.scale == 1.0 and not args.debug_mode
        also = "none"
        if full_c3 and want_cpu and args.form == "graph" and world == 1:
            also = ("c3_ordered_codes,c3_float_stream,c3_lookahead,c3_lut_materialized,c2,c2_float_columns,c5,bench_pdfz,bench_pdfz_group,"
                    "cpp_host,c3_dropin,c3_1e5_walk,c4_per_gpu")
        elif full_c3 and args.form == "graph" and world > 1:
            also = "cpp_multi_gpu"        # the C++ one-process runner over the same N cards
    if world > 1:
        # every rank lets go of its card before rank 0 starts the C++ runner on all of them
        leg.close()
        if rccl is not None:
            rccl.close()
        torch.cuda.empty_cache()
        dist.barrier()
        dist.shutdown()
    failed_leg = None
    if also != "none" and rank == 0:
        if world == 1:
            leg.close()
        recs = {}
        names = [x.strip() for x in also.split(",") if x.strip()]
        cpp_names = [n for n in names if n in ("cpp_host", "c3_dropin", "c3_1e5_walk", "c4_per_gpu")]
        for name in names:
            if name == "c3_lookahead":            # the same walk taken one or two steps per pass (two evaluations per pass)
                recs[name] = also_record(args, torch, dev, "c3", "graph", False, 3 * args.also_steps, 20, exp_seed, "all",
                                         lookahead=True)
            elif name == "c3_float_stream":       # the headline's walk with the ordered table's FLOAT columns streamed
                recs[name] = also_record(args, torch, dev, "c3", "graph", False, args.also_steps, 20, exp_seed, "all",
                                         overrides={"no_codes": True})
            elif name == "c3_ordered_codes":      # ... with the ORDERED form over codes (rounds 4-5's headline kernel)
                recs[name] = also_record(args, torch, dev, "c3", "graph", False, args.also_steps, 20, exp_seed, "all",
                                         overrides={"no_boxes": True})
            elif name == "c3_lut_materialized":
                recs[name] = also_record(args, torch, dev, "c3", "graph", True, args.also_steps, 20, exp_seed, "all")
            elif name == "c2":
                recs[name] = also_record(args, torch, dev, "c2", "graph", False, args.also_steps, 20, exp_seed, "all")
            elif name == "c2_float_columns":      # config 2 with both observables streamed as floats (no pre-binned column)
                recs[name] = also_record(args, torch, dev, "c2", "graph", False, args.also_steps, 20, exp_seed, "all",
                                         overrides={"no_prebin": True})
            elif name == "c5":
                recs[name] = also_record(args, torch, dev, "c5", "graph", False, max(10, args.also_steps // 4), 10,
                                         exp_seed, "ends")
            elif name in ("bench_pdfz", "bench_pdfz_group"):
                recs[name] = pdfz_record(args, torch, dev, name, exp_seed)
            elif name in cpp_names:
                if name == cpp_names[0]:      # ONE bench_cpp process for all of them (the tables are built once)
                    lm = recs.get("c3_lut_materialized") or {}
                    recs.update(cpp_host_records(args, cpp_names, lut_materialized_value=lm.get("value")))
            elif name == "cpp_multi_gpu":
                recs[name] = cpp_multi_gpu_record(args, world, collective)
            else:
                raise SystemExit("unknown --also entry %r" % name)
            if isinstance(recs.get(name), dict) and "failed" in recs[name]:
                failed_leg = name
        result["also"] = recs
        fs = recs.get("c3_float_stream")
        if isinstance(fs, dict) and "value" in fs:
            # the pure-f64 figure, named where the headline's fraction is read: the same walk streaming float columns
            result["roofline"]["f64_stream"] = {"evals_per_sec": fs["value"], "fill_kernel_us": fs.get("fill_kernel_us"),
                                                "frac": fs.get("frac"), "record": "also.c3_float_stream"}
        oc = recs.get("c3_ordered_codes")
        if isinstance(oc, dict) and "value" in oc and "boxed" in str(result["roofline"].get("fill_form")):
            # the ORDERED form of the same walk (what the fill takes once the resolution parameter has moved away from 0:
            # 81 us whatever the parameters, where the boxed form takes 62 us at 0, 73 at 0.05, 123 at 0.2)
            result["roofline"]["ordered_form"] = {"evals_per_sec": oc["value"], "fill_kernel_us": oc.get("fill_kernel_us"),
                                                  "frac": oc.get("frac"), "record": "also.c3_ordered_codes"}

    if rank == 0:
        emit(result, json_out)
    if world == 1:
        dist.shutdown()
    # a failed leg fails the job at every N -- AFTER the line is out: the ranks' own measurement stands in it, and the
    # exit code says that something beside it did not run (at N > 1 that is the C++ one-process runner over the same
    # cards: hiding its first failure on a real node would be worse than a red run)
    if failed_leg:
        raise SystemExit("bench.py: the %s leg FAILED (its record says how): %s"
                         % (failed_leg, result["also"][failed_leg].get("failed")))


if __name__ == "__main__":
    main()
