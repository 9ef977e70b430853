"""Synthetic TNtuple-shaped workloads for the parity tests and the bench harness.

The shapes restate BASELINE.json's configs as SURVEY.md section 8(d) fixes them (the reference's
own inputs are ROOT files that do not exist here).  Everything is generated from stated seeds
with numpy; `scale` shrinks the sample counts for tests without changing the shape.
"""
import numpy as np


class Signal:
    """One signal's MC sample table + how it enters the fit."""

    def __init__(self, samples, nfields, nexpected, source_id, dataset=0):
        self.samples = samples              # float32 [n, nfields] row-major
        self.nfields = nfields
        self.nexpected = float(nexpected)
        self.source_id = int(source_id)
        self.dataset = dataset
        self.n_mc_total = None              # simulated events BEFORE cuts, when they differ (signal.cpp:28)

    @property
    def n_mc(self):
        return self.samples.shape[0] if self.n_mc_total is None else self.n_mc_total


class Workload:
    def __init__(self, name, nobs, lower, upper, nbins, signals, systematics, syst_sigmas, events,
                 description):
        self.name = name
        self.nobs = nobs
        self.lower, self.upper, self.nbins = list(lower), list(upper), list(nbins)
        self.signals = signals
        # systematics: list of dicts (type, obs, true_obs, pars) with pars = indices relative to
        # the start of the systematic block of the parameter vector (mcmc.cpp:238 binds the
        # evaluators at offset nsources)
        self.systematics = systematics
        self.syst_sigmas = list(syst_sigmas)
        self.events = events                # float32 [E, nobs+1] (last column = dataset id)
        self.description = description

    @property
    def nsignals(self):
        return len(self.signals)

    @property
    def nsources(self):
        return 1 + max(s.source_id for s in self.signals)

    @property
    def nsyst_pars(self):
        return len(self.syst_sigmas)

    @property
    def nparameters(self):
        return self.nsources + self.nsyst_pars

    @property
    def nsamples_total(self):
        return sum(s.n_mc for s in self.signals)

    def parameter_means(self):
        return np.concatenate([np.ones(self.nsources), np.zeros(self.nsyst_pars)])

    def parameter_sigmas(self):
        return np.concatenate([np.zeros(self.nsources), np.asarray(self.syst_sigmas, dtype=np.float64)])


def _events_from_mixture(rng, signals, nobs, nevents, lower, upper):
    """Draw data events from the signals' own samples (observable columns), dataset id 0."""
    per = max(1, nevents // len(signals))
    rows = []
    for s in signals:
        idx = rng.integers(0, s.n_mc, size=per)
        rows.append(s.samples[idx, :nobs])
    ev = np.concatenate(rows, axis=0)[:nevents]
    if ev.shape[0] < nevents:
        ev = np.concatenate([ev, ev[: nevents - ev.shape[0]]], axis=0)
    out = np.zeros((nevents, nobs + 1), dtype=np.float32)
    out[:, :nobs] = ev
    return out


def split_counts(total, parts):
    base = total // parts
    counts = [base] * parts
    counts[-1] += total - base * parts
    return counts


def config1(scale=1.0, seed=1):
    """C1 plumbing: S=2, N=1e4, D=1 energy in [5,15) 10 bins, fields [energy, mc_energy, DATASET],
    scale + resolution_scale systematics (config/example.json:15-50 restated)."""
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(split_counts(int(10000 * scale), 2)):
        mc_e = rng.uniform(4.0, 16.0, size=n)
        e = mc_e + rng.normal(0.0, 0.5, size=n)
        tab = np.stack([e, mc_e, np.zeros(n)], axis=1).astype(np.float32)
        signals.append(Signal(tab, 3, nexpected=50.0 * (j + 1), source_id=j))
    systs = [dict(type="scale", obs=0, pars=[0]),
             dict(type="resolution_scale", obs=0, true_obs=1, pars=[1])]
    ev = _events_from_mixture(rng, signals, 1, 200, [5.0], [15.0])
    return Workload("C1", 1, [5.0], [15.0], [10], signals, systs, [1e-2, 1e-3], ev,
                    "S=2, N=1e4, D=1, 10 bins, scale+resolution_scale")


def config2(scale=1.0, seed=2, nevents=100000):
    """C2: N=1e7, D=2 (energy, radius), S=6, no systematics, 50x50 bins on [0,10)x[0,6)."""
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(split_counts(int(1e7 * scale), 6)):
        e = rng.normal(3.0 + j, 1.5, size=n)
        r = 6.0 * rng.uniform(0.0, 1.0, size=n) ** (1.0 / 3.0)
        tab = np.stack([e, r, np.zeros(n)], axis=1).astype(np.float32)
        signals.append(Signal(tab, 3, nexpected=100.0 + 10 * j, source_id=j))
    ev = _events_from_mixture(rng, signals, 2, nevents, None, None)
    return Workload("C2", 2, [0.0, 0.0], [10.0, 6.0], [50, 50], signals, [], [], ev,
                    "S=6, N=1e7, D=2, 50x50 bins, no systematics")


C3_SYSTS = [dict(type="shift", obs=1, pars=[0]),
            dict(type="scale", obs=0, pars=[1]),
            dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
C3_SIGMAS = [0.05, 0.01, 0.05]


def config3_signal_table(rng, j, n):
    """One C3 signal: fields [e, r, c, e_true, DATASET]."""
    e_true = rng.normal(2.0 + 0.5 * j, 1.2, size=n)
    e = e_true + rng.normal(0.0, 0.3, size=n)
    r = 6.0 * rng.uniform(0.0, 1.0, size=n) ** (1.0 / 3.0)
    c = rng.uniform(-1.0, 1.0, size=n)
    return np.stack([e, r, c, e_true, np.zeros(n)], axis=1).astype(np.float32)


def config3(scale=1.0, seed=3, nevents=100000):
    """C3: N=1e8, D=3, S=12, F=5, 20^3 bins on [0,10)x[0,6)x[-1,1), shift(r) + scale(e) +
    resolution_scale(e | e_true), one coefficient each, P = 12 + 3."""
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(split_counts(int(1e8 * scale), 12)):
        signals.append(Signal(config3_signal_table(rng, j, n), 5, nexpected=80.0 + 5 * j, source_id=j))
    ev = _events_from_mixture(rng, signals, 3, nevents, None, None)
    return Workload("C3", 3, [0.0, 0.0, -1.0], [10.0, 6.0, 1.0], [20, 20, 20], signals, C3_SYSTS,
                    C3_SIGMAS, ev, "S=12, N=1e8, D=3, 20^3 bins, shift+scale+resolution_scale")


def config5(scale=1.0, seed=5, nevents=100000, nbins=(200, 200, 200, 4, 4)):
    """C5 stress: D=5, N=1e9 over S=20, F=7 (5 obs + truth + DATASET); literal 200^5 bins is
    infeasible, (200,200,200,4,4) = 1.28e8 bins is the HBM-resident-histogram regime."""
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(split_counts(int(round(1e9 * scale)), 20)):
        e_true = rng.normal(2.0 + 0.3 * j, 1.2, size=n)
        e = e_true + rng.normal(0.0, 0.3, size=n)
        cols = [e, 6.0 * rng.uniform(size=n) ** (1.0 / 3.0), rng.uniform(-1, 1, size=n),
                rng.uniform(0, 1, size=n), rng.uniform(0, 1, size=n), e_true, np.zeros(n)]
        signals.append(Signal(np.stack(cols, axis=1).astype(np.float32), 7, nexpected=50.0 + j, source_id=j))
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])]
    ev = _events_from_mixture(rng, signals, 5, nevents, None, None)
    return Workload("C5", 5, [0.0, 0.0, -1.0, 0.0, 0.0], [10.0, 6.0, 1.0, 1.0, 1.0], list(nbins), signals,
                    systs, C3_SIGMAS, ev, "S=20, N=1e9, D=5, fine bins")


def bench_pdfz(scale=1.0, seed=7, nevents=100000):
    """The reference's own benchmark shape (bench/bench_sxmc.cpp:34-102): 1e7 N(0,1) samples, 1-D,
    1000 bins on [-3,3), one shift systematic, 1e5 clamped-gaussian evaluation points."""
    rng = np.random.default_rng(seed)
    n = int(1e7 * scale)
    tab = rng.normal(size=(n, 1)).astype(np.float32)
    pts = rng.normal(size=nevents * 2).astype(np.float32)
    pts = pts[(pts >= -3.0) & (pts < 3.0)][:nevents]
    ev = np.zeros((pts.size, 2), dtype=np.float32)
    ev[:, 0] = pts
    sig = [Signal(tab, 1, nexpected=100.0, source_id=0)]
    return Workload("bench_pdfz", 1, [-3.0], [3.0], [1000], sig, [dict(type="shift", obs=0, pars=[0])],
                    [0.1], ev, "bench_sxmc pdfz: N=1e7, 1-D, 1000 bins, 1 shift systematic")


GROUP_SIZES = [1e3, 2e5, 1e4, 1e3, 1e3, 3e6, 5e5, 1e6, 8e4, 2e4] + [1e3] * 19    # bench_sxmc.cpp:121-151


def bench_pdfz_group(scale=1.0, seed=8, nevents=100000):
    """The reference's second benchmark shape (bench/bench_sxmc.cpp:105-225): 29 evaluators with 1e3 ... 3e6
    N(0,1) samples each, 1-D, 1000 bins on [-3,3), one shift systematic per PDF, launched all-then-wait."""
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(GROUP_SIZES):
        tab = rng.normal(size=(max(1, int(n * scale)), 1)).astype(np.float32)
        signals.append(Signal(tab, 1, nexpected=10.0 + j, source_id=j))
    pts = rng.normal(size=nevents * 2).astype(np.float32)
    pts = pts[(pts >= -3.0) & (pts < 3.0)][:nevents]
    ev = np.zeros((pts.size, 2), dtype=np.float32)
    ev[:, 0] = pts
    return Workload("bench_pdfz_group", 1, [-3.0], [3.0], [1000], signals, [dict(type="shift", obs=0, pars=[0])],
                    [0.1], ev, "bench_sxmc pdfz_group: 29 PDFs of 1e3..3e6 samples, 1-D, 1000 bins, 1 shift systematic")
