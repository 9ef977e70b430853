"""The ensemble layer around the hot path: fake data sets, interval extraction, and the experiment
loop that shards over GPUs.  ROOT-free restatements of

  make_fake_dataset                 src/generator.cpp:10-48
  EvalHist::RandomSample            src/pdfz.cpp:817-922   (ROOT TH1::GetRandom / GetRandom2 / GetRandom3)
  Signal::get_efficiency            src/signal.cpp:172-199
  LikelihoodSpace::get_contour      src/likelihood.cpp:90-102
  Contour::get_interval             src/error_estimators/contour.cpp:18-69
  Projection::get_interval          src/error_estimators/projection.cpp:14-77 (see note)
  ensemble()                        src/sxmc.cpp:44-145

Parity with the reference is statistical only ("parity unpinned"): every one of these draws on ROOT
(TRandom, TH1::GetRandom, TMath::ChisquareQuantile, TH1::Fit) and the reference holds no test for them.
Poisson and uniform deviates come from numpy's PCG64 here.  Note on Projection: the reference fits a
Gaussian to an auto-binned TH1F with ROOT's minimiser to find the mode; here the mode is the vertex of a
parabola fitted to the log-counts of the bins above half maximum of a 100-bin histogram (a Gaussian fit
to the peak); the interval walk over the bins is the reference's.
"""
import math

import numpy as np

from . import capi
from .capi import DeviceArray
from .mcmc import MCMC

INTERVAL_FIELDS = ("point_estimate", "lower", "upper", "coverage")


# ------------------------------------------------------------------------------------ fake data
def get_efficiency(ev, nsyst_pars, syst_means, n_mc=None, want_bins=True):
    """Signal::get_efficiency (signal.cpp:172-199): in-domain count with every systematic at its mean,
    divided by the signal's n_mc -- the number of simulated events BEFORE cuts (signal.cpp:198), which is
    what the NLL kernels divide by too.  n_mc=None: the evaluator's row count (no cuts were applied).
    Leaves the evaluator bound to scratch buffers (kept alive by the evaluator until it is re-bound);
    returns (efficiency, bins, in-domain count)."""
    params = DeviceArray(np.asarray(syst_means, dtype=np.float64) if nsyst_pars else np.zeros(1))
    norm = DeviceArray.zeros(1, np.uint32)
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(params)
    ev.EvalAsync(False)
    ev.EvalFinished()
    n = int(norm.get()[0])
    return n / float(ev.nsamples if n_mc is None else n_mc), (ev.GetBins() if want_bins else None), n


def random_sample(rng, bins, lower, upper, nbins, nobserved):
    """TH1::GetRandom (1-3 D): pick a bin with probability proportional to its content, then a point
    uniform inside the bin.  bins: flat row-major counts."""
    nbins = np.asarray(nbins)
    D = nbins.size
    if D > 3:
        raise ValueError("Cannot sample histograms of more than 3 dimensions")   # pdfz.cpp:499-501
    total = float(bins.sum())
    if nobserved == 0 or total <= 0:
        return np.zeros((0, D), np.float32)
    cdf = np.cumsum(bins, dtype=np.float64) / total
    flat = np.searchsorted(cdf, rng.random(nobserved), side="right")
    flat = np.minimum(flat, bins.size - 1)
    idx = np.stack(np.unravel_index(flat, nbins), axis=1)
    width = (np.asarray(upper, np.float64) - np.asarray(lower, np.float64)) / nbins
    pts = np.asarray(lower, np.float64) + (idx + rng.random((nobserved, D))) * width
    return pts.astype(np.float32)


def make_fake_dataset(rng, workload, evaluators, poisson=True):
    """generator.cpp:10-48: per signal nexpected x efficiency events (Poisson fluctuated), drawn from
    the signal's histogram at the mean systematics; rows of nobs + 1 floats (last = dataset id)."""
    w = workload
    syst_means = w.parameter_means()[w.nsources:]
    rows, observed = [], []
    for sig, ev in zip(w.signals, evaluators):
        if w.nobs > 3:
            raise ValueError("Cannot sample histograms of more than 3 dimensions")   # pdfz.cpp:499-501
        eff, _, indomain = get_efficiency(ev, w.nsyst_pars, syst_means, sig.n_mc, want_bins=False)
        nevents = sig.nexpected * eff
        n = int(rng.poisson(nevents)) if poisson else int(math.floor(nevents + 0.5))
        if indomain == 0:
            n = 0                                        # an empty histogram yields no events
        # drawn on the device from the histogram get_efficiency just filled: it never leaves HBM
        rows.append(ev.RandomSample(n, int(rng.integers(0, 2 ** 63 - 1))) if n else
                    np.zeros((0, w.nobs + 1), np.float32))
        observed.append(n)
    return np.concatenate(rows, axis=0).astype(np.float32), observed


# ------------------------------------------------------------------------------------ intervals
def chisquare_quantile_1dof(cl):
    """TMath::ChisquareQuantile(cl, 1) = (Phi^-1((1 + cl) / 2))^2, by bisection on erf."""
    target = cl
    lo, hi = 0.0, 40.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if math.erf(math.sqrt(mid / 2.0)) < target:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def as_printed(v):
    """What `ostream << float` writes (6 significant digits, %g), read back.  The reference builds its selections
    as TEXT -- "likelihood+" << -lmin << "<" << delta (likelihood.cpp:93-94, contour.cpp:45-46) -- so the offset
    and threshold it actually applies are the printed, rounded ones: with |lmin| of a few 1e5 (BASELINE config 3)
    the offset is off by up to 0.5, which moves the contour.  Reproduced because the intervals are results."""
    return float("%g" % float(np.float32(v)))


def contour_intervals(chain, cl=0.9):
    """Contour::get_interval for every parameter (contour.cpp:17-69, likelihood.cpp:90-102).  chain: [n, P + 1]
    float32 (last column = NLL).  Returns float32 [P, 4]: point_estimate, lower, upper, coverage (-999 as in
    the reference)."""
    chain = np.asarray(chain, np.float32)
    nll = chain[:, -1].astype(np.float64)
    lmin = np.float32(chain[:, -1].min())
    delta = np.float32(0.5 * chisquare_quantile_1dof(cl))                     # contour.cpp:19: a float
    inside = nll + as_printed(-lmin) < as_printed(delta)                      # likelihood.cpp:90-102
    if not inside.any():       # (the reference asserts here: the printed offset lost the minimum; use the exact one)
        inside = chain[:, -1] - lmin < delta
    contour = chain[inside]
    cn = contour[:, -1].astype(np.float64)
    coff = as_printed(-np.float32(contour[:, -1].min()))
    dnll = np.float32(0.13)
    while True:                                              # contour.cpp:39-53: 0.13, 0.65, 3.25, ...
        near = contour[cn + coff < as_printed(dnll)]
        dnll = np.float32(dnll * np.float32(5))
        if near.shape[0] >= 1:
            break
    P = chain.shape[1] - 1
    out = np.zeros((P, 4), np.float32)
    out[:, 0] = (near[:, :P].min(axis=0) + near[:, :P].max(axis=0)) / 2
    out[:, 1] = contour[:, :P].min(axis=0)
    out[:, 2] = contour[:, :P].max(axis=0)
    out[:, 3] = -999
    return out


def _g(x, precision=6):
    """`ostream << float` at the stream's precision (general format)."""
    return "%.*g" % (precision, float(np.float32(x)))


def interval_str(row, cl=0.9, one_sided=False):
    """Interval::str (interval.cpp:6-20) of (point_estimate, lower, upper, ...): "point -lower_error +upper_error", or
    "point <upper (cl% CL)" for a one-sided interval (its own string stream there: always six significant digits)."""
    point, lower, upper = (np.float32(row[k]) for k in range(3))
    if one_sided:
        return "%s <%s (%s%% CL)" % (_g(point), _g(upper), _g(np.float32(100) * np.float32(cl)))
    return "%s -%s +%s" % (_g(point), _g(point - lower), _g(upper - point))


def correlation_matrix(chain):
    """get_correlation_matrix (utils.cpp:29-77) of the chain's parameter columns (all but the last, the likelihood):
    float32 [P, P].  As there, sums accumulate in float32 in row order, the square root is a double's, and only the
    diagonal and the entries to its right are computed -- the ones below stay 0."""
    chain = np.asarray(chain, np.float32)
    n, P = chain.shape[0], chain.shape[1] - 1
    cols = chain[:, :P]

    def fsum(v):                      # a float accumulator over the rows, in order (np.sum would add pairwise)
        return np.cumsum(v, dtype=np.float32)[-1] if v.shape[0] else np.float32(0)
    means = np.array([fsum(cols[:, j]) / np.float32(n) for j in range(P)], np.float32)
    d = cols - means[None, :]
    out = np.zeros((P, P), np.float32)
    for i in range(P):
        for j in range(i, P):
            t, dx2, dy2 = fsum(d[:, i] * d[:, j]), fsum(d[:, i] * d[:, i]), fsum(d[:, j] * d[:, j])
            with np.errstate(invalid="ignore", divide="ignore"):
                out[i, j] = np.float32(np.float64(t) / np.sqrt(np.float64(np.float32(dx2 * dy2))))
    return out


def format_best_fit(names, intervals, nll_min, cl=0.9, one_sided=None, precision=6):
    """LikelihoodSpace::print_best_fit (likelihood.cpp:34-45): the parameters in NAME order (a std::map there: byte
    order), then the minimum of the likelihood column.  names: the chain's parameter columns; intervals: [P, >= 3].
    precision: the output stream's -- it reaches only the NLL line (print_correlations leaves it at 3, so from the
    second experiment on the reference prints the NLL with three digits)."""
    lines = ["-- Best fit --"]
    for name, k in sorted(((n, k) for k, n in enumerate(names) if n != "likelihood"), key=lambda t: t[0].encode()):
        lines.append(" %s: %s" % (name, interval_str(intervals[k], cl, bool(one_sided[k]) if one_sided is not None else False)))
    lines.append(" NLL: %s" % _g(nll_min, precision))
    return "\n".join(lines) + "\n"


def format_correlations(names, matrix):
    """LikelihoodSpace::print_correlations (likelihood.cpp:48-72): names in column order, right-aligned to the longest;
    entries fixed, three decimals, eight columns wide."""
    names = [n for n in names if n != "likelihood"]
    width = max([len(n) for n in names] + [0])
    lines = ["-- Correlation matrix --"]
    for i, n in enumerate(names):
        lines.append("%*s " % (width, n) + "".join("%8.3f" % float(matrix[i, j]) for j in range(len(names))))
    return "\n".join(lines) + "\n"


def gaus_fit(centers, counts, iterations=200):
    """What `TH1::Fit("gaus")` minimises (projection.cpp:22-23): chi2 = sum over the non-empty bins of
    ((n_i - A exp(-(x_i - mu)^2 / (2 sigma^2))) / sqrt(n_i))^2, the function taken at the bin centre, started
    from the histogram's maximum, mean and RMS as TH1's InitGaus does.  Minimised by Levenberg-Marquardt (ROOT:
    Minuit MIGRAD; the minimum is the same, the path is not).  Returns (A, mu, sigma) or None when it does not
    converge to a positive width."""
    x = np.asarray(centers, np.float64)
    y = np.asarray(counts, np.float64)
    keep = y > 0
    x, y = x[keep], y[keep]
    if x.size < 3:
        return None
    e = np.sqrt(y)
    mean = float(np.sum(x * y) / np.sum(y))
    rms = float(np.sqrt(max(np.sum(y * (x - mean) ** 2) / np.sum(y), 0.0)))
    if rms <= 0:
        return None
    p = np.array([float(y.max()), mean, rms])

    def residuals(q):
        return (y - q[0] * np.exp(-0.5 * ((x - q[1]) / q[2]) ** 2)) / e

    lam, chi2 = 1e-3, float(np.sum(residuals(p) ** 2))
    for _ in range(iterations):
        g = np.exp(-0.5 * ((x - p[1]) / p[2]) ** 2)
        jac = np.stack([g, p[0] * g * (x - p[1]) / p[2] ** 2, p[0] * g * (x - p[1]) ** 2 / p[2] ** 3], axis=1) / e[:, None]
        r = residuals(p)
        a, b = jac.T @ jac, jac.T @ r
        try:
            step = np.linalg.solve(a + lam * np.diag(np.diag(a) + 1e-300), b)
        except np.linalg.LinAlgError:
            return None
        trial = p + step
        c2 = float(np.sum(residuals(trial) ** 2)) if trial[2] > 0 else np.inf
        if c2 <= chi2:
            converged = chi2 - c2 <= 1e-12 * max(chi2, 1e-300) and np.all(np.abs(step) <= 1e-10 * (np.abs(p) + 1e-300))
            p, chi2, lam = trial, c2, max(lam * 0.3, 1e-12)
            if converged:
                break
        else:
            lam *= 10.0
            if lam > 1e12:
                break
    return (float(p[0]), float(p[1]), float(p[2])) if p[2] > 0 and np.all(np.isfinite(p)) else None


def projection_interval(values, cl=0.9, nbins=100):
    """Projection::get_interval on one parameter's samples (projection.cpp:14-77): the parameter's samples are
    histogrammed (ROOT's TTree::Draw picks the range and 100 bins by its own "nice limits" rule, which is not
    reproduced: [min, max] here), a Gaussian is fitted (gaus_fit) and its mean is the point estimate; the limits
    walk outwards from the mean's bin until cl / 2 of the samples lie on either side (or one-sided from the low
    edge when less than cl / 2 lies below the mean).  Returns (point_estimate, lower, upper, coverage, one_sided)."""
    values = np.asarray(values, np.float64)
    lo, hi = float(values.min()), float(values.max())
    if hi <= lo:
        return float(lo), float(lo), float(hi), 1.0, False
    # TH1 conventions (TAxis::FindBin, GetBinCenter, GetBinLowEdge): bin = 1 + int(nbins (x - xmin) / (xmax - xmin)),
    # the maximum counted in the last bin (see the docstring: the range is ours, not TTree::Draw's)
    width = (hi - lo) / nbins
    idx = np.minimum((nbins * (values - lo) / (hi - lo)).astype(np.int64), nbins - 1)
    counts = np.bincount(idx, minlength=nbins)
    total = counts.sum()
    centers = lo + (np.arange(nbins) + 0.5) * width
    edges = lo + np.arange(nbins + 1) * width
    fit = gaus_fit(centers, counts)
    mu = fit[1] if fit is not None else float(centers[np.argmax(counts)])     # (no fit: the mode's bin)
    # 1-based bin of the mean (TH1::FindBin): 0 below the range, nbins + 1 at or beyond its end
    imax = 0 if mu < lo else nbins + 1 if mu >= hi else 1 + int(nbins * (mu - lo) / (hi - lo))
    if imax < 1:                                                               # projection.cpp:28-31
        imax, mu = 1, float(edges[0])
    imax = min(imax, nbins)
    csum = np.concatenate([[0], np.cumsum(counts)])                            # csum[i] = bins 1..i
    ilo, ihi = 1, 0
    if csum[imax] / total < cl / 2:                                            # projection.cpp:36-45
        one_sided = True
        for i in range(0, nbins + 1):
            if csum[i] / total >= cl:
                ihi = i
                break
    else:
        one_sided = False
        for i in range(imax, 0, -1):
            if (csum[imax] - csum[i - 1]) / total >= cl / 2:
                ilo = i
                break
        for i in range(imax + 1, nbins + 1):
            if (csum[i] - csum[imax]) / total >= cl / 2:
                ihi = i
                break
    ihi = max(ihi, ilo) if ihi else nbins
    coverage = (csum[ihi] - csum[ilo - 1]) / total
    # projection.cpp:72-73: GetBinLowEdge(ilo); GetBinLowEdge(ihi) + GetBinWidth(ihi)
    return float(mu), float(edges[ilo - 1]), float(edges[ihi - 1] + width), float(coverage), one_sided


# ------------------------------------------------------------------------------------ the ensemble
def run_experiment(workload, seed, nsteps, burnin_fraction=0.1, cl=0.9, sync_interval=10000, mcmc=None,
                   form="fused", graph_steps=0, lookahead=False):
    """One iteration of the loop in sxmc.cpp:59-145: fake data -> MCMC -> intervals.
    Reuses `mcmc` (evaluators with the MC tables resident in HBM) across experiments when given.
    lookahead: walk with two evaluations per pass over the tables (MCMC.walk(lookahead=True): the same chain, about a
    quarter more steps per second when one experiment has the GPU to itself; `mcmc` then needs a created stream).
    Returns (intervals float32 [P, 4], chain, accepted)."""
    rng = np.random.default_rng(seed)
    if mcmc is None:
        mcmc = MCMC(workload, seed=seed & 0xFFFFFFFF, fused={"fused": True, "step": "step", "reference": False}[form],
                    lut_output=False, consume=True)
    else:
        mcmc.reseed(seed & 0xFFFFFFFF)
    data, _ = make_fake_dataset(rng, workload, mcmc.pdfs, poisson=True)
    chain, accepted = mcmc.walk(data, nsteps, burnin_fraction, sync_interval=sync_interval, graph_steps=graph_steps,
                                lookahead=lookahead and mcmc.stream is not None and mcmc.consume)
    return contour_intervals(chain, cl), chain, accepted


def run_experiments_in_lockstep(workload, seeds, nsteps, sets, burnin_fraction=0.1, cl=0.9, sync_interval=10000,
                                graph_steps=0, timing=None):
    """len(seeds) fake experiments at once on one GPU, the chains grouped into lockstep sets (mcmc.LockstepChains:
    the chains of a set share a stream and ONE fill pass per step; different sets run on different streams, so one
    set's step ends overlap another's fill).  Same schedule of re-tunings and flushes as a single walk; every
    chain walks what it walks alone.  Returns a list of (intervals, chain, accepted) in the order of `seeds`.
    timing (a dict, optional): "stepping_seconds" is increased by the wall time of the stepping alone -- from the last
    chain's set-up (fake data, evaluation points, first evaluation) to the last flush, re-tunings and flushes
    included -- which is what an experiment of BASELINE's 1e5 steps consists of."""
    import time
    chains = [m for st in sets for m in st.chains]
    assert len(seeds) == len(chains)
    for st in sets:
        st.drop_graph()                              # new data: recorded steps hold the old evaluation-point tables
    for m, seed in zip(chains, seeds):
        rng = np.random.default_rng(seed)
        m.reseed(seed & 0xFFFFFFFF)
        data, _ = make_fake_dataset(rng, workload, m.pdfs, poisson=True)
        m.walk_begin(data, nsteps, burnin_fraction, sync_interval=sync_interval)
    if timing is not None:
        capi.synchronize()
    t0 = time.perf_counter()
    i = 0
    for f in chains[0].flush_schedule():            # the same schedule for every chain
        for m in chains:
            m._retune_if_due(i)
        n = f - i + 1
        while n > 0:
            k = graph_steps if graph_steps > 0 and n >= graph_steps else n
            for st in sets:
                st.steps(k, graph_steps)
            n -= k
        for m in chains:
            m._flush_if_due(f)
        i = f + 1
    if timing is not None:
        capi.synchronize()
        timing["stepping_seconds"] = timing.get("stepping_seconds", 0.0) + (time.perf_counter() - t0)
    out = []
    for m in chains:
        chain, accepted = m.walk_end()
        out.append((contour_intervals(chain, cl), chain, accepted))
    return out


def run_experiments_concurrently(workload, seeds, nsteps, chains, burnin_fraction=0.1, cl=0.9, sync_interval=10000,
                                 graph_steps=0):
    """len(chains) fake experiments at once on one GPU (BASELINE config 4: one experiment per stream):
    the chains share one resident copy of the MC tables (MCMC(..., share_with=...), own non-blocking
    streams) and are advanced in turn, so one experiment's small kernels overlap another's fill.
    graph_steps = K > 0 advances each chain K recorded steps (one HIP-graph replay) per turn.
    Returns a list of (intervals, chain, accepted) in the order of `seeds`."""
    assert len(seeds) == len(chains)
    for m, seed in zip(chains, seeds):
        rng = np.random.default_rng(seed)
        m.reseed(seed & 0xFFFFFFFF)
        data, _ = make_fake_dataset(rng, workload, m.pdfs, poisson=True)
        m.walk_begin(data, nsteps, burnin_fraction, sync_interval=sync_interval)
    if graph_steps <= 0:
        for i in range(nsteps):
            for m in chains:
                m.walk_advance(i)
    else:
        i = 0
        for f in chains[0].flush_schedule():        # the same schedule for every chain
            for m in chains:
                m._retune_if_due(i)
            n = f - i + 1
            while n > 0:
                k = graph_steps if n >= graph_steps else n
                for m in chains:
                    m.steps(k, graph_steps)
                n -= k
            for m in chains:
                m._flush_if_due(f)
            i = f + 1
    out = []
    for m in chains:
        chain, accepted = m.walk_end()
        out.append((contour_intervals(chain, cl), chain, accepted))
    return out
