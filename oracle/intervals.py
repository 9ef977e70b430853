"""Brute-force restatement of the reference's interval extraction, for the tests.  TEST INFRASTRUCTURE ONLY (like the
rest of oracle/): plain Python loops over the chain, statement by statement after

  LikelihoodSpace::get_contour    src/likelihood.cpp:90-102
  Contour::Contour / get_interval src/error_estimators/contour.cpp:17-69
  Projection::get_interval        src/error_estimators/projection.cpp:14-77   (+ TH1 conventions it relies on)

PARITY UNPINNED: the reference holds no test or fixture for these, and they lean on ROOT (TTree::Draw's text
selections, TH1::Fit, TMath::ChisquareQuantile), which is not in this image.  What IS reproduced exactly is the
arithmetic the reference's own statements fix:

* the selections are TEXT -- `sel << "likelihood+" << -lmin << "<" << delta` -- so the offset and the threshold that
  TTree::Draw / CopyTree apply are the numbers as `ostream << float` prints them (6 significant digits), evaluated in
  double per row;
* TAxis::FindBin: bin = 1 + int(nbins * (x - xmin) / (xmax - xmin)); TH1::Integral(a, b) sums bin contents a..b,
  Integral(0, -1) everything incl. under/overflow; GetBinLowEdge(i) = xmin + (i - 1) * width.

Not reproducible without ROOT and therefore a stated deviation shared with the product code: the histogram's range
and binning (TTree::Draw's "nice limits"; here 100 bins over [min, max], the maximum counted in the last bin) and
Minuit's path to the chi2 minimum of TH1::Fit("gaus") (here scipy's MINPACK Levenberg-Marquardt on the same chi2).
"""
import math

import numpy as np


def printed(x):
    """`ostream << float`: default floatfield, precision 6 == printf %g."""
    return "%g" % float(np.float32(x))


def chisquare_quantile_1dof(cl):
    """TMath::ChisquareQuantile(cl, 1): the square of the standard normal quantile at (1 + cl) / 2."""
    from scipy import stats
    return float(stats.chi2.ppf(float(cl), 1))


def select(likelihood, lmin, threshold):
    """Rows passing `"likelihood+" << -lmin << "<" << threshold` (both numbers as printed; the formula is evaluated
    in double on the float branch value)."""
    off, thr = float(printed(-np.float32(lmin))), float(printed(np.float32(threshold)))
    return [i for i, v in enumerate(likelihood) if float(v) + off < thr]


def contour_intervals(chain, cl):
    """-> list of (point_estimate, lower, upper, coverage) per parameter.  chain: [n, P + 1] float32, last column
    the likelihood."""
    chain = np.asarray(chain, np.float32)
    n, ncol = chain.shape
    like = [chain[i, ncol - 1] for i in range(n)]
    delta = np.float32(0.5 * chisquare_quantile_1dof(np.float32(cl)))      # contour.cpp:19 (a float)
    lmin = min(like)                                                        # likelihood.cpp:91
    rows = select(like, lmin, delta)                                        # likelihood.cpp:93-97 (CopyTree)
    if not rows:
        raise AssertionError("the reference asserts here (likelihood.cpp:99): the printed offset lost the minimum")
    clike = [like[i] for i in rows]
    out = []
    cmin = min(clike)                                                       # contour.cpp:40 (over the contour points)
    dnll = np.float32(0.13)
    while True:                                                             # contour.cpp:43-53
        near = [rows[k] for k in select(clike, cmin, dnll)]
        dnll = np.float32(dnll * np.float32(5))
        if len(near) >= 1:
            break
    for p in range(ncol - 1):
        nv = [chain[i, p] for i in near]
        cv = [chain[i, p] for i in rows]
        point = np.float32((np.float32(min(nv)) + np.float32(max(nv))) / np.float32(2))   # contour.cpp:56-57 (floats)
        out.append((float(point), float(min(cv)), float(max(cv)), -999.0))
    return out


def gaus_chi2_fit(centers, counts):
    """The minimum of what TH1::Fit("gaus") minimises: chi2 over the non-empty bins, errors sqrt(n), the function
    at the bin centre; started from the histogram's maximum, mean and RMS.  scipy / MINPACK."""
    from scipy import optimize
    x = np.array([c for c, y in zip(centers, counts) if y > 0], np.float64)
    y = np.array([y for y in counts if y > 0], np.float64)
    if x.size < 3:
        return None
    mean = float(np.sum(x * y) / np.sum(y))
    rms = math.sqrt(max(float(np.sum(y * (x - mean) ** 2) / np.sum(y)), 0.0))
    if not rms > 0:
        return None

    def res(q):
        return (y - q[0] * np.exp(-0.5 * ((x - q[1]) / q[2]) ** 2)) / np.sqrt(y)
    sol = optimize.least_squares(res, [float(y.max()), mean, rms], method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                                 max_nfev=20000)
    return tuple(float(v) for v in sol.x) if sol.x[2] > 0 else None


def projection_interval(values, cl, nbins=100):
    """Projection::get_interval for one parameter -> (point_estimate, lower, upper, coverage, one_sided)."""
    vals = [float(np.float32(v)) for v in values]
    xmin, xmax = min(vals), max(vals)
    if not xmax > xmin:
        return xmin, xmin, xmax, 1.0, False
    width = (xmax - xmin) / nbins
    content = [0.0] * (nbins + 2)                        # TH1: 0 underflow, 1..nbins, nbins + 1 overflow
    for v in vals:
        b = 1 + int(nbins * (v - xmin) / (xmax - xmin))  # TAxis::FindBin
        content[min(b, nbins)] += 1.0                    # (the maximum belongs to the last bin: see the module text)

    def integral(a, b):                                  # TH1::Integral(a, b); (0, -1): everything
        if b < 0:
            b = nbins + 1
        return sum(content[max(a, 0):min(b, nbins + 1) + 1])

    def low_edge(i):
        return xmin + (i - 1) * width
    centers = [xmin + (i - 0.5) * width for i in range(1, nbins + 1)]
    fit = gaus_chi2_fit(centers, content[1:nbins + 1])
    mu = fit[1] if fit is not None else centers[max(range(nbins), key=lambda i: content[i + 1])]
    imax = 1 + int(math.floor(nbins * (mu - xmin) / (xmax - xmin))) if xmin <= mu < xmax else (0 if mu < xmin else nbins + 1)
    point = mu
    if imax < 1:                                         # projection.cpp:28-31
        imax, point = 1, low_edge(1)
    imax = min(imax, nbins)                              # (a mean at or beyond the last edge: the last bin)
    total = integral(0, -1)
    ilo, ihi = 1, 0
    if integral(0, imax) / total < cl / 2:               # projection.cpp:38-47
        one_sided = True
        for i in range(0, nbins + 1):
            if integral(0, i) / total >= cl:
                ihi = i
                break
    else:
        one_sided = False
        for i in range(imax, 0, -1):                     # projection.cpp:52-58
            if integral(i, imax) / total >= cl / 2:
                ilo = i
                break
        for i in range(imax + 1, nbins + 1):             # projection.cpp:61-67
            if integral(imax + 1, i) / total >= cl / 2:
                ihi = i
                break
    if ihi == 0:                                         # (no bin reached the level: the reference would report bin 0's
        ihi = nbins                                      # edges; product and oracle close the interval at the last bin)
    ihi = max(ihi, ilo)
    coverage = integral(ilo, ihi) / total
    return point, low_edge(ilo), low_edge(ihi) + width, coverage, one_sided


def correlation_matrix(chain):
    """get_correlation_matrix, src/utils.cpp:29-77, statement by statement: float accumulators over the rows in order,
    means = sum / nentries, the entries on and to the right of the diagonal t / sqrt(dx2 * dy2) with the square root
    taken in double (TMath::Sqrt), everything else left at 0.  chain: [n, P + 1], last column the likelihood."""
    chain = np.asarray(chain, np.float32)
    n, P = chain.shape[0], chain.shape[1] - 1
    f = np.float32
    means = [f(0)] * P
    for k in range(n):
        for j in range(P):
            means[j] = f(means[j] + chain[k, j])
    means = [f(m / f(n)) for m in means]
    out = [[0.0] * P for _ in range(P)]
    for i in range(P):
        for j in range(i, P):
            t = dx2 = dy2 = f(0)
            for k in range(n):
                x1 = f(chain[k, i] - means[i])
                x2 = f(chain[k, j] - means[j])
                t = f(t + f(x1 * x2))
                dx2 = f(dx2 + f(x1 * x1))
                dy2 = f(dy2 + f(x2 * x2))
            out[i][j] = float(f(float(t) / math.sqrt(float(f(dx2 * dy2)))))
    return out


def interval_text(point, lower, upper):
    """Interval::str, src/interval.cpp:6-20, two-sided form (a string stream of its own: six significant digits)."""
    f = np.float32
    g = lambda x: "%g" % float(f(x))
    return "%s -%s +%s" % (g(point), g(f(point) - f(lower)), g(f(upper) - f(point)))
