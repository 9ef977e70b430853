"""Soak test of the ordered-observable fill (runs on the GPU box): random shapes, programs and parameters, sample
values hugging the transformed bin edges; the ordered evaluation must give the histograms and norms of the
unordered one, bit for bit (and of the oracle for every tenth case).  Usage: python tools/soak_ordered.py [first] [count]
Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import oracle  # noqa: E402  (checker only)
from sxmc_amd import nll, pdfz  # noqa: E402
from sxmc_amd.capi import DeviceArray  # noqa: E402
from sxmc_amd.mcmc import make_systematic  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    nobs = int(rng.integers(1, 4))
    nextra = int(rng.integers(0, 2))
    nfields = nobs + nextra + 1
    nbins = [int(rng.choice([2, 3, 7, 20, 50, 200])) for _ in range(nobs)]
    while int(np.prod(nbins)) > 40000:
        nbins[int(np.argmax(nbins))] //= 2
    ordered = int(rng.integers(0, nobs))
    kinds = ["shift", "scale", "ctscale"]
    systs, npar = [], 0
    for _ in range(int(rng.integers(1, 3))):                          # one or two monotone systematics on `ordered`
        systs.append(dict(type=kinds[int(rng.integers(0, 3))], obs=ordered, pars=[npar]))
        npar += 1
    for k in range(nobs):                                            # maybe something on another observable
        if k != ordered and rng.uniform() < 0.5:
            kind = ["scale", "resolution_scale", "shift"][int(rng.integers(0, 3))]
            d = dict(type=kind, obs=k, pars=list(range(npar, npar + (2 if rng.uniform() < 0.3 else 1))))
            npar += len(d["pars"])
            if kind == "resolution_scale":
                choices = [f for f in range(nobs + nextra) if f not in (k, ordered)]
                if not choices:
                    continue
                d["true_obs"] = int(rng.choice(choices))
            systs.append(d)
    n = int(rng.choice([300, 5000, 70001, 300000]))
    tab = rng.uniform(-0.2, 1.2, size=(n, nfields)).astype(np.float32)
    # the ordered column: values on and within ulps of bin edges (of the untransformed grid and of a shifted one),
    # long runs of one value, specials
    edges = (np.arange(nbins[ordered] + 1) / nbins[ordered]).astype(np.float32)
    col = tab[:, ordered].copy()
    m = rng.uniform(size=n) < 0.5
    col[m] = rng.choice(edges, size=int(m.sum())) - np.float32(rng.choice([0.0, 0.0137, -0.021]))
    for _ in range(2):
        up = rng.uniform(size=n) < 0.5
        mv = rng.uniform(size=n) < 0.4
        col = np.where(mv, np.nextafter(col, np.where(up, np.float32(9), np.float32(-9))), col).astype(np.float32)
    if n > 3000:
        col[100:100 + 700] = col[100]
    special = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0], np.float32)
    sp = rng.uniform(size=n) < 0.01
    col[sp] = rng.choice(special, size=int(sp.sum()))
    tab[:, ordered] = col
    tab[:, -1] = 0.0
    ev = pdfz.EvalHist(tab, nfields, nobs, [0.0] * nobs, [1.0] * nobs, nbins)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.zeros(max(npar, 1)))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    bad = []
    for trial in range(4):
        params = rng.normal(0, 0.03, max(npar, 1))
        if trial == 1:
            params[0] = rng.choice([0.0137, -0.021, 0.0, -2.5, -1.0])     # edge-aligned shifts, reversed / zero scales
        if trial == 3:
            params = rng.normal(0, 1.0, max(npar, 1))
        pbuf.set(params)
        got = {}
        for order in (True, False):
            group.SetOrdering(order, force=True)
            info = group.LaunchInfo()
            if order and "ordered" not in info:
                got = None
                break
            group.EvalAsync(False)
            group.EvalFinished()
            got[order] = (ev.GetBins(), int(norm.get()[0]))
        if got is None:
            break
        if not (np.array_equal(got[True][0], got[False][0]) and got[True][1] == got[False][1]):
            bad.append((seed, trial, "ordered != unordered", systs, list(params), nbins, n))
        if seed % 10 == 0 and trial < 2:
            geom = oracle.HistGeometry([0.0] * nobs, [1.0] * nobs, nbins)
            bins, nrm = oracle.bin_samples(geom, tab, nfields, systs, params)
            if not (np.array_equal(got[True][0], bins) and got[True][1] == nrm):
                bad.append((seed, trial, "ordered != oracle", systs, list(params), nbins, n))
    used = got is not None
    group.close()
    ev.close()
    return bad, used


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    failures, ordered_cases = [], 0
    for seed in range(first, first + count):
        bad, used = one_case(seed)
        ordered_cases += int(used)
        for b in bad:
            print("FAIL", b, flush=True)
        failures += bad
        if (seed - first) % 50 == 49:
            print("... %d cases, %d ordered, %d failures" % (seed - first + 1, ordered_cases, len(failures)), flush=True)
    print("soak_ordered: %d cases (%d ran the ordered fill), %d failures" % (count, ordered_cases, len(failures)))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
