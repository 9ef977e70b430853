"""Soak test of the BOXED form of the fill (runs on the GPU box): random shapes, random programs -- one observable with
1-3 one-coefficient systematics of which at least one is a resolution scale against a truth field (the boxed one),
one other observable with 1-2 shift / scale / cos-theta scale (streamed as codes), the rest untouched --, parameters
from tiny to wild, tables whose truth field is near or far from the observable, values that are not finite, long runs of
equal values.  The boxed form must give the histograms and norms of the ordered / bucketed form of the same launch, bit
for bit, and of the oracle for every fourth case.  Usage: python tools/soak_boxed.py [first] [count]"""
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
    nobs = int(rng.integers(2, 5))
    nfields = nobs + 2
    truth = nobs                                      # the truth field
    nbins = [int(rng.choice([3, 5, 8, 20, 40])) for _ in range(nobs)]
    while int(np.prod(nbins)) > 20000:
        nbins[int(np.argmax(nbins))] //= 2
    bx, st = [int(k) for k in rng.choice(nobs, size=2, replace=False)]
    systs, npar = [], 0
    kinds = ["resolution_scale"] + [["shift", "scale", "ctscale", "resolution_scale"][int(rng.integers(0, 4))]
                                    for _ in range(int(rng.integers(0, 3)))]
    rng.shuffle(kinds)
    for kind in kinds:
        d = dict(type=kind, obs=bx, pars=[npar])
        npar += 1
        if kind == "resolution_scale":
            d["true_obs"] = truth
        systs.append(d)
    for _ in range(int(rng.integers(1, 3))):
        systs.insert(int(rng.integers(0, len(systs) + 1)),
                     dict(type=["shift", "scale", "ctscale"][int(rng.integers(0, 3))], obs=st, pars=[npar]))
        npar += 1
    n = int(rng.choice([3000, 70000, 300000]))
    lo, hi = [0.0] * nobs, [1.0] * nobs
    tab = rng.uniform(-0.2, 1.2, size=(n, nfields)).astype(np.float32)
    spread = float(rng.choice([0.0, 0.01, 0.1, 1.0]))
    tab[:, truth] = (tab[:, bx] + rng.normal(0, 1, n) * spread).astype(np.float32)
    if rng.uniform() < 0.5:
        idx = rng.choice(n, size=max(1, n // 200), replace=False)
        special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e30, -3e38], np.float32)
        tab[idx, int(rng.choice([bx, st, truth]))] = rng.choice(special, size=idx.size)
    if rng.uniform() < 0.5:
        a = int(rng.integers(0, max(1, n - 700)))
        tab[a:a + 600, bx] = tab[a, bx]
    tab[:, nfields - 1] = 0.0
    param_sets = []
    for _ in range(3):
        scale = float(rng.choice([1e-9, 1e-3, 0.02, 0.3, 3.0]))
        p = rng.normal(0, scale, npar)
        if rng.uniform() < 0.1:
            p[int(rng.integers(0, npar))] = float(rng.choice([np.nan, np.inf, -1.0, -2.5, 1e200]))
        param_sets.append(p)
    ev = pdfz.EvalHist(tab, nfields, nobs, lo, hi, nbins)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.asarray(param_sets[0], np.float64))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    group.SetBoxes(True)
    boxed = "boxed" in group.LaunchInfo()
    bad = []
    for k, p in enumerate(param_sets):
        pbuf.set(np.asarray(p, np.float64))
        res = []
        for boxes in ((True, False) if boxed else (False,)):
            group.SetBoxes(boxes)
            group.EvalAsync(False)
            group.EvalFinished()
            res.append((ev.GetBins(), int(norm.get()[0])))
        if boxed and not (np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]):
            bad.append("seed %d set %d: boxed != ordered" % (seed, k))
        if seed % 4 == 0:
            geom = oracle.HistGeometry(lo, hi, nbins)
            ob, on = oracle.bin_samples(geom, tab, nfields, systs, np.asarray(p, np.float64))
            if not (np.array_equal(res[0][0], ob) and res[0][1] == on):
                bad.append("seed %d set %d: != oracle" % (seed, k))
    return boxed, bad


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    nboxed, failures = 0, []
    for seed in range(first, first + count):
        boxed, bad = one_case(seed)
        nboxed += int(boxed)
        for b in bad:
            print(b, flush=True)
        failures += bad
        if (seed - first) % 25 == 24:
            print("... %d cases, %d boxed, %d mismatches" % (seed - first + 1, nboxed, len(failures)), flush=True)
    print("soak_boxed: %d cases (%d took the boxed form) x 3 parameter sets, %d mismatches" % (count, nboxed, len(failures)))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
