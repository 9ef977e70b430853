"""Soak test of the fill over CODES (runs on the GPU box): random shapes, programs of one-coefficient systematics
and parameters -- ordinary, large, edge-aligned --, tables with values outside the windows and values that are not
finite, and samples placed within ulps of where the program's affine map puts the bin edges.  The evaluation over
codes must give the histograms and norms of the same launch streaming the float columns, bit for bit, and of the
oracle for every fifth case.  Usage: python tools/soak_codes.py [first] [count]
Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import oracle  # noqa: E402  (checker only)
from sxmc_amd import nll, pdfz  # noqa: E402
from sxmc_amd.capi import DeviceArray  # noqa: E402
from sxmc_amd.mcmc import make_systematic  # noqa: E402


def affine(systs, params, nslot):
    """The program composed in float64: x_k = sum_m A[k][m] field_m + C[k] (for placing samples on edges)."""
    A, C = np.eye(nslot), np.zeros(nslot)
    for s in systs:
        p, k = params[s["pars"][0]], s["obs"]
        if s["type"] == "shift":
            C[k] += p
        elif s["type"] == "scale":
            A[k] *= 1 + p
            C[k] *= 1 + p
        elif s["type"] == "ctscale":
            A[k] *= 1 + p
            C[k] = 1 + (C[k] - 1) * (1 + p)
        else:
            e = s["true_obs"]
            A[k] = A[k] + p * (A[k] - A[e])
            C[k] = C[k] + p * (C[k] - C[e])
    return A, C


def one_case(seed):
    rng = np.random.default_rng(seed)
    nobs = int(rng.integers(2, 5))
    nextra = int(rng.integers(1, 3))
    nfields = nobs + nextra + 1
    nbins = [int(rng.choice([2, 3, 5, 7, 20, 50])) for _ in range(nobs)]
    while int(np.prod(nbins)) > 30000:
        nbins[int(np.argmax(nbins))] //= 2
    # the ordered observable: the one with the fewest bins among those written by monotone systematics only; make it
    # unique by construction -- observable `o` gets a shift / scale / cos-theta scale and strictly the fewest bins
    o = int(rng.integers(0, nobs))
    nbins[o] = max(2, min(nbins) - 1) if min(nbins) > 2 else 2
    for k in range(nobs):
        if k != o and nbins[k] <= nbins[o]:
            nbins[k] = nbins[o] + 1 + int(rng.integers(0, 5))
    systs, npar = [], 0
    systs.append(dict(type=["shift", "scale", "ctscale"][int(rng.integers(0, 3))], obs=o, pars=[npar]))
    npar += 1
    # one or two observables binned from codes: each written by 1-3 systematics, at least one resolution scale so that
    # two or more fields are streamed
    binned = [k for k in range(nobs) if k != o]
    rng.shuffle(binned)
    binned = binned[:int(rng.integers(1, min(2, len(binned)) + 1))]
    streamed = set(binned)
    for j, k in enumerate(binned):
        nops = int(rng.integers(1, 4))
        for i in range(nops):
            kind = ["shift", "scale", "ctscale", "resolution_scale"][int(rng.integers(0, 4))]
            if j == 0 and i == 0:
                kind = "resolution_scale"
            d = dict(type=kind, obs=k, pars=[npar if rng.uniform() < 0.8 else int(rng.integers(0, npar))])
            if d["pars"][0] == npar:
                npar += 1
            if kind == "resolution_scale":
                choices = [f for f in list(range(nobs, nobs + nextra)) + binned if f != k]
                others = [f for f in range(nobs) if f not in binned and f != o]     # an untouched observable as truth
                if others and rng.uniform() < 0.2:
                    choices = others
                d["true_obs"] = int(rng.choice(choices))
                streamed.add(d["true_obs"])
            systs.append(d)
    if len(streamed) < 2 or len(streamed) > 4:
        return [], False
    rng.shuffle(systs)
    n = int(rng.choice([3000, 70001, 300000, 600001]))
    tab = rng.uniform(-0.3, 1.3, size=(n, nfields)).astype(np.float32)
    tab[:, -1] = 0.0
    # outliers and values that are not finite in the streamed fields
    for f in streamed:
        far = rng.uniform(size=n) < 0.003
        tab[far, f] = rng.uniform(-40, 40, size=int(far.sum())).astype(np.float32)
        bad = rng.uniform(size=n) < 0.002
        tab[bad, f] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), size=int(bad.sum()))
    base_params = rng.normal(0, 0.05, max(npar, 1))
    # samples on the transformed bin edges of the first binned observable, for base_params
    k0 = binned[0]
    A, C = affine(systs, base_params, nobs + nextra)
    if abs(A[k0][k0]) > 1e-3:
        m = rng.uniform(size=n) < 0.4
        edges = rng.integers(0, nbins[k0] + 1, size=n) / nbins[k0]
        rest = C[k0] + sum(A[k0][f] * tab[:, f].astype(np.float64) for f in range(nobs + nextra) if f != k0)
        x = ((edges - rest) / A[k0][k0]).astype(np.float32)
        for _ in range(2):
            up = rng.uniform(size=n) < 0.5
            mv = rng.uniform(size=n) < 0.5
            x = np.where(mv, np.nextafter(x, np.where(up, np.float32(99), np.float32(-99))), x).astype(np.float32)
        ok = m & np.isfinite(x)
        tab[ok, k0] = x[ok]
    ev = pdfz.EvalHist(tab, nfields, nobs, [0.0] * nobs, [1.0] * nobs, nbins)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.zeros(max(npar, 1)))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    group.SetOrdering(True, force=True)
    group.SetCodes(True)
    if "ordered+codes" not in group.LaunchInfo():
        group.close()
        ev.close()
        return [], False
    bad = []
    for trial in range(5):
        params = base_params.copy()
        if trial == 1:
            params = rng.normal(0, 0.05, max(npar, 1))
        if trial == 2:
            params = rng.normal(0, 1.0, max(npar, 1))
        if trial == 3:
            params = rng.normal(0, 12.0, max(npar, 1))                     # wide error bounds: queues fill, codes switch off
        if trial == 4:
            params[int(rng.integers(0, max(npar, 1)))] = rng.choice([-1.0, 0.0, np.nextafter(base_params[0], 1.0)])
        pbuf.set(params)
        got = {}
        for codes in (True, False):
            group.SetCodes(codes)
            group.EvalAsync(False)
            group.EvalFinished()
            got[codes] = (ev.GetBins(), int(norm.get()[0]))
        if not (np.array_equal(got[True][0], got[False][0]) and got[True][1] == got[False][1]):
            bad.append((seed, trial, "codes != floats", systs, list(params), nbins, n))
        if seed % 5 == 0 and trial in (0, 2):
            geom = oracle.HistGeometry([0.0] * nobs, [1.0] * nobs, nbins)
            bins, nrm = oracle.bin_samples(geom, tab, nfields, systs, params)
            if not (np.array_equal(got[True][0], bins) and got[True][1] == nrm):
                bad.append((seed, trial, "codes != oracle", systs, list(params), nbins, n))
    group.close()
    ev.close()
    return bad, True


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    failures, used_cases = [], 0
    for seed in range(first, first + count):
        bad, used = one_case(seed)
        used_cases += int(used)
        for b in bad:
            print("FAIL", b, flush=True)
        failures += bad
        if (seed - first) % 50 == 49:
            print("... %d cases, %d over codes, %d failures" % (seed - first + 1, used_cases, len(failures)), flush=True)
    print("soak_codes: %d cases (%d ran the fill over codes), %d failures" % (count, used_cases, len(failures)))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
