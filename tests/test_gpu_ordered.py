"""GPU parity: the ORDERED-observable form of the bucketed fill (fill_ordered_kernel, sxmc_group_set_ordering).

An observable written only by one-coefficient shift / scale / cos-theta-scale systematics is a monotone function of
the sample's raw value, so along rows sorted by that value its bin is a step function: one constant per 256-sample
granule except where a granule straddles a bin edge.  Everything here compares histograms and norms, bit for bit,
with the oracle (bin_samples, /root/reference/src/pdfz.cpp:349-408 restated) and with the same launch without
ordering / without bucketing."""
import numpy as np
import pytest

from oracle import oracle
from sxmc_amd import nll, pdfz
from sxmc_amd.capi import DeviceArray
from sxmc_amd.mcmc import make_systematic
from tests.test_gpu_pdfz import build_group, oracle_eval, table

pytestmark = pytest.mark.gpu

C3 = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
      dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]

CASES = [
    # one observable, one shift (bench_sxmc pdfz): nothing is streamed but the granule words
    ("1d-shift", 1, [1000], [dict(type="shift", obs=0, pars=[0])], [[0.013], [-0.4], [0.0], [3.0]], 2),
    # 1 + p < 0: the map reverses the order of the rows
    ("1d-negative-scale", 1, [17], [dict(type="scale", obs=0, pars=[0])], [[-2.3], [-1.0], [0.2]], 1),
    # BASELINE config 3: r ordered, c in the buckets, e binned per sample
    ("c3", 3, [20, 20, 20], C3, [[0.02, -0.01, 0.07], [-0.3, 0.0, 0.0], [0.7, 0.1, -0.2]], 5),
    # two systematics on the ordered observable, the other observable untouched
    ("two-ops", 2, [30, 40], [dict(type="ctscale", obs=1, pars=[0]), dict(type="shift", obs=1, pars=[1])],
     [[0.04, -0.02], [-1.5, 0.6]], 3),
    # ordered + another written observable + nothing untouched
    ("no-bucket-key", 2, [9, 7], [dict(type="shift", obs=0, pars=[0]),
                                  dict(type="resolution_scale", obs=1, true_obs=2, pars=[1])], [[0.05, 0.3]], 3),
    # two candidates: the one with fewer bins is ordered, the other binned per sample
    ("two-candidates", 3, [40, 6, 5], [dict(type="shift", obs=0, pars=[0]), dict(type="scale", obs=1, pars=[1])],
     [[0.02, 0.05], [-0.02, -0.6]], 4),
    # a polynomial on another observable (hiprtc), the ordered one with a cos-theta scale
    ("poly-elsewhere", 3, [12, 9, 10], [dict(type="shift", obs=0, pars=[0, 1, 2]), dict(type="ctscale", obs=2, pars=[3])],
     [[0.02, -0.03, 0.01, 0.05]], 4),
    # C5's systematics on a histogram that fits LDS
    ("5d", 5, [6, 5, 4, 3, 2], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                                dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])], [[0.02, -0.01, 0.07]], 7),
]


@pytest.mark.parametrize("name,nobs,nbins,systs,param_sets,nfields", CASES, ids=[c[0] for c in CASES])
def test_ordered_observable_gives_identical_histograms(name, nobs, nbins, systs, param_sets, nfields):
    rng = np.random.default_rng(31)
    sizes = [70001, 3, 123457, 0, 255, 257, 256]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, param_sets[0], nfields=nfields)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    assert "ordered" in group.LaunchInfo()
    for params in param_sets:
        pbuf.set(np.asarray(params, np.float64))
        results = []
        for order, bucket in ((True, True), (False, True), (False, False)):
            group.SetOrdering(order, force=True)
            group.SetBucketing(bucket)
            for partition in ((0, 1, 2) if order else (0,)):
                group.SetPartition(partition)
                group.EvalAsync(False)
                group.EvalFinished()
                results.append(([e.GetBins() for e in evs], norms.get(), group.AlgorithmicBytes()["fill_read"]))
        group.SetPartition(0)
        # teams of workgroups over contiguous parts of the sorted table (what sxmc_group_optimize may choose): the
        # same counts, whatever the team count (ordered + bucketed forms)
        group.SetPartition(2)
        for order in (True, False):
            group.SetOrdering(order, force=True)
            group.SetBucketing(True)
            for teams in (3, 7, 64):
                group.SetPartitionTeams(teams)
                if order and teams == 3:
                    assert "teams=3" in group.LaunchInfo() or "partition=1" in group.LaunchInfo()
                group.EvalAsync(False)
                group.EvalFinished()
                results.append(([e.GetBins() for e in evs], norms.get(), 0.0))
        group.SetPartitionTeams(0)
        group.SetPartition(0)
        assert results[0][2] < results[3][2] <= results[4][2]          # fewer bytes to stream
        for j, t in enumerate(tabs):
            o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params)
            for k, (bins, nrm, _) in enumerate(results):
                assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"], (params, j, k)
    got = evs[0].GetSamples().reshape(-1, nobs + 1)                     # the caller's row order is untouched
    assert np.array_equal(got[:, :nobs].view(np.uint32), tabs[0][:, :nobs].view(np.uint32))


def hugging(rng, n, nbins, shift):
    """Values within a few ulps of where the bin edges land once `shift` is added, + the special values."""
    edges = (np.arange(nbins + 1, dtype=np.float64) / nbins - shift).astype(np.float32)
    x = rng.choice(edges, size=n)
    for _ in range(3):
        up = rng.uniform(size=n) < 0.5
        x = np.where(rng.uniform(size=n) < 0.6, np.nextafter(x, np.where(up, np.float32(9), np.float32(-9))), x)
    special = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, np.float32(1) - np.float32(2 ** -24)],
                       np.float32)
    return np.where(rng.uniform(size=n) < 0.02, rng.choice(special, size=n), x).astype(np.float32)


@pytest.mark.parametrize("kind", ["shift", "scale", "ctscale"])
def test_ordered_observable_samples_on_the_bin_edges(kind):
    """Samples placed within ulps of the (transformed) bin edges, long runs of identical values that span several
    granules, NaN of both signs, infinities and signed zeros in the ordered column: granules that straddle an
    edge must take the per-sample path, all others the constant -- the counts must be the oracle's."""
    rng = np.random.default_rng(37)
    n, nb = 300000, [50, 3]
    p = {"shift": 0.0137, "scale": 0.031, "ctscale": -0.027}[kind]
    tab = table(rng, n, 3)
    if kind == "shift":
        col = hugging(rng, n, nb[0], p)
    elif kind == "scale":      # x (1 + p) = edge  <=>  x = edge / (1 + p)
        col = (hugging(rng, n, nb[0], 0.0).astype(np.float64) / (1 + p)).astype(np.float32)
    else:                      # 1 + (x - 1)(1 + p) = edge
        col = (1 + (hugging(rng, n, nb[0], 0.0).astype(np.float64) - 1) / (1 + p)).astype(np.float32)
    col[1000:3000] = col[1000]                       # 2000 identical values: whole granules of one value
    tab[:, 0] = col
    systs = [dict(type=kind, obs=0, pars=[0])]
    geom = oracle.HistGeometry([0.0, 0.0], [1.0, 1.0], nb)
    for order in (True, False):
        ev = pdfz.EvalHist(tab, 3, 2, [0.0, 0.0], [1.0, 1.0], nb)
        ev.AddSystematic(make_systematic(systs[0]))
        norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.array([p]))
        ev.SetNormalizationBuffer(norm)
        ev.SetParameterBuffer(pbuf)
        group = nll.EvalGroup([ev])
        group.SetOrdering(order, force=True)
        assert ("ordered" in group.LaunchInfo()) == order
        for pv in (p, np.nextafter(p, 1.0), np.nextafter(p, -1.0), 0.0, -p):
            pbuf.set(np.array([pv]))
            group.EvalAsync(False)
            group.EvalFinished()
            bins, nrm = oracle.bin_samples(geom, tab, 3, systs, np.array([pv]))
            assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm, (order, pv)
        group.close()
        ev.close()


def test_ordering_by_default_only_where_it_pays():
    """Default mode: a table needs at least twice as many granules as can straddle a bin edge of the ordered
    observable (buckets x (nbins + 1)); smaller tables keep the unordered bucketed layout."""
    rng = np.random.default_rng(47)
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
    for n, nbins, want in ((400001, [6, 5, 4], True), (400001, [6, 50, 40], False)):
        evs, tabs, lut, norms, pbuf = build_group(rng, [n], 3, nbins, systs, [0.02, -0.01, 0.07], nfields=5)
        group = nll.EvalGroup(evs)
        assert ("ordered" in group.LaunchInfo()) == want, group.LaunchInfo()
        group.EvalAsync(False)
        group.EvalFinished()
        o = oracle_eval(tabs[0], 5, [0.0] * 3, [1.0] * 3, nbins, systs, [0.02, -0.01, 0.07])
        assert np.array_equal(evs[0].GetBins(), o["bins"]) and norms.get()[0] == o["norm"]
        group.close()


def test_ordered_observable_with_wild_parameters():
    """NaN and infinite coefficients: no monotone-map argument is made, every granule takes the per-sample path,
    and the result is whatever the unordered evaluation gives (nothing in the domain for NaN / +-inf shifts)."""
    rng = np.random.default_rng(41)
    sizes = [50001, 777]
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=1, pars=[1])]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 2, [11, 13], systs, [0.0, 0.0], nfields=3)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    assert "ordered" in group.LaunchInfo()
    for params in ([np.nan, 0.0], [0.0, np.nan], [np.inf, 0.0], [0.1, -np.inf], [0.0, np.inf], [0.01, -1.0],
                   [1e300, 1e300], [0.02, 0.03]):
        pbuf.set(np.asarray(params, np.float64))
        out = []
        for order in (True, False):
            group.SetOrdering(order, force=True)
            group.EvalAsync(False)
            group.EvalFinished()
            out.append(([e.GetBins() for e in evs], norms.get()))
        for j in range(len(sizes)):
            assert np.array_equal(out[0][0][j], out[1][0][j]) and out[0][1][j] == out[1][1][j], params
        if np.all(np.isfinite(params)):
            for j, t in enumerate(tabs):
                o = oracle_eval(t, 3, [0.0] * 2, [1.0] * 2, [11, 13], systs, params)
                assert np.array_equal(out[0][0][j], o["bins"]) and out[0][1][j] == o["norm"], params


def test_ordered_observable_lookup_and_reuse_across_groups():
    """Evaluation for lookup through the ordered fill (lut bits = the oracle's), and a second group over evaluators
    that share the same table re-uses the ordered copy."""
    rng = np.random.default_rng(43)
    sizes = [90001, 4001]
    pts = np.concatenate([table(rng, 500, 3), rng.integers(0, 2, size=(500, 1)).astype(np.float32)], axis=1)
    params = [0.02, -0.01, 0.07]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, [20, 20, 20], C3, params, nfields=5, points=pts)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)     # (tables this small have too few granules per bin edge for the default)
    assert "ordered" in group.LaunchInfo()
    group.EvalAsync(True)
    group.EvalFinished()
    got = lut.get().reshape(len(sizes), -1)
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], C3, params, points=pts, dataset=j % 2)
        assert np.array_equal(got[j].view(np.uint32), np.asarray(o["out"], np.float32).view(np.uint32))
        assert norms.get()[j] == o["norm"]
    shared = [pdfz.EvalHist.Shared(e) for e in evs]        # (the systematics come with the table)
    norms2, pbuf2 = DeviceArray.zeros(len(sizes), np.uint32), DeviceArray(np.asarray([0.1, 0.0, 0.0]))
    for j, s in enumerate(shared):
        s.SetNormalizationBuffer(norms2, j)
        s.SetParameterBuffer(pbuf2, 0, 1)
    g2 = nll.EvalGroup(shared)
    g2.SetOrdering(True, force=True)
    assert "ordered" in g2.LaunchInfo()
    g2.EvalAsync(False)
    g2.EvalFinished()
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], C3, [0.1, 0.0, 0.0])
        assert np.array_equal(shared[j].GetBins(), o["bins"]) and norms2.get()[j] == o["norm"]
