"""GPU parity: the CODES of an ordered table (fill_ordered_body's kCodes path, sxmc_group_set_codes).

One-coefficient shift / scale / cos-theta scale / resolution scale compose into an affine map of the sample's fields,
so the fill may bin a sample from 16-bit codes of its fields whenever its bin coordinate lies further from a bin edge
than a bound on everything the codes and the arithmetic leave unknown; the few samples that lie closer are binned at
the end of the stream from their float values with the reference's arithmetic.  Everything here compares histograms
and norms, bit for bit, with the oracle (bin_samples, /root/reference/src/pdfz.cpp:349-408 restated) and with the
same launch streaming the float columns."""
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
    # BASELINE config 3: r ordered, c in the buckets, e (scale, resolution scale against e_true) from codes
    ("c3", 3, [20, 20, 20], C3, [[0.02, -0.01, 0.07], [-0.3, 0.0, 0.0], [0.7, 0.1, -0.2], [0.0, 0.9, 1.4],
                                 [0.01, -0.7, -0.9], [0.0, -1.0, 0.3], [0.0, -2.5, 0.0]], 5),
    # the same program on five observables (BASELINE config 5's systematics, histogram in LDS)
    ("5d", 5, [6, 5, 4, 3, 2], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                                dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])],
     [[0.02, -0.01, 0.07], [0.1, 0.3, -0.4]], 7),
    # nothing untouched: one bucket; obs 1 against a truth field, obs 0 ordered
    ("no-bucket-key", 2, [9, 7], [dict(type="shift", obs=0, pars=[0]),
                                  dict(type="resolution_scale", obs=1, true_obs=2, pars=[1])],
     [[0.05, 0.3], [-0.2, -0.6]], 3),
    # two observables binned from codes (run-time compiled), the third ordered
    ("two-binned", 3, [5, 7, 6], [dict(type="shift", obs=0, pars=[0]), dict(type="scale", obs=1, pars=[1]),
                                  dict(type="ctscale", obs=2, pars=[2])], [[0.02, 0.05, -0.1], [-0.3, -0.4, 0.6]], 4),
    # an UNTOUCHED observable as the truth field of another: its value is an input, its bin the bucket's
    ("obs-as-truth", 3, [8, 6, 5], [dict(type="resolution_scale", obs=0, true_obs=1, pars=[0]),
                                    dict(type="shift", obs=2, pars=[1])], [[0.4, 0.03], [-0.8, -0.2]], 4),
    # three streamed fields: two words of codes per row
    ("three-fields", 3, [7, 9, 5], [dict(type="scale", obs=0, pars=[0]),
                                    dict(type="resolution_scale", obs=1, true_obs=3, pars=[1]),
                                    dict(type="shift", obs=2, pars=[2])], [[0.03, 0.5, 0.01], [-0.5, -0.3, -0.1]], 5),
    # four streamed fields; the second observable's systematic reads the FIRST one after that was written
    ("four-fields", 3, [6, 6, 4], [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0]),
                                   dict(type="resolution_scale", obs=1, true_obs=0, pars=[1]),
                                   dict(type="resolution_scale", obs=1, true_obs=4, pars=[3]),
                                   dict(type="shift", obs=2, pars=[2])],
     [[0.2, -0.3, 0.02, 0.1], [-0.6, 0.8, -0.05, -0.2]], 6),
    # five systematics on one observable, every kind, parameters shared
    ("five-ops", 2, [25, 4], [dict(type="shift", obs=0, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                              dict(type="ctscale", obs=0, pars=[2]), dict(type="shift", obs=0, pars=[1]),
                              dict(type="resolution_scale", obs=0, true_obs=2, pars=[0]),
                              dict(type="scale", obs=1, pars=[3])],
     [[0.01, 0.02, -0.03, 0.05], [0.3, -0.2, 0.5, -0.1]], 4),
]


def evaluate(group, evs, norms):
    group.EvalAsync(False)
    group.EvalFinished()
    return [e.GetBins() for e in evs], norms.get()


@pytest.mark.parametrize("name,nobs,nbins,systs,param_sets,nfields", CASES, ids=[c[0] for c in CASES])
def test_codes_give_identical_histograms(name, nobs, nbins, systs, param_sets, nfields):
    rng = np.random.default_rng(131)
    sizes = [70001, 3, 123457, 0, 255, 257, 256]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, param_sets[0], nfields=nfields)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    group.SetCodes(True)
    assert "ordered+codes" in group.LaunchInfo(), group.LaunchInfo()
    members, rows, exact_rows, never_rows = group.CodesInfo()
    assert members >= 3 and exact_rows == 0 and never_rows < 256 * members * (max(nbins) + 2) * 64
    for params in param_sets:
        pbuf.set(np.asarray(params, np.float64))
        results = []
        for codes in (True, False):
            group.SetCodes(codes)
            assert ("ordered+codes" in group.LaunchInfo()) == codes
            for partition in (0, 1, 2):
                group.SetPartition(partition)
                bins, nrm = evaluate(group, evs, norms)
                results.append((bins, nrm, group.AlgorithmicBytes()["fill_read"]))
        group.SetPartition(0)
        assert results[0][2] < results[3][2]                     # fewer bytes to stream
        for j, t in enumerate(tabs):
            o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params)
            for k, (bins, nrm, _) in enumerate(results):
                assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"], (params, j, k)
    group.close()


def test_codes_samples_on_the_transformed_bin_edges():
    """Samples whose transformed value lies within ulps of a bin edge -- on it, just below, just above: every one of
    them is ambiguous to the codes and must be decided by the reference's arithmetic on the float values."""
    rng = np.random.default_rng(137)
    n, nb = 400000, [40, 3, 2]
    p1, p2 = 0.031, -0.17                                       # scale, resolution scale
    tab = table(rng, n, 5)
    t = tab[:, 3].astype(np.float64)
    edges = rng.integers(0, nb[0] + 1, size=n) / nb[0]
    # x (1 + p1) + p2 (x (1 + p1) - t) = edge  <=>  x = (edge + p2 t) / ((1 + p1)(1 + p2))
    x = ((edges + p2 * t) / ((1 + p1) * (1 + p2))).astype(np.float32)
    for _ in range(3):
        up = rng.uniform(size=n) < 0.5
        x = np.where(rng.uniform(size=n) < 0.6, np.nextafter(x, np.where(up, np.float32(9), np.float32(-9))), x)
    keep = rng.uniform(size=n) < 0.7                             # 70 % on the edges, the rest anywhere
    tab[:, 0] = np.where(keep, x, tab[:, 0])
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
    geom = oracle.HistGeometry([0.0] * 3, [1.0] * 3, nb)
    ev = pdfz.EvalHist(tab, 5, 3, [0.0] * 3, [1.0] * 3, nb)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.array([0.0, p1, p2]))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    assert "ordered+codes" in group.LaunchInfo()
    for pv in ([0.0, p1, p2], [0.0, np.nextafter(p1, 1.0), p2], [0.0, p1, np.nextafter(p2, -1.0)], [0.0, 0.0, 0.0],
               [0.01, -p1, -p2]):
        pbuf.set(np.array(pv))
        bins, nrm = oracle.bin_samples(geom, tab, 5, systs, np.array(pv))
        for codes in (True, False):
            group.SetCodes(codes)
            group.EvalAsync(False)
            group.EvalFinished()
            assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm, (codes, pv)
    group.close()
    ev.close()


def test_codes_rows_outside_the_windows_and_values_that_are_not_finite():
    """Values far outside the domain get the code "ask the exact columns" -- and a scale that brings them back in
    must count them; NaN and infinities get "never counted"."""
    rng = np.random.default_rng(139)
    n, nb = 200001, [20, 5, 4]
    tab = table(rng, n, 5)
    far = rng.uniform(size=n) < 0.01
    tab[far, 0] = rng.uniform(20.0, 60.0, size=far.sum()).astype(np.float32)       # e far above the domain [0, 1)
    tab[far, 3] = tab[far, 0] + rng.normal(0, 0.5, size=far.sum()).astype(np.float32)
    special = np.array([np.nan, -np.nan, np.inf, -np.inf], np.float32)
    bad = rng.uniform(size=n) < 0.005
    tab[bad, 0] = rng.choice(special, size=bad.sum())
    bad2 = rng.uniform(size=n) < 0.005
    tab[bad2, 3] = rng.choice(special, size=bad2.sum())
    geom = oracle.HistGeometry([0.0] * 3, [1.0] * 3, nb)
    ev = pdfz.EvalHist(tab, 5, 3, [0.0] * 3, [1.0] * 3, nb)
    for s in C3:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.zeros(3))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    assert "ordered+codes" in group.LaunchInfo()
    members, rows, exact_rows, never_rows = group.CodesInfo()
    assert members == 1 and exact_rows > 0.002 * n and never_rows > 0.002 * n
    counted_far = 0
    for pv in ([0.0, 0.0, 0.0], [0.0, -0.98, 0.0], [0.02, -0.975, 0.3], [0.0, 0.5, -0.5]):
        pbuf.set(np.array(pv))
        bins, nrm = oracle.bin_samples(geom, tab, 5, C3, np.array(pv))
        only_near, _ = oracle.bin_samples(geom, tab[~far], 5, C3, np.array(pv))
        counted_far += int(bins.sum() - only_near.sum())
        for codes in (True, False):
            group.SetCodes(codes)
            group.EvalAsync(False)
            group.EvalFinished()
            assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm, (codes, pv)
    assert counted_far > 100          # (the scale of -0.98 did bring rows from outside the windows into the domain)
    group.close()
    ev.close()


def test_codes_with_large_and_wild_parameters():
    """Parameters that widen the error bound until many samples are ambiguous (the queue in LDS overflows and rows are
    decided where they are met), parameters beyond that (the codes are switched off for the evaluation), and
    parameters that are not finite."""
    rng = np.random.default_rng(149)
    sizes = [400001, 777, 50001]
    nbins = [20, 6, 5]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, nbins, C3, [0.0, 0.0, 0.0], nfields=5)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    assert "ordered+codes" in group.LaunchInfo()
    for params in ([0.0, 30.0, 0.0], [0.0, 0.0, 40.0], [0.0, -25.0, 12.0], [0.0, 150.0, 0.0], [0.0, 1e4, -1e4],
                   [0.0, 1e300, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [0.0, np.nan, 0.0], [0.0, 0.0, np.inf],
                   [np.nan, 0.0, 0.0], [0.0, -np.inf, 0.1], [0.02, -0.01, 0.07]):
        pbuf.set(np.asarray(params, np.float64))
        out = []
        for codes in (True, False):
            group.SetCodes(codes)
            out.append(evaluate(group, evs, norms))
        for j in range(len(sizes)):
            assert np.array_equal(out[0][0][j], out[1][0][j]) and out[0][1][j] == out[1][1][j], (params, j)
        if np.all(np.isfinite(params)):
            for j, t in enumerate(tabs):
                o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, nbins, C3, params)
                assert np.array_equal(out[0][0][j], o["bins"]) and out[0][1][j] == o["norm"], (params, j)
    group.close()


@pytest.mark.parametrize("queue_log", [9, 10])
def test_codes_queue_overflow(queue_log):
    """Queues capped at 2^9 and 2^10 entries per workgroup (sxmc_group_set_codes_queue_log): they fill up and are emptied
    in the middle of the stream, whole granules are handed to the float columns; the counts do not change."""
    rng = np.random.default_rng(157)
    sizes = [500001, 30001]
    nbins = [20, 6, 5]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, nbins, C3, [0.0, 0.0, 0.0], nfields=5)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    group.SetCodesQueueLog(queue_log)
    assert "ordered+codes" in group.LaunchInfo()
    for params in ([0.0, 30.0, 0.0], [0.02, -0.01, 0.07], [0.0, -20.0, 15.0]):
        pbuf.set(np.asarray(params, np.float64))
        bins, nrm = evaluate(group, evs, norms)
        for j, t in enumerate(tabs):
            o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, nbins, C3, params)
            assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"], (params, j)
    group.close()


def test_codes_lookup_and_tables_shared_between_groups():
    """Evaluation for lookup through the codes (lut bits = the oracle's), and a second group over evaluators that
    share the table re-uses the codes."""
    rng = np.random.default_rng(151)
    sizes = [90001, 4001]
    pts = np.concatenate([table(rng, 500, 3), rng.integers(0, 2, size=(500, 1)).astype(np.float32)], axis=1)
    params = [0.02, -0.01, 0.07]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, [20, 20, 20], C3, params, nfields=5, points=pts)
    group = nll.EvalGroup(evs)
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    assert "ordered+codes" in group.LaunchInfo()
    group.EvalAsync(True)
    group.EvalFinished()
    got = lut.get().reshape(len(sizes), -1)
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], C3, params, points=pts, dataset=j % 2)
        assert np.array_equal(got[j].view(np.uint32), np.asarray(o["out"], np.float32).view(np.uint32))
        assert norms.get()[j] == o["norm"]
    shared = [pdfz.EvalHist.Shared(e) for e in evs]
    norms2, pbuf2 = DeviceArray.zeros(len(sizes), np.uint32), DeviceArray(np.asarray([0.1, 0.2, -0.3]))
    for j, s in enumerate(shared):
        s.SetNormalizationBuffer(norms2, j)
        s.SetParameterBuffer(pbuf2, 0, 1)
    g2 = nll.EvalGroup(shared)
    g2.SetOrdering(True, force=True)
    g2.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    assert "ordered+codes" in g2.LaunchInfo()
    g2.EvalAsync(False)
    g2.EvalFinished()
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], C3, [0.1, 0.2, -0.3])
        assert np.array_equal(shared[j].GetBins(), o["bins"]) and norms2.get()[j] == o["norm"]


def test_codes_degenerate_columns():
    """Columns the windows cannot be cut from in the ordinary way: one value only, no finite value at all, values at the
    ends of the float range, a truth field in other units (no overlap with the observable's window), tables of a few rows."""
    rng = np.random.default_rng(163)
    nb = [12, 5, 4]
    geom = oracle.HistGeometry([0.0] * 3, [1.0] * 3, nb)
    cases = {}
    for name in ("constant-truth", "nan-truth", "huge-values", "other-units", "tiny"):
        n = 7 if name == "tiny" else 150001
        tab = table(rng, n, 5)
        if name == "constant-truth":
            tab[:, 3] = np.float32(0.4375)
        if name == "nan-truth":
            tab[:, 3] = np.nan
        if name == "huge-values":
            big = rng.uniform(size=n) < 0.01
            tab[big, 0] = rng.choice(np.array([3e38, -3e38, 1e30, -1e-30, 1e-45], np.float32), size=int(big.sum()))
            tab[big, 3] = rng.choice(np.array([3e38, -3e38, 7e20], np.float32), size=int(big.sum()))
        if name == "other-units":
            tab[:, 3] = (tab[:, 3] * 1000.0 + 5000.0).astype(np.float32)
        cases[name] = tab
    for name, tab in cases.items():
        ev = pdfz.EvalHist(tab, 5, 3, [0.0] * 3, [1.0] * 3, nb)
        for s in C3:
            ev.AddSystematic(make_systematic(s))
        norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.zeros(3))
        ev.SetNormalizationBuffer(norm)
        ev.SetParameterBuffer(pbuf)
        group = nll.EvalGroup([ev])
        group.SetOrdering(True, force=True)
        group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
        for pv in ([0.0, 0.0, 0.0], [0.03, -0.02, 0.1], [-0.2, 0.4, -0.6], [0.0, 0.0, 1e-4]):
            pbuf.set(np.array(pv))
            bins, nrm = oracle.bin_samples(geom, tab, 5, C3, np.array(pv))
            for codes in (True, False):
                group.SetCodes(codes)
                group.EvalAsync(False)
                group.EvalFinished()
                assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm, (name, codes, pv)
        group.close()
        ev.close()


@pytest.mark.parametrize("offset", [3.0e4, 1.0e6, -2.5e5])
def test_codes_with_windows_far_from_zero(offset):
    """A domain far from zero: the codes' windows then sit at |centre| / step of 10^8 .. 10^9 and beyond, where the
    table's own check of a row's position in its cell (centre = base + (code + 1/2) step, in double) is itself rounded
    at the level its 2^-20 slack was meant for -- the bound books that under its double-precision term
    (fill_kernels.inc.h, "THE BOUND": Q borrows from D).  Histograms over codes, over the float columns and the
    oracle's must agree bit for bit; float32 values are coarse out there (spacing 1/16 at 10^6), so many samples sit
    exactly on transformed edges."""
    rng = np.random.default_rng(163)
    n, nb = 300000, [20, 6, 4]
    lo, hi = [offset, 0.0, -1.0], [offset + 10.0, 6.0, 1.0]
    tab = np.zeros((n, 5), np.float32)
    t = offset + rng.uniform(-4.0, 14.0, n)
    tab[:, 3] = t
    tab[:, 0] = t + rng.normal(0, 0.6, n)
    tab[:, 1] = np.array([0.5, 1.5, 2.5, 3.5, 4.5, 5.5], np.float32)[rng.integers(0, 6, n)]    # (runs of equal r)
    tab[:, 2] = rng.uniform(-1.0, 1.0, n)
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
    geom = oracle.HistGeometry(lo, hi, nb)
    ev = pdfz.EvalHist(tab, 5, 3, lo, hi, nb)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.zeros(3))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(pbuf)
    group = nll.EvalGroup([ev])
    group.SetOrdering(True, force=True)
    group.SetBoxes(False)      # (the ordered form: the boxed one has tests/test_gpu_boxed.py)
    group.SetCodes(True)
    assert "ordered+codes" in group.LaunchInfo(), group.LaunchInfo()
    base, step = group.CodesWindows(0)
    assert len(base) == 2 and abs(base[0]) / step[0] > 5e7          # (far from zero in units of the code step)
    rel = 1.0 / abs(offset)
    for pv in ([0.0, 0.0, 0.0], [0.02, 2.0 * rel, 0.2], [-0.03, -3.0 * rel, -0.35], [0.0, 0.5 * rel, 1.5],
               [0.01, 40.0 * rel, 0.05]):                            # (a scale of 40 / offset moves e by four bins)
        pbuf.set(np.array(pv))
        bins, nrm = oracle.bin_samples(geom, tab, 5, systs, np.array(pv))
        for codes in (True, False):
            group.SetCodes(codes)
            group.EvalAsync(False)
            group.EvalFinished()
            assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm, (offset, codes, pv)
        assert nrm > 0.2 * n or pv[1] > 10 * rel
    group.close()
    ev.close()
