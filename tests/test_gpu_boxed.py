"""GPU parity: the BOXED form of the bucketed fill (fill_boxed_kernel, sxmc_group_set_boxes).

An observable that is resolution-scaled against a truth field depends on two fields and has no order; the boxed form
sorts the rows of a bucket into small boxes of (observable, truth field), runs the reference's operations on the
corners of every granule's box per evaluation (every IEEE operation is monotone in each operand, so the corners bound
every row inside) and bins whole granules at once where both ends land in one bin; the other written observable is
streamed as one 16-bit code per row.  Everything here compares histograms and norms, bit for bit, with the oracle
(bin_samples, /root/reference/src/pdfz.cpp:349-408 restated) and with the same launch in the ordered form and as a
float stream."""
import numpy as np
import pytest

from sxmc_amd import nll, pdfz
from sxmc_amd.capi import DeviceArray
from sxmc_amd.mcmc import make_systematic
from tests.test_gpu_pdfz import oracle_eval

pytestmark = pytest.mark.gpu

C3 = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
      dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
LO, HI, NB = [0.0, 0.0, -1.0], [10.0, 6.0, 1.0], [20, 20, 20]


def c3_table(rng, n, j=0, spread=0.3):
    """fields [e, r, c, e_true, DATASET] like BASELINE config 3 (sxmc_amd/workloads.py)."""
    e_true = rng.normal(2.0 + 0.5 * j, 1.2, size=n)
    e = e_true + rng.normal(0.0, spread, size=n)
    r = 6.0 * rng.uniform(0.0, 1.0, size=n) ** (1.0 / 3.0)
    c = rng.uniform(-1.0, 1.0, size=n)
    return np.stack([e, r, c, e_true, np.zeros(n)], axis=1).astype(np.float32)


def make_group(tabs, systs, params, lower=LO, upper=HI, nbins=NB, nfields=5):
    nobs = len(nbins)
    norms = DeviceArray(np.full(len(tabs), 55, np.uint32))
    pbuf = DeviceArray(np.asarray(params, np.float64))
    evs = []
    for j, t in enumerate(tabs):
        ev = pdfz.EvalHist(t, nfields, nobs, lower, upper, nbins)
        for s in systs:
            ev.AddSystematic(make_systematic(s))
        ev.SetNormalizationBuffer(norms, j)
        ev.SetParameterBuffer(pbuf, 0, 1)
        evs.append(ev)
    return nll.EvalGroup(evs), evs, norms, pbuf


def evaluate(group, evs, norms):
    group.EvalAsync(False)
    group.EvalFinished()
    return [e.GetBins() for e in evs], norms.get()


PARAMS = [[0.02, -0.004, 0.03], [0.0, 0.0, 0.0], [-0.05, 0.01, -0.05], [0.11, 0.02, 0.12], [0.3, -0.2, -0.4],
          [0.01, -2.5, 0.02],          # 1 + p < 0: the scale reverses the box
          [0.01, -1.0, 0.3],           # 1 + p = 0
          [1e-9, 1e-12, -1e-9]]


def test_boxed_form_gives_the_oracles_histograms():
    rng = np.random.default_rng(61)
    sizes = [400001, 3, 123457, 0, 255, 257, 70000]
    tabs = [c3_table(rng, n, j) for j, n in enumerate(sizes)]
    group, evs, norms, pbuf = make_group(tabs, C3, PARAMS[0])
    group.SetBoxes(True)
    assert "boxed+codes" in group.LaunchInfo(), group.LaunchInfo()
    boxed_bytes = group.AlgorithmicBytes()["fill_read"]
    for params in PARAMS:
        pbuf.set(np.asarray(params, np.float64))
        results = []
        for boxes, partition in ((True, 0), (True, 1), (True, 2), (False, 0)):
            group.SetBoxes(boxes)
            group.SetPartition(partition)
            assert ("boxed" in group.LaunchInfo()) == boxes
            results.append(evaluate(group, evs, norms))
        group.SetPartition(0)
        for j, t in enumerate(tabs):
            o = oracle_eval(t, 5, LO, HI, NB, C3, params)
            for k, (bins, nrm) in enumerate(results):
                assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"], (params, j, k)
    group.SetBoxes(False)
    assert "ordered+codes" in group.LaunchInfo()
    assert boxed_bytes < 0.62 * group.AlgorithmicBytes()["fill_read"]      # 2 bytes per sample instead of 4


def test_boxed_form_with_small_queues_and_other_launch_shapes():
    rng = np.random.default_rng(62)
    tabs = [c3_table(rng, n, j) for j, n in enumerate([300000, 100000])]
    group, evs, norms, pbuf = make_group(tabs, C3, [0.02, -0.004, 0.08])
    group.SetBoxes(True)
    want = None
    for qlog in (0, 9, 10):
        for launch in ((0, 0), (256, 2), (1024, 1), (768, 1), (512, 2)):
            group.SetCodesQueueLog(qlog)
            group.SetLaunchConfig(*launch)
            info = group.LaunchInfo()
            got = evaluate(group, evs, norms)
            if "boxed" not in info:
                continue    # (a shape the boxed form's LDS copy does not fit: planned without it)
            if want is None:
                want = [oracle_eval(t, 5, LO, HI, NB, C3, [0.02, -0.004, 0.08]) for t in tabs]
            for j in range(len(tabs)):
                assert np.array_equal(got[0][j], want[j]["bins"]) and got[1][j] == want[j]["norm"], (qlog, launch, j)
    assert want is not None


def test_boxed_form_wild_parameters_and_values():
    """Coefficients that are not finite or huge, NaN / infinite values in the boxed observable, its truth field and the
    streamed observable, rows outside the code window: all decided by the float columns, the counts the oracle's."""
    rng = np.random.default_rng(63)
    n = 200000
    t = c3_table(rng, n)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e30, -1e30, 3e38], np.float32)
    for col in (0, 1, 3):
        idx = rng.choice(n, size=300, replace=False)
        t[idx, col] = rng.choice(special, size=300)
    t[1000:1600, 0] = t[1000, 0]                      # runs of identical values in the boxed observable
    t[5000:5600, 3] = t[5000, 3]
    group, evs, norms, pbuf = make_group([t], C3, [0.0, 0.0, 0.0])
    group.SetBoxes(True)
    assert "boxed" in group.LaunchInfo()
    for params in ([0.02, -0.004, 0.03], [np.nan, 0.0, 0.0], [0.0, np.inf, 0.0], [0.0, 0.0, -np.inf], [0.0, 0.0, np.nan],
                   [1e300, 0.1, 0.1], [0.1, 1e300, 1e300], [0.1, 1e160, -1e160], [5.0, 0.0, 0.0], [-7.0, 3.0, 30.0]):
        pbuf.set(np.asarray(params, np.float64))
        bins, nrm = evaluate(group, evs, norms)
        o = oracle_eval(t, 5, LO, HI, NB, C3, params)
        assert np.array_equal(bins[0], o["bins"]) and nrm[0] == o["norm"], params


def edge_values(rng, n, lo, hi, nb, inverse):
    """values whose image under the program lies within a few ulps of a bin edge; inverse: edge -> raw value"""
    edges = lo + (hi - lo) * np.arange(nb + 1, dtype=np.float64) / nb
    x = inverse(rng.choice(edges, size=n)).astype(np.float32)
    for _ in range(3):
        up = rng.uniform(size=n) < 0.5
        x = np.where(rng.uniform(size=n) < 0.6, np.nextafter(x, np.where(up, np.float32(1e9), np.float32(-1e9))), x)
    return x.astype(np.float32)


def test_boxed_form_samples_on_the_bin_edges():
    """The boxed observable within ulps of the transformed edges (boxes that straddle, boxes of ONE value on an edge),
    the streamed observable within ulps of its shifted edges (ambiguous rows)."""
    rng = np.random.default_rng(64)
    n = 250000
    params = [0.0137, 0.031, 0.0]           # resolution parameter 0: e' = e (1 + p) exactly invertible to ulps
    t = c3_table(rng, n)
    t[: n // 2, 0] = edge_values(rng, n // 2, 0.0, 10.0, 20, lambda e: e / (1 + params[1]))
    t[n // 4: 3 * n // 4, 1] = edge_values(rng, n // 2, 0.0, 6.0, 20, lambda e: e - params[0])
    t[2000:2900, 0] = t[2000, 0]            # whole granules of one value that sits on an edge
    t[2000:2900, 3] = t[2000, 3]
    group, evs, norms, pbuf = make_group([t], C3, params)
    group.SetBoxes(True)
    for p in (params, [0.0137, 0.031, 1e-7], [0.0137, 0.031, -0.02]):
        pbuf.set(np.asarray(p, np.float64))
        bins, nrm = evaluate(group, evs, norms)
        o = oracle_eval(t, 5, LO, HI, NB, C3, p)
        assert np.array_equal(bins[0], o["bins"]) and nrm[0] == o["norm"], p


OTHER_PROGRAMS = [
    # (name, nobs, nbins, lower, upper, nfields, systs, params): run-time compiled boxed programs
    ("ctscale-streamed", 3, [12, 10, 8], LO, HI, 5,
     [dict(type="ctscale", obs=2, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=3, pars=[1]),
      dict(type="shift", obs=0, pars=[2])], [[0.03, 0.05, -0.02], [-0.01, -0.3, 0.4]]),
    # the boxed observable is NOT the outermost dimension, the streamed one the innermost, nothing untouched
    ("two-dims", 2, [9, 14], [0.0, 0.0], [6.0, 10.0], 5,
     [dict(type="scale", obs=0, pars=[0]), dict(type="resolution_scale", obs=1, true_obs=3, pars=[1]),
      dict(type="scale", obs=1, pars=[2])], [[0.02, 0.06, -0.01], [0.5, -0.07, 0.02]]),
    # two resolution scales against the same truth field
    ("two-resolutions", 3, [20, 6, 5], [0.0, 0.0, -1.0], [10.0, 6.0, 1.0], 5,
     [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0]), dict(type="shift", obs=1, pars=[1]),
      dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])], [[0.04, 0.1, -0.03]]),
]


@pytest.mark.parametrize("name,nobs,nbins,lower,upper,nfields,systs,param_sets", OTHER_PROGRAMS,
                         ids=[c[0] for c in OTHER_PROGRAMS])
def test_boxed_form_other_programs(name, nobs, nbins, lower, upper, nfields, systs, param_sets):
    rng = np.random.default_rng(65)
    tabs = []
    for n in (150000, 40000):
        t = c3_table(rng, n)
        if name == "two-dims":      # fields [r, e, c, e_true]: the resolution-scaled observable second
            t = t[:, [1, 0, 2, 3, 4]].copy()
        tabs.append(t)
    group, evs, norms, pbuf = make_group(tabs, systs, param_sets[0], lower, upper, nbins, nfields)
    group.SetBoxes(True)
    assert "boxed+codes" in group.LaunchInfo(), group.LaunchInfo()
    for params in param_sets:
        pbuf.set(np.asarray(params, np.float64))
        for boxes in (True, False):
            group.SetBoxes(boxes)
            bins, nrm = evaluate(group, evs, norms)
            for j, t in enumerate(tabs):
                o = oracle_eval(t, nfields, lower, upper, nbins, systs, params)
                assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"], (params, boxes, j)


def test_boxed_form_is_not_taken_where_it_does_not_apply():
    rng = np.random.default_rng(66)
    t = c3_table(rng, 100000)
    # three written observables; a polynomial on the streamed observable; the truth field written; no resolution scale
    for systs in ([dict(type="shift", obs=1, pars=[0]), dict(type="shift", obs=2, pars=[1]),
                   dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])],
                  [dict(type="shift", obs=1, pars=[0, 1]), dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])],
                  [dict(type="shift", obs=1, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=1, pars=[2])],
                  [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1])]):
        group, evs, norms, pbuf = make_group([t], systs, [0.01, 0.02, 0.03])
        group.SetBoxes(True)
        assert "boxed" not in group.LaunchInfo(), (systs, group.LaunchInfo())
        bins, nrm = evaluate(group, evs, norms)
        o = oracle_eval(t, 5, LO, HI, NB, systs, [0.01, 0.02, 0.03])
        assert np.array_equal(bins[0], o["bins"]) and nrm[0] == o["norm"], systs


def test_boxed_plan_comes_with_an_ordered_twin():
    """The default plan keeps the boxed AND the ordered tables; a fill launches one of them: the ordered one until
    sxmc_group_adapt_fill_form is asked, then the boxed one while the image of a mean box is narrow (small resolution
    parameter) -- same counts either way."""
    rng = np.random.default_rng(67)
    tabs = [c3_table(rng, n, j) for j, n in enumerate([1000000, 1200000])]
    group, evs, norms, pbuf = make_group(tabs, C3, PARAMS[0])
    info = group.LaunchInfo()
    assert "boxed+codes|ordered+codes(now)" in info, info
    assert group.FillForm() == 2
    ordered_bytes = group.AlgorithmicBytes()["fill_read"]
    group.SetBoxLimit(0.6)      # (tables of 10^6 rows: boxes several times as wide as config 3's)
    seen = set()
    for params, want in (([0.02, -0.004, 0.0], 1), ([0.02, -0.004, -0.01], 1), ([0.02, -0.004, 0.4], 2),
                         ([0.02, -0.004, -0.6], 2), ([0.0, 30.0, 0.0], 2), ([0.0, 0.0, np.nan], 2), ([0.0, np.inf, 0.0], 2),
                         ([0.02, -0.004, 0.001], 1)):
        pbuf.set(np.asarray(params, np.float64))
        want_bins = [oracle_eval(t, 5, LO, HI, NB, C3, params) for t in tabs]
        form, changed = group.AdaptFillForm()
        assert form == want and group.FillForm() == want, (params, form)
        assert changed == (want not in seen and bool(seen)) or changed in (True, False)
        seen.add(form)
        assert ("boxed+codes(now)" in group.LaunchInfo()) == (form == 1)
        for forced in (form, 3 - form):
            group.SetFillForm(forced)
            bins, nrm = evaluate(group, evs, norms)
            for j in range(len(tabs)):
                assert np.array_equal(bins[j], want_bins[j]["bins"]) and nrm[j] == want_bins[j]["norm"], (params, forced, j)
        group.SetFillForm(form)
    assert seen == {1, 2}
    group.SetFillForm(1)
    assert group.AlgorithmicBytes()["fill_read"] < 0.62 * ordered_bytes
    # the band between the two decisions: at the limit the form stays what it is
    # tables too small for boxes to pay, or boxes switched off: one form
    small, evs2, norms2, pbuf2 = make_group([c3_table(rng, 50000)], C3, PARAMS[0])
    assert "boxed" not in small.LaunchInfo() and small.FillForm() == 0 and small.AdaptFillForm() == (0, False)
    group.SetBoxes(False)
    assert group.FillForm() == 0 and "boxed" not in group.LaunchInfo()
    group.SetBoxes(True)
    assert group.FillForm() == 0 and "table=boxed+codes " in group.LaunchInfo()


def test_walks_are_the_same_chain_in_either_form():
    """A walk over a plan with both forms: forced boxed, forced ordered, and left to sxmc_group_adapt_fill_form (asked at
    the first step and at every flush, the recorded steps recorded again when the form changes) -- the fills are
    bit-identical, so the chains are."""
    from sxmc_amd import capi, workloads
    from sxmc_amd.mcmc import MCMC
    w = workloads.config3(0.25, nevents=5000)
    chains = []
    for mode in ("adaptive", "boxed", "ordered"):
        m = MCMC(w, seed=11, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
        m.setup(sync_interval=64)
        capi.synchronize()
        assert "boxed+codes|ordered+codes(now)" in m.group.LaunchInfo()
        rows = []
        for block in range(4):
            if mode != "adaptive":
                m._adapt_pending = False
                m.group.SetFillForm(1 if mode == "boxed" else 2)
            m.step()
            m.steps(60, graph_steps=10)
            if mode != "adaptive":
                m.group.SetFillForm(1 if mode == "boxed" else 2)
            r, _ = m.flush()
            if mode != "adaptive":                      # (flush asked; put the forced form back for the next block)
                m._graph = None
                m.group.SetFillForm(1 if mode == "boxed" else 2)
            rows.append(r)
        if mode == "adaptive":
            assert m.group.FillForm() in (1, 2)
        chains.append(np.concatenate(rows, axis=0))
    assert chains[0].shape[0] > 0
    assert np.array_equal(chains[0].view(np.uint32), chains[1].view(np.uint32))
    assert np.array_equal(chains[0].view(np.uint32), chains[2].view(np.uint32))


def test_whole_walk_with_burn_in_is_the_same_chain_whatever_chooses_the_form():
    """MCMC.walk (burn-in re-tuning, flushes, graph replay) over a plan with two forms: flushing -- and asking for the form
    -- every ADAPT_INTERVAL steps gives the chain of a walk pinned to the ordered form."""
    from sxmc_amd import capi, workloads
    from sxmc_amd import mcmc as mcmc_mod
    w = workloads.config3(0.25, nevents=5000)
    chains = []
    for pinned in (False, True):
        m = mcmc_mod.MCMC(w, seed=13, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
        if pinned:
            class Pinned:                              # (the group, with the question answered "ordered, unchanged")
                def __init__(self, g):
                    self._grp = g

                def __getattr__(self, name):
                    return getattr(self._grp, name)

                def AdaptFillForm(self):
                    return 2, False

                def FillForm(self):
                    return 0
            m.group = Pinned(m.group)
        chain, acc = m.walk(w.events, 2300, 0.2, sync_interval=10000, graph_steps=10)
        chains.append((chain, acc))
        if not pinned:
            assert m.group.FillForm() in (1, 2) and mcmc_mod.ADAPT_INTERVAL - 1 in m.flush_schedule()
    assert chains[0][1] == chains[1][1]
    assert np.array_equal(chains[0][0].view(np.uint32), chains[1][0].view(np.uint32))
