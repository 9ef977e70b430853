"""GPU parity tests of the pdfz::EvalHist half of the path: the HIP kernels (through the C ABI)
against the CPU oracle and the reference's own known answers.  Bit-exact: bins, norm, read_bins,
lut.  Mirrors the structure of the reference's test/test_pdfz*.cpp."""
import math

import numpy as np
import pytest

from oracle import oracle
from sxmc_amd import capi, nll, pdfz
from sxmc_amd.capi import DeviceArray
from sxmc_amd.mcmc import make_systematic
from tests.helpers import check_case_values, eval_points_with_dataset

pytestmark = pytest.mark.gpu


def gpu_eval(samples, nfields, lower, upper, nbins, systs, params, points=None, dataset=0,
             param_offset=0, param_stride=1, pdf_offset=0, pdf_stride=1, pdf_size=None,
             norm_offset=0, norm_init=None, launch=None, do_eval_pdf=True):
    """One evaluator, the reference call sequence (test_pdfz_fixtures.h + test_pdfz.cpp:79-96)."""
    nobs = len(nbins)
    ev = pdfz.EvalHist(samples, nfields, nobs, lower, upper, nbins, dataset=dataset)
    for s in systs:
        ev.AddSystematic(make_systematic(s))
    npoints = 0
    if points is not None:
        points = np.ascontiguousarray(points, dtype=np.float32)
        npoints = points.size // (nobs + 1)
        ev.SetEvalPoints(points)
    if pdf_size is None:
        pdf_size = pdf_offset + max(npoints, 1) * pdf_stride
    pdf_values = DeviceArray(np.full(pdf_size, 12345.0, dtype=np.float32))
    norm = DeviceArray(np.array(norm_init if norm_init is not None else [0, 0, 0], dtype=np.uint32))
    pbuf = DeviceArray(np.ascontiguousarray(params, dtype=np.float64))
    ev.SetPDFValueBuffer(pdf_values, pdf_offset, pdf_stride)
    ev.SetNormalizationBuffer(norm, norm_offset)
    ev.SetParameterBuffer(pbuf, param_offset, param_stride)
    if launch:
        ev.SetLaunchConfig(*launch)
    ev.EvalAsync(do_eval_pdf)
    ev.EvalFinished()
    res = dict(norm=norm.get(), out=pdf_values.get(),
               read_bins=ev.GetReadBins() if points is not None else None, ev=ev)
    try:
        res["bins"] = ev.GetBins()
    except capi.SxmcError:
        # a histogram beyond LDS capacity evaluated for lookup counts only the event bins (sparse mode);
        # the dense histogram comes from an evaluation with do_eval_pdf = False, as in CreateHistogram
        ev.EvalAsync(False)
        ev.EvalFinished()
        res["bins"] = ev.GetBins()
        res["sparse"] = True
        assert np.array_equal(norm.get(), res["norm"])
    return res


def oracle_eval(samples, nfields, lower, upper, nbins, systs, params, points=None, dataset=0,
                param_offset=0, param_stride=1, pdf_offset=0, pdf_stride=1, pdf_size=None):
    geom = oracle.HistGeometry(lower, upper, nbins)
    params = np.ascontiguousarray(params, dtype=np.float64)
    bins, norm = oracle.bin_samples(geom, samples, nfields, systs, params[param_offset:], param_stride)
    res = dict(bins=bins, norm=norm, geom=geom)
    if points is not None:
        rb = oracle.set_eval_points(geom, points, dataset)
        n = rb.size
        if pdf_size is None:
            pdf_size = pdf_offset + max(n, 1) * pdf_stride
        out = np.full(pdf_size, 12345.0, dtype=np.float32)
        oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=out, offset=pdf_offset, stride=pdf_stride)
        res.update(read_bins=rb, out=out)
    return res


def assert_same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def compare(kw, norm_offset=0, **gpu_kw):
    g = gpu_eval(norm_offset=norm_offset, **kw, **gpu_kw)
    o = oracle_eval(**kw)
    assert np.array_equal(g["bins"], o["bins"])
    assert int(g["norm"][norm_offset]) == o["norm"]
    if kw.get("points") is not None:
        assert np.array_equal(g["read_bins"], o["read_bins"])
        assert_same_bits(g["out"], o["out"])
    return g, o


# ---------------------------------------------------------------- the reference's known answers
def test_reference_known_answers_on_gpu(golden):
    """Every EvalHist case of test/test_pdfz.cpp, test_pdfz_2d.cpp, test_pdfz_syst.cpp."""
    for case in golden["cases"]:
        kw = dict(samples=np.asarray(case["samples"], np.float32), nfields=case["nfields"],
                  lower=case["lower"], upper=case["upper"], nbins=case["nbins"],
                  systs=case["systematics"], params=case["params"],
                  points=eval_points_with_dataset(case), pdf_offset=case["pdf_offset"],
                  pdf_stride=case["pdf_stride"], pdf_size=case["pdf_size"])
        g = gpu_eval(norm_offset=case["norm_offset"], norm_init=case["norm_init"], **kw)
        check_case_values(case, g["out"], g["norm"])
        o = oracle_eval(**kw)
        assert np.array_equal(g["bins"], o["bins"]), case["name"]
        assert_same_bits(g["out"], o["out"])          # untouched slots keep the sentinel too
        g["ev"].close()


def test_histogram_without_eval_points():
    # pdfz.cpp:472-476: EvalAsync(false) / no SetEvalPoints only fills (CreateHistogram path)
    rng = np.random.default_rng(0)
    samples = rng.normal(0.5, 0.3, size=(5000, 1)).astype(np.float32)
    kw = dict(samples=samples, nfields=1, lower=[0.0], upper=[1.0], nbins=[16], systs=[], params=[0.0])
    g = gpu_eval(**kw, do_eval_pdf=False)
    o = oracle_eval(**kw)
    assert np.array_equal(g["bins"], o["bins"]) and g["norm"][0] == o["norm"]
    assert g["bins"].sum() == o["norm"]


# ---------------------------------------------------------------- seeded parity, all systematic kinds
def table(rng, n, ncols, lo=-0.5, hi=1.5):
    return rng.uniform(lo, hi, size=(n, ncols)).astype(np.float32)


SYST_CASES = {
    "none": ([], [0.0]),
    "shift": ([dict(type="shift", obs=0, pars=[0])], [0.07]),
    "scale": ([dict(type="scale", obs=0, pars=[0])], [-0.03]),
    "ctscale": ([dict(type="ctscale", obs=0, pars=[0])], [0.11]),
    "resolution": ([dict(type="resolution_scale", obs=0, true_obs=-1, pars=[0])], [0.21]),
    "chain3": ([dict(type="shift", obs=-2, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                dict(type="resolution_scale", obs=0, true_obs=-1, pars=[2])], [0.05, -0.01, 0.08]),
    "poly2": ([dict(type="shift", obs=0, pars=[0, 1])], [0.02, 0.03]),
    "poly3_scale": ([dict(type="scale", obs=0, pars=[2, 0, 1])], [0.01, -0.02, 0.03]),
    "five_ops": ([dict(type="shift", obs=0, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                  dict(type="ctscale", obs=0, pars=[2]), dict(type="shift", obs=0, pars=[1]),
                  dict(type="resolution_scale", obs=0, true_obs=-1, pars=[0])], [0.01, 0.02, -0.03]),
}


def resolve(systs, nobs, nfields):
    """-1 -> last field (a truth column), -2 -> last observable."""
    out = []
    for s in systs:
        s = dict(s)
        if s.get("true_obs") == -1:
            s["true_obs"] = nfields - 1
        if s["obs"] == -2:
            s["obs"] = nobs - 1
        out.append(s)
    return out


@pytest.mark.parametrize("nobs,nbins", [(1, [37]), (2, [13, 7]), (3, [6, 5, 4]), (4, [3, 4, 2, 3]),
                                        (5, [2, 3, 2, 2, 3])])
@pytest.mark.parametrize("syst", sorted(SYST_CASES))
def test_parity_all_systematics(nobs, nbins, syst):
    rng = np.random.default_rng(100 * nobs + len(syst))
    nfields = nobs + 2                       # observables, one unused column, one truth column
    n = 20011                                # not a multiple of 4: exercises the NaN padding
    samples = table(rng, n, nfields)
    systs, params = SYST_CASES[syst]
    systs = resolve(systs, nobs, nfields)
    pts = np.concatenate([table(rng, 257, nobs), rng.integers(0, 2, size=(257, 1)).astype(np.float32)], axis=1)
    kw = dict(samples=samples, nfields=nfields, lower=[0.0] * nobs, upper=[1.0] * nobs, nbins=nbins,
              systs=systs, params=params, points=pts, dataset=1)
    compare(kw)


def test_parameter_offset_stride_and_output_offset_stride():
    rng = np.random.default_rng(5)
    samples = table(rng, 9001, 3)
    systs = [dict(type="shift", obs=0, pars=[1]), dict(type="resolution_scale", obs=1, true_obs=2, pars=[0])]
    params = [9.0, 9.0, 0.3, 9.0, -0.04, 9.0]          # offset 2 stride 2 -> p0 = 0.3, p1 = -0.04
    pts = np.concatenate([table(rng, 100, 2), np.zeros((100, 1), np.float32)], axis=1)
    kw = dict(samples=samples, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=[8, 9], systs=systs,
              params=params, param_offset=2, param_stride=2, points=pts, pdf_offset=5, pdf_stride=3)
    g, o = compare(kw, norm_offset=2, norm_init=[77, 88, 99])
    assert list(g["norm"][:2]) == [77, 88]


def test_systematic_on_a_non_observable_field():
    # a shift on the truth column changes what the later resolution systematic sees
    rng = np.random.default_rng(6)
    samples = table(rng, 5000, 3)
    systs = [dict(type="shift", obs=2, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=2, pars=[1])]
    compare(dict(samples=samples, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=[5, 5], systs=systs,
                 params=[0.2, 0.5]))


def test_shapes_without_specialization_use_the_generic_kernel():
    rng = np.random.default_rng(7)
    # 6 observables (no specialization) and 3 extra referenced fields (nslot = nobs + 3)
    samples = table(rng, 7001, 7)
    compare(dict(samples=samples, nfields=7, lower=[0.0] * 6, upper=[1.0] * 6, nbins=[2, 3, 2, 2, 2, 2],
                 systs=[dict(type="scale", obs=1, pars=[0]),
                        dict(type="resolution_scale", obs=0, true_obs=6, pars=[1])], params=[0.05, 0.1]))
    samples = table(rng, 7001, 6)
    systs = [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0]),
             dict(type="resolution_scale", obs=1, true_obs=4, pars=[0]),
             dict(type="resolution_scale", obs=1, true_obs=5, pars=[1, 0])]
    compare(dict(samples=samples, nfields=6, lower=[0.0] * 2, upper=[1.0] * 2, nbins=[9, 9], systs=systs,
                 params=[0.05, 0.1]))


def test_large_histogram_uses_global_atomics():
    rng = np.random.default_rng(8)
    samples = table(rng, 200003, 3, lo=-0.1, hi=1.1)
    nb = [300, 300]                                    # 90000 bins > LDS capacity
    pts = np.concatenate([table(rng, 1000, 2), np.zeros((1000, 1), np.float32)], axis=1)
    g, o = compare(dict(samples=samples, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=nb,
                        systs=[dict(type="shift", obs=0, pars=[0])], params=[0.01], points=pts))
    assert g.get("sparse")                             # the lookup evaluation counted only the event bins
    g, o = compare(dict(samples=samples, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=nb,
                        systs=[dict(type="shift", obs=0, pars=[0])], params=[0.01]), do_eval_pdf=False)
    assert not g.get("sparse")


@pytest.mark.parametrize("launch", [(256, 1), (256, 4), (512, 2), (1024, 1), (1024, 2)])
def test_launch_shapes_give_identical_counts(launch):
    rng = np.random.default_rng(9)
    samples = table(rng, 300007, 4)
    systs = resolve(SYST_CASES["chain3"][0], 3, 4)
    compare(dict(samples=samples, nfields=4, lower=[0.0] * 3, upper=[1.0] * 3, nbins=[20, 20, 20], systs=systs,
                 params=SYST_CASES["chain3"][1]), launch=launch)


def test_edge_inputs():
    # empty sample table; single sample; NaN and +-inf samples; samples exactly on the edges
    compare(dict(samples=np.zeros((0, 1), np.float32), nfields=1, lower=[0.0], upper=[1.0], nbins=[4],
                 systs=[], params=[0.0], points=np.array([[0.5, 0.0]], np.float32)))
    compare(dict(samples=np.array([[0.25]], np.float32), nfields=1, lower=[0.0], upper=[1.0], nbins=[4],
                 systs=[], params=[0.0]))
    edge = np.array([[np.nan], [np.inf], [-np.inf], [0.0], [1.0], [np.nextafter(np.float32(1), np.float32(0))],
                     [-0.0], [0.5]], np.float32)
    g, o = compare(dict(samples=edge, nfields=1, lower=[0.0], upper=[1.0], nbins=[4], systs=[], params=[0.0]))
    assert o["norm"] == 4
    # empty evaluation point list
    g = gpu_eval(np.array([[0.25]], np.float32), 1, [0.0], [1.0], [4], [], [0.0],
                 points=np.zeros((0, 2), np.float32))
    assert g["norm"][0] == 1 and g["read_bins"].size == 0


def test_determinism_and_reevaluation():
    # integer accumulation is order independent: two evaluations are bitwise identical, and the
    # histogram is re-zeroed on every evaluation (pdfz.cpp:454-458)
    rng = np.random.default_rng(10)
    samples = table(rng, 100003, 2)
    ev = pdfz.EvalHist(samples, 2, 2, [0.0, 0.0], [1.0, 1.0], [30, 30])
    ev.AddSystematic(pdfz.ShiftSystematic(0, 0))
    norm = DeviceArray.zeros(1, np.uint32)
    par = DeviceArray(np.array([0.05]))
    ev.SetNormalizationBuffer(norm)
    ev.SetParameterBuffer(par)
    runs = []
    for _ in range(3):
        ev.EvalAsync(False)
        ev.EvalFinished()
        runs.append((ev.GetBins(), norm.get()[0]))
    assert all(np.array_equal(r[0], runs[0][0]) and r[1] == runs[0][1] for r in runs)
    par.set(np.array([-0.05]))                       # parameter buffer is re-read each evaluation
    ev.EvalAsync(False)
    ev.EvalFinished()
    geom = oracle.HistGeometry([0.0, 0.0], [1.0, 1.0], [30, 30])
    ob, on = oracle.bin_samples(geom, samples, 2, [dict(type="shift", obs=0, pars=[0])], np.array([-0.05]))
    assert np.array_equal(ev.GetBins(), ob) and norm.get()[0] == on


def test_get_samples_roundtrip():
    rng = np.random.default_rng(11)
    samples = table(rng, 1003, 4)
    ev = pdfz.EvalHist(samples, 4, 2, [0.0, 0.0], [1.0, 1.0], [3, 3], dataset=7)
    got = ev.GetSamples().reshape(-1, 3)
    assert np.array_equal(got[:, :2], samples[:, :2]) and np.all(got[:, 2] == 7.0)


def test_eval_before_binding_buffers_is_an_error():
    ev = pdfz.EvalHist(np.zeros((4, 1), np.float32), 1, 1, [0.0], [1.0], [2])
    with pytest.raises(capi.SxmcError):
        ev.EvalAsync()


# ---------------------------------------------------------------- group (batched) evaluation
def build_group(rng, sizes, nobs, nbins, systs, params, nfields=None, points=None, lo=-0.5, hi=1.5):
    nfields = nfields or nobs + 1
    evs, tabs = [], []
    S = len(sizes)
    E = 0 if points is None else points.shape[0]
    lut = DeviceArray(np.full(max(1, S * E), 777.0, np.float32))
    norms = DeviceArray(np.full(S, 55, np.uint32))
    pbuf = DeviceArray(np.asarray(params, np.float64))
    for j, n in enumerate(sizes):
        t = table(rng, n, nfields, lo, hi)
        tabs.append(t)
        ev = pdfz.EvalHist(t, nfields, nobs, [0.0] * nobs, [1.0] * nobs, nbins, dataset=j % 2)
        for s in systs:
            ev.AddSystematic(make_systematic(s))
        if points is not None:
            ev.SetEvalPoints(points)
            ev.SetPDFValueBuffer(lut, j * E, 1)
        ev.SetNormalizationBuffer(norms, j)
        ev.SetParameterBuffer(pbuf, 0, 1)
        evs.append(ev)
    return evs, tabs, lut, norms, pbuf


def test_group_matches_oracle_per_signal_ragged_sizes():
    rng = np.random.default_rng(12)
    sizes = [1000, 2000000, 10007, 3, 0, 500001, 1, 999, 1234567]   # ragged, one empty, tiny ones
    nobs, nbins = 2, [16, 11]
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=2, pars=[1])]
    params = [0.03, -0.2]
    pts = np.concatenate([table(rng, 333, 2), rng.integers(0, 2, size=(333, 1)).astype(np.float32)], axis=1)
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, params, nfields=3, points=pts)
    group = nll.EvalGroup(evs)
    for launch, partition in [((0, 0), 0), ((256, 4), 1), ((1024, 1), 2), ((512, 2), 2), ((256, 1), 1)]:
        group.SetLaunchConfig(*launch)
        group.SetPartition(partition)
        group.EvalAsync(True)
        group.EvalFinished()
        got_lut = lut.get().reshape(len(sizes), -1)
        got_norms = norms.get()
        for j, t in enumerate(tabs):
            o = oracle_eval(t, 3, [0.0] * 2, [1.0] * 2, nbins, systs, params, points=pts, dataset=j % 2)
            assert np.array_equal(evs[j].GetBins(), o["bins"]), (launch, j)
            assert got_norms[j] == o["norm"]
            assert_same_bits(got_lut[j], o["out"])


def test_group_with_mixed_shapes_and_histogram_modes():
    rng = np.random.default_rng(13)
    # member 0: 1-D LDS; member 1: 2-D global-atomic (too many bins); member 2: 3-D with truth column
    t0, t1, t2 = table(rng, 40001, 1), table(rng, 50001, 2), table(rng, 30001, 4)
    e0 = pdfz.EvalHist(t0, 1, 1, [0.0], [1.0], [1000])
    e1 = pdfz.EvalHist(t1, 2, 2, [0.0, 0.0], [1.0, 1.0], [250, 250])
    e2 = pdfz.EvalHist(t2, 4, 3, [0.0] * 3, [1.0] * 3, [8, 8, 8])
    e2.AddSystematic(pdfz.ResolutionScaleSystematic(0, 3, 0))
    norms = DeviceArray.zeros(3, np.uint32)
    pbuf = DeviceArray(np.array([0.1]))
    for j, e in enumerate((e0, e1, e2)):
        e.SetNormalizationBuffer(norms, j)
        e.SetParameterBuffer(pbuf)
    g = nll.EvalGroup([e0, e1, e2])
    g.EvalAsync(False)
    g.EvalFinished()
    o0 = oracle_eval(t0, 1, [0.0], [1.0], [1000], [], [0.1])
    o1 = oracle_eval(t1, 2, [0.0, 0.0], [1.0, 1.0], [250, 250], [], [0.1])
    o2 = oracle_eval(t2, 4, [0.0] * 3, [1.0] * 3, [8, 8, 8],
                     [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0])], [0.1])
    for e, o in ((e0, o0), (e1, o1), (e2, o2)):
        assert np.array_equal(e.GetBins(), o["bins"])
    assert list(norms.get()) == [o0["norm"], o1["norm"], o2["norm"]]


def test_interleaved_and_sliced_partitions_agree_on_equal_members():
    rng = np.random.default_rng(14)
    sizes = [400003] * 12
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]        # the C3 program (static kernel)
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, [20, 20, 20], systs, [0.02, -0.01, 0.07], nfields=5)
    group = nll.EvalGroup(evs)
    results = []
    for partition in (1, 2, 0):
        group.SetPartition(partition)
        group.EvalAsync(False)
        group.EvalFinished()
        results.append(([e.GetBins() for e in evs], norms.get()))
    for bins, nrm in results[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(bins, results[0][0])) and np.array_equal(nrm, results[0][1])
    o = oracle_eval(tabs[5], 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], systs, [0.02, -0.01, 0.07])
    assert np.array_equal(results[0][0][5], o["bins"]) and results[0][1][5] == o["norm"]


@pytest.mark.parametrize("nobs,nbins,systs,params", [
    (3, [20, 20, 20], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                       dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])], [0.02, -0.01, 0.07]),   # 1 byte
    (3, [10, 30, 40], [dict(type="scale", obs=0, pars=[0])], [0.03]),                                       # 2 bytes
    (3, [4, 3, 5], [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0])], [0.2]),
    (2, [200, 200], [dict(type="shift", obs=1, pars=[0])], [-0.02]),                                        # 40000 bins
    (2, [9, 7], [dict(type="scale", obs=0, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=2, pars=[1])],
     [0.01, 0.1]),
])
def test_prebinned_observables_give_identical_histograms(nobs, nbins, systs, params):
    """Observables no systematic writes are streamed as one pre-binned column (static programs): the
    histograms must be those of the oracle and of the same launch with pre-binning off."""
    rng = np.random.default_rng(15)
    nfields = nobs + 2
    sizes = [70001, 3, 123457]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, params, nfields=nfields)
    # edge cases in the pre-binned columns: NaN, exact edges, out of domain
    group = nll.EvalGroup(evs)
    group.SetBucketing(False)          # (bucketing would take precedence: test_bucketed_table_* covers it)
    results = []
    for prebin in (True, False):
        group.SetPrebinning(prebin)
        group.EvalAsync(False)
        group.EvalFinished()
        results.append(([e.GetBins() for e in evs], norms.get()))
        fr = group.AlgorithmicBytes()["fill_read"]
        results[-1] += (fr,)
    assert results[0][2] < results[1][2]                   # fewer bytes to stream
    for j, t in enumerate(tabs):
        o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params)
        for bins, nrm, _ in results:
            assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"]


C5_LIKE = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
           dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])]


@pytest.mark.parametrize("nobs,nbins,systs,params,nfields", [
    (3, [20, 20, 20], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                       dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])], [0.02, -0.01, 0.07], 5),  # C3
    (3, [10, 30, 40], [dict(type="scale", obs=0, pars=[0])], [0.03], 5),        # 1271 buckets: small tables fall back
    (3, [4, 3, 5], [dict(type="resolution_scale", obs=0, true_obs=3, pars=[0])], [0.2], 5),
    (2, [200, 200], [dict(type="shift", obs=1, pars=[0])], [-0.02], 4),         # the written observable is the last one
    (2, [9, 7], [dict(type="scale", obs=0, pars=[0]), dict(type="resolution_scale", obs=0, true_obs=2, pars=[1])],
     [0.01, 0.1], 4),
    (3, [5, 5, 8], [dict(type="ctscale", obs=2, pars=[0])], [0.04], 4),
    (3, [6, 7, 8], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=1, pars=[1])], [0.01, -0.02], 4),
    (5, [6, 5, 4, 3, 2], C5_LIKE, [0.02, -0.01, 0.07], 7),                      # C5's shape, histogram in LDS
    (5, [60, 50, 8, 3, 2], C5_LIKE, [0.02, -0.01, 0.07], 7),                    # ... and beyond LDS (144000 bins)
])
def test_bucketed_table_gives_identical_histograms(nobs, nbins, systs, params, nfields):
    """Samples grouped by the bins of the observables no systematic writes; the fill streams only the columns
    that change + one bin offset per 256-sample granule.  Histograms and norms must be those of the oracle and
    of the same launch with bucketing off (pre-binned column) and with both off."""
    rng = np.random.default_rng(15)
    sizes = [70001, 3, 123457, 0, 255, 257]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, params, nfields=nfields)
    group = nll.EvalGroup(evs)
    group.SetOrdering(False)           # plain bucketing here; tests/test_gpu_ordered.py covers the ordered form
    results = []
    for bucket, prebin in ((True, True), (False, True), (False, False)):
        group.SetBucketing(bucket)
        group.SetPrebinning(prebin)
        for partition in ((0, 1, 2) if bucket else (0,)):
            group.SetPartition(partition)
            group.EvalAsync(False)
            group.EvalFinished()
            results.append(([e.GetBins() for e in evs], norms.get(), group.AlgorithmicBytes()["fill_read"]))
    group.SetPartition(0)
    assert results[0][2] < results[-1][2]                  # fewer bytes to stream
    for j, t in enumerate(tabs):
        o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params)
        for bins, nrm, _ in results:
            assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"]
    # the caller's row order is untouched (GetSamples reads the original table, pdfz.h:542-556)
    got = evs[0].GetSamples().reshape(-1, nobs + 1)
    assert np.array_equal(got[:, :nobs].view(np.uint32), tabs[0][:, :nobs].view(np.uint32))


def test_bucketed_table_edge_values():
    """The untouched observables carry NaN / inf / exact edges / -0.0 / values one float below an edge (the bucket
    keys are formed with the fill's own arithmetic, pdfz.cpp:388-398): the bucketed evaluation must count
    exactly what the oracle counts."""
    rng = np.random.default_rng(19)
    n = 40000
    tab = table(rng, n, 4)
    edge = np.array([np.nan, np.inf, -np.inf, 0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), -0.0, 0.5,
                     np.nextafter(np.float32(0.5), np.float32(0)), 0.999], np.float32)
    tab[:, 1] = np.where(rng.uniform(size=n) < 0.3, rng.choice(edge, size=n), tab[:, 1])
    tab[:, 2] = np.where(rng.uniform(size=n) < 0.3, rng.choice(edge, size=n), tab[:, 2])
    lower, upper, nbins = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [7, 3, 3]
    geom = oracle.HistGeometry(lower, upper, nbins)
    systs = [dict(type="shift", obs=0, pars=[0])]
    for bucket in (True, False):
        ev = pdfz.EvalHist(tab, 4, 3, lower, upper, nbins)
        ev.AddSystematic(make_systematic(systs[0]))
        norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.array([0.013]))
        ev.SetNormalizationBuffer(norm)
        ev.SetParameterBuffer(pbuf)
        group = nll.EvalGroup([ev])
        group.SetBucketing(bucket)
        group.SetOrdering(False)
        group.EvalAsync(False)
        group.EvalFinished()
        bins, nrm = oracle.bin_samples(geom, tab, 4, systs, np.array([0.013]))
        assert np.array_equal(ev.GetBins(), bins) and norm.get()[0] == nrm
        group.close()
        ev.close()


def test_prebinned_column_edge_values():
    # the untouched observable carries NaN / inf / exact edges / -0.0: same accept/reject as the oracle
    edge = np.array([np.nan, np.inf, -np.inf, 0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), -0.0, 0.5,
                     0.25, 0.999], np.float32)
    tab = np.zeros((edge.size * 3, 3), np.float32)
    tab[:, 0] = np.tile(np.array([0.1, 0.5, 0.9], np.float32), edge.size)
    tab[:, 1] = np.repeat(edge, 3)
    systs = [dict(type="shift", obs=0, pars=[0])]
    kw = dict(samples=tab, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=[4, 5], systs=systs, params=[0.05])
    g, o = compare(kw)
    assert o["norm"] == 3 * 6 and g["bins"].sum() == o["norm"]     # 0, nextbelow(1), -0, 0.5, 0.25, 0.999


def test_sparse_counting_matches_dense_lookup():
    """Histograms beyond LDS capacity, evaluated for lookup: only the distinct event bins are counted
    (bit filter + hash table).  lut and norms must equal the dense evaluation's, bit for bit, including
    events outside the domain (-1), of another dataset (-2), many events in one bin, and no usable event."""
    rng = np.random.default_rng(16)
    nb = [120, 110, 7]                                   # 92400 bins
    sizes = [150001, 70001, 9]
    pts = np.concatenate([table(rng, 4000, 3, lo=-0.2, hi=1.2), rng.integers(0, 2, size=(4000, 1)).astype(np.float32)],
                         axis=1)
    pts[:500, :3] = pts[0, :3]                           # 500 events share one bin
    systs = [dict(type="scale", obs=0, pars=[0]), dict(type="resolution_scale", obs=1, true_obs=3, pars=[1, 0])]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, nb, systs, [0.02, 0.1], nfields=5, points=pts)
    group = nll.EvalGroup(evs)
    out = {}
    for sparse in (True, False):
        group.SetSparse(sparse)
        lut.set(np.full(lut.size, 777.0, np.float32))
        group.EvalAsync(True)
        group.EvalFinished()
        out[sparse] = (lut.get(), norms.get())
        if sparse:
            with pytest.raises(capi.SxmcError):
                evs[0].GetBins()
    assert np.array_equal(out[True][0].view(np.uint32), out[False][0].view(np.uint32))
    assert np.array_equal(out[True][1], out[False][1])
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 5, [0.0] * 3, [1.0] * 3, nb, systs, [0.02, 0.1], points=pts, dataset=j % 2)
        assert_same_bits(out[True][0].reshape(len(sizes), -1)[j], o["out"])
        assert out[True][1][j] == o["norm"]
        assert np.array_equal(evs[j].GetBins(), o["bins"])          # dense evaluation ran last
    # no event inside the domain / of this dataset: nothing to count, everything NaN or 0
    far = np.concatenate([np.full((5, 3), 7.0, np.float32), np.zeros((5, 1), np.float32)], axis=1)
    for e in evs:
        e.SetEvalPoints(far)
    group.SetSparse(True)
    group.EvalAsync(True)
    group.EvalFinished()
    assert np.all(np.isnan(lut.get()[:5]))


@pytest.mark.parametrize("nobs,nb,systs,params,nfields", [
    (5, [60, 50, 8, 3, 2], C5_LIKE, [0.02, -0.01, 0.07], 7),                      # C5's shape: 144000 bins
    (3, [4000, 4, 3], [dict(type="scale", obs=0, pars=[0])], [0.02], 4),          # one written observable
    (3, [5, 3000, 4], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=1, pars=[1])], [0.01, -0.02], 4),
    (2, [3, 20000], [dict(type="shift", obs=1, pars=[0])], [0.0001], 3),          # 3 buckets, > 1024 event bins each
])
def test_sparse_counting_over_bucketed_runs(nobs, nb, systs, params, nfields):
    """Histograms beyond LDS capacity whose table is bucketed: every wave walks its own run of granules, the event
    bins are grouped by bucket and counted in per-wave LDS tables that are flushed when the wave leaves the bucket
    (fill_sparse_kernel).  lut and norms must equal the dense evaluation's and the oracle's, bit for bit --
    including events outside the domain (-1), of another dataset (-2), many events in one bin, buckets with
    more event bins than a wave's table holds, and a second set of evaluation points."""
    rng = np.random.default_rng(26)
    sizes = [150001, 70001, 9, 0, 30011]
    pts = np.concatenate([table(rng, 6000, nobs, lo=-0.2, hi=1.2), rng.integers(0, 2, size=(6000, 1)).astype(np.float32)],
                         axis=1)
    pts[:500, :nobs] = pts[0, :nobs]                      # 500 events share one bin
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nb, systs, params, nfields=nfields, points=pts,
                                              lo=-0.05, hi=1.05)
    group = nll.EvalGroup(evs)
    fr = group.AlgorithmicBytes()["fill_read"]
    assert fr < 4.0 * sum(sizes) * (nobs + 1) * 0.8       # the big members really are bucketed

    def check(points):
        out = {}
        for sparse, bucket, order in ((True, True, True), (True, True, False), (False, True, True), (False, True, False),
                                      (True, False, False)):
            group.SetSparse(sparse)
            group.SetBucketing(bucket)
            group.SetOrdering(order, force=True)       # (on: the written observable that is only shifted / scaled is ordered)
            info = group.LaunchInfo()
            assert "failed" not in info, info
            lut.set(np.full(lut.size, 777.0, np.float32))
            group.EvalAsync(True)
            group.EvalFinished()
            out[(sparse, bucket) if not order else (sparse, bucket, order)] = (lut.get(), norms.get())
        ne = points.shape[0]
        for k, v in out.items():
            assert np.array_equal(v[0].view(np.uint32), out[(True, True)][0].view(np.uint32)), k
            assert np.array_equal(v[1], out[(True, True)][1]), k
        for j, t in enumerate(tabs):
            o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nb, systs, params, points=points, dataset=j % 2)
            got = out[(True, True)][0][j * 6000: j * 6000 + ne]
            assert_same_bits(got, o["out"])
            assert out[(True, True)][1][j] == o["norm"]

    check(pts)
    group.SetBucketing(True)
    group.SetOrdering(True, force=True)
    pts2 = pts[1000:3500].copy()                          # a new data set: the bucket tables are rebuilt
    for e in evs:
        e.SetEvalPoints(pts2)
    check(pts2)


RTC_CASES = [
    # BASELINE config 3's program + a cos-theta scale on c: every observable is written, no table entry
    ("c3+ctscale", 3, [20, 20, 20], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                                     dict(type="resolution_scale", obs=0, true_obs=3, pars=[2]),
                                     dict(type="ctscale", obs=2, pars=[3])], [0.02, -0.01, 0.07, 0.03], 5, "rows"),
    # a 4-D shape, one observable left alone: bucketed, program not in the table
    ("4d", 4, [8, 7, 6, 5], [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
                             dict(type="resolution_scale", obs=0, true_obs=4, pars=[2]),
                             dict(type="ctscale", obs=3, pars=[3])], [0.02, -0.01, 0.07, 0.03], 6, "bucketed"),
    # polynomial systematics (3 and 2 coefficients), one observable left alone
    ("poly", 3, [12, 9, 10], [dict(type="shift", obs=0, pars=[0, 1, 2]), dict(type="scale", obs=1, pars=[3, 0])],
     [0.02, -0.03, 0.01, 0.05], 4, "bucketed"),
    # the histogram fills LDS to the last word (40 800 bins = 163 KB of dynamic LDS through the module launch)
    ("lds-full", 2, [200, 204], [dict(type="ctscale", obs=0, pars=[0]), dict(type="shift", obs=1, pars=[1])],
     [0.01, 0.02], 3, "rows"),
]


@pytest.mark.parametrize("name,nobs,nbins,systs,params,nfields,table_kind", RTC_CASES, ids=[c[0] for c in RTC_CASES])
def test_runtime_specialised_kernels_match_the_decoded_program_and_the_oracle(name, nobs, nbins, systs, params, nfields,
                                                                              table_kind):
    """Programs of systematics that are not in the library's table are compiled at set-up (hiprtc) from the same
    kernel template; the result must be that of the run-time decoded program and of the oracle, bit for bit."""
    rng = np.random.default_rng(31)
    sizes = [90001, 5, 40013]
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, nobs, nbins, systs, params, nfields=nfields, lo=-0.1, hi=1.1)
    group = nll.EvalGroup(evs)
    group.SetOrdering(False)           # (tests/test_gpu_ordered.py covers the ordered form; once below as well)
    info = group.LaunchInfo()
    assert "program=runtime" in info and "table=" + table_kind in info, info
    assert "failed" not in info, info
    results = []
    for rtc, order in ((True, False), (True, True), (False, False)):
        group.SetRuntimeKernels(rtc)
        group.SetOrdering(order, force=True)
        group.EvalAsync(False)
        group.EvalFinished()
        results.append(([e.GetBins() for e in evs], norms.get()))
    assert "program=runtime" not in group.LaunchInfo()
    for j, t in enumerate(tabs):
        o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params)
        for bins, nrm in results:
            assert np.array_equal(bins[j], o["bins"]) and nrm[j] == o["norm"]


def test_runtime_specialised_sparse_runs_kernel():
    """Histograms beyond LDS + a program that is not built in: the run-time kernels cover the dense fill of the
    bucketed table and the sparse counting over runs."""
    rng = np.random.default_rng(32)
    nobs, nb, nfields = 5, [60, 50, 8, 3, 2], 7
    systs = C5_LIKE + [dict(type="ctscale", obs=2, pars=[1])]
    params = [0.02, -0.01, 0.07]
    pts = np.concatenate([table(rng, 5000, nobs, lo=-0.2, hi=1.2), np.zeros((5000, 1), np.float32)], axis=1)
    evs, tabs, lut, norms, pbuf = build_group(rng, [150001, 60001], nobs, nb, systs, params, nfields=nfields, points=pts,
                                              lo=-0.05, hi=1.05)
    group = nll.EvalGroup(evs)
    out = {}
    for order, kind in ((False, "bucketed"), (True, "ordered")):      # (ordered: c, the observable with the fewest bins)
        group.SetOrdering(order, force=True)
        info = group.LaunchInfo()
        assert "program=runtime" in info and "table=%s+runs(runtime)" % kind in info and "failed" not in info, info
        for sparse in (True, False):
            group.SetSparse(sparse)
            lut.set(np.full(lut.size, 777.0, np.float32))
            group.EvalAsync(True)
            group.EvalFinished()
            out[(sparse, order)] = (lut.get(), norms.get())
    for k in list(out):
        assert np.array_equal(out[k][0].view(np.uint32), out[(False, False)][0].view(np.uint32)), k
        assert np.array_equal(out[k][1], out[(False, False)][1]), k
    out = {True: out[(True, True)], False: out[(False, False)]}
    for j, t in enumerate(tabs):
        o = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nb, systs, params, points=pts, dataset=j % 2)
        assert_same_bits(out[True][0].reshape(2, -1)[j], o["out"])
        assert out[True][1][j] == o["norm"]


def test_polynomial_powers_are_rounded_once_like_libm_pow():
    """p = sum_i c_i * pow(x, i) (pdfz.cpp:310-314): the reference's x^i is libm's pow -- the correctly rounded
    power for these small integer exponents, except where libm itself misrounds.  The kernels keep the running
    power as an unevaluated sum (error-free FMA product) and round once: every x^i must equal the exactly
    rounded rational power; repeated multiplication would be off in the last bit for ~1 in 5 values at i >= 3."""
    import ctypes as C
    from fractions import Fraction
    rng = np.random.default_rng(41)
    x = np.concatenate([rng.uniform(-3, 12, 6000).astype(np.float32).astype(np.float64),     # widened sample fields
                        rng.uniform(-3, 12, 6000) * (1 + rng.normal(0, 1e-3, 6000))])        # ... and transformed ones
    dx, dout = DeviceArray(x), DeviceArray.zeros(x.size, np.float64)
    naive_off = libm_off = 0
    hooks = capi.measure_lib()      # (a test hook of the measurement build: pow_step of the same fill_kernels.inc.h; the
    for i in range(0, 8):           #  product's use of it is what the polynomial cases below hold against the oracle)
        assert hooks.sxmc_debug_pow_int(capi.ptr(dx), x.size, i, capi.ptr(dout)) == capi.OK, hooks.sxmc_last_error()
        got = dout.get()
        exact = np.array([float(Fraction(v) ** i) for v in x])          # Fraction -> float rounds correctly
        assert np.array_equal(got, exact), "x^%d is not the correctly rounded power" % i
        naive = np.ones_like(x)
        for _ in range(i):
            naive = naive * x
        naive_off += int(np.count_nonzero(naive != exact))
        libm_off += int(np.count_nonzero(np.array([math.pow(v, i) for v in x]) != exact))
    assert naive_off > 1000                  # what the kernels did before: thousands of last-bit differences
    assert libm_off < naive_off // 20        # libm's pow is (almost always) the correctly rounded power


def test_polynomial_systematics_at_a_million_samples():
    """Three- and four-coefficient systematics on 1.2e6 samples: bins and norm equal the oracle's (libm pow)."""
    rng = np.random.default_rng(42)
    systs = [dict(type="shift", obs=0, pars=[0, 1, 2, 3]), dict(type="scale", obs=1, pars=[4, 5, 6])]
    params = [0.01, -0.02, 0.015, -0.004, 0.01, 0.03, -0.02]
    t = table(rng, 1200000, 3, lo=-0.1, hi=1.1)
    kw = dict(samples=t, nfields=3, lower=[0.0, 0.0], upper=[1.0, 1.0], nbins=[60, 50], systs=systs, params=params)
    compare(kw)


@pytest.mark.parametrize("seed", range(40))
def test_random_programs_every_path_matches_the_oracle(seed):
    """Random shapes (1-5 observables, 0-2 extra fields), random programs (1-4 systematics of any kind on any
    observable, truth field an extra field or another observable, 1-3 polynomial coefficients), with and without
    points: whatever kernel the plan picks -- built in, compiled at run time, decoded; rows, pre-binned, bucketed --
    bins, norm and lookup values must be the oracle's, and the same with bucketing / run-time kernels switched off."""
    rng = np.random.default_rng(1000 + seed)
    nobs = int(rng.integers(1, 6))
    nextra = int(rng.integers(0, 3))
    nfields = nobs + nextra + 1                                   # + the dataset column
    nbins = [int(rng.integers(2, 9)) for _ in range(nobs)]
    if seed % 5 == 4:
        nbins[int(rng.integers(0, nobs))] = 50000                 # beyond LDS: global histogram / sparse counting
    kinds = ["shift", "scale", "ctscale", "resolution_scale"]
    systs, npar = [], 0
    for _ in range(int(rng.integers(1, 5))):
        kind = kinds[int(rng.integers(0, 4))]
        ncoef = int(rng.choice([1, 1, 1, 2, 3]))
        d = dict(type=kind, obs=int(rng.integers(0, nobs)), pars=list(range(npar, npar + ncoef)))
        npar += ncoef
        if kind == "resolution_scale":
            choices = [f for f in range(nobs + nextra) if f != d["obs"]] or [d["obs"]]
            d["true_obs"] = int(rng.choice(choices))
        systs.append(d)
    params = list(rng.normal(0, 0.03, npar))
    pts = np.concatenate([table(rng, 300, nobs, lo=-0.1, hi=1.1), np.zeros((300, 1), np.float32)], axis=1)
    evs, tabs, lut, norms, pbuf = build_group(rng, [20011, 4001], nobs, nbins, systs, params, nfields=nfields,
                                              points=pts, lo=-0.1, hi=1.1)
    group = nll.EvalGroup(evs)
    want = [oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, systs, params, points=pts, dataset=j % 2)
            for j, t in enumerate(tabs)]
    for bucket, rtc, order in ((True, True, True), (True, True, False), (False, True, False), (True, False, True),
                               (False, False, False)):
        group.SetBucketing(bucket)
        group.SetRuntimeKernels(rtc)
        group.SetOrdering(order, force=True)
        info = group.LaunchInfo()
        assert "failed" not in info, info
        lut.set(np.full(lut.size, 777.0, np.float32))
        group.EvalAsync(True)
        group.EvalFinished()
        got_lut, got_norms = lut.get().reshape(2, -1), norms.get()
        group.EvalAsync(False)
        group.EvalFinished()
        for j in range(2):
            assert got_norms[j] == want[j]["norm"], (info, systs)
            assert_same_bits(got_lut[j], want[j]["out"])
            assert np.array_equal(evs[j].GetBins(), want[j]["bins"]), (info, systs)


def test_shared_table_with_different_prebinned_columns():
    """Two evaluators over one sample table whose systematics leave different observables untouched: each
    group keeps its own pre-binned column, and evaluating one does not disturb the other."""
    rng = np.random.default_rng(17)
    nobs, nfields, nbins = 3, 5, [12, 9, 10]
    t = table(rng, 250001, nfields)
    sa = [dict(type="scale", obs=0, pars=[0])]                                     # untouched: 1, 2
    sb = [dict(type="shift", obs=1, pars=[1]), dict(type="scale", obs=0, pars=[0]),
          dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]              # untouched: 2
    params = [0.03, -0.04, 0.2]
    pbuf = DeviceArray(np.asarray(params, np.float64))
    norms = DeviceArray.zeros(2, np.uint32)
    base = pdfz.EvalHist(t, nfields, nobs, [0.0] * nobs, [1.0] * nobs, nbins)
    ea, eb = pdfz.EvalHist.Shared(base), pdfz.EvalHist.Shared(base)
    for e, systs in ((ea, sa), (eb, sb)):
        for s in systs:
            e.AddSystematic(make_systematic(s))
    for j, e in enumerate((ea, eb)):
        e.SetNormalizationBuffer(norms, j)
        e.SetParameterBuffer(pbuf)
    ga, gb = nll.EvalGroup([ea]), nll.EvalGroup([eb])
    assert ga.AlgorithmicBytes()["fill_read"] < gb.AlgorithmicBytes()["fill_read"] < 4 * 4 * 250001 * 1.01
    oa = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, sa, params)
    ob = oracle_eval(t, nfields, [0.0] * nobs, [1.0] * nobs, nbins, sb, params)
    for g, e, o, j in ((ga, ea, oa, 0), (gb, eb, ob, 1), (ga, ea, oa, 0), (gb, eb, ob, 1)):
        g.EvalAsync(False)
        g.EvalFinished()
        assert np.array_equal(e.GetBins(), o["bins"]) and norms.get()[j] == o["norm"]


def test_maximum_sizes_of_the_interface():
    """MAX_NFIELDS = 10 fields (pdfz.cpp:17), SXMC_MAX_SYST = 16 systematics on one evaluator, 8 polynomial
    coefficients in one systematic; one more of each is refused."""
    rng = np.random.default_rng(18)
    nfields, nobs = 10, 4
    samples = table(rng, 30011, nfields, lo=0.05, hi=0.95)
    kinds = ["shift", "scale", "ctscale", "resolution_scale"]
    systs = []
    for q in range(15):
        s = dict(type=kinds[q % 4], obs=q % nobs, pars=[q % 5])
        if s["type"] == "resolution_scale":
            s["true_obs"] = 4 + (q % 6)                      # every extra field gets referenced
        systs.append(s)
    systs.append(dict(type="shift", obs=1, pars=[0, 1, 2, 3, 4, 0, 1, 2]))   # 8 coefficients
    params = [0.004, -0.003, 0.002, 0.001, -0.002]
    pts = np.concatenate([table(rng, 300, nobs), np.zeros((300, 1), np.float32)], axis=1)
    kw = dict(samples=samples, nfields=nfields, lower=[0.0] * nobs, upper=[1.0] * nobs, nbins=[5, 4, 3, 6],
              systs=systs, params=params, points=pts)
    g, o = compare(kw)
    assert o["norm"] > 1000
    ev = g["ev"]
    with pytest.raises(pdfz.Error):
        ev.AddSystematic(make_systematic(dict(type="shift", obs=0, pars=[0])))          # a 17th
    with pytest.raises(pdfz.Error):
        pdfz.EvalHist(np.zeros((3, 11), np.float32), 11, 2, [0.0, 0.0], [1.0, 1.0], [2, 2])   # 11 fields
    e2 = pdfz.EvalHist(samples, nfields, nobs, [0.0] * nobs, [1.0] * nobs, [5, 4, 3, 6])
    with pytest.raises(pdfz.Error):
        e2.AddSystematic(make_systematic(dict(type="shift", obs=0, pars=[0] * 9)))      # 9 coefficients


@pytest.mark.parametrize("nbins", [[40832], [40833], [232, 176], [232, 177]])
def test_histograms_at_the_lds_capacity_boundary(nbins):
    """40 832 bins is the largest LDS-private histogram; one more bin switches to the HBM-resident modes."""
    rng = np.random.default_rng(19)
    nobs = len(nbins)
    samples = table(rng, 150001, nobs + 1, lo=-0.05, hi=1.05)
    pts = np.concatenate([table(rng, 500, nobs), np.zeros((500, 1), np.float32)], axis=1)
    kw = dict(samples=samples, nfields=nobs + 1, lower=[0.0] * nobs, upper=[1.0] * nobs, nbins=nbins,
              systs=[dict(type="scale", obs=0, pars=[0])], params=[0.01], points=pts)
    compare(kw)                                       # lookup evaluation (sparse counters beyond the boundary)
    kw.pop("points")
    compare(kw, do_eval_pdf=False)                    # dense histogram only


@pytest.mark.parametrize("lut_output", [True, False])
def test_fused_evaluation_without_events(lut_output):
    """No data events at all: the fused evaluation still fills the histograms and returns a zero event sum."""
    rng = np.random.default_rng(20)
    evs, tabs, lut, norms, pbuf = build_group(rng, [5001, 7003], 2, [6, 5], [dict(type="shift", obs=0, pars=[0])],
                                              [0.02], points=np.zeros((0, 3), np.float32))
    group = nll.EvalGroup(evs)
    group.SetLutOutput(lut_output)
    sums = DeviceArray(np.full(1024, 7.0))
    n = group.EvalNllAsync(None, pbuf, DeviceArray(np.ones(2)), DeviceArray(np.array([5001, 7003], np.uint32)),
                           DeviceArray(np.zeros(2, np.int16)), norms, sums)
    group.EvalFinished()
    assert n >= 1 and np.all(sums.get()[:n] == 0.0)
    for j, t in enumerate(tabs):
        o = oracle_eval(t, 3, [0.0] * 2, [1.0] * 2, [6, 5], [dict(type="shift", obs=0, pars=[0])], [0.02])
        assert np.array_equal(evs[j].GetBins(), o["bins"]) and norms.get()[j] == o["norm"]


def test_optimize_keeps_results_and_only_acts_on_pure_streams():
    """sxmc_group_optimize (EvalHist::Optimize for the batched launch): trial launches pick a lane count per CU;
    histograms afterwards are those of any other shape; groups with a hand-set shape, a long run-time program
    or little work are left alone."""
    rng = np.random.default_rng(21)
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
    params = [0.02, -0.01, 0.07]
    sizes = [4600003] * 12                # 55e6 samples, half of them inside the untouched observable's domain,
    evs, tabs, lut, norms, pbuf = build_group(rng, sizes, 3, [20, 20, 20], systs, params, nfields=5)   # x 8 B: long
    group = nll.EvalGroup(evs)
    chosen = group.Optimize()
    assert chosen in (448, 512, 576, 640, 768, 896, 1024)           # (the larger shapes: tables streamed as codes)
    group.EvalAsync(False)
    group.EvalFinished()
    o = oracle_eval(tabs[3], 5, [0.0] * 3, [1.0] * 3, [20, 20, 20], systs, params)
    assert np.array_equal(evs[3].GetBins(), o["bins"]) and norms.get()[3] == o["norm"]
    group.SetLaunchConfig(256, 2)
    assert group.Optimize() == 0                                        # shape set by hand: untouched
    small = build_group(rng, [5000, 7000], 2, [9, 7], [dict(type="shift", obs=0, pars=[0])], [0.01])
    assert nll.EvalGroup(small[0]).Optimize() == 0                      # short launches take all the waves
