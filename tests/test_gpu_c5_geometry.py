"""GPU parity at BASELINE config 5's REAL geometry: 5 observables binned (200, 200, 200, 4, 4) = 1.28e8 bins
per signal (512 MB dense histograms, far beyond LDS), F = 7 fields, shift + scale + resolution_scale
floated, 1e5 data events with out-of-domain (-1), other-dataset (-2) and duplicate-bin events.  The
filter / table / counter sizes of the sparse (event-bin) evaluation depend on this geometry, so it is
tested here and not only at the 40x40x40x4x4 miniature of test_gpu_nll.py.

Oracle: bin_samples / eval_pdf / the NLL chain of oracle/sxmc_oracle.c (pdfz.cpp:349-436,
nll_kernels.cpp:89-188), one signal at a time (a dense oracle histogram is 512 MB).
Bar: lookup-table bit patterns, normalisations and (dense evaluation) every bin count equal; NLL within
1e-12 relative (north_star asks 1e-6)."""
import numpy as np
import pytest

from oracle import oracle
from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC

pytestmark = pytest.mark.gpu

NBINS = (200, 200, 200, 4, 4)
COUNTS = (2_200_003, 2_000_001, 2_500_000)       # ragged on purpose (not multiples of 4 / 64 / 256)


def c5_workload(counts=COUNTS, nevents=100_000, seed=55):
    rng = np.random.default_rng(seed)
    signals = []
    for j, n in enumerate(counts):
        e_true = rng.normal(2.0 + 0.3 * j, 1.2, size=n)
        e = e_true + rng.normal(0.0, 0.3, size=n)
        cols = [e, 6.0 * rng.uniform(size=n) ** (1.0 / 3.0), rng.uniform(-1, 1, size=n),
                rng.uniform(0, 1, size=n), rng.uniform(0, 1, size=n), e_true, np.zeros(n)]
        tab = np.stack(cols, axis=1).astype(np.float32)
        tab[::1013, 2] = np.nan                 # NaN fields are outside the domain
        tab[5::997, 3] = 1.0                    # exactly the upper edge: outside
        tab[7::991, 4] = 0.0                    # exactly the lower edge: inside
        signals.append(workloads.Signal(tab, 7, nexpected=50.0 + j, source_id=j))
    systs = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
             dict(type="resolution_scale", obs=0, true_obs=5, pars=[2])]
    ev = workloads._events_from_mixture(rng, signals, 5, nevents, None, None)
    ev[::17, 0] = -3.0                          # outside the domain           -> read_bins -1
    ev[3::29, 5] = 1.0                          # another experiment's data set -> read_bins -2
    ev[1::7] = ev[0::7][: ev[1::7].shape[0]]    # duplicates: several events in one bin
    return workloads.Workload("C5", 5, [0.0, 0.0, -1.0, 0.0, 0.0], [10.0, 6.0, 1.0, 1.0, 1.0], list(NBINS), signals,
                              systs, workloads.C3_SIGMAS, ev, "C5 geometry, %d signals" % len(counts))


@pytest.fixture(scope="module")
def c5():
    w = c5_workload()
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    assert geom.total_nbins == 128_000_000
    return w, geom


def oracle_at(w, geom, vector, keep_bins):
    ne = w.events.shape[0]
    lut = np.zeros((w.nsignals, ne), np.float32)
    norms = np.zeros(w.nsignals, np.uint32)
    kept = []
    for j, s in enumerate(w.signals):
        rb = oracle.set_eval_points(geom, w.events, s.dataset)
        bins, norm = oracle.bin_samples(geom, s.samples, s.nfields, w.systematics, vector[w.nsources:])
        oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=lut[j])
        norms[j] = norm
        kept.append(bins if keep_bins else None)
    val, _ = oracle.full_nll(lut, vector, ne, w.nsignals, w.nsources, w.parameter_means(), w.parameter_sigmas(),
                             [s.nexpected for s in w.signals], [s.n_mc for s in w.signals],
                             [s.source_id for s in w.signals], norms)
    return val, kept, norms, lut


def close_chain(m):
    for p in m.pdfs:
        p.close()
    m.group.close()


@pytest.mark.parametrize("sparse,order", [(True, True), (False, True), (True, False), (False, False)])
def test_c5_real_geometry_whole_step_against_the_oracle(c5, sparse, order):
    w, geom = c5
    m = MCMC(w, seed=21, fused=True)
    m.group.SetSparse(sparse)
    m.group.SetOrdering(order, force=True)          # forced on: r, which is only shifted, is the ordered observable
    # (off by default at this geometry: 61 granules per bucket against 200 bins of r)
    m.setup(sync_interval=8)
    assert ("ordered" in m.group.LaunchInfo()) == order
    rb0 = m.pdfs[0].GetReadBins()
    assert np.array_equal(rb0, oracle.set_eval_points(geom, w.events, 0))
    assert (rb0 == -1).sum() > 1000 and (rb0 == -2).sum() > 1000
    assert np.unique(rb0[rb0 >= 0]).size < (rb0 >= 0).sum()          # duplicate bins are present
    proposal = m.proposed_vector.get()
    assert np.all(proposal[w.nsources:] != 0)                        # the three systematics really move the samples
    m.step(debug_mode=True)
    rows, nacc = m.flush()
    want, bins, norms, lut = oracle_at(w, geom, proposal, keep_bins=not sparse)
    assert np.array_equal(m.normalizations.get(), norms)
    assert np.array_equal(m.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
    got = m.proposed_nll.get()[0]
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)
    assert nacc == 1 and abs(rows[0, -1] - np.float32(want)) <= 1e-6 * abs(want)
    if sparse:
        with pytest.raises(capi.SxmcError):
            m.pdfs[0].GetBins()                                      # only the event bins were counted
        m.group.EvalAsync(False, m.stream)                           # what CreateHistogram does: dense fill
        m.group.EvalFinished()
        _, bins, norms2, _ = oracle_at(w, geom, m.proposed_vector.get(), keep_bins=True)
        assert np.array_equal(m.normalizations.get(), norms2)
    for j, p in enumerate(m.pdfs):
        assert np.array_equal(p.GetBins(), bins[j]), "signal %d: dense histogram differs" % j
        bins[j] = None
    close_chain(m)


@pytest.mark.parametrize("sparse", [True, False])
def test_c5_real_geometry_walk_forms_agree(c5, sparse):
    """The forms the bench and the drivers use (event classes instead of the lookup table, step end that
    clears for the next step, graph replay) walk the same chain as the plain fused form at this geometry,
    and its NLL column follows the oracle."""
    w, geom = c5
    plain = MCMC(w, seed=5, fused=True)
    plain.group.SetSparse(sparse)
    plain.setup(sync_interval=16)
    first = plain.proposed_vector.get()
    want_chain, want_acc = plain.run(12, debug_mode=True)           # debug mode: every proposal is accepted
    want_nll, _, _, _ = oracle_at(w, geom, first, keep_bins=False)
    assert abs(want_chain[0, -1] - np.float32(want_nll)) <= 1e-6 * abs(want_nll)
    close_chain(plain)
    m = MCMC(w, seed=5, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
    m.group.SetSparse(sparse)
    m.setup(sync_interval=16)
    m.step(debug_mode=True)
    m.steps(11, graph_steps=4, debug_mode=True)
    chain, acc = m.flush()
    assert acc == want_acc == 12
    assert np.array_equal(chain[:, :-1], want_chain[:, :-1])
    assert np.allclose(chain[:, -1], want_chain[:, -1], rtol=1e-6, atol=0)
    close_chain(m)


def test_strides_beyond_24_bits_take_the_general_path():
    """fill_sparse_kernel forms idx * stride + bin with ONE signed 24-bit multiply-add; the host sends it only geometries
    whose bin counts and strides are below 2^23.  100 x 2900 x 2900 bins (8.4e8, a 3.4 GB dense histogram): the first
    observable's stride is 8 410 000 >= 2^23, so the plan must NOT use the runs kernel -- and the lookup values,
    normalisations and the NLL must still be the oracle's."""
    rng = np.random.default_rng(77)
    nb, lower, upper = [100, 2900, 2900], [0.0, 0.0, -1.0], [10.0, 6.0, 1.0]
    signals = []
    for j in range(2):
        n = 3_000_001          # (enough rows per bucket of c -- 2 900 of them -- for the table to be bucketed at all)
        e_true = rng.normal(4.0 + j, 1.5, n)
        tab = np.stack([e_true + rng.normal(0, 0.3, n), 6.0 * rng.uniform(size=n) ** (1 / 3), rng.uniform(-1, 1, n),
                        e_true, np.zeros(n)], axis=1).astype(np.float32)
        signals.append(workloads.Signal(tab, 5, nexpected=500.0 + 100 * j, source_id=j))
    ev = workloads._events_from_mixture(rng, signals, 3, 20000, None, None)
    ev[::19, 1] = 7.0                                    # outside the domain
    w = workloads.Workload("wide-strides", 3, lower, upper, nb, signals, workloads.C3_SYSTS, workloads.C3_SIGMAS, ev,
                           "strides beyond 2^23")
    geom = oracle.HistGeometry(lower, upper, nb)
    assert geom.total_nbins == 841_000_000
    m = MCMC(w, seed=4, fused=True)
    m.setup(sync_interval=8)
    info = m.group.LaunchInfo()
    assert "hist=global" in info and "+runs" not in info, info
    assert "table=bucketed" in info and "note:" in info and "2^23" in info, info   # (the plan says why it fell back)
    proposal = m.proposed_vector.get()
    m.step(debug_mode=True)
    rows, nacc = m.flush()
    want, _, norms, lut = oracle_at(w, geom, proposal, keep_bins=False)
    assert np.array_equal(m.normalizations.get(), norms)
    assert np.array_equal(m.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
    got = m.proposed_nll.get()[0]
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)
    close_chain(m)
    # the same table with 8 x fewer bins in the last observable: strides fit again, the runs kernel is back
    w2 = workloads.Workload("narrow-strides", 3, lower, upper, [100, 2900, 360], signals, workloads.C3_SYSTS,
                            workloads.C3_SIGMAS, ev, "strides below 2^23")
    m2 = MCMC(w2, seed=4, fused=True)
    m2.setup(sync_interval=8)
    assert "+runs" in m2.group.LaunchInfo() and "note:" not in m2.group.LaunchInfo(), m2.group.LaunchInfo()
    proposal = m2.proposed_vector.get()
    m2.step(debug_mode=True)
    m2.flush()
    geom2 = oracle.HistGeometry(lower, upper, [100, 2900, 360])
    want, _, norms, lut = oracle_at(w2, geom2, proposal, keep_bins=False)
    assert np.array_equal(m2.normalizations.get(), norms)
    assert np.array_equal(m2.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
    close_chain(m2)
