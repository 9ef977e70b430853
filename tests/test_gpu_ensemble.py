"""GPU: the ensemble layer on the device path -- efficiency, fake data sets drawn from the evaluated
histograms, and one whole fake experiment (fake data -> MCMC walk with burn-in -> contour intervals)."""
import numpy as np
import pytest

from oracle import oracle
from sxmc_amd import capi, ensemble, workloads
from sxmc_amd.mcmc import MCMC

pytestmark = pytest.mark.gpu


def test_efficiency_and_fake_dataset_follow_the_histograms():
    w = workloads.config2(0.01, nevents=100)
    for s in w.signals:
        s.nexpected = 20000.0
    m = MCMC(w, seed=1)
    rng = np.random.default_rng(2)
    data, observed = ensemble.make_fake_dataset(rng, w, m.pdfs, poisson=False)
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    start = 0
    for j, s in enumerate(w.signals):
        bins, norm = oracle.bin_samples(geom, s.samples, s.nfields, [], np.zeros(1))
        eff = norm / s.n_mc
        assert observed[j] == int(np.floor(s.nexpected * eff + 0.5))      # nint(nexpected * efficiency)
        ev = data[start:start + observed[j]]
        start += observed[j]
        assert np.all(ev[:, 2] == s.dataset)
        assert np.all((ev[:, 0] >= 0) & (ev[:, 0] < 10) & (ev[:, 1] >= 0) & (ev[:, 1] < 6))
        # drawn events, re-binned, follow the PDF: Pearson chi2 over well-populated bins
        rb = oracle.set_eval_points(geom, ev, s.dataset)
        got = np.bincount(rb, minlength=geom.total_nbins).astype(np.float64)
        exp = bins.astype(np.float64) / norm * observed[j]
        sel = exp > 20
        chi2 = np.sum((got[sel] - exp[sel]) ** 2 / exp[sel])
        assert chi2 < sel.sum() + 6 * np.sqrt(2 * sel.sum())
        assert got[bins == 0].sum() == 0                                   # empty bins are never drawn
    assert start == data.shape[0]


def test_device_random_sample_is_reproducible_and_respects_cuts():
    """sxmc_hist_random_sample: events drawn on the device from the histogram of the last evaluation (it never
    leaves HBM): same seed -> same events, another seed -> other events, cuts are honoured (pdfz.cpp:853-857),
    empty bins are never drawn, and new evaluation points of the same size re-use the evaluator's buffers."""
    w = workloads.config3(0.003, nevents=100)
    m = MCMC(w, seed=1)
    ev = m.pdfs[2]
    eff, bins, n_in = ensemble.get_efficiency(ev, w.nsyst_pars, w.parameter_means()[w.nsources:], w.signals[2].n_mc)
    assert 0 < eff <= 1 and n_in == bins.sum()
    a = ev.RandomSample(5000, 77)
    b = ev.RandomSample(5000, 77)
    c = ev.RandomSample(5000, 78)
    assert a.shape == (5000, 4) and np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.all(a[:, 3] == w.signals[2].dataset)
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    rb = oracle.set_eval_points(geom, a, w.signals[2].dataset)
    assert np.all(rb >= 0) and np.all(bins[rb] > 0)
    cut = ev.RandomSample(3000, 5, lowers=[1.0, 2.0, -0.5], uppers=[4.0, 5.0, 0.5])
    assert np.all((cut[:, 0] >= 1.0) & (cut[:, 0] <= 4.0) & (cut[:, 1] >= 2.0) & (cut[:, 1] <= 5.0) &
                  (np.abs(cut[:, 2]) <= 0.5))
    # cuts that leave nothing of the histogram: the reference would redraw for ever (pdfz.cpp:838-905); here the call
    # FAILS after 1024 attempts per event -- it does not hand out points that are outside the cuts
    with pytest.raises(capi.SxmcError) as err:
        ev.RandomSample(100, 5, lowers=[20.0, 2.0, -0.5], uppers=[30.0, 5.0, 0.5])
    assert "could not be drawn inside the cuts" in str(err.value)
    again = ev.RandomSample(5000, 77)
    assert np.array_equal(again, a)                      # and the evaluator is as usable as before
    # a second and third data set on the same evaluators: descriptors are patched, results are those of a fresh chain
    for seed in (3, 4):
        data, _ = ensemble.make_fake_dataset(np.random.default_rng(seed), w, m.pdfs, poisson=True)
        m.setup(data=data, sync_interval=8)
        v = m.proposed_vector.get()
        m.step(debug_mode=True)
        rows, _ = m.flush()
        w2 = workloads.Workload.__new__(workloads.Workload)
        w2.__dict__.update(w.__dict__)
        w2.events = data
        lut = np.zeros((w.nsignals, data.shape[0]), np.float32)
        norms = np.zeros(w.nsignals, np.uint32)
        for j, s_ in enumerate(w.signals):
            rbj = oracle.set_eval_points(geom, data, s_.dataset)
            bj, nj = oracle.bin_samples(geom, s_.samples, s_.nfields, w.systematics, v[w.nsources:])
            oracle.eval_pdf(rbj, bj, nj, geom.bin_volume, out=lut[j])
            norms[j] = nj
        want, _ = oracle.full_nll(lut, v, data.shape[0], w.nsignals, w.nsources, w.parameter_means(),
                                  w.parameter_sigmas(), [s_.nexpected for s_ in w.signals],
                                  [s_.n_mc for s_ in w.signals], [s_.source_id for s_ in w.signals], norms)
        assert abs(rows[0, -1] - np.float32(want)) <= 1e-6 * abs(want)


def test_efficiency_divides_by_the_count_before_cuts():
    """signal.cpp:198: efficiency = in-domain count / n_mc, the number of simulated events BEFORE cuts -- the
    same n_mc the NLL kernels divide the norms by -- not the number of rows that survived the cuts."""
    w = workloads.config2(0.01, nevents=100)
    for j, s in enumerate(w.signals):
        s.nexpected = 20000.0
        s.n_mc_total = 2 * s.samples.shape[0] + 17 * j          # cuts removed more than half of the rows
    m = MCMC(w, seed=1)
    _, observed = ensemble.make_fake_dataset(np.random.default_rng(2), w, m.pdfs, poisson=False)
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    for j, s in enumerate(w.signals):
        _, norm = oracle.bin_samples(geom, s.samples, s.nfields, [], np.zeros(1))
        assert observed[j] == int(np.floor(s.nexpected * norm / s.n_mc_total + 0.5))
        assert observed[j] < int(np.floor(s.nexpected * norm / s.samples.shape[0] + 0.5))


def test_one_fake_experiment_end_to_end():
    w = workloads.config3(0.003, nevents=100)
    for s in w.signals:
        s.nexpected = 400.0
    m = MCMC(w, seed=3, fused=True)
    iv, chain, acc = ensemble.run_experiment(w, seed=1234, nsteps=1500, burnin_fraction=0.2, mcmc=m,
                                             sync_interval=500)
    P = w.nparameters
    assert iv.shape == (P, 4) and chain.shape == (1500 - 600, P + 1)
    assert 0 < acc < 1500
    assert np.all(np.isfinite(iv[:, :3])) and np.all(iv[:, 1] <= iv[:, 0]) and np.all(iv[:, 0] <= iv[:, 2])
    # the systematics stay near their constraints, the rates stay positive
    assert np.all(np.abs(iv[w.nsources:, 0]) < 0.3)
    assert np.all(iv[: w.nsources, 1] >= 0)
    # a second experiment on the same evaluators (tables stay resident) gives a different data set
    iv2, chain2, _ = ensemble.run_experiment(w, seed=99, nsteps=300, burnin_fraction=0.1, mcmc=m)
    assert chain2.shape[0] == 300 - 60 and not np.array_equal(iv, iv2)


def test_fake_experiment_with_the_lookahead_walk_gives_the_same_intervals():
    """run_experiment(lookahead=True): same fake data (same seed), same chain, same intervals, fewer passes."""
    w = workloads.config3(0.003, nevents=100)
    for s in w.signals:
        s.nexpected = 400.0
    out = []
    for look in (False, True):
        m = MCMC(w, seed=3, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
        iv, chain, acc = ensemble.run_experiment(w, seed=4321, nsteps=600, burnin_fraction=0.2, mcmc=m,
                                                 sync_interval=250, graph_steps=4 if look else 0, lookahead=look)
        out.append((iv, chain, acc))
        if look:
            assert 0 < m.lookahead_passes < 600
    assert out[0][2] == out[1][2] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])


def test_run_config_end_to_end(tmp_path):
    """A fit described in the reference's JSON schema, from tables on disk to chains on disk."""
    from sxmc_amd import io
    from tests.test_io_cpu import EXAMPLE
    rng = np.random.default_rng(0)
    for name, n in (("a.npz", 40000), ("b.npz", 60000)):
        mc = rng.uniform(4, 16, n).astype(np.float32)
        io.write_table(tmp_path / name, np.stack([mc + rng.normal(0, 0.5, n).astype(np.float32),
                                                  rng.uniform(0, 12, n).astype(np.float32), mc], axis=1),
                       ["energy", "radius", "mc_energy"])
    (tmp_path / "fit.json").write_text(EXAMPLE)
    iv, limits, names = io.run_config(str(tmp_path / "fit.json"), out_dir=str(tmp_path), nexperiments=2, nsteps=400)
    assert iv.shape == (2, 5, 4) and len(limits) == 2 and names[-1] == "likelihood"
    assert np.all(iv[:, :, 1] <= iv[:, :, 2]) and np.all(np.isfinite(iv[:, :, :3]))
    with np.load(tmp_path / "fit_test_1.npz") as z:
        assert set(z.files) == set(names) and z["likelihood"].shape[0] == 400 - 80
        assert np.all(np.isfinite(z["likelihood"]))


def test_concurrent_experiments_with_graph_replay_match_eager():
    """Three experiments in flight, each advancing a HIP-graph replay of 8 steps per turn: the same chains and
    intervals as the step-by-step schedule."""
    from sxmc_amd import capi
    from sxmc_amd.mcmc import MCMC
    w = workloads.config3(0.003, nevents=2000)
    seeds = [101, 102, 103]
    base = MCMC(w, seed=1, stream=capi.new_stream())
    pool = [base] + [MCMC(w, seed=1, stream=capi.new_stream(), share_with=base) for _ in range(2)]
    eager = ensemble.run_experiments_concurrently(w, seeds, 150, pool, burnin_fraction=0.1, sync_interval=64)
    graph = ensemble.run_experiments_concurrently(w, seeds, 150, pool, burnin_fraction=0.1, sync_interval=64,
                                                  graph_steps=8)
    for (ia, ca, na), (ib, cb, nb) in zip(eager, graph):
        assert na == nb and np.array_equal(ca, cb) and np.array_equal(ia, ib)
