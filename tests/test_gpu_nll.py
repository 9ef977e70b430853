"""GPU parity tests of the NLL half of the path (nll_kernels) and of the whole MCMC step,
against the CPU oracle.  Tolerance: 1e-6 relative on the summed NLL is what BASELINE.json's
north_star asks; the kernels are held to 1e-12 here (only the summation order differs)."""
import math

import numpy as np
import pytest

from oracle import oracle
from sxmc_amd import capi, nll, workloads
from sxmc_amd.capi import DeviceArray
from sxmc_amd.mcmc import MCMC
from tests.helpers import philox4x32_10

pytestmark = pytest.mark.gpu

NLL_RTOL = 1e-12          # kernels vs oracle (north_star bound: 1e-6)


def random_nll_inputs(rng, ne, ns, nsources, nsyst=2):
    lut = rng.uniform(0.0, 2.0, size=(ns, ne)).astype(np.float32)
    lut[rng.uniform(size=lut.shape) < 0.05] = np.nan          # empty-histogram lookups
    lut[rng.uniform(size=lut.shape) < 0.05] = 0.0
    P = nsources + nsyst
    pars = np.concatenate([rng.uniform(0.5, 1.5, nsources), rng.normal(0, 0.1, nsyst)])
    means = np.concatenate([np.ones(nsources), np.zeros(nsyst)])
    sigmas = np.concatenate([np.zeros(nsources), np.full(nsyst, 0.1)])
    sigmas[0] = 0.3
    nexpected = rng.uniform(10, 100, ns)
    n_mc = rng.integers(1000, 100000, ns).astype(np.uint32)
    norms = (n_mc * rng.uniform(0.3, 1.0, ns)).astype(np.uint32)
    source_id = (np.arange(ns) % nsources).astype(np.int16)
    return dict(lut=lut, pars=pars, means=means, sigmas=sigmas, nexpected=nexpected, n_mc=n_mc, norms=norms,
                source_id=source_id, P=P)


def gpu_nll(x, ne, ns, nsources, grid=64, block=256, reduce_threads=128):
    d = {k: DeviceArray(v) for k, v in x.items() if isinstance(v, np.ndarray)}
    sums = DeviceArray(np.full(grid * block, 1e300))         # stale garbage must be overwritten
    total = DeviceArray.zeros(1, np.float64)
    out = DeviceArray.zeros(1, np.float64)
    nll.nll_event_chunks(grid, block, None, d["lut"], d["pars"], ne, ns, d["nexpected"], d["n_mc"],
                         d["source_id"], d["norms"], sums)
    nll.nll_event_reduce(1, reduce_threads, None, grid * block, sums, total)
    nll.nll_total(1, 1, None, x["P"], d["pars"], ns, nsources, d["means"], d["sigmas"], total, d["nexpected"],
                  d["n_mc"], d["source_id"], d["norms"], out)
    capi.synchronize()
    return out.get()[0], total.get()[0], sums.get()


@pytest.mark.parametrize("ne,ns,nsources", [(1, 1, 1), (7, 2, 2), (1000, 12, 12), (100000, 12, 5), (4097, 29, 29)])
def test_nll_chain_matches_oracle(ne, ns, nsources):
    rng = np.random.default_rng(ne + ns)
    x = random_nll_inputs(rng, ne, ns, nsources)
    got, ev, sums = gpu_nll(x, ne, ns, nsources)
    want, want_ev = oracle.full_nll(x["lut"], x["pars"], ne, ns, nsources, x["means"], x["sigmas"],
                                    x["nexpected"], x["n_mc"], x["source_id"], x["norms"])
    assert abs(ev - want_ev) <= NLL_RTOL * abs(want_ev) + 1e-300
    assert abs(got - want) <= NLL_RTOL * abs(want)
    assert np.all(np.abs(sums) < 1e299)


@pytest.mark.parametrize("grid,block,red", [(1, 64, 64), (3, 192, 256), (64, 256, 128), (16, 1024, 1024),
                                            (5, 96, 96), (2, 33, 1)])   # partly filled waves in the reduction
def test_nll_launch_shapes(grid, block, red):
    rng = np.random.default_rng(3)
    x = random_nll_inputs(rng, 5003, 6, 6)
    got, _, _ = gpu_nll(x, 5003, 6, 6, grid, block, red)
    want, _ = oracle.full_nll(x["lut"], x["pars"], 5003, 6, 6, x["means"], x["sigmas"], x["nexpected"],
                              x["n_mc"], x["source_id"], x["norms"])
    assert abs(got - want) <= NLL_RTOL * abs(want)


def test_nll_penalties():
    rng = np.random.default_rng(4)
    x = random_nll_inputs(rng, 100, 3, 3)
    x["pars"][1] = -0.01                                      # negative rate -> 1e18
    got, _, _ = gpu_nll(x, 100, 3, 3)
    assert got == 1e18
    x = random_nll_inputs(rng, 100, 3, 3)
    x["pars"][3] = -0.5                                       # negative systematic is allowed
    got, _, _ = gpu_nll(x, 100, 3, 3)
    want, _ = oracle.full_nll(x["lut"], x["pars"], 100, 3, 3, x["means"], x["sigmas"], x["nexpected"],
                              x["n_mc"], x["source_id"], x["norms"])
    assert got < 1e17 and abs(got - want) <= NLL_RTOL * abs(want)


def test_philox_known_answers_and_stream_layout():
    # Random123 known-answer vectors for philox4x32-10, and the (offset, subsequence, seed) layout
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    hooks = capi.measure_lib()      # (the raw words: a test hook of the measurement build, same nll_device.h)
    for ctr, key, want in kat:
        assert philox4x32_10(ctr, key) == want
        st = np.array([key[0] | (key[1] << 32), ctr[2] | (ctr[3] << 32), ctr[0] | (ctr[1] << 32), 0],
                      dtype=np.uint64)
        d_st, d_out = DeviceArray(st), DeviceArray.zeros(8, np.uint32)
        assert hooks.sxmc_debug_philox_dump(capi.ptr(d_st), capi.ptr(d_out), 2) == capi.OK, hooks.sxmc_last_error()
        got = d_out.get()
        assert tuple(int(v) for v in got[:4]) == want
        nxt = (ctr[0] | (ctr[1] << 32)) + 1 & 0xFFFFFFFFFFFFFFFF
        assert tuple(int(v) for v in got[4:]) == philox4x32_10(
            (nxt & 0xFFFFFFFF, nxt >> 32, ctr[2], ctr[3]), key)
        assert int(d_st.get()[2]) == (ctr[0] | (ctr[1] << 32)) + 2 & 0xFFFFFFFFFFFFFFFF
        # ... and through the PRODUCT library's own entry point: pick_new_vector (nll_kernels.cpp:30-53) of one
        # parameter at 0 with width 1 proposes the unit normal of exactly those words (Box-Muller in double)
        d_st = DeviceArray(st)
        cur, prop = DeviceArray(np.zeros(1)), DeviceArray(np.zeros(1))
        width = DeviceArray(np.ones(1, np.float32))
        capi.call("sxmc_launch_pick_new_vector", 1, 64, None, 1, capi.ptr(d_st), capi.ptr(width), capi.ptr(cur),
                  capi.ptr(prop))
        capi.synchronize()
        u1, u2 = (want[0] + 1.0) * 2.0 ** -32, want[1] * 2.0 ** -32
        normal = math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)
        assert abs(prop.get()[0] - normal) <= 1e-12 * max(1.0, abs(normal))
        assert int(d_st.get()[2]) == (ctr[0] | (ctr[1] << 32)) + 1 & 0xFFFFFFFFFFFFFFFF


def test_init_rngs_and_proposal_statistics():
    P = 300
    rng = nll.make_rngs(P, seed=42)
    st = rng.get().reshape(P, 4)
    assert np.all(st[:, 0] == 42) and np.array_equal(st[:, 1], np.arange(P, dtype=np.uint64))
    assert np.all(st[:, 2] == 0)
    cur = DeviceArray(np.linspace(-1, 1, P))
    jw = np.full(P, 0.5, np.float32)
    jw[::7] = -1.0                                            # fixed parameters
    d_jw = DeviceArray(jw)
    prop = DeviceArray.zeros(P, np.float64)
    draws = []
    for _ in range(200):
        nll.pick_new_vector(1, 64, None, P, rng, d_jw, cur, prop)
        draws.append(prop.get())
    draws = np.array(draws)
    z = (draws - np.linspace(-1, 1, P)) / 0.5
    assert np.all(z[:, ::7] == 0.0)
    zf = np.delete(z, np.arange(0, P, 7), axis=1).ravel()
    assert abs(zf.mean()) < 0.02 and abs(zf.std() - 1.0) < 0.02
    assert abs(np.mean(zf ** 3)) < 0.06 and abs(np.mean(zf ** 4) - 3.0) < 0.15
    assert np.all(rng.get().reshape(P, 4)[1:7, 2] == 200)


def test_finish_combo_debug_mode_matches_oracle():
    """debug_mode accepts every step (nll_kernels.cpp:71): the chain is then deterministic given
    the proposals, so the whole fused kernel can be checked against the oracle step by step."""
    rng = np.random.default_rng(5)
    ne, ns, nsources = 2000, 4, 4
    x = random_nll_inputs(rng, ne, ns, nsources)
    P = x["P"]
    d = {k: DeviceArray(v) for k, v in x.items() if isinstance(v, np.ndarray)}
    grid, block = 8, 256
    sums = DeviceArray.zeros(grid * block, np.float64)
    rngs = nll.make_rngs(P, 7)
    v_cur, v_prop = DeviceArray(x["pars"].copy()), DeviceArray(x["pars"] * 1.01)
    nll_cur, nll_prop = DeviceArray(np.array([1e9])), DeviceArray.zeros(1, np.float64)
    acc, cnt = DeviceArray.zeros(1, np.int32), DeviceArray.zeros(1, np.int32)
    nsteps = 5
    jb = DeviceArray.zeros(nsteps * (P + 1), np.float32)
    jw = DeviceArray(np.full(P, 0.01, np.float32))
    proposals = []
    for _ in range(nsteps):
        proposals.append(v_prop.get())
        nll.nll_event_chunks(grid, block, None, d["lut"], v_prop, ne, ns, d["nexpected"], d["n_mc"],
                             d["source_id"], d["norms"], sums)
        nll.finish_nll_jump_pick_combo(1, 128, None, grid * block, sums, ns, nsources, d["means"], d["sigmas"],
                                       rngs, nll_cur, nll_prop, v_cur, v_prop, acc, cnt, jb, P, jw,
                                       d["nexpected"], d["n_mc"], d["source_id"], d["norms"], True)
    capi.synchronize()
    assert acc.get()[0] == nsteps and cnt.get()[0] == nsteps
    rows = jb.get().reshape(nsteps, P + 1)
    for k, v in enumerate(proposals):
        want, _ = oracle.full_nll(x["lut"], v, ne, ns, nsources, x["means"], x["sigmas"], x["nexpected"],
                                  x["n_mc"], x["source_id"], x["norms"])
        assert np.array_equal(rows[k, :P], v.astype(np.float32))
        assert rows[k, P] == np.float32(want) or abs(rows[k, P] - want) <= 2e-7 * abs(want)
    # the next proposal is centred on the accepted vector
    assert np.all(np.abs(v_prop.get() - v_cur.get()) < 0.01 * 8)
    assert np.array_equal(v_cur.get(), proposals[-1])


def test_metropolis_acceptance_rule():
    """Without debug mode: downhill always accepted; uphill by 50 never (exp(-50) ~ 2e-22)."""
    P, ns, nsources, ne = 2, 1, 1, 10
    lut = DeviceArray(np.ones((1, ne), np.float32))
    means, sigmas = DeviceArray(np.array([1.0, 0.0])), DeviceArray(np.array([0.0, 0.0]))
    nexp, n_mc = DeviceArray(np.array([10.0])), DeviceArray(np.array([100], np.uint32))
    sid, norms = DeviceArray(np.array([0], np.int16)), DeviceArray(np.array([100], np.uint32))
    rngs = nll.make_rngs(P, 1)
    sums = DeviceArray.zeros(64, np.float64)
    jw = DeviceArray(np.array([-1.0, -1.0], np.float32))
    jb = DeviceArray.zeros(4 * (P + 1), np.float32)
    acc, cnt = DeviceArray.zeros(1, np.int32), DeviceArray.zeros(1, np.int32)
    v_cur, v_prop = DeviceArray(np.array([1.0, 0.0])), DeviceArray(np.array([1.2, 0.0]))
    nll.nll_event_chunks(1, 64, None, lut, v_prop, ne, ns, nexp, n_mc, sid, norms, sums)
    capi.synchronize()
    ev = sums.get().sum()
    nll_at_prop = -ev + 1.2 * 10.0
    for nll_cur0, accept in [(nll_at_prop + 1.0, True), (nll_at_prop - 50.0, False)]:
        v_cur.set(np.array([1.0, 0.0])); acc.set(np.zeros(1, np.int32)); cnt.set(np.zeros(1, np.int32))
        nll_cur, nll_prop = DeviceArray(np.array([nll_cur0])), DeviceArray.zeros(1, np.float64)
        nll.finish_nll_jump_pick_combo(1, 128, None, 64, sums, ns, nsources, means, sigmas, rngs, nll_cur,
                                       nll_prop, v_cur, v_prop, acc, cnt, jb, P, jw, nexp, n_mc, sid, norms, False)
        capi.synchronize()
        assert abs(nll_prop.get()[0] - nll_at_prop) < 1e-12
        assert (acc.get()[0] == 1) == accept
        assert v_cur.get()[0] == (1.2 if accept else 1.0)
        assert nll_cur.get()[0] == (nll_prop.get()[0] if accept else nll_cur0)


# ---------------------------------------------------------------- the whole path on BASELINE shapes
def oracle_nll_of_workload(w, vector):
    """MCMC::nll on the CPU: evaluate every signal's PDF at `vector`, then the three NLL stages."""
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    ne = w.events.shape[0]
    lut = np.zeros((w.nsignals, ne), np.float32)
    norms = np.zeros(w.nsignals, np.uint32)
    all_bins = []
    for j, s in enumerate(w.signals):
        rb = oracle.set_eval_points(geom, w.events, s.dataset)
        bins, norm = oracle.bin_samples(geom, s.samples, s.nfields, w.systematics, vector[w.nsources:])
        oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=lut[j])
        norms[j] = norm
        all_bins.append(bins)
    val, ev = oracle.full_nll(lut, vector, ne, w.nsignals, w.nsources, w.parameter_means(), w.parameter_sigmas(),
                              [s.nexpected for s in w.signals], [s.n_mc for s in w.signals],
                              [s.source_id for s in w.signals], norms)
    return val, all_bins, norms, lut


@pytest.mark.parametrize("make,scale,nevents", [(workloads.config1, 1.0, None), (workloads.config2, 0.02, 5000),
                                                (workloads.config3, 0.004, 5000), (workloads.bench_pdfz, 0.03, 4000)])
@pytest.mark.parametrize("fused", [False, True, "step"])
def test_mcmc_step_matches_oracle_on_baseline_shapes(make, scale, nevents, fused):
    w = make(scale) if nevents is None else make(scale, nevents=nevents)
    m = MCMC(w, seed=99, fused=fused)
    m.setup(sync_interval=16)
    capi.synchronize()
    # initial NLL at the means (mcmc.cpp:244-250)
    want0, bins0, norms0, lut0 = oracle_nll_of_workload(w, w.parameter_means())
    assert abs(m.current_nll.get()[0] - want0) <= 1e-9 * abs(want0)
    assert np.array_equal(m.normalizations.get(), norms0)
    assert np.array_equal(m.lut.get().view(np.uint32), lut0.ravel().view(np.uint32))
    # three accepted steps: every row of the chain is (proposal, NLL(proposal))
    proposals = []
    for _ in range(3):
        proposals.append(m.proposed_vector.get())
        m.step(debug_mode=True)
    rows, nacc = m.flush()
    assert nacc == 3 and rows.shape == (3, w.nparameters + 1)
    for k, v in enumerate(proposals):
        want, bins, norms, lut = oracle_nll_of_workload(w, v)
        assert np.array_equal(rows[k, :-1], v.astype(np.float32))
        assert abs(rows[k, -1] - want) <= 1e-6 * abs(want)              # north_star bound (float32 store)
    # state after the last step: histograms, norms and lut are those of the last proposal, bit for bit
    for j, p in enumerate(m.pdfs):
        assert np.array_equal(p.GetBins(), bins[j])
    assert np.array_equal(m.normalizations.get(), norms)
    assert np.array_equal(m.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
    assert abs(m.proposed_nll.get()[0] - want) <= NLL_RTOL * abs(want)


def test_free_chain_runs_and_accepts_some():
    w = workloads.config3(0.002, nevents=2000)
    m = MCMC(w, seed=5, fused="step")
    m.setup(sync_interval=64)
    chain, acc = m.run(128)
    assert chain.shape == (128, w.nparameters + 1)
    assert 0 < acc < 128
    assert np.all(np.isfinite(chain))


@pytest.mark.parametrize("make,scale,plan", [(workloads.config2, 1.0, "table=prebinned"),
                                             (workloads.config3, 1.0, "table=boxed+codes"),
                                             (workloads.config3, 1.0, "table=ordered+codes")])
def test_baseline_configs_at_size_against_the_oracle(make, scale, plan):
    """BASELINE config 2 (10^7 samples) and config 3 (10^8 samples, 12 signals, shift + scale + resolution_scale), both
    at their FULL size and through the kernels the bench measures (config 3: the bucketed table with the energy
    observable boxed and the radius streamed as codes, fill_boxed_kernel -- the headline kernel at the headline size): one whole MCMC step in the
    walk's default form (event classes, graph-recorded launches) against the oracle -- every bin, the norms, the
    lookup table bits and the NLL."""
    w = make(scale, nevents=100000)
    m = MCMC(w, seed=31, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
    if plan == "table=ordered+codes":            # (rounds 4-5's headline kernel, fill_ordered_kernel over two-field codes)
        m.group.SetBoxes(False)
    m.setup(sync_interval=8)
    capi.synchronize()                           # (the first proposal is drawn on the chain's non-blocking stream)
    assert plan in m.group.LaunchInfo(), m.group.LaunchInfo()
    proposal = m.proposed_vector.get()
    m.step(debug_mode=True)
    if plan == "table=boxed+codes":              # (the walk asked at its first step: a small resolution parameter)
        assert m.group.FillForm() == 1 and "boxed+codes(now)" in m.group.LaunchInfo()
    m.steps(4, graph_steps=2, debug_mode=True)
    rows, nacc = m.flush()
    assert nacc == 5
    ncores = 16
    geom = oracle.HistGeometry(w.lower, w.upper, w.nbins)
    ne = w.events.shape[0]
    lut = np.zeros((w.nsignals, ne), np.float32)
    norms = np.zeros(w.nsignals, np.uint32)
    m.proposed_vector.set(proposal)
    m.group.EvalAsync(True, m.stream)            # the same vector again, histograms and lookup table readable
    m.group.EvalFinished()
    for j, s in enumerate(w.signals):
        rb = oracle.set_eval_points(geom, w.events, s.dataset)
        bins, norm = oracle.bin_samples(geom, s.samples, s.nfields, w.systematics, proposal[w.nsources:], nthreads=ncores)
        oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=lut[j])
        norms[j] = norm
        assert np.array_equal(m.pdfs[j].GetBins(), bins), "signal %d" % j
    assert np.array_equal(m.normalizations.get(), norms)
    assert np.array_equal(m.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
    want, _ = oracle.full_nll(lut, proposal, ne, w.nsignals, w.nsources, w.parameter_means(), w.parameter_sigmas(),
                              [s.nexpected for s in w.signals], [s.n_mc for s in w.signals],
                              [s.source_id for s in w.signals], norms)
    assert abs(rows[0, -1] - np.float32(want)) <= 1e-6 * abs(want)     # the walk's own step, float32 in the chain
    m.nll(m.proposed_vector, m.proposed_nll)
    capi.synchronize()
    assert abs(m.proposed_nll.get()[0] - want) <= NLL_RTOL * abs(want)


@pytest.mark.parametrize("nchains,graph_steps,joint_ends",
                         [(2, 0, True), (2, 6, True), (3, 4, True), (4, 0, True), (3, 4, False), (2, 0, False)])
def test_lockstep_chains_walk_what_they_walk_alone(nchains, graph_steps, joint_ends):
    """sxmc_multigroup_step_async: chains over the same sample tables (different seeds, different data) advanced
    together -- ONE fill pass per step bins every sample under each chain's parameters -- must walk, bit for
    bit, the chains they walk when stepped alone (same kernels for everything but the fill; counts are integers).
    joint_ends: the chains' step ends share two launches (chain = blockIdx.y; the chains have different numbers of
    event classes, so different numbers of workgroups in their event sums) or are launched chain by chain."""
    from sxmc_amd.mcmc import LockstepChains
    w = workloads.config3(0.004, nevents=3000)
    rng = np.random.default_rng(5)
    datas = [w.events[rng.permutation(w.events.shape[0])[: 2000 + 100 * c]] for c in range(nchains)]
    nsteps = 41
    base = MCMC(w, seed=100, lut_output=False, consume=True, stream=capi.new_stream())
    alone = []
    for c in range(nchains):
        m = MCMC(w, seed=200 + c, lut_output=False, consume=True, stream=capi.new_stream(), share_with=base)
        m.setup(data=datas[c], sync_interval=64)
        alone.append(m.run(nsteps))
    stream = capi.new_stream()
    chains = [MCMC(w, seed=200 + c, lut_output=False, consume=True, stream=stream, share_with=base)
              for c in range(nchains)]
    for c, m in enumerate(chains):
        m.setup(data=datas[c], sync_interval=64)
    ls = LockstepChains(chains)
    ls.mg.SetJointStepEnd(joint_ends)
    ls.step()                                     # (the first step is launched; recording needs the plans in place)
    ls.steps(nsteps - 1, graph_steps)
    for c, m in enumerate(chains):
        rows, nacc = m.flush()
        assert nacc == alone[c][1] and 0 < nacc < nsteps
        assert np.array_equal(rows, alone[c][0]), "chain %d" % c
        # its share: the fill pass + the set's two step-end launches (lookup / event sum, step end + clearing); chain by
        # chain every chain's step end is ONE cooperative launch (step_end_kernel)
        assert m.group.LastStepLaunches() == (3 if joint_ends else 2)
    ls.close()


@pytest.mark.parametrize("graph_passes,threads", [(0, 0), (6, 0), (4, 1024)])
def test_lookahead_walk_is_the_sequential_chain(graph_passes, threads):
    """sxmc_multigroup_lookahead_step_async: one pass over the tables evaluates the step's proposal AND the vector the
    next step would propose after a rejection; the step end decides one or two steps.  Every row of the jump buffer,
    the accept count and the chain's continuation must be those of the walk stepped one evaluation at a time --
    in pieces of odd lengths (the stop is honoured exactly), eager and replayed from graphs."""
    from sxmc_amd.mcmc import LookaheadWalk
    w = workloads.config3(0.004, nevents=3000)
    nsteps = 150
    plain = MCMC(w, seed=77, lut_output=False, consume=True, stream=capi.new_stream())
    plain.setup(sync_interval=256)
    want_rows, want_acc = plain.run(nsteps)
    assert 0 < want_acc < nsteps
    m = MCMC(w, seed=77, lut_output=False, consume=True, stream=capi.new_stream())
    m.setup(sync_interval=256)
    la = LookaheadWalk(m, threads=threads)
    la.bind()
    done = 0
    for piece in (17, 1, 2, 40, 23, 67):
        done = la.steps(piece, graph_passes=graph_passes, count0=done)
        assert done == sum((17, 1, 2, 40, 23, 67)[: (17, 1, 2, 40, 23, 67).index(piece) + 1])
    rows, nacc = m.flush()
    assert rows.shape[0] == nsteps and nacc == want_acc
    assert np.array_equal(rows, want_rows)
    assert la.passes < nsteps                    # rejections were decided two steps per pass
    # the chain goes on identically stepped the ordinary way (generator states, current vector, proposal)
    more_plain, _ = plain.run(10)
    more, _ = m.run(10)
    assert np.array_equal(more, more_plain)
    # debug mode accepts everything: every pass is one step, still the same chain
    p2 = MCMC(w, seed=5, lut_output=False, consume=True, stream=capi.new_stream())
    p2.setup(sync_interval=64)
    want2, _ = p2.run(12, debug_mode=True)
    m2 = MCMC(w, seed=5, lut_output=False, consume=True, stream=capi.new_stream())
    m2.setup(sync_interval=64)
    la2 = LookaheadWalk(m2, threads=threads)
    la2.bind()
    la2.steps(12, debug_mode=True, count0=0)
    got2, _ = m2.flush()
    assert np.array_equal(got2, want2) and la2.passes == 12
    la.close()
    la2.close()


def test_lookahead_event_sum_is_partitioned_like_the_sequential_step():
    """the look-ahead pass must cut its event sum exactly like the sequential step, or the two round the NLL
    differently and the chains part at an accept boundary.  (a) MORE than 65 536 event classes -- beyond the cap the
    look-ahead sum used to have (512 blocks of 128 rows per candidate against the sequential step's 1 024): four data
    sets, so that the members' event-bin tables differ and the distinct tuples of event bins outnumber the bins of
    one histogram (two of which must fit LDS for the look-ahead pass).  (b) a problem so small that the sequential step ends in the one-workgroup form: the look-ahead pass is
    not offered, says so, and a walk asked to look ahead falls back to the sequential chain."""
    from sxmc_amd.mcmc import LookaheadWalk
    rng = np.random.default_rng(12)
    nb, lower, upper = [26, 26, 26], [0.0, 0.0, -1.0], [10.0, 6.0, 1.0]          # 17 576 bins: two histograms fit LDS
    signals = []
    for j in range(4):
        n = 200000
        e_true = rng.uniform(0, 10, n).astype(np.float32)
        tab = np.stack([e_true + rng.normal(0, 0.2, n).astype(np.float32), rng.uniform(0, 6, n).astype(np.float32),
                        rng.uniform(-1, 1, n).astype(np.float32), e_true, np.full(n, j, np.float32)], axis=1)
        signals.append(workloads.Signal(np.ascontiguousarray(tab, np.float32), 5, 4000.0 + 500 * j, j, dataset=j))
    nev = 260000
    events = np.zeros((nev, 4), np.float32)
    events[:, 0] = rng.uniform(0, 10, nev)
    events[:, 1] = rng.uniform(0, 6, nev)
    events[:, 2] = rng.uniform(-1, 1, nev)
    events[:, 3] = rng.integers(0, 4, nev)
    w = workloads.Workload("four-datasets", 3, lower, upper, nb, signals, workloads.C3_SYSTS, workloads.C3_SIGMAS, events,
                           "many event classes")
    nsteps = 40
    plain = MCMC(w, seed=3, lut_output=False, consume=True, stream=capi.new_stream())
    plain.setup(sync_interval=64)
    want_rows, want_acc = plain.run(nsteps)
    m = MCMC(w, seed=3, lut_output=False, consume=True, stream=capi.new_stream())
    m.setup(sync_interval=64)
    assert m.group.LookaheadSupported()
    la = LookaheadWalk(m, threads=0)
    la.bind()
    la.steps(nsteps, graph_passes=4, count0=0)
    rows, nacc = m.flush()
    # the classes really outnumber the old cap (K distinct tuples = the occupied bins of every data set)
    geom = oracle.HistGeometry(lower, upper, nb)
    K = sum(np.unique(oracle.set_eval_points(geom, events, j)[events[:, 3] == j]).size for j in range(4))
    assert K > 65536, K
    assert 0 < want_acc < nsteps and nacc == want_acc and np.array_equal(rows, want_rows)
    la.close()
    # (b) config 1: 10 bins x 2 signals -> at most 20 look-ups per step
    w1 = workloads.config1()
    seq = MCMC(w1, seed=9, lut_output=False, consume=True, stream=capi.new_stream())
    chain_seq, acc_seq = seq.walk(w1.events, 300, 0.1, sync_interval=100, graph_steps=4)
    ahead = MCMC(w1, seed=9, lut_output=False, consume=True, stream=capi.new_stream())
    chain_la, acc_la = ahead.walk(w1.events, 300, 0.1, sync_interval=100, graph_steps=4, lookahead=True)
    assert not ahead.group.LookaheadSupported() and ahead.lookahead_passes == 0
    assert acc_la == acc_seq and np.array_equal(chain_la, chain_seq)
    la1 = LookaheadWalk(ahead, threads=0)
    la1.bind()
    with pytest.raises(capi.SxmcError) as err:
        la1.one_pass()
    assert "not offered for this shape" in str(err.value)
    la1.close()


@pytest.mark.parametrize("width_scale", [0.05, 12.0])
def test_lookahead_walk_at_high_and_low_acceptance(width_scale):
    """Nearly every step accepted (every pass one step) and nearly every step rejected (every pass two steps,
    incl. a negative rate -> the 1e18 penalty): the same chain as stepping one evaluation at a time."""
    from sxmc_amd.mcmc import LookaheadWalk
    w = workloads.config3(0.003, nevents=2000)
    nsteps = 120
    chains = []
    for look in (False, True):
        m = MCMC(w, seed=91, lut_output=False, consume=True, stream=capi.new_stream())
        jw = (m.initial_jump_widths() * np.float32(width_scale)).astype(np.float32)
        m.setup(sync_interval=256, jump_width=jw)
        if look:
            la = LookaheadWalk(m, threads=0)
            la.bind()
            la.steps(nsteps, graph_passes=5, count0=0)
            passes = la.passes
            rows, nacc = m.flush()
            la.close()
        else:
            rows, nacc = m.run(nsteps)
        chains.append((rows, nacc))
    assert chains[0][1] == chains[1][1] and np.array_equal(chains[0][0], chains[1][0])
    frac = chains[0][1] / nsteps
    assert (frac > 0.8) if width_scale < 1 else (frac < 0.5), frac
    assert (passes > 0.8 * nsteps) if width_scale < 1 else (passes < 0.85 * nsteps), passes


def test_whole_walk_with_lookahead_is_the_same_walk():
    """MCMC.walk(lookahead=True): burn-in re-tunings (the look-ahead vector is formed anew with the new widths),
    jump-buffer flushes (exact stops) and graph replays -- the chain and the accept count of the ordinary walk."""
    w = workloads.config3(0.004, nevents=3000)
    plain = MCMC(w, seed=41, lut_output=False, consume=True, stream=capi.new_stream())
    want = plain.walk(w.events, 260, 0.15, sync_interval=70)
    for gs in (0, 5):
        m = MCMC(w, seed=41, lut_output=False, consume=True, stream=capi.new_stream())
        got = m.walk(w.events, 260, 0.15, sync_interval=70, graph_steps=gs, lookahead=True)
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), gs
        assert 0 < m.lookahead_passes < 260


def test_lockstep_refuses_chains_that_cannot_share_a_pass():
    from sxmc_amd.mcmc import LockstepChains
    w = workloads.config3(0.002, nevents=500)
    stream = capi.new_stream()
    a = MCMC(w, seed=1, lut_output=False, consume=True, stream=stream)
    b = MCMC(w, seed=2, lut_output=False, consume=True, stream=stream)      # its own copy of the tables
    for m in (a, b):
        m.setup(sync_interval=8)
    ls = LockstepChains([a, b])
    with pytest.raises(capi.SxmcError, match="share"):
        ls.step()
    ls.close()


def test_step_forms_walk_the_same_chain():
    """Same seed => same proposals and uniforms: the three step forms must accept the same steps and
    store the same chain (NLL equal to summation-order precision, compared as stored floats)."""
    w = workloads.config3(0.003, nevents=3000)
    chains = []
    for form in (False, True, "step"):
        m = MCMC(w, seed=11, fused=form)
        m.setup(sync_interval=64)
        chain, acc = m.run(64)
        chains.append((chain, acc))
    for chain, acc in chains[1:]:
        assert acc == chains[0][1]
        assert np.array_equal(chain[:, :-1], chains[0][0][:, :-1])
        assert np.allclose(chain[:, -1], chains[0][0][:, -1], rtol=1e-6, atol=0)


def test_c5_shape_five_observables_hbm_resident_histograms():
    """BASELINE config 5's shape at test size: 5 observables + truth + dataset (F = 7), histograms far
    beyond LDS capacity (global-atomic mode), shift + scale + resolution_scale floated; one whole MCMC
    step against the oracle, bit for bit."""
    w = workloads.config5(3e-5, nevents=2000, nbins=(40, 40, 40, 4, 4))      # 1.0e6 bins per signal
    w.signals = w.signals[:4]
    for sparse in (True, False):      # event-bin counters (default) and the dense HBM-resident histogram
        m = MCMC(w, seed=21, fused=True)
        m.group.SetSparse(sparse)
        m.setup(sync_interval=8)
        proposal = m.proposed_vector.get()
        m.step(debug_mode=True)
        rows, nacc = m.flush()
        want, bins, norms, lut = oracle_nll_of_workload(w, proposal)
        if not sparse:
            for j, p in enumerate(m.pdfs):
                assert np.array_equal(p.GetBins(), bins[j])
        assert np.array_equal(m.normalizations.get(), norms)
        assert np.array_equal(m.lut.get().view(np.uint32), lut.ravel().view(np.uint32))
        assert abs(m.proposed_nll.get()[0] - want) <= NLL_RTOL * abs(want)
        assert nacc == 1


def test_concurrent_chains_share_one_sample_table():
    """BASELINE config 4's per-GPU shape: several experiments at once, one per stream, over ONE resident
    copy of the MC tables.  Each chain must produce exactly what it produces alone."""
    w = workloads.config3(0.004, nevents=3000)
    solo = []
    for seed in (5, 6, 7):
        m = MCMC(w, seed=seed, fused=True)
        m.setup(sync_interval=64)
        solo.append(m.run(48))
        del m
    base = MCMC(w, seed=5, fused=True, stream=capi.new_stream())
    chains = [base] + [MCMC(w, seed=sd, fused=True, stream=capi.new_stream(), share_with=base) for sd in (6, 7)]
    for m in chains:
        m.setup(sync_interval=64)
    capi.synchronize()
    for _ in range(48):                       # interleaved: the three walks are in flight together
        for m in chains:
            m.step()
    for m, (chain, acc) in zip(chains, solo):
        got, nacc = m.flush()
        assert nacc == acc and np.array_equal(got, chain)


@pytest.mark.parametrize("fused", [True, False])
def test_graph_replayed_steps_walk_the_same_chain(fused):
    """HIP-graph capture of the per-step sequence: a walk whose steps are replayed from a
    recorded graph gives the chain of the walk launched step by step, bit for bit -- including across
    the burn-in re-tuning points, the jump-buffer flushes and a second walk on other data."""
    w = workloads.config3(0.004, nevents=3000)
    eager = MCMC(w, seed=11, fused=fused)
    want = eager.walk(w.events, 203, 0.2, sync_interval=50)
    want2 = eager.walk(w.events[:2000], 90, 0.1, sync_interval=1000)
    m = MCMC(w, seed=11, fused=fused, stream=capi.new_stream())
    got = m.walk(w.events, 203, 0.2, sync_interval=50, graph_steps=8)
    assert m._graph is not None
    got2 = m.walk(w.events[:2000], 90, 0.1, sync_interval=1000, graph_steps=16)
    for (a, na), (b, nb) in ((want, got), (want2, got2)):
        assert na == nb and a.shape == b.shape and np.array_equal(a, b)


def test_graph_capture_refuses_a_stale_group_and_the_default_stream():
    w = workloads.config1()
    m = MCMC(w, seed=3, stream=capi.new_stream())
    m.setup(sync_interval=16)                       # parameter buffers re-pointed: the launch plan is stale
    with pytest.raises(capi.SxmcError):
        m.capture_steps(2)
    capi.synchronize()
    m.step()                                        # one eager step brings it up to date
    g = m.capture_steps(2)
    g.launch(3)
    rows, _ = m.flush()
    assert rows.shape[0] == 7
    with pytest.raises(capi.SxmcError):
        capi.call("sxmc_graph_begin_capture", capi.ptr(None))


@pytest.mark.parametrize("make,scale,nevents", [(workloads.config1, 1.0, None), (workloads.config2, 0.02, 5000),
                                                (workloads.config3, 0.004, 5000)])
@pytest.mark.parametrize("fused", [True, "step"])
def test_event_classes_give_the_same_nll_without_the_lookup_table(make, scale, nevents, fused):
    """lut_output=False: the event sum runs over distinct tuples of event bins weighted by multiplicity.
    NLL of each step against the oracle (which walks the events one by one), histograms and norms bit for
    bit; the lookup table keeps the values of setup()."""
    w = make(scale) if nevents is None else make(scale, nevents=nevents)
    w.events[:40] = w.events[0]                      # many events in one bin
    w.events[40:45, :w.nobs] = 1e9                   # outside the domain
    m = MCMC(w, seed=99, fused=fused, lut_output=False)
    m.setup(sync_interval=16)
    capi.synchronize()
    lut_setup = m.lut.get().copy()
    proposals = []
    for _ in range(3):
        proposals.append(m.proposed_vector.get())
        m.step(debug_mode=True)
    rows, nacc = m.flush()
    assert nacc == 3
    for k, v in enumerate(proposals):
        want, bins, norms, lut = oracle_nll_of_workload(w, v)
        assert abs(rows[k, -1] - want) <= 1e-6 * abs(want)
    assert abs(m.proposed_nll.get()[0] - want) <= NLL_RTOL * abs(want)
    for j, p in enumerate(m.pdfs):
        assert np.array_equal(p.GetBins(), bins[j])
    assert np.array_equal(m.normalizations.get(), norms)
    assert np.array_equal(m.lut.get().view(np.uint32), lut_setup.view(np.uint32))
    # new evaluation points: the classes are rebuilt
    m.setup(data=w.events[: max(3, w.events.shape[0] // 2)], sync_interval=16)
    v = m.proposed_vector.get()
    m.step(debug_mode=True)
    rows, _ = m.flush()
    w2 = workloads.Workload.__new__(workloads.Workload)
    w2.__dict__.update(w.__dict__)
    w2.events = w.events[: max(3, w.events.shape[0] // 2)]
    want, _, _, _ = oracle_nll_of_workload(w2, v)
    assert abs(rows[0, -1] - want) <= 1e-6 * abs(want)


def test_event_classes_with_sparse_counters_and_two_datasets():
    """Members of two data sets (their event-bin tables differ) and histograms beyond LDS capacity (sparse
    counter slots): class keys are tuples over the distinct tables."""
    w = workloads.config5(3e-5, nevents=2000, nbins=(40, 40, 40, 4, 4))
    w.signals = w.signals[:4]
    w.signals[1].dataset = 1
    w.signals[3].dataset = 1
    w.events[::3, -1] = 1.0
    for sparse in (True, False):
        m = MCMC(w, seed=21, fused=True, lut_output=False)
        m.group.SetSparse(sparse)
        m.setup(sync_interval=8)
        proposal = m.proposed_vector.get()
        m.step(debug_mode=True)
        rows, nacc = m.flush()
        want, bins, norms, lut = oracle_nll_of_workload(w, proposal)
        assert np.array_equal(m.normalizations.get(), norms)
        assert abs(m.proposed_nll.get()[0] - want) <= NLL_RTOL * abs(want)


def test_no_device_memory_is_leaked_by_chains_and_graphs():
    """Evaluators, groups (launch plans, sparse structures, event classes, pre-binned columns), chains and
    recorded graphs give their device memory back."""
    import ctypes as C
    w = workloads.config3(0.002, nevents=2000)
    w5 = workloads.config5(2e-5, nevents=500, nbins=(40, 40, 40, 4, 4))
    w5.signals = w5.signals[:3]

    def cycle():
        for wl in (w, w5):
            m = MCMC(wl, seed=3, stream=capi.new_stream(), lut_output=False)
            m.walk(wl.events, 40, 0.1, sync_interval=16, graph_steps=4)
            other = MCMC(wl, seed=4, share_with=m)
            other.setup(sync_interval=8)
            other.step()
            other.flush()
            for p in other.pdfs + m.pdfs:
                p.close()
            other.group.close()
            m.group.close()
            if m._graph is not None:
                m._graph.close()
            capi.call("sxmc_stream_destroy", capi.ptr(m.stream))
            del m, other
        import gc
        gc.collect()
        capi.synchronize()

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        capi.call("sxmc_mem_info", C.byref(f), C.byref(t))
        return f.value

    cycle()                                          # first cycle may grow pools (code objects, streams)
    free0 = free_bytes()
    for _ in range(3):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 8 << 20, (free0, free1)


@pytest.mark.parametrize("make,scale,nevents,sparse", [(workloads.config3, 0.004, 3000, True),
                                                       (workloads.config1, 1.0, None, True),
                                                       (workloads.config5, 3e-5, 1500, True),
                                                       (workloads.config5, 3e-5, 1500, False)])
def test_step_end_that_clears_for_the_next_step_walks_the_same_chain(make, scale, nevents, sparse):
    """consume=True: finish_nll_jump_pick_combo and the next step's zeroing in one launch (3 launches per
    step).  Same chain bit for bit, step by step and replayed from graphs; histograms and normalisations are
    cleared between steps and come back with an ordinary evaluation."""
    kw = {} if nevents is None else dict(nevents=nevents)
    if make is workloads.config5:
        kw["nbins"] = (40, 40, 40, 4, 4)
    w = make(scale, **kw)
    if make is workloads.config5:
        w.signals = w.signals[:3]
    plain = MCMC(w, seed=17, lut_output=False)
    plain.group.SetSparse(sparse)
    want = plain.walk(w.events, 90, 0.1, sync_interval=40)
    for graph_steps in (0, 8):
        m = MCMC(w, seed=17, lut_output=False, consume=True, stream=capi.new_stream())
        m.group.SetSparse(sparse)
        got = m.walk(w.events, 90, 0.1, sync_interval=40, graph_steps=graph_steps)
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
        # between steps nothing is left to read ...
        assert np.all(m.normalizations.get() == 0)
        with pytest.raises(capi.SxmcError):
            m.pdfs[0].GetBins()
        # ... and an ordinary evaluation at the same vector brings both back
        m.group.EvalAsync(False, m.stream)
        m.group.EvalFinished()
        vec = m.proposed_vector.get()
        _, bins, norms, _ = oracle_nll_of_workload(w, vec)
        assert np.array_equal(m.normalizations.get(), norms)
        for j, p in enumerate(m.pdfs):
            assert np.array_equal(p.GetBins(), bins[j])
        # one more consuming step after that evaluation still starts from cleared histograms
        v = m.proposed_vector.get()
        m.step(debug_mode=True)
        rows, _ = m.flush(device_wide=False)
        wantv, _, _, _ = oracle_nll_of_workload(w, v)
        assert abs(rows[-1, -1] - wantv) <= 1e-6 * abs(wantv)


def test_cleared_histograms_are_not_trusted_after_another_evaluation_of_a_member():
    """After a consuming step the group skips its next zero launch -- unless a member was evaluated on its own
    (or through another group) in between, which leaves counts behind."""
    w = workloads.config3(0.003, nevents=2000)
    m = MCMC(w, seed=23, lut_output=False, consume=True)
    m.setup(sync_interval=16)
    m.step(debug_mode=True)
    capi.synchronize()
    # a member evaluated alone, with other parameters, between two steps of the walk
    alone = DeviceArray(np.array([0.3, 0.1, -0.2]))
    keep = m.pdfs[4]
    keep.SetParameterBuffer(alone)
    keep.EvalAsync(False)
    keep.EvalFinished()
    keep.SetParameterBuffer(m.proposed_vector, m.nsources)
    v = m.proposed_vector.get()
    m.step(debug_mode=True)
    rows, _ = m.flush()
    want, _, norms, _ = oracle_nll_of_workload(w, v)
    assert abs(rows[-1, -1] - want) <= 1e-6 * abs(want)
