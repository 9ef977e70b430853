"""The step end in ONE cooperative launch (step_end_kernel): look-ups + event sum over workgroups
that publish their partial sums and count themselves in, a finisher workgroup that waits for the count inside the
kernel, the histograms cleared by the others once every look-up is done.  Same virtual blocks of the event sum, same
order of the partial sums: the chain must be the two-launch form's bit for bit -- step by step, replayed from graphs,
with the lookup table materialised or not, with several chains' step ends in flight at once."""
import numpy as np
import pytest

from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC

pytestmark = pytest.mark.gpu


def close(m):
    capi.synchronize()
    if m._graph is not None:
        m._graph.close()
    for p in m.pdfs:
        p.close()
    m.group.close()


def walk(w, coop, nsteps, graph_steps=0, lut_output=False, seed=5, burnin=0.1, fused=False, force_ordering=False):
    m = MCMC(w, seed=seed, lut_output=lut_output, consume=True, stream=capi.new_stream())
    m.group.SetCooperativeStepEnd(coop)
    m.group.SetFusedStep(fused)
    if force_ordering:
        m.group.SetOrdering(True, force=True)     # (a table this small would not be ordered by default)
    chain, acc = m.walk(w.events, nsteps, burnin, sync_interval=50, graph_steps=graph_steps)
    launches, timeouts = m.group.LastStepLaunches(), m.group.StepEndTimeouts()
    close(m)
    return chain, acc, launches, timeouts


@pytest.mark.parametrize("make,scale,nevents", [(workloads.config2, 0.02, 5000), (workloads.config3, 0.004, 3000),
                                                (workloads.config3, 0.004, 40000)])
@pytest.mark.parametrize("lut_output", [False, True])
def test_one_launch_step_end_walks_the_two_launch_chain(make, scale, nevents, lut_output):
    w = make(scale, nevents=nevents)
    want = walk(w, False, 130, lut_output=lut_output)
    assert want[2] == 3 and want[3] == 0                          # fill + event sum + step end
    for graph_steps in (0, 7):
        got = walk(w, True, 130, graph_steps, lut_output=lut_output)
        rows = nevents if lut_output else None                    # (event classes: far fewer rows than events)
        if rows is not None and rows > 128 * 128:
            assert got[2] == 3                                    # beyond 128 workers the two-launch form stays
        else:
            assert got[2] == 2, got[2]                            # fill + ONE step-end launch
        assert got[3] == 0
        assert got[1] == want[1] and 0 < got[1] < 130
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))


def test_several_chains_step_ends_in_flight_together():
    """Four chains on four streams over one copy of the tables, each ending its steps with a kernel that waits inside:
    they must all drain (no timeouts) and walk the chains they walk alone."""
    w = workloads.config3(0.004, nevents=3000)
    alone = [walk(w, True, 200, graph_steps=10, seed=31 + k) for k in range(4)]
    base = MCMC(w, seed=31, lut_output=False, consume=True, stream=capi.new_stream())
    chains = [base] + [MCMC(w, seed=31 + k, lut_output=False, consume=True, stream=capi.new_stream(), share_with=base)
                       for k in range(1, 4)]
    for m in chains:
        m.group.SetCooperativeStepEnd(True)
        m.group.SetFusedStep(False)
        m.walk_begin(w.events, 200, 0.1, sync_interval=50)
    # advance the four walks in turn, a run of steps at a time, so that their kernels are queued side by side
    schedules = [m.flush_schedule() for m in chains]
    done = [0] * 4
    for f in schedules[0]:
        for k, m in enumerate(chains):
            m._retune_if_due(done[k])
            m.steps(f - done[k] + 1, 10, False)
        for k, m in enumerate(chains):
            m._flush_if_due(f)
            done[k] = f + 1
    for k, m in enumerate(chains):
        rows, acc = m.walk_end()
        assert m.group.StepEndTimeouts() == 0
        assert acc == alone[k][1] and np.array_equal(rows.view(np.uint32), alone[k][0].view(np.uint32))
    for m in chains[::-1]:
        close(m)


def test_lookahead_walk_still_partitions_like_the_sequential_step():
    """The look-ahead pass sums each candidate in the sequential step's virtual blocks; the sequential step now ends
    cooperatively -- same blocks, so the two walks must still be one chain."""
    w = workloads.config3(0.004, nevents=20000)
    m = MCMC(w, seed=9, lut_output=False, consume=True, stream=capi.new_stream())
    m.group.SetCooperativeStepEnd(True)
    m.group.SetFusedStep(False)
    want = m.walk(w.events, 150, 0.1, sync_interval=50, graph_steps=5)
    assert m.group.LastStepLaunches() == 2
    close(m)
    m = MCMC(w, seed=9, lut_output=False, consume=True, stream=capi.new_stream())
    got = m.walk(w.events, 150, 0.1, sync_interval=50, graph_steps=5, lookahead=True)
    assert m.lookahead_passes > 0
    close(m)
    assert got[1] == want[1] and np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))


@pytest.mark.parametrize("make,scale,nevents,plan", [(workloads.config3, 0.004, 3000, "table=ordered"),
                                                     (workloads.config3, 0.02, 40000, "table=ordered"),
                                                     (workloads.config2, 0.02, 5000, "table=prebinned")])
def test_whole_step_in_one_launch_walks_the_same_chain(make, scale, nevents, plan):
    """fill_step_kernel: the fill's workgroups and the step end's finisher + workers as ONE grid (the roles wait for the
    fill's workgroups to count themselves done).  Same arithmetic and partial sums as the separate launches: the chain
    must be theirs bit for bit, eager and graph-replayed; 1 launch per step; no timeouts."""
    w = make(scale, nevents=nevents)
    ordered = "ordered" in plan
    want = walk(w, False, 150, fused=False, force_ordering=ordered)
    assert want[2] == 3
    for graph_steps in (0, 6):
        got = walk(w, True, 150, graph_steps, fused=True, force_ordering=ordered)
        assert got[2] == 1, got[2]
        assert got[3] == 0
        assert got[1] == want[1] and 0 < got[1] < 150
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32))


def test_fused_steps_of_several_chains_share_the_device():
    """Four chains on four streams, each step ONE launch whose later blocks wait for its earlier ones: the launches of
    different chains interleave on the device and must all drain; every chain walks what it walks alone."""
    w = workloads.config3(0.004, nevents=3000)
    alone = [walk(w, True, 160, graph_steps=8, seed=41 + k, fused=False, force_ordering=True) for k in range(4)]
    base = MCMC(w, seed=41, lut_output=False, consume=True, stream=capi.new_stream())
    chains = [base] + [MCMC(w, seed=41 + k, lut_output=False, consume=True, stream=capi.new_stream(), share_with=base)
                       for k in range(1, 4)]
    for m in chains:
        m.group.SetOrdering(True, force=True)
        m.group.SetFusedStep(True)
        m.walk_begin(w.events, 160, 0.1, sync_interval=40)
    schedules = chains[0].flush_schedule()
    done = [0] * 4
    for f in schedules:
        for k, m in enumerate(chains):
            m._retune_if_due(done[k])
            m.steps(f - done[k] + 1, 8, False)
        for k, m in enumerate(chains):
            m._flush_if_due(f)
            done[k] = f + 1
    for k, m in enumerate(chains):
        rows, acc = m.walk_end()
        assert "ordered" in m.group.LaunchInfo() and m.group.LastStepLaunches() == 1
        assert m.group.StepEndTimeouts() == 0
        assert acc == alone[k][1] and np.array_equal(rows.view(np.uint32), alone[k][0].view(np.uint32))
    for m in chains[::-1]:
        close(m)


def test_many_stepping_chains_fall_back_to_two_launches_and_walk_the_same_chain():
    """RESIDENCY.  The waits inside the cooperative step end need the waited-for workgroups resident; the
    library takes that form only while the step ends of ALL chains stepping in the process fit half of what the device
    holds of that kernel (129 workgroups per chain against the runtime's occupancy figure x CUs), decided at a group's
    first step.  Here so many chains step that the later ones must take the two-launch form -- and every chain, whichever
    form it got, walks the chain it walks alone."""
    w = workloads.config3(0.004, nevents=3000)
    want = walk(w, True, 40, seed=77)
    assert want[2] == 2 and want[3] == 0
    base = MCMC(w, seed=77, lut_output=False, consume=True, stream=capi.new_stream())
    chains = [base] + [MCMC(w, seed=77, lut_output=False, consume=True, stream=capi.new_stream(), share_with=base)
                       for _ in range(39)]
    forms = []
    for m in chains:
        chain, acc = m.walk(w.events, 40, 0.1, sync_interval=50)
        forms.append(m.group.LastStepLaunches())
        assert m.group.StepEndTimeouts() == 0
        assert acc == want[1] and np.array_equal(chain.view(np.uint32), want[0].view(np.uint32))
    # the first chains end their steps in one launch (2 per step), the late ones in two (3 per step)
    assert forms[0] == 2 and forms[-1] == 3, forms
    assert forms == sorted(forms), forms
    for m in chains[::-1]:
        close(m)
    # ... and once those groups are gone a new chain gets the cooperative form again
    again = walk(w, True, 40, seed=77)
    assert again[2] == 2 and np.array_equal(again[0].view(np.uint32), want[0].view(np.uint32))
