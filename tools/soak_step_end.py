"""Soak test of round 4's two step forms (runs on the GPU box): random shapes, event counts, seeds, walk lengths, graph
replay or not, lookup table on or off.
  * the cooperative one-launch step end (step_end_kernel) against the two-launch form: the jump buffer and the accept
    count must be identical bit for bit, the timeout counter 0;
  * the unchanged call sequence (S x EvalAsync, S x EvalFinished, nll_event_chunks, finish_nll_jump_pick_combo: the
    library batches the deferred evaluations) against the explicit group call followed by the same kernels.
Usage: python tools/soak_step_end.py [ncases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from sxmc_amd import capi, workloads          # noqa: E402
from sxmc_amd.mcmc import MCMC                # noqa: E402


def close(m):
    capi.synchronize()
    if m._graph is not None:
        m._graph.close()
    for p in m.pdfs:
        p.close()
    m.group.close()


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = 0
    for case in range(ncases):
        rng = np.random.default_rng(1000 + case)
        make = [workloads.config2, workloads.config3][int(rng.integers(0, 2))]
        scale = float(rng.choice([0.002, 0.005, 0.02]))
        nevents = int(rng.choice([300, 2500, 9000, 40000]))
        w = make(scale, nevents=nevents)
        nsteps = int(rng.integers(30, 260))
        gs = int(rng.choice([0, 4, 9]))
        lut = bool(rng.integers(0, 2))
        seed = int(rng.integers(1, 1 << 30))
        scale_w = float(rng.choice([0.3, 1.0, 3.0]))
        res = []
        for coop in (False, True):
            m = MCMC(w, seed=seed, lut_output=lut, consume=True, stream=capi.new_stream())
            m.group.SetCooperativeStepEnd(coop)
            chain = m.walk(w.events, nsteps, 0.1, sync_interval=64, graph_steps=gs)
            res.append((chain, m.group.LastStepLaunches(), m.group.StepEndTimeouts()))
            close(m)
        same = res[0][0][1] == res[1][0][1] and np.array_equal(res[0][0][0].view(np.uint32), res[1][0][0].view(np.uint32))
        ok = same and res[1][2] == 0
        # the unchanged caller against the group call, same kernels after the evaluation
        drop = []
        for form in (False, "dropin"):
            m = MCMC(w, seed=seed, fused=form)
            jw = (m.initial_jump_widths() * np.float32(scale_w)).astype(np.float32)
            m.setup(sync_interval=nsteps + 1, jump_width=jw)
            for _ in range(min(nsteps, 60)):
                m.step()
            drop.append(m.flush())
            close(m)
        same2 = drop[0][1] == drop[1][1] and np.array_equal(drop[0][0].view(np.uint32), drop[1][0].view(np.uint32))
        ok = ok and same2
        bad += 0 if ok else 1
        print("case %2d %s scale %.3f events %5d steps %3d graph %d lut %d: step end %s (launches %d vs %d, timeouts %d), "
              "drop-in %s, accepted %d / %d" % (case, w.name, scale, nevents, nsteps, gs, lut, "same" if same else "DIFFERENT",
                                                 res[0][1], res[1][1], res[1][2], "same" if same2 else "DIFFERENT",
                                                 res[1][0][1], drop[1][1]), flush=True)
    print("soak_step_end: %d cases, %d mismatches" % (ncases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
