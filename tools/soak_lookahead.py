"""Soak test of the look-ahead walk (runs on the GPU box): random seeds, jump-width scales (acceptance 0.3 ... 0.85),
walk lengths, piece sizes, eager and graph-replayed passes, lane counts; the jump buffer and the accept count must be
those of the chain stepped one evaluation at a time.  Usage: python tools/soak_lookahead.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC, LookaheadWalk
bad = 0
w = workloads.config3(0.003, nevents=2500)
for seed in range(100, 124):
    rng = np.random.default_rng(seed)
    scale = float(rng.choice([0.1, 0.5, 1.0, 2.0, 5.0]))
    nsteps = int(rng.integers(50, 400))
    gp = int(rng.choice([0, 3, 7]))
    res = []
    for look in (False, True):
        m = MCMC(w, seed=seed, lut_output=False, consume=True, stream=capi.new_stream())
        jw = (m.initial_jump_widths() * np.float32(scale)).astype(np.float32)
        m.setup(sync_interval=512, jump_width=jw)
        if look:
            la = LookaheadWalk(m, threads=int(rng.choice([0, 1024])))
            la.bind()
            done = 0
            while done < nsteps:
                piece = min(nsteps - done, int(rng.integers(1, 90)))
                done = la.steps(piece, graph_passes=gp, count0=done)
            rows, nacc = m.flush()
            passes = la.passes
            la.close()
        else:
            rows, nacc = m.run(nsteps)
        res.append((rows, nacc))
        for p in m.pdfs:
            p.close()
        m.group.close()
    same = res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
    bad += 0 if same else 1
    print(seed, "scale", scale, "steps", nsteps, "graph", gp, "acc %.2f" % (res[0][1] / nsteps), "passes", passes, "OK" if same else "MISMATCH", flush=True)
print("soak_lookahead: %d mismatches" % bad)
sys.exit(1 if bad else 0)
