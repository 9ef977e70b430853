"""Soak test of the lockstep sets (runs on the GPU box): random numbers of chains (2-4), seeds, data sets of different
sizes per chain (different numbers of event classes, so different numbers of workgroups in the chains' event sums),
jump-width scales, walk lengths, eager and graph-replayed steps, the set's step ends in two launches for the set or
chain by chain; every chain's jump buffer and accept count must be those of the chain stepped alone.
Usage: python tools/soak_lockstep.py [cases=24]"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC, LockstepChains
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
w = workloads.config3(0.003, nevents=3000)
base = MCMC(w, seed=1, lut_output=False, consume=True, stream=capi.new_stream())
for seed in range(300, 300 + ncases):
    rng = np.random.default_rng(seed)
    nchains = int(rng.integers(2, 5))
    scale = float(rng.choice([0.2, 1.0, 3.0]))
    nsteps = int(rng.integers(20, 160))
    gs = int(rng.choice([0, 4, 9]))
    joint = bool(rng.integers(0, 2))
    datas = [w.events[rng.permutation(w.events.shape[0])[: int(rng.integers(200, 3000))]] for _ in range(nchains)]
    alone = []
    for c in range(nchains):
        m = MCMC(w, seed=seed * 10 + c, lut_output=False, consume=True, stream=capi.new_stream(), share_with=base)
        jw = (m.initial_jump_widths() * np.float32(scale)).astype(np.float32)
        m.setup(data=datas[c], sync_interval=256, jump_width=jw)
        alone.append(m.run(nsteps))
        for p in m.pdfs:
            p.close()
        m.group.close()
    stream = capi.new_stream()
    chains = [MCMC(w, seed=seed * 10 + c, lut_output=False, consume=True, stream=stream, share_with=base)
              for c in range(nchains)]
    for c, m in enumerate(chains):
        jw = (m.initial_jump_widths() * np.float32(scale)).astype(np.float32)
        m.setup(data=datas[c], sync_interval=256, jump_width=jw)
    ls = LockstepChains(chains)
    ls.mg.SetJointStepEnd(joint)
    ls.step()
    ls.steps(nsteps - 1, gs)
    same = True
    for c, m in enumerate(chains):
        rows, nacc = m.flush()
        same = same and nacc == alone[c][1] and np.array_equal(rows, alone[c][0])
    ls.close()
    for m in chains:
        for p in m.pdfs:
            p.close()
        m.group.close()
    bad += 0 if same else 1
    print(seed, "chains", nchains, "events", [d.shape[0] for d in datas], "scale", scale, "steps", nsteps, "graph", gs,
          "joint" if joint else "per-chain", "acc %.2f" % (alone[0][1] / nsteps), "OK" if same else "MISMATCH", flush=True)
print("soak_lockstep: %d mismatches" % bad)
sys.exit(1 if bad else 0)
