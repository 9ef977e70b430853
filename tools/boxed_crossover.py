"""Where the boxed and the ordered form of the fill take the same time: BASELINE config 3 at full size, the resolution
parameter swept, each form forced (sxmc_group_set_box_limit), the fill timed by the dispatch's own events
(sxmc_group_profile).  Run on the GPU box: python tools/boxed_crossover.py > gpurun_out/boxed_crossover.log"""
import sys
import numpy as np
sys.path.insert(0, ".")
from sxmc_amd import capi, workloads
from sxmc_amd.mcmc import MCMC

w = workloads.config3(1.0, nevents=100000)
m = MCMC(w, seed=5, fused=True, lut_output=False, consume=True, stream=capi.new_stream())
m.setup(sync_interval=8)
capi.synchronize()
print(m.group.LaunchInfo().strip(), flush=True)
base = m.proposed_vector.get().copy()


def timed():
    m.group.Profile(True, 64)
    for _ in range(40):                       # (fills only, at the vector set above)
        m.group.EvalAsync(False, m.stream)
    m.group.EvalFinished()
    total, n = m.group.ProfileRead()
    m.group.Profile(False, 0)
    return 1e3 * total / max(n, 1)


for p in (0.0, 0.005, 0.01, 0.02, 0.03, 0.04, 0.05, 0.07, 0.1, 0.15, 0.2, 0.3, -0.05, -0.2):
    v = base.copy()
    v[w.nsources + 2] = p
    m.proposed_vector.set(v)
    form, _ = m.group.AdaptFillForm()
    m.group.SetFillForm(1)
    tb = timed()
    m.group.SetFillForm(2)
    to = timed()
    m.group.SetFillForm(form)
    print("p_res %+.3f  boxed %.2f us  ordered %.2f us  form chosen: %s" % (p, tb, to, {1: "boxed", 2: "ordered"}[form]),
          flush=True)
