"""Where do codes stop paying?  BASELINE config 3's shape at a tenth of its size with 20 ... 100 bins of e (the observable
binned from codes): the share of ambiguous samples grows with the bin count (code step in units of a bin); the fill is
timed over codes and over the float columns (runs on the GPU box).  The planner's gate (get_bucket_codes: expected
share <= 2e-3) is lifted for the measurement with SXMC_CODES_GATE=1.  Usage: python tools/codes_gate_probe.py [scale]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SXMC_CODES_GATE", "1")

from sxmc_amd import capi, nll, pdfz, workloads  # noqa: E402
from sxmc_amd.capi import DeviceArray  # noqa: E402
from sxmc_amd.mcmc import make_systematic  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    w = workloads.config3(scale, nevents=1000)
    params = np.array([0.02, -0.01, 0.03])
    for nbe in (20, 40, 60, 80, 100):
        nbins = [nbe, 20, 20]
        evs = []
        norms = DeviceArray.zeros(w.nsignals, np.uint32)
        pbuf = DeviceArray(params)
        for j, s in enumerate(w.signals):
            ev = pdfz.EvalHist(s.samples, s.nfields, w.nobs, w.lower, w.upper, nbins)
            for sy in w.systematics:
                ev.AddSystematic(make_systematic(sy))
            ev.SetNormalizationBuffer(norms, j)
            ev.SetParameterBuffer(pbuf, 0, 1)
            evs.append(ev)
        group = nll.EvalGroup(evs)
        group.SetOrdering(True, force=True)
        out = {}
        for codes in (True, False):
            group.SetCodes(codes)
            info = group.LaunchInfo()
            for _ in range(5):
                group.EvalAsync(False)
            group.EvalFinished()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                group.EvalAsync(False)
            group.EvalFinished()
            out[codes] = (1e6 * (time.perf_counter() - t0) / n, "codes" in info, norms.get().sum())
        share = nbe * (15.9 / 10.0) / 65532.0
        print("bins of e %3d  expected ambiguous share %.1e | codes %s %7.1f us | floats %7.1f us | ratio %.2f" % (
            nbe, share, "(on) " if out[True][1] else "(OFF)", out[True][0], out[False][0], out[False][0] / out[True][0]),
            flush=True)
        group.close()
        for ev in evs:
            ev.close()


if __name__ == "__main__":
    main()
