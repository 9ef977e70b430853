"""GPU: the margin of the codes' error bound, MEASURED (fill_ordered_body, sxmc_amd/csrc/fill_kernels.inc.h "THE BOUND").

Bit-exact histograms from 16-bit codes rest on an inequality: a sample is binned from its codes only when its bin
coordinate lies further from every bin edge than a bound e = Q + R on what codes and arithmetic leave unknown (Q: half a
code step per field, an identity; R: bounds on roundings, hand-derived).  Samples within ulps of an edge never test that
inequality -- they are all ambiguous and take the exact path.  The samples that do are those sitting ON the threshold,
fields at the extreme edges of their code cells: tests/codes_margin_worker.py builds them (per parameter set, for every
bin edge), and
  * the PRODUCT library must bin every one of them like the oracle (first test, in this process, no hook involved);
  * the measurement build (libsxmc_hip_measure.so, a child process) repeats the fill with R scaled by s and with the
    whole threshold scaled by t: the smallest s at which everything is still binned like the oracle is s_min -- the
    measured margin 1 / s_min of the hand-set part -- and t = 1/2 MUST misplace samples (the negative control: a bound
    half as large as needed does not get past these tests).
Reference arithmetic: /root/reference/src/pdfz.cpp:306-331, 388-398."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sxmc_amd import capi
from tests import codes_margin_worker as worker

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rows_built_on_the_threshold_are_binned_like_the_oracle():
    """The product library, unscaled, on the rows built to sit on the threshold of the codes' test."""
    assert not capi.is_measurement_build() or os.environ.get("SXMC_HIP_LIB")
    rng = np.random.default_rng(20252)
    built = 0
    for params in worker.PARAM_SETS:
        rec = worker.one_parameter_set(rng, params, 60000, scales=False)
        assert rec["unscaled"] == 0, rec
        assert rec["rows_built"] > 100 and rec["median_distance_to_edge_bins"] < 2e-3, rec
        built += rec["rows_built"]
    assert built > 8000


def test_margin_of_the_bound_and_negative_control():
    env = dict(os.environ, SXMC_HIP_LIB=capi.MEASURE_LIB_PATH)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "codes_margin_worker.py"), "100000"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "codes_margin.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["rows_built"] > 8000
    # the bound as shipped: nothing misplaced
    assert rec["misplaced_unscaled"] == 0
    for s in rec["sets"]:
        assert s["rounding_scale"]["1.0"] == 0, s
    # MEASURED MARGIN: the hand-set part of the bound (everything but the half code step) can be halved -- at least --
    # before a single one of these samples is misplaced
    assert rec["s_min"] is not None and rec["s_min"] <= 0.5, rec["s_min"]
    # ... and it is a MEASUREMENT, not a test that cannot fail: below s_min samples ARE misplaced (the roundings the
    # bound provides for are real: with no room for them at all, s = 0, a good part of the built rows goes wrong)
    below = [x for x in rec["rounding_scales"] if x < rec["s_min"]]
    assert below and rec["misplaced_at_zero"] > 0, rec
    assert sum(st["rounding_scale"][str(max(below))] for st in rec["sets"]) > 0, rec
    # NEGATIVE CONTROL on the whole threshold: the bound is Q + R with R about a tenth of Q, so a threshold at 85 % of
    # the bound is already below the half code step and misplaces samples; one half as large as the bound misplaces them
    # by the thousand -- a bound that is too small does not get past these tests
    assert rec["misplaced_at_85_percent_threshold"] > 100, rec
    assert rec["misplaced_at_half_threshold"] > 1000, rec
    for st in rec["sets"]:
        assert st["total_scale"]["0.5"] > 0 and st["total_scale"]["0.75"] > 0, st
