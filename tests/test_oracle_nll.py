"""CPU: hand-derived checks of the oracle's NLL restatement (nll_kernels.cpp:30-188).
The reference holds no test for these functions, so the NLL part of the oracle is
"parity unpinned" (oracle/sxmc_oracle.h); these cases pin it to the formulas only."""
import math

import numpy as np

from oracle import oracle


def setup_case():
    ne, ns = 4, 2
    lut = np.array([[0.5, 1.0, np.nan, 0.0],       # signal 0
                    [2.0, 0.0, 1.0, 0.0]], dtype=np.float32)   # signal 1, lut[j*ne+i]
    pars = np.array([1.5, 0.5, 0.1])               # 2 sources + 1 systematic
    nexpected = np.array([10.0, 20.0])
    n_mc = np.array([100, 200], dtype=np.uint32)
    norms = np.array([50, 150], dtype=np.uint32)
    source_id = np.array([0, 1], dtype=np.int16)
    return ne, ns, lut, pars, nexpected, n_mc, source_id, norms


def test_event_chunks_formula():
    ne, ns, lut, pars, nexpected, n_mc, source_id, norms = setup_case()
    sums = oracle.nll_event_chunks(lut, pars, ne, ns, nexpected, n_mc, source_id, norms)
    eff = [np.float32(50 / 100), np.float32(150 / 200)]
    a = [pars[0] * nexpected[0] * float(eff[0]), pars[1] * nexpected[1] * float(eff[1])]
    s = [a[0] * 0.5 + a[1] * 2.0, a[0] * 1.0, a[1] * 1.0, 0.0]   # NaN -> 0; s=0 skipped
    expect = math.log(s[0]) + math.log(s[1]) + math.log(s[2])
    assert abs(sums[0] - expect) < 1e-14


def test_eff_is_rounded_to_float():
    # nll_kernels.cpp:105: `float eff = 1.0 * norms[j] / n_mc[j]`
    lut = np.array([[1.0]], dtype=np.float32)
    sums = oracle.nll_event_chunks(lut, np.array([1.0]), 1, 1, np.array([1.0]),
                                   np.array([3], np.uint32), np.array([0], np.int16),
                                   np.array([1], np.uint32))
    assert sums[0] == math.log(float(np.float32(1.0 / 3.0)))
    assert sums[0] != math.log(1.0 / 3.0)


def test_total_formula_and_penalties():
    ne, ns, lut, pars, nexpected, n_mc, source_id, norms = setup_case()
    means = np.array([1.0, 1.0, 0.0])
    sigmas = np.array([0.0, 0.25, 0.05])
    nll, ev = oracle.full_nll(lut, pars, ne, ns, 2, means, sigmas, nexpected, n_mc, source_id, norms)
    expect = -ev + 1.5 * 10.0 * 50 / 100 + 0.5 * 20.0 * 150 / 200 \
        + 0.5 * ((0.5 - 1.0) / 0.25) ** 2 + 0.5 * ((0.1 - 0.0) / 0.05) ** 2
    assert abs(nll - expect) < 1e-12
    # negative source rate -> 1e18 (nll_kernels.cpp:175-178); negative systematic is fine
    bad = pars.copy(); bad[1] = -0.1
    assert oracle.nll_total(bad, ns, 2, means, sigmas, [ev], nexpected, n_mc, source_id, norms) == 1e18
    ok = pars.copy(); ok[2] = -0.1
    assert oracle.nll_total(ok, ns, 2, means, sigmas, [ev], nexpected, n_mc, source_id, norms) < 1e17
    # NaN event sum -> 1e18 (nll_kernels.cpp:162-165)
    assert oracle.nll_total(pars, ns, 2, means, sigmas, [np.nan], nexpected, n_mc, source_id, norms) == 1e18


def test_reduce_is_a_plain_sum():
    sums = np.arange(1, 101, dtype=np.float64)
    assert oracle.nll_event_reduce(sums)[0] == 5050.0


def test_jump_decider_and_proposal():
    P = 3
    vcur = np.array([1.0, 2.0, 3.0]); vprop = np.array([1.5, 2.5, 3.5])
    nc = np.array([10.0]); npr = np.array([11.0])
    acc = np.zeros(1, np.int32); cnt = np.zeros(1, np.int32)
    buf = np.zeros(3 * (P + 1), np.float32)
    # u > exp(-1): rejected, current appended
    oracle.jump_decider(0.9, nc, npr, vcur, vprop, acc, cnt, buf)
    assert acc[0] == 0 and cnt[0] == 1 and list(buf[:4]) == [1.0, 2.0, 3.0, 10.0]
    # u <= exp(-1): accepted
    oracle.jump_decider(0.3, nc, npr, vcur, vprop, acc, cnt, buf)
    assert acc[0] == 1 and cnt[0] == 2 and list(buf[4:8]) == [1.5, 2.5, 3.5, 11.0] and nc[0] == 11.0
    # downhill always accepted
    npr[0] = 5.0; vprop[:] = [0.0, 0.0, 0.0]
    oracle.jump_decider(0.999, nc, npr, vcur, vprop, acc, cnt, buf)
    assert acc[0] == 2 and list(buf[8:12]) == [0.0, 0.0, 0.0, 5.0]
    # proposal: width<=0 means fixed (nll_kernels.cpp:39-51)
    out = oracle.pick_new_vector([1.0, 1.0, -2.0], [0.5, -1.0, 0.25], [1.0, 2.0, 3.0])
    assert list(out) == [1.5, 2.0, 2.5]
