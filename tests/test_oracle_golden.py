"""CPU: the oracle reproduces every known answer the reference's own tests hold for pdfz
(tests/golden/pdfz_known_answers.json <- /root/reference/test/test_pdfz*.cpp)."""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import check_case_values, eval_points_with_dataset


def run_case_on_oracle(case, nthreads=1):
    geom = oracle.HistGeometry(case["lower"], case["upper"], case["nbins"])
    pts = eval_points_with_dataset(case)
    rb = oracle.set_eval_points(geom, pts, dataset=0)
    params = np.asarray(case["params"], dtype=np.float64)
    bins, norm = oracle.bin_samples(geom, case["samples"], case["nfields"],
                                    case["systematics"], params, nthreads=nthreads)
    out = np.full(case["pdf_size"], 12345.0, dtype=np.float32)
    oracle.eval_pdf(rb, bins, norm, geom.bin_volume, out=out,
                    offset=case["pdf_offset"], stride=case["pdf_stride"])
    norm_buf = np.array(case["norm_init"] or [0, 0, 0], dtype=np.uint32)
    norm_buf[case["norm_offset"]] = norm
    return geom, rb, bins, norm_buf, out


def test_all_reference_known_answers(golden):
    assert len(golden["cases"]) == 13
    for case in golden["cases"]:
        geom, rb, bins, norm_buf, out = run_case_on_oracle(case)
        check_case_values(case, out, norm_buf)
        assert int(bins.sum()) <= case["expected_norm"]
        # slots the evaluator must not touch keep their sentinel
        touched = {case["pdf_offset"] + i * case["pdf_stride"] for i in range(rb.size)}
        for i in range(case["pdf_size"]):
            if i not in touched:
                assert out[i] == np.float32(12345.0)


def test_multithreaded_oracle_matches_serial(golden):
    for case in golden["cases"]:
        _, _, bins1, n1, _ = run_case_on_oracle(case, nthreads=1)
        _, _, bins4, n4, _ = run_case_on_oracle(case, nthreads=4)
        assert np.array_equal(bins1, bins4) and np.array_equal(n1, n4)


def test_dataset_mismatch_gives_zero():
    # pdfz.cpp:289-300: in-domain point of another dataset -> -2 -> 0.0; out of domain wins -> NaN
    geom = oracle.HistGeometry([0.0], [1.0], [2])
    pts = np.array([[0.25, 0.0], [0.25, 1.0], [1.5, 1.0]], dtype=np.float32)
    rb = oracle.set_eval_points(geom, pts, dataset=0)
    assert list(rb) == [0, -2, -1]
    bins, norm = oracle.bin_samples(geom, [0.1, 0.2, 0.7], 1, [], np.zeros(1))
    out = oracle.eval_pdf(rb, bins, norm, geom.bin_volume)
    assert out[0] == np.float32(2 / (3 * 0.5)) and out[1] == 0.0 and np.isnan(out[2])


def test_geometry_row_major():
    geom = oracle.HistGeometry([0, 0, -1], [10, 6, 1], [20, 30, 40])
    assert list(geom.bin_stride) == [1200, 40, 1] and geom.total_nbins == 24000
    assert geom.bin_volume == (10 / 20) * (6 / 30) * (2 / 40)


def test_chained_systematics_are_sequential():
    # apply_systematic works in place: the scale sees the shifted value (pdfz.cpp:382-385)
    geom = oracle.HistGeometry([0.0], [10.0], [10])
    systs = [dict(type="shift", obs=0, pars=[0]), dict(type="scale", obs=0, pars=[1])]
    bins, norm = oracle.bin_samples(geom, [1.0], 1, systs, np.array([1.0, 1.0]))
    assert norm == 1 and bins[4] == 1          # (1+1)*(1+1) = 4
    systs = systs[::-1]
    bins, norm = oracle.bin_samples(geom, [1.0], 1, systs, np.array([1.0, 1.0]))
    assert bins[3] == 1                        # 1*(1+1) + 1 = 3


def test_polynomial_parameter():
    # p = sum p_i x^i evaluated at the CURRENT x (pdfz.cpp:310-314)
    geom = oracle.HistGeometry([0.0], [100.0], [100])
    systs = [dict(type="shift", obs=0, pars=[0, 1, 2])]
    bins, norm = oracle.bin_samples(geom, [3.0], 1, systs, np.array([0.5, 2.0, 1.0]))
    assert bins[int(3 + 0.5 + 6 + 9)] == 1


def test_ctscale():
    geom = oracle.HistGeometry([-1.0], [1.0], [4])
    systs = [dict(type="ctscale", obs=0, pars=[0])]
    bins, norm = oracle.bin_samples(geom, [0.5, -0.9], 1, systs, np.array([0.5]))
    # 1 + (0.5-1)*1.5 = 0.25 -> bin 2 ; 1 + (-1.9)*1.5 = -1.85 -> out
    assert norm == 1 and bins[2] == 1


def test_param_offset_and_stride():
    geom = oracle.HistGeometry([0.0], [10.0], [10])
    systs = [dict(type="shift", obs=0, pars=[1])]
    params = np.array([9.0, 9.0, 9.0, 9.0, 2.0])     # stride 2, index 1 -> params[2]... offset applied by caller
    bins, norm = oracle.bin_samples(geom, [1.0], 1, systs, params[2:], param_stride=2)
    assert bins[3] == 1


def test_nan_and_edge_samples_are_safe():
    geom = oracle.HistGeometry([0.0], [1.0], [2])
    bins, norm = oracle.bin_samples(geom, [np.nan, 1.0, np.nextafter(np.float32(1), np.float32(0))], 1,
                                    [], np.zeros(1))
    assert norm == 1 and bins[1] == 1 and bins[0] == 0


def test_empty_samples():
    geom = oracle.HistGeometry([0.0], [1.0], [2])
    bins, norm = oracle.bin_samples(geom, np.zeros(0, np.float32), 1, [], np.zeros(1))
    assert norm == 0 and bins.sum() == 0
    out = oracle.eval_pdf(np.array([0, -1, -2], np.int32), bins, norm, geom.bin_volume)
    assert np.isnan(out[0]) and np.isnan(out[1]) and out[2] == 0.0   # 0/0 -> NaN (pdfz.cpp:431)
