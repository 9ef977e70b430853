#!/usr/bin/env python3
"""Writes tests/golden/pdfz_known_answers.json.

The cases are the known-answer tests the reference itself holds for pdfz::EvalHist
(/root/reference/test/test_pdfz.cpp, test_pdfz_2d.cpp, test_pdfz_syst.cpp with the
fixtures in test_pdfz_fixtures.h / test_pdfz_fixtures_2d.h), transcribed here as DATA
(inputs and expected outputs), each with the file:line it comes from.  Nothing is
computed: every expected number below is the literal the reference test asserts.

Two adaptations, because the reference's tests were written against an older API than the
reference's current headers:
  * evaluation points get a trailing dataset column (= 0, the evaluator's default
    dataset): SetEvalPoints requires nobs+1 columns (pdfz.cpp:246, 289-293);
  * `ShiftSystematic(obs, par)` becomes a one-coefficient systematic whose parameter
    index list is [par] (pdfz.h:152-156).
The reference asserts floats with ASSERT_FLOAT_EQ, i.e. within 4 float ulps; the JSON
records that as "float_ulps": 4.
"""
import json
import os

NAN = "nan"

SAMPLES_1D = [0.1, 0.2, 0.3, 0.4, 0.5, 1.1, -0.1]            # test_pdfz_fixtures.h:12-19
EVAL_1D = [-0.1, 0.0, 0.25, 0.5, 0.75, 1.0]                   # test_pdfz_fixtures.h:48-54
HIST_1D = dict(nfields=1, nobs=1, lower=[0.0], upper=[1.0], nbins=[2])  # :10-11, 23-29

# test_pdfz_syst.cpp:168-176: same observable column plus a truth column fixed at 0.7
SAMPLES_RES = [0.1, 0.7, 0.2, 0.7, 0.3, 0.7, 0.4, 0.7, 0.5, 0.7, 1.1, 0.7, -0.1, 0.7]
HIST_RES = dict(nfields=2, nobs=1, lower=[0.0], upper=[1.0], nbins=[2])

SAMPLES_2D = [0.4, 10.5, 0.5, 11.0, 0.75, 11.0, 0.6, 11.5,
              0.6, 11.8, 0.9, 11.5, 0.4, 12.0]               # test_pdfz_fixtures_2d.h:12-19
EVAL_2D = [0.2, 10.2, 0.7, 10.4, 0.5, 11.0, 0.25, 11.8,
           0.9, 11.9, 0.3, 12.0, 0.3, 13.0, 0.3, 5.0]        # test_pdfz_fixtures_2d.h:50-58
HIST_2D = dict(nfields=2, nobs=2, lower=[0.0, 10.0], upper=[1.0, 12.0], nbins=[2, 3])  # :10-11, 23-29
N2D = 6 * (0.5 * (2.0 / 3.0))                                 # test_pdfz_2d.cpp:56


def case(name, ref, hist, samples, eval_points, expected_norm, expected, syst=None,
         param=None, pdf_offset=0, pdf_stride=1, pdf_size=20, norm_offset=0,
         norm_init=None, norm_expected=None, note=None):
    c = dict(name=name, ref=ref, samples=samples, eval_points=eval_points,
             systematics=[] if syst is None else [syst],
             params=[0.0] * 5 if param is None else [param, 0.0, 0.0, 0.0, 0.0],
             pdf_offset=pdf_offset, pdf_stride=pdf_stride, pdf_size=pdf_size,
             norm_offset=norm_offset, norm_size=3, norm_init=norm_init,
             expected_norm=expected_norm, expected_norm_buffer=norm_expected,
             expected_values=expected, float_ulps=4)
    c.update(hist)
    if note:
        c["note"] = note
    return c


SHIFT = dict(type="shift", obs=0, pars=[0])                    # test_pdfz_syst.cpp:26
SCALE = dict(type="scale", obs=0, pars=[0])                    # test_pdfz_syst.cpp:96
RES = dict(type="resolution_scale", obs=0, true_obs=1, pars=[0])  # test_pdfz_syst.cpp:202

BASE = {"1": 1.6, "2": 1.6, "3": 0.4, "4": 0.4}


def vals(v1, v2, v3, v4):
    # indices are evaluation-point indices; 0 and 5 are outside [0,1) -> NaN
    return {"0": NAN, "1": v1, "2": v2, "3": v3, "4": v4, "5": NAN}


cases = [
    case("eval_1d", "test/test_pdfz.cpp:79-96", HIST_1D, SAMPLES_1D, EVAL_1D, 5,
         vals(1.6, 1.6, 0.4, 0.4)),
    case("eval_1d_offset_stride", "test/test_pdfz.cpp:98-126", HIST_1D, SAMPLES_1D, EVAL_1D, 5,
         vals(1.6, 1.6, 0.4, 0.4), pdf_offset=3, pdf_stride=2, norm_offset=1,
         norm_init=[77, 88, 99], norm_expected=[77, 5, 99]),
    case("shift_zero", "test/test_pdfz_syst.cpp:39-53", HIST_1D, SAMPLES_1D, EVAL_1D, 5,
         vals(1.6, 1.6, 0.4, 0.4), syst=SHIFT, param=0.0),
    case("shift_neg", "test/test_pdfz_syst.cpp:56-70", HIST_1D, SAMPLES_1D, EVAL_1D, 4,
         vals(1.5, 1.5, 0.5, 0.5), syst=SHIFT, param=-0.25),
    case("shift_pos", "test/test_pdfz_syst.cpp:73-87", HIST_1D, SAMPLES_1D, EVAL_1D, 6,
         vals(1.0, 1.0, 1.0, 1.0), syst=SHIFT, param=0.25),
    case("scale_zero", "test/test_pdfz_syst.cpp:109-123", HIST_1D, SAMPLES_1D, EVAL_1D, 5,
         vals(1.6, 1.6, 0.4, 0.4), syst=SCALE, param=0.0),
    case("scale_neg", "test/test_pdfz_syst.cpp:126-140", HIST_1D, SAMPLES_1D, EVAL_1D, 6,
         vals(5.0 / 3, 5.0 / 3, 1.0 / 3, 1.0 / 3), syst=SCALE, param=-0.1),
    case("scale_pos", "test/test_pdfz_syst.cpp:143-157", HIST_1D, SAMPLES_1D, EVAL_1D, 4,
         vals(1.0, 1.0, 1.0, 1.0), syst=SCALE, param=1.0),
    case("resolution_zero", "test/test_pdfz_syst.cpp:224-238", HIST_RES, SAMPLES_RES, EVAL_1D, 5,
         vals(1.6, 1.6, 0.4, 0.4), syst=RES, param=0.0),
    case("resolution_neg", "test/test_pdfz_syst.cpp:241-255", HIST_RES, SAMPLES_RES, EVAL_1D, 7,
         vals(2.0 * 5 / 7, 2.0 * 5 / 7, 2.0 * 2 / 7, 2.0 * 2 / 7), syst=RES, param=-0.30),
    case("resolution_pos", "test/test_pdfz_syst.cpp:258-272", HIST_RES, SAMPLES_RES, EVAL_1D, 4,
         vals(2.0, 2.0, 0.0, 0.0), syst=RES, param=0.30),
    case("eval_2d", "test/test_pdfz_2d.cpp:45-68", HIST_2D, SAMPLES_2D, EVAL_2D, 6,
         {"0": 1 / N2D, "1": 0 / N2D, "2": 2 / N2D, "3": 0 / N2D, "4": 3 / N2D,
          "5": NAN, "6": NAN, "7": NAN}, pdf_size=40,
         note="the reference test asserts NaN at output slots 6,7,8 (slot 8 is past the 8 "
              "points); point 5 = (0.3, 12.0) is NaN by the same rule and is asserted by "
              "the offset/stride variant (slot 13)"),
    case("eval_2d_offset_stride", "test/test_pdfz_2d.cpp:71-104", HIST_2D, SAMPLES_2D, EVAL_2D, 6,
         {"0": 1 / N2D, "1": 0 / N2D, "2": 2 / N2D, "3": 0 / N2D, "4": 3 / N2D,
          "5": NAN, "6": NAN, "7": NAN}, pdf_offset=3, pdf_stride=2, pdf_size=40,
         norm_offset=1, norm_init=[77, 88, 99], norm_expected=[77, 6, 99]),
]

# Constructor validation: each must raise pdfz::Error (test_pdfz.cpp:42-73, test_pdfz_2d.cpp:8-39)
ctor_errors = [
    dict(name="wrong_sample_size_1d", ref="test/test_pdfz.cpp:42-44", nsamples_floats=7,
         nfields=2, nobs=1, lower=[0.0], upper=[1.0], nbins=[2]),
    dict(name="nobs_larger_than_nfields_1d", ref="test/test_pdfz.cpp:47-49", nsamples_floats=7,
         nfields=1, nobs=7, lower=[0.0], upper=[1.0], nbins=[2]),
    dict(name="wrong_lower_size_1d", ref="test/test_pdfz.cpp:52-55", nsamples_floats=7,
         nfields=1, nobs=1, lower=[0.0, 0.0], upper=[1.0], nbins=[2]),
    dict(name="wrong_upper_size_1d", ref="test/test_pdfz.cpp:58-61", nsamples_floats=7,
         nfields=1, nobs=1, lower=[0.0], upper=[1.0, 0.0], nbins=[2]),
    dict(name="wrong_nbins_size_1d", ref="test/test_pdfz.cpp:64-67", nsamples_floats=7,
         nfields=1, nobs=1, lower=[0.0], upper=[1.0], nbins=[2, 0]),
    dict(name="zero_bins_1d", ref="test/test_pdfz.cpp:70-73", nsamples_floats=7,
         nfields=1, nobs=1, lower=[0.0], upper=[1.0], nbins=[0]),
    dict(name="wrong_sample_size_2d", ref="test/test_pdfz_2d.cpp:8-10", nsamples_floats=14,
         nfields=3, nobs=2, lower=[0.0, 10.0], upper=[1.0, 12.0], nbins=[2, 3]),
    dict(name="nobs_larger_than_nfields_2d", ref="test/test_pdfz_2d.cpp:13-15", nsamples_floats=14,
         nfields=2, nobs=7, lower=[0.0, 10.0], upper=[1.0, 12.0], nbins=[2, 3]),
    dict(name="wrong_lower_size_2d", ref="test/test_pdfz_2d.cpp:18-21", nsamples_floats=14,
         nfields=2, nobs=2, lower=[0.0], upper=[1.0, 12.0], nbins=[2, 3]),
    dict(name="wrong_upper_size_2d", ref="test/test_pdfz_2d.cpp:24-27", nsamples_floats=14,
         nfields=2, nobs=2, lower=[0.0, 10.0], upper=[1.0], nbins=[2, 3]),
    dict(name="wrong_nbins_size_2d", ref="test/test_pdfz_2d.cpp:30-33", nsamples_floats=14,
         nfields=2, nobs=2, lower=[0.0, 10.0], upper=[1.0, 12.0], nbins=[2]),
    dict(name="zero_bins_2d", ref="test/test_pdfz_2d.cpp:36-39", nsamples_floats=14,
         nfields=2, nobs=2, lower=[0.0, 10.0], upper=[1.0, 12.0], nbins=[2, 0]),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pdfz_known_answers.json")
    with open(out, "w") as f:
        json.dump(dict(source="reference gtest known answers (see make_pdfz_known_answers.py)",
                       cases=cases, ctor_errors=ctor_errors), f, indent=1)
    print("wrote", out, len(cases), "cases,", len(ctor_errors), "ctor error cases")
