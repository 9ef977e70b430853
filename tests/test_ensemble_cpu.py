"""CPU: interval extraction and the histogram sampler of the ensemble layer (sxmc_amd/ensemble.py),
against closed forms (the reference holds no test for likelihood.cpp / contour.cpp / generator.cpp)."""
import math

import numpy as np

from sxmc_amd import ensemble


def test_chisquare_quantile():
    # TMath::ChisquareQuantile(p, 1): textbook values
    assert abs(ensemble.chisquare_quantile_1dof(0.9) - 2.705543454) < 1e-8
    assert abs(ensemble.chisquare_quantile_1dof(0.6826894921) - 1.0) < 1e-8
    assert abs(ensemble.chisquare_quantile_1dof(0.95) - 3.841458821) < 1e-8


def test_contour_interval_on_a_parabola():
    # NLL = 0.5 ((x - 3) / 0.5)^2 sampled on a grid: the 90% contour is |x - 3| < 0.5 sqrt(2.7055)
    x = np.linspace(0, 6, 6001)
    y = np.linspace(-1, 1, 6001)
    nll = 0.5 * ((x - 3.0) / 0.5) ** 2 + 10.0
    chain = np.stack([x, y, nll], axis=1).astype(np.float32)
    iv = ensemble.contour_intervals(chain, cl=0.9)
    half = 0.5 * math.sqrt(2.705543454)
    assert abs(iv[0, 0] - 3.0) < 2e-3                    # mid-range of the points within dNLL < 0.13
    assert abs(iv[0, 1] - (3.0 - half)) < 2e-3 and abs(iv[0, 2] - (3.0 + half)) < 2e-3
    assert iv[0, 3] == -999
    assert iv[1, 1] < iv[1, 0] < iv[1, 2]


def test_contour_widens_until_it_finds_a_point():
    # only two samples, 2 apart in NLL: the best point alone defines the estimate (contour.cpp:41-53)
    chain = np.array([[1.0, 5.0], [2.0, 7.0]], np.float32)
    iv = ensemble.contour_intervals(chain, cl=0.9)
    assert iv[0, 0] == 1.0 and iv[0, 1] == 1.0 and iv[0, 2] == 1.0


DELTA90 = float(np.float32(0.5 * 2.705543454))            # contour.cpp:19: 1.3527717 as a float, printed "1.35277"


def test_contour_known_answers_threshold_is_strict_and_exact():
    # NLL values straddling lmin + delta by 1e-3: strictly-less-than selects the first two (likelihood.cpp:93-96)
    chain = np.array([[1.0, 10.0], [2.0, 10.0 + 1.35277 - 1e-3], [3.0, 10.0 + 1.35277 + 1e-3], [0.5, 10.05]],
                     np.float32)
    iv = ensemble.contour_intervals(chain, cl=0.9)
    assert iv[0, 1] == 0.5 and iv[0, 2] == 2.0             # min / max of the parameter inside the contour
    assert iv[0, 0] == 0.75                                # mid-range of the points with dNLL < 0.13: x = 1 and 0.5
    assert iv[0, 3] == -999


def test_contour_known_answers_the_printed_offset_moves_the_contour():
    # lmin = -348086.3125 is written into the selection as "348086" (6 significant digits): a point 1.55 above the
    # minimum satisfies likelihood + 348086 < 1.35277 although 1.55 > delta -- the reference's contour, reproduced
    lmin = np.float32(-348086.3125)
    chain = np.array([[0.0, lmin], [5.0, lmin + np.float32(1.5625)], [7.0, lmin + np.float32(1.6875)]], np.float32)
    assert ensemble.as_printed(-lmin) == 348086.0
    iv = ensemble.contour_intervals(chain, cl=0.9)
    assert iv[0, 1] == 0.0 and iv[0, 2] == 5.0             # -0.3125 + 1.5625 = 1.25 < 1.35277; 1.375 is not
    assert iv[0, 0] == 0.0


def test_contour_known_answers_widening_by_fives():
    # lmin = -348085.6875 prints as "348086": even the minimum is 0.3125 above the printed offset, so the first
    # pass (0.13) finds nothing and the second (0.65) takes the points up to 0.65 - 0.3125 above the minimum
    lmin = np.float32(-348085.6875)
    chain = np.array([[0.0, lmin], [4.0, lmin + np.float32(0.25)], [9.0, lmin + np.float32(0.9375)],
                      [20.0, lmin + np.float32(1.0625)]], np.float32)
    assert ensemble.as_printed(-lmin) == 348086.0
    iv = ensemble.contour_intervals(chain, cl=0.9)
    assert iv[0, 0] == 2.0                                 # (0 + 4) / 2: 0.3125 and 0.5625 < 0.65, 1.25 is not
    assert iv[0, 1] == 0.0 and iv[0, 2] == 9.0             # 0.3125 + 0.9375 = 1.25 < 1.35277 <= 0.3125 + 1.0625


def test_gaus_fit_known_answers():
    # bin contents that ARE a Gaussian at the bin centres: chi2 = 0 exactly at (A, mu, sigma), which the fit must find
    x = np.linspace(-2.95, 8.95, 120)
    for a, mu, sigma in ((1000.0, 3.2, 0.9), (50.0, -0.5, 2.5), (1e6, 7.0, 0.3)):
        y = a * np.exp(-0.5 * ((x - mu) / sigma) ** 2)
        fit = ensemble.gaus_fit(x, y)
        assert fit is not None
        assert abs(fit[0] - a) <= 1e-6 * a and abs(fit[1] - mu) <= 1e-8 and abs(fit[2] - sigma) <= 1e-8
    assert ensemble.gaus_fit(x, np.zeros_like(x)) is None                      # nothing to fit
    assert ensemble.gaus_fit([0.0, 1.0], [5.0, 5.0]) is None                   # fewer than three non-empty bins


def test_projection_interval_central_and_one_sided():
    rng = np.random.default_rng(0)
    v = rng.normal(5.0, 1.0, 200000)
    mu, lo, hi, cov, one_sided = ensemble.projection_interval(v, cl=0.9)
    assert not one_sided and abs(mu - 5.0) < 0.02
    # the reference's walk counts the mode bin on the lower side only, so the upper limit overshoots
    assert abs(lo - (5.0 - 1.645)) < 0.2 and 5.0 + 1.645 - 0.1 < hi < 5.0 + 2.1 and 0.9 <= cov < 0.96
    v = np.abs(rng.normal(0.0, 1.0, 200000))                   # a rate piled up at its lower boundary
    mu, lo, hi, cov, one_sided = ensemble.projection_interval(v, cl=0.9)
    assert one_sided and lo <= 1e-3 and cov >= 0.9 and abs(hi - 1.645) < 0.1


def test_histogram_sampler_follows_the_histogram():
    rng = np.random.default_rng(1)
    bins = np.array([0, 10, 0, 30, 60, 0], np.uint32)       # 2 x 3, row-major
    pts = ensemble.random_sample(rng, bins, [0.0, 10.0], [1.0, 13.0], [2, 3], 100000)
    assert pts.shape == (100000, 2) and pts.dtype == np.float32
    ix = np.minimum((pts[:, 0] * 2).astype(int), 1)
    iy = np.minimum((pts[:, 1] - 10.0).astype(int), 2)
    got = np.bincount(ix * 3 + iy, minlength=6) / 100000.0
    assert np.all(np.abs(got - bins / 100.0) < 0.01)
    assert got[0] == 0 and got[2] == 0 and got[5] == 0      # empty bins are never drawn
    assert pts[:, 0].min() >= 0 and pts[:, 0].max() < 1 and pts[:, 1].min() >= 10 and pts[:, 1].max() < 13
    assert ensemble.random_sample(rng, np.zeros(4, np.uint32), [0.0], [1.0], [4], 5).shape == (0, 1)
