// ensemble.h -- ROOT-free forms of the code around the MCMC driver in the reference's experiment loop
// (src/sxmc.cpp:44-145): fake data sets, interval extraction, the loop itself.
//
//   make_fake_dataset            src/generator.cpp:10-48
//   EvalHist::RandomSample       src/pdfz.cpp:817-922 (TH1::GetRandom / GetRandom2 / GetRandom3: pick a bin
//                                with probability proportional to its content, then uniform inside the bin)
//   LikelihoodSpace::get_contour src/likelihood.cpp:90-102
//   Contour::get_interval        src/error_estimators/contour.cpp:18-69
//   Interval                     src/interval.h:11-29
//   median                       src/utils.h:76-90
//
// Parity with the reference is statistical only: these draw on ROOT's generators and TMath, and the
// reference holds no test for them.  Deviates come from std::mt19937_64 here.  Experiments are
// independent, so a multi-GPU run gives experiment k to rank k mod G (sxmc_amd/dist.py) and gathers
// the intervals once at the end.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <exception>
#include <functional>
#include <iomanip>
#include <map>
#include <memory>
#include <mutex>
#include <limits>
#include <ostream>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "mcmc.h"

namespace sxmc {

/** interval.h:11-29 */
struct Interval {
  bool one_sided = false;
  float point_estimate = -1;
  float lower = -1;
  float upper = -1;
  float cl = -1;
  float coverage = -1;
};

/** utils.h:76-90 */
template <typename T>
T median(std::vector<T> v) {
  std::sort(v.begin(), v.end());
  const size_t half = v.size() / 2;
  return v.size() % 2 == 0 ? (T)(1.0 * (v[half - 1] + v[half]) / 2) : v[half];
}

/** TMath::ChisquareQuantile(cl, 1) = (Phi^-1((1 + cl) / 2))^2, by bisection on erf. */
inline double chisquare_quantile_1dof(double cl) {
  double lo = 0.0, hi = 40.0;
  for (int i = 0; i < 200; i++) {
    const double mid = 0.5 * (lo + hi);
    (std::erf(std::sqrt(mid / 2.0)) < cl ? lo : hi) = mid;
  }
  return 0.5 * (lo + hi);
}

/** What `ostream << float` writes (6 significant digits, %g) read back: the reference builds its selections as
 *  TEXT -- "likelihood+" << -lmin << "<" << delta (likelihood.cpp:93-94, contour.cpp:45-46) -- so the offset and
 *  the threshold it actually applies are the printed, rounded ones.  With |lmin| of a few 1e5 (BASELINE config 3)
 *  the offset is off by up to 0.5, which moves the contour; reproduced here because the intervals are results. */
inline double as_printed(float v) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%g", (double)v);
  return std::strtod(buf, nullptr);
}

/** Contour::get_interval for every parameter of a chain (contour.cpp:17-69, likelihood.cpp:90-102). */
inline std::vector<Interval> contour_intervals(const Chain& chain, float cl = 0.9f) {
  const size_t ncol = chain.names.size(), P = ncol - 1, n = chain.nrows();
  float lmin = chain.at(0, P);
  for (size_t r = 1; r < n; r++) lmin = std::min(lmin, chain.at(r, P));
  const float delta = 0.5 * chisquare_quantile_1dof(cl);   // contour.cpp:19 (a float there too)
  // likelihood.cpp:90-102: rows with likelihood + (-lmin as printed) < (delta as printed)
  std::vector<size_t> contour;
  const double off = as_printed(-lmin), dprinted = as_printed(delta);
  for (size_t r = 0; r < n; r++)
    if ((double)chain.at(r, P) + off < dprinted) contour.push_back(r);
  if (contour.empty()) {   // (the reference asserts here: the printed offset lost the minimum; use the exact one)
    for (size_t r = 0; r < n; r++)
      if (chain.at(r, P) - lmin < delta) contour.push_back(r);
  }
  // contour.cpp:39-53: points near the maximum-likelihood point, widened 0.13, 0.65, 3.25, ... until one is found;
  // the offset is the minimum over the contour points, printed the same way
  float cmin = chain.at(contour[0], P);
  for (size_t r : contour) cmin = std::min(cmin, chain.at(r, P));
  const double coff = as_printed(-cmin);
  std::vector<size_t> near;
  float dnll = 0.13f;
  do {
    near.clear();
    const double dn = as_printed(dnll);
    for (size_t r : contour)
      if ((double)chain.at(r, P) + coff < dn) near.push_back(r);
    dnll *= 5;
  } while (near.empty());
  std::vector<Interval> out(P);
  for (size_t p = 0; p < P; p++) {
    Interval iv;
    iv.cl = cl;
    iv.one_sided = false;
    iv.coverage = -999;
    float nlo = chain.at(near[0], p), nhi = nlo, clo = chain.at(contour[0], p), chi = clo;
    for (size_t r : near) {
      nlo = std::min(nlo, chain.at(r, p));
      nhi = std::max(nhi, chain.at(r, p));
    }
    for (size_t r : contour) {
      clo = std::min(clo, chain.at(r, p));
      chi = std::max(chi, chain.at(r, p));
    }
    iv.point_estimate = (nlo + nhi) / 2;
    iv.lower = clo;
    iv.upper = chi;
    out[p] = iv;
  }
  return out;
}

/** What `TH1::Fit("gaus")` minimises (projection.cpp:22-23): chi2 over the non-empty bins of
 *  ((n_i - A exp(-(x_i - mu)^2 / (2 sigma^2))) / sqrt(n_i))^2, the function taken at the bin centre, started from
 *  the histogram's maximum, mean and RMS (TH1's InitGaus).  Levenberg-Marquardt here, Minuit MIGRAD in ROOT: the
 *  same minimum.  false when it does not converge to a positive width. */
inline bool gaus_fit(const std::vector<double>& centers, const std::vector<double>& counts, double& A, double& mu,
                     double& sigma) {
  std::vector<double> x, y;
  for (size_t i = 0; i < centers.size(); i++)
    if (counts[i] > 0) {
      x.push_back(centers[i]);
      y.push_back(counts[i]);
    }
  const size_t n = x.size();
  if (n < 3) return false;
  double sy = 0, sxy = 0, ymax = 0;
  for (size_t i = 0; i < n; i++) {
    sy += y[i];
    sxy += x[i] * y[i];
    ymax = std::max(ymax, y[i]);
  }
  const double mean = sxy / sy;
  double var = 0;
  for (size_t i = 0; i < n; i++) var += y[i] * (x[i] - mean) * (x[i] - mean);
  const double rms = std::sqrt(std::max(var / sy, 0.0));
  if (!(rms > 0)) return false;
  double p[3] = {ymax, mean, rms};
  auto chi2_of = [&](const double* q) {
    if (!(q[2] > 0)) return std::numeric_limits<double>::infinity();
    double c = 0;
    for (size_t i = 0; i < n; i++) {
      const double z = (x[i] - q[1]) / q[2], r = (y[i] - q[0] * std::exp(-0.5 * z * z)) / std::sqrt(y[i]);
      c += r * r;
    }
    return c;
  };
  double lam = 1e-3, chi2 = chi2_of(p);
  for (int it = 0; it < 200; it++) {
    double a[3][3] = {{0}}, b[3] = {0};
    for (size_t i = 0; i < n; i++) {
      const double e = std::sqrt(y[i]), d = x[i] - p[1], g = std::exp(-0.5 * d * d / (p[2] * p[2]));
      const double j[3] = {g / e, p[0] * g * d / (p[2] * p[2]) / e, p[0] * g * d * d / (p[2] * p[2] * p[2]) / e};
      const double r = (y[i] - p[0] * g) / e;
      for (int u = 0; u < 3; u++) {
        b[u] += j[u] * r;
        for (int v = 0; v < 3; v++) a[u][v] += j[u] * j[v];
      }
    }
    double m[3][4];
    for (int u = 0; u < 3; u++) {
      for (int v = 0; v < 3; v++) m[u][v] = a[u][v] + (u == v ? lam * (a[u][u] + 1e-300) : 0.0);
      m[u][3] = b[u];
    }
    bool singular = false;
    for (int c = 0; c < 3 && !singular; c++) {   // Gauss-Jordan with partial pivoting
      int piv = c;
      for (int r = c + 1; r < 3; r++)
        if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
      if (m[piv][c] == 0.0) singular = true;
      for (int k = 0; k < 4 && !singular; k++) std::swap(m[c][k], m[piv][k]);
      for (int r = 0; r < 3 && !singular; r++) {
        if (r == c) continue;
        const double f = m[r][c] / m[c][c];
        for (int k = c; k < 4; k++) m[r][k] -= f * m[c][k];
      }
    }
    if (singular) return false;
    const double step[3] = {m[0][3] / m[0][0], m[1][3] / m[1][1], m[2][3] / m[2][2]};
    const double trial[3] = {p[0] + step[0], p[1] + step[1], p[2] + step[2]};
    const double c2 = chi2_of(trial);
    if (c2 <= chi2) {
      bool small = chi2 - c2 <= 1e-12 * std::max(chi2, 1e-300);
      for (int u = 0; u < 3; u++) small = small && std::fabs(step[u]) <= 1e-10 * (std::fabs(p[u]) + 1e-300);
      for (int u = 0; u < 3; u++) p[u] = trial[u];
      chi2 = c2;
      lam = std::max(lam * 0.3, 1e-12);
      if (small) break;
    } else {
      lam *= 10.0;
      if (lam > 1e12) break;
    }
  }
  if (!(p[2] > 0) || !std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return false;
  A = p[0];
  mu = p[1];
  sigma = p[2];
  return true;
}

/** Projection::get_interval on one parameter's samples (projection.cpp:14-77): histogram (ROOT's TTree::Draw
 *  picks range and binning by its own "nice limits" rule, not reproduced: `nbins` bins over [min, max] here),
 *  Gaussian fit for the point estimate, limits walked outwards from the mean's bin until cl / 2 of the samples
 *  lie on either side (one-sided from the low edge when less than cl / 2 lies below the mean). */
inline Interval projection_interval(const std::vector<float>& values, float cl = 0.9f, int nbins = 100) {
  Interval iv;
  iv.cl = cl;
  double lo = values.at(0), hi = lo;
  for (float v : values) {
    lo = std::min<double>(lo, v);
    hi = std::max<double>(hi, v);
  }
  if (!(hi > lo)) {
    iv.point_estimate = iv.lower = (float)lo;
    iv.upper = (float)hi;
    iv.coverage = 1;
    return iv;
  }
  const double width = (hi - lo) / nbins;
  std::vector<double> counts((size_t)nbins, 0.0), centers((size_t)nbins), csum((size_t)nbins + 1, 0.0);
  // TH1 conventions (TAxis::FindBin): bin = 1 + int(nbins (x - xmin) / (xmax - xmin)); the maximum counts in the last bin
  for (float v : values) counts[(size_t)std::min<long>(nbins - 1, (long)(nbins * ((double)v - lo) / (hi - lo)))] += 1;
  for (int i = 0; i < nbins; i++) centers[(size_t)i] = lo + (i + 0.5) * width;
  double total = 0;
  for (int i = 0; i < nbins; i++) csum[(size_t)i + 1] = (total += counts[(size_t)i]);   // csum[i] = bins 1..i
  double a = 0, mu = 0, sigma = 0;
  if (!gaus_fit(centers, counts, a, mu, sigma)) {
    mu = centers[(size_t)(std::max_element(counts.begin(), counts.end()) - counts.begin())];
  }
  // 1-based bin of the mean (TH1::FindBin): 0 below the range, nbins + 1 at or beyond its end
  long imax = mu < lo ? 0 : mu >= hi ? nbins + 1 : 1 + (long)(nbins * (mu - lo) / (hi - lo));
  if (imax < 1) {                                         // projection.cpp:28-31
    imax = 1;
    mu = lo;
  }
  imax = std::min<long>(imax, nbins);
  long ilo = 1, ihi = 0;
  if (csum[(size_t)imax] / total < cl / 2) {              // projection.cpp:36-45
    iv.one_sided = true;
    for (long i = 0; i <= nbins; i++)
      if (csum[(size_t)i] / total >= cl) {
        ihi = i;
        break;
      }
  } else {
    iv.one_sided = false;
    for (long i = imax; i > 0; i--)
      if ((csum[(size_t)imax] - csum[(size_t)i - 1]) / total >= cl / 2) {
        ilo = i;
        break;
      }
    for (long i = imax + 1; i <= nbins; i++)
      if ((csum[(size_t)i] - csum[(size_t)imax]) / total >= cl / 2) {
        ihi = i;
        break;
      }
  }
  ihi = ihi ? std::max(ihi, ilo) : nbins;
  iv.point_estimate = (float)mu;
  iv.coverage = (float)((csum[(size_t)ihi] - csum[(size_t)ilo - 1]) / total);
  iv.lower = (float)(lo + (ilo - 1) * width);
  iv.upper = (float)(lo + (ihi - 1) * width + width);   // projection.cpp:73: GetBinLowEdge(ihi) + GetBinWidth(ihi)
  return iv;
}

/** Projection::get_interval for every parameter of a chain. */
inline std::vector<Interval> projection_intervals(const Chain& chain, float cl = 0.9f) {
  const size_t P = chain.names.size() - 1;
  std::vector<Interval> out;
  for (size_t p = 0; p < P; p++) {
    std::vector<float> col;
    for (size_t r = 0; r < chain.nrows(); r++) col.push_back(chain.at(r, p));
    out.push_back(projection_interval(col, cl));
  }
  return out;
}

/** error_estimator.h: how the intervals are taken from the sampled likelihood space (fit.error_type). */
enum ErrorType { ERROR_CONTOUR, ERROR_PROJECTION };

/** LikelihoodSpace::extract_best_fit (likelihood.cpp:104-137): every parameter's interval by the chosen estimator. */
inline std::vector<Interval> extract_intervals(const Chain& chain, float cl, ErrorType error_type) {
  return error_type == ERROR_PROJECTION ? projection_intervals(chain, cl) : contour_intervals(chain, cl);
}

/** Interval::str (interval.cpp:6-20): "point -lower_error +upper_error", or "point <upper (cl% CL)". */
inline std::string interval_str(const Interval& iv) {
  const float lower_error = iv.point_estimate - iv.lower, upper_error = iv.upper - iv.point_estimate;
  std::ostringstream ss;
  ss << iv.point_estimate;
  if (iv.one_sided) ss << " <" << iv.upper << " (" << 100 * iv.cl << "% CL)";
  else ss << " -" << lower_error << " +" << upper_error;
  return ss.str();
}

/** get_correlation_matrix (utils.cpp:29-77) of a chain's parameter columns (every column but `likelihood`), row-major
 *  [P][P].  As there: sums, means and products accumulate in float in row order, the square root is taken in double,
 *  and only the diagonal and what is to the right of it is computed -- the entries below stay 0. */
inline std::vector<float> correlation_matrix(const Chain& chain) {
  const size_t P = chain.names.size() - 1, n = chain.nrows();
  std::vector<float> matrix(P * P, 0.0f), means(P, 0.0f);
  for (size_t k = 0; k < n; k++)
    for (size_t j = 0; j < P; j++) means[j] += chain.at(k, j);
  for (size_t j = 0; j < P; j++) means[j] /= (int)n;
  for (size_t i = 0; i < P; i++) {
    for (size_t j = i; j < P; j++) {
      float t = 0, dx2 = 0, dy2 = 0;
      for (size_t k = 0; k < n; k++) {
        const float x1 = chain.at(k, i) - means[i], x2 = chain.at(k, j) - means[j];
        t += x1 * x2;
        dx2 += x1 * x1;
        dy2 += x2 * x2;
      }
      matrix[i * P + j] = (float)(t / std::sqrt((double)(dx2 * dy2)));
    }
  }
  return matrix;
}

/** LikelihoodSpace::print_best_fit (likelihood.cpp:34-45): the parameters in NAME order (a std::map there), then
 *  the minimum of the likelihood column (likelihood.cpp:134). */
inline void print_best_fit(std::ostream& os, const Chain& chain, const std::vector<Interval>& intervals) {
  const size_t P = chain.names.size() - 1;
  std::map<std::string, Interval> by_name;
  for (size_t p = 0; p < P && p < intervals.size(); p++) by_name[chain.names[p]] = intervals[p];
  os << "-- Best fit --" << std::endl;
  for (const auto& kv : by_name) {
    if (kv.first == "likelihood") continue;
    os << " " << kv.first << ": " << interval_str(kv.second) << std::endl;
  }
  float lmin = chain.nrows() ? chain.at(0, P) : 0.0f;
  for (size_t r = 1; r < chain.nrows(); r++) lmin = std::min(lmin, chain.at(r, P));
  os << " NLL: " << lmin << std::endl;
}

/** LikelihoodSpace::print_correlations (likelihood.cpp:48-72): names in column order, right-aligned to the longest,
 *  entries fixed with three decimals in eight columns. */
inline void print_correlations(std::ostream& os, const Chain& chain) {
  const size_t P = chain.names.size() - 1;
  const std::vector<float> c = correlation_matrix(chain);
  os << "-- Correlation matrix --" << std::endl;
  int maxlen = 0;
  for (size_t i = 0; i < P; i++) maxlen = std::max(maxlen, (int)chain.names[i].length());
  for (size_t i = 0; i < P; i++) {
    os << std::setw(maxlen) << chain.names[i] << " ";
    for (size_t j = 0; j < P; j++) {
      os << std::setiosflags(std::ios::fixed) << std::setprecision(3) << std::setw(8) << c[j + i * P];
    }
    os << std::resetiosflags(std::ios::fixed) << std::endl;
  }
}

/** RandomSample on a flat row-major histogram (1-3 D). */
inline void random_sample(std::mt19937_64& rng, const std::vector<unsigned>& bins, const std::vector<Observable>& obs,
                          size_t nobserved, unsigned dataset, std::vector<float>& events) {
  const size_t D = obs.size();
  if (D > 3) throw pdfz::Error("Cannot EvalHist::CreateHistogram for dimensions greater than 3!");
  std::vector<double> cdf(bins.size());
  double total = 0;
  for (size_t i = 0; i < bins.size(); i++) cdf[i] = (total += bins[i]);
  if (total <= 0) return;
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  for (size_t e = 0; e < nobserved; e++) {
    size_t flat = std::upper_bound(cdf.begin(), cdf.end(), uni(rng) * total) - cdf.begin();
    flat = std::min(flat, bins.size() - 1);
    std::vector<size_t> idx(D);
    for (size_t k = D; k-- > 0;) {
      idx[k] = flat % obs[k].bins;
      flat /= obs[k].bins;
    }
    for (size_t k = 0; k < D; k++) {
      const double width = ((double)obs[k].upper - (double)obs[k].lower) / (double)obs[k].bins;
      events.push_back((float)((double)obs[k].lower + ((double)idx[k] + uni(rng)) * width));
    }
    events.push_back((float)dataset);
  }
}

/** make_fake_dataset (generator.cpp:10-48).  observables must be in field order. */
inline std::vector<float> make_fake_dataset(std::mt19937_64& rng, std::vector<Signal>& signals,
                                            std::vector<Systematic>& systematics,
                                            std::vector<Observable>& observables, bool poisson,
                                            std::vector<unsigned>* observed_out = nullptr) {
  std::vector<float> events;
  for (Signal& s : signals) {
    const double eff = get_efficiency(s, systematics);
    const double nevents = s.nexpected * eff;
    size_t observed;
    if (poisson) {
      observed = nevents > 0 ? std::poisson_distribution<long long>(nevents)(rng) : 0;
    } else {
      observed = (size_t)std::floor(nevents + 0.5);
    }
    if (observables.size() > 3) throw pdfz::Error("Cannot EvalHist::CreateHistogram for dimensions greater than 3!");
    if (eff <= 0) observed = 0;   // an empty histogram yields no events
    // drawn on the device from the histogram get_efficiency just filled: it never leaves HBM
    if (observed) dynamic_cast<pdfz::EvalHist*>(s.histogram)->SampleEvents(events, observed, rng());
    if (observed_out) observed_out->push_back((unsigned)observed);
  }
  return events;
}

/** sxmc.cpp:130-141 writes every experiment's sampled likelihood space ("ls") to <output_prefix>_<i>.root.  Set this to
 *  receive the chains (experiment index, chain) -- e.g. to write them with write_chain_npz (config.h).  Called on the
 *  experiment's own host thread; calls are serialised.  Empty by default: chains are dropped once their intervals are
 *  taken. */
inline std::function<void(unsigned, const Chain&)>& chain_sink() {
  static std::function<void(unsigned, const Chain&)> sink;
  return sink;
}

/** Set by a caller whose configuration lists data sets (sxmc.cpp:71-80): asked for experiment k's events (rows of
 *  nobservables + 1 floats); true = `rows` holds them, false = the experiment samples a fake data set as usual.  May be
 *  called from several host threads at once (one per chain in flight): it must only read. */
inline std::function<bool(unsigned, std::vector<float>&)>& data_source() {
  static std::function<bool(unsigned, std::vector<float>&)> source;
  return source;
}

/** Set by a caller that wants what sxmc.cpp:100-101 prints for every experiment -- the text of print_best_fit followed
 *  by print_correlations -- handed over as (experiment index, text), one call at a time. */
inline std::function<void(unsigned, const std::string&)>& report_sink() {
  static std::function<void(unsigned, const std::string&)> sink;
  return sink;
}

/** A chain from a table of columns (parameter names..., "likelihood"): what read_table (config.h) returns for a file
 *  written by write_chain_npz / sxmc_amd/io.py -- the `fit.samples` path of sxmc.cpp:84-94, where a saved likelihood
 *  space replaces the walk. */
inline Chain chain_from_table(const std::vector<float>& matrix, const std::vector<std::string>& fields) {
  if (fields.empty() || fields.back() != "likelihood" || matrix.size() % fields.size() != 0 || matrix.empty()) {
    throw pdfz::Error("a saved chain needs at least one row and \"likelihood\" as its last column");
  }
  Chain c;
  c.names = fields;
  c.rows = matrix;
  return c;
}

struct ExperimentResult {
  unsigned index = 0;
  std::vector<Interval> intervals;  //!< one per parameter
  size_t accepted = 0;
  size_t nevents = 0;
  /** where the experiment's host time went (seconds): its data (fake-data draw or configured files), the chain's
   *  construction, the walk's set-up (buffers, SetEvalPoints, first evaluation), its steps, its tear-down (buffers
   *  freed), the chain's destruction, the intervals.  What a short experiment spends outside `steps` is what an
   *  ensemble of short experiments loses (bench_cpp prints the sums). */
  struct Phases {
    double data = 0, construct = 0, walk_setup = 0, steps = 0, walk_teardown = 0, destroy = 0, intervals = 0;
  } phases;
};

/** The experiment loop of sxmc.cpp:59-145 over the given experiment indices (all of them on one GPU,
 *  or this rank's share).  Evaluators (and their MC tables in HBM) are reused by every experiment. */
/** Per-experiment seed (the reference's single sequential gRandom stream cannot be sharded). */
inline unsigned long long experiment_seed(unsigned long long base_seed, unsigned k) {
  unsigned long long x = base_seed * 0x9E3779B97F4A7C15ull + (k + 1ull) * 0xBF58476D1CE4E5B9ull;
  x ^= x >> 31;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 29;
  return x;
}

/** Where the lanes of ensemble_concurrent meet, twice per round of experiments: when every lane's walk is set up (before
 *  any of them queues its first long run of steps) and when every lane's last step has finished (before any tears
 *  down).  Set-up and tear-down synchronise the whole device; beside a chain that has a second of graph replays queued
 *  each of those calls waits that second out, under the set-up lock the other lanes need -- which is how eight lanes came
 *  to walk one at a time at 1e5 steps per experiment.  A lane that fails breaks the barrier: nobody waits for it. */
class LaneBarrier {
 public:
  void arrive_and_wait(size_t expected) {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return;
    const unsigned long long gen = generation;
    if (++count >= expected) {
      count = 0;
      generation++;
      cv.notify_all();
      return;
    }
    cv.wait(lk, [&] { return generation != gen || broken; });
  }
  void break_all() {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }

 private:
  std::mutex m;
  std::condition_variable cv;
  size_t count = 0;
  unsigned long long generation = 0;
  bool broken = false;
};

/** One iteration of sxmc.cpp:59-145: fake data -> MCMC -> intervals, on `stream` (null: default). */
inline ExperimentResult run_experiment(unsigned k, unsigned long long base_seed, std::vector<Source>& sources,
                                       std::vector<Signal>& signals, std::vector<Systematic>& systematics,
                                       std::vector<Observable>& observables, unsigned nsteps, float burnin_fraction,
                                       float cl, unsigned sync_interval, unsigned graph_steps = 0,
                                       sxmc_stream_t stream = nullptr, SetupLock* exclusive = nullptr,
                                       LockstepSet* lockstep = nullptr, size_t lockstep_index = 0,
                                       ErrorType error_type = ERROR_CONTOUR, LaneBarrier* meet = nullptr,
                                       size_t meet_lanes = 0) {
  const unsigned long long x = experiment_seed(base_seed, k);
  std::mt19937_64 rng(x);
  // `exclusive` (one chain per host thread): held over everything that allocates, copies through the
  // legacy stream or synchronises the device -- see MCMC::exclusive
  std::unique_lock<SetupLock> lock;
  if (exclusive) lock = std::unique_lock<SetupLock>(*exclusive);
  typedef std::chrono::steady_clock PhaseClock;
  auto since = [](PhaseClock::time_point t) { return std::chrono::duration<double>(PhaseClock::now() - t).count(); };
  ExperimentResult r;
  PhaseClock::time_point t = PhaseClock::now();
  std::vector<float> data;
  if (!(data_source() && data_source()(k, data))) {
    data = make_fake_dataset(rng, signals, systematics, observables, true);
  }
  r.phases.data = since(t);
  t = PhaseClock::now();
  std::unique_ptr<MCMC> mcmc(new MCMC(sources, signals, systematics, observables, x, stream));
  mcmc->graph_steps = graph_steps;
  mcmc->exclusive = exclusive;
  mcmc->lockstep = lockstep;
  mcmc->lockstep_index = lockstep_index;
  if (lockstep) mcmc->optimize = false;   // chains that share a fill pass share ONE launch shape: the default one
  // (lanes of one round meet when all are set up and when all have stepped; a walk that ends without having passed a
  // meeting point -- a throw, a form that has none -- passes it afterwards, so that nobody waits for it)
  bool met_setup = false, met_steps = false;
  if (meet && meet_lanes > 1) {
    mcmc->on_setup_done = [&]() {
      met_setup = true;
      meet->arrive_and_wait(meet_lanes);
    };
    mcmc->on_steps_done = [&]() {
      met_steps = true;
      meet->arrive_and_wait(meet_lanes);
    };
  }
  r.phases.construct = since(t);
  if (exclusive) lock.unlock();   // the walk takes it itself
  t = PhaseClock::now();
  Chain chain;
  try {
    chain = (*mcmc)(data, nsteps, burnin_fraction, false, sync_interval);
  } catch (...) {
    if (meet) meet->break_all();
    throw;
  }
  if (meet && meet_lanes > 1) {
    if (!met_setup) meet->arrive_and_wait(meet_lanes);
    if (!met_steps) meet->arrive_and_wait(meet_lanes);
  }
  r.phases.walk_setup = chain.setup_seconds;
  r.phases.steps = chain.steps_seconds;
  r.phases.walk_teardown = since(t) - chain.setup_seconds - chain.steps_seconds;
  t = PhaseClock::now();
  if (exclusive) lock.lock();
  mcmc.reset();
  if (exclusive) lock.unlock();
  r.phases.destroy = since(t);
  if (chain_sink()) {
    static std::mutex sink_mutex;
    std::lock_guard<std::mutex> guard(sink_mutex);
    chain_sink()(k, chain);
  }
  t = PhaseClock::now();
  r.index = k;
  r.intervals = extract_intervals(chain, cl, error_type);
  r.phases.intervals = since(t);
  r.accepted = chain.accepted;
  r.nevents = data.size() / (observables.size() + 1);
  if (report_sink()) {
    std::ostringstream os;
    print_best_fit(os, chain, r.intervals);
    print_correlations(os, chain);
    static std::mutex report_mutex;
    std::lock_guard<std::mutex> guard(report_mutex);
    report_sink()(k, os.str());
  }
  return r;
}

inline std::vector<ExperimentResult> ensemble(const std::vector<unsigned>& experiments, unsigned long long base_seed,
                                              std::vector<Source>& sources, std::vector<Signal>& signals,
                                              std::vector<Systematic>& systematics,
                                              std::vector<Observable>& observables, unsigned nsteps,
                                              float burnin_fraction, float cl = 0.9f, unsigned sync_interval = 10000,
                                              unsigned graph_steps = 0, ErrorType error_type = ERROR_CONTOUR) {
  PoolScope pool;   // the experiments' arrays recycle their blocks instead of allocating and freeing (device_array.h)
  std::vector<ExperimentResult> out;
  for (unsigned k : experiments) {
    out.push_back(run_experiment(k, base_seed, sources, signals, systematics, observables, nsteps, burnin_fraction,
                                 cl, sync_interval, graph_steps, nullptr, nullptr, nullptr, 0, error_type));
  }
  return out;
}

/** The same loop with `nconcurrent` experiments in flight on this GPU (BASELINE config 4's per-GPU shape:
 *  one experiment per stream).  Each lane is a host thread with its own non-blocking stream and its own
 *  evaluators, which share the resident sample tables of `signals` (share_pdfz); lane t runs experiments
 *  t, t + nconcurrent, ...  Results come back in the order of `experiments` and are the ones `ensemble`
 *  gives (every experiment is seeded by its index). */
inline std::vector<ExperimentResult> ensemble_concurrent(const std::vector<unsigned>& experiments,
                                                         unsigned long long base_seed, std::vector<Source>& sources,
                                                         std::vector<Signal>& signals,
                                                         std::vector<Systematic>& systematics,
                                                         std::vector<Observable>& observables, unsigned nsteps,
                                                         float burnin_fraction, unsigned nconcurrent, float cl = 0.9f,
                                                         unsigned sync_interval = 10000, unsigned graph_steps = 0,
                                                         int device = -1, SetupLock* device_exclusive = nullptr,
                                                         ErrorType error_type = ERROR_CONTOUR) {
  PoolScope pool;   // the experiments' arrays recycle their blocks instead of allocating and freeing (device_array.h)
  const size_t lanes = std::max<size_t>(1, std::min<size_t>(nconcurrent, experiments.size()));
  std::vector<ExperimentResult> out(experiments.size());
  std::vector<std::exception_ptr> errors(lanes);
  // set-up, graph recording and tear-down of the lanes of this device, one at a time (ensemble_multi_gpu passes the
  // device's lock so that lanes of the same card started from different calls still take turns)
  SetupLock own_exclusive;
  SetupLock& exclusive = device_exclusive ? *device_exclusive : own_exclusive;
  LaneBarrier meet;    // the lanes walk in ROUNDS: set up one after the other, step side by side, tear down
  std::vector<std::thread> threads;
  for (size_t t = 0; t < lanes; t++) {
    threads.emplace_back([&, t]() {
      sxmc_stream_t strm = nullptr;
      std::vector<Signal> mine;
      try {
        if (device >= 0) check(sxmc_set_device(device));   // (the current device is a per-thread setting)
        {
          std::lock_guard<SetupLock> lock(exclusive);
          check(sxmc_stream_create_nonblocking(&strm));
          transfer_stream() = strm;
          for (const Signal& s : signals) mine.push_back(share_pdfz(s));
        }
        std::vector<Source> src = sources;
        std::vector<Systematic> sys = systematics;
        std::vector<Observable> obs = observables;
        for (size_t i = t; i < experiments.size(); i += lanes) {
          // (lanes in this round: all of them, or what is left of the list in its last round)
          const size_t round_lanes = std::min(lanes, experiments.size() - (i - t));
          out[i] = run_experiment(experiments[i], base_seed, src, mine, sys, obs, nsteps, burnin_fraction, cl,
                                  sync_interval, graph_steps, strm, &exclusive, nullptr, 0, error_type, &meet, round_lanes);
        }
      } catch (...) {
        errors[t] = std::current_exception();
        meet.break_all();
      }
      {
        std::lock_guard<SetupLock> lock(exclusive);
        for (Signal& s : mine) delete s.histogram;
        transfer_stream() = nullptr;
        if (strm) sxmc_stream_destroy(strm);
      }
    });
  }
  for (std::thread& th : threads) th.join();
  for (std::exception_ptr& e : errors)
    if (e) std::rethrow_exception(e);
  return out;
}

/** The same loop with the experiments in flight advanced in LOCKSTEP sets (BASELINE config 4's per-GPU shape, taken
 *  further): `nsets` sets of `chains_per_set` chains; the chains of a set walk on one stream and share ONE pass
 *  over the sample tables per step (LockstepSet / sxmc_multigroup_step_async: the bytes streamed per evaluation
 *  divide by the chains per set; config 3: 9 500 steps/s with 2 sets of 4 against 5 500 with a fill per chain),
 *  different sets run on different streams so that one set's step ends overlap another's fill.  Every lane is a
 *  host thread, as in ensemble_concurrent; experiments that do not fill a whole round of nsets x chains_per_set
 *  lanes run through ensemble_concurrent at the end.  Results are those of `ensemble`, in the order of
 *  `experiments`. */
inline std::vector<ExperimentResult> ensemble_lockstep(const std::vector<unsigned>& experiments,
                                                       unsigned long long base_seed, std::vector<Source>& sources,
                                                       std::vector<Signal>& signals, std::vector<Systematic>& systematics,
                                                       std::vector<Observable>& observables, unsigned nsteps,
                                                       float burnin_fraction, unsigned chains_per_set, unsigned nsets,
                                                       float cl = 0.9f, unsigned sync_interval = 10000,
                                                       unsigned graph_steps = 10, int device = -1,
                                                       SetupLock* device_exclusive = nullptr,
                                                       ErrorType error_type = ERROR_CONTOUR) {
  PoolScope pool;   // the experiments' arrays recycle their blocks instead of allocating and freeing (device_array.h)
  const size_t L = std::max(2u, std::min(4u, chains_per_set)), S = std::max(1u, nsets), lanes = L * S;
  const size_t usable = experiments.size() / lanes * lanes;
  std::vector<ExperimentResult> out(experiments.size());
  SetupLock own_exclusive;
  SetupLock& exclusive = device_exclusive ? *device_exclusive : own_exclusive;
  if (usable) {
    std::vector<std::unique_ptr<LockstepSet>> sets;
    std::vector<sxmc_stream_t> streams(S, nullptr);
    if (device >= 0) check(sxmc_set_device(device));
    for (size_t k = 0; k < S; k++) {
      check(sxmc_stream_create_nonblocking(&streams[k]));
      sets.emplace_back(new LockstepSet(L, streams[k], &exclusive));
    }
    std::vector<std::exception_ptr> errors(lanes);
    std::vector<std::thread> threads;
    for (size_t t = 0; t < lanes; t++) {
      threads.emplace_back([&, t]() {
        LockstepSet& set = *sets[t / L];
        std::vector<Signal> mine;
        try {
          if (device >= 0) check(sxmc_set_device(device));
          {
            std::lock_guard<SetupLock> lock(exclusive);
            transfer_stream() = set.stream;
            for (const Signal& s : signals) mine.push_back(share_pdfz(s));
          }
          std::vector<Source> src = sources;
          std::vector<Systematic> sys = systematics;
          std::vector<Observable> obs = observables;
          for (size_t i = t; i < usable; i += lanes) {
            out[i] = run_experiment(experiments[i], base_seed, src, mine, sys, obs, nsteps, burnin_fraction, cl,
                                    sync_interval, graph_steps, set.stream, &exclusive, &set, t % L, error_type);
          }
        } catch (const pdfz::Error& e) {
          errors[t] = std::current_exception();
          set.abandon(e.msg);
        } catch (const std::exception& e) {
          errors[t] = std::current_exception();
          set.abandon(e.what());
        } catch (...) {
          errors[t] = std::current_exception();
          set.abandon("a chain of the set failed");
        }
        std::lock_guard<SetupLock> lock(exclusive);
        for (Signal& s : mine) delete s.histogram;
        transfer_stream() = nullptr;
      });
    }
    for (std::thread& th : threads) th.join();
    sets.clear();
    for (sxmc_stream_t st : streams)
      if (st) sxmc_stream_destroy(st);
    for (std::exception_ptr& e : errors)
      if (e) std::rethrow_exception(e);
  }
  if (usable < experiments.size()) {
    std::vector<unsigned> rest(experiments.begin() + (std::ptrdiff_t)usable, experiments.end());
    std::vector<ExperimentResult> r = ensemble_concurrent(rest, base_seed, sources, signals, systematics, observables,
                                                          nsteps, burnin_fraction, (unsigned)lanes, cl, sync_interval,
                                                          graph_steps, device, &exclusive, error_type);
    for (size_t i = 0; i < r.size(); i++) out[usable + i] = r[i];
  }
  return out;
}


/** How the device threads of ensemble_multi_gpu meet before the collective: every thread arrives exactly once, with
 *  "my part went well" or not, and all of them learn whether EVERY part went well.  A collective is entered by all
 *  ranks or by none -- a rank that failed on the way must never leave its peers waiting inside ncclAllGather. */
class Rendezvous {
 public:
  explicit Rendezvous(size_t n_) : n(n_) {}
  bool arrive(bool ok) {
    std::unique_lock<std::mutex> lock(m);
    all_ok = all_ok && ok;
    if (++arrived == n) {
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return arrived == n; });
    }
    return all_ok;
  }

 private:
  std::mutex m;
  std::condition_variable cv;
  size_t n, arrived = 0;
  bool all_ok = true;
};

struct MultiGpuOptions {
  float cl = 0.9f;
  unsigned sync_interval = 10000;
  unsigned graph_steps = 0;
  unsigned nconcurrent = 4;       //!< experiments in flight per device when lockstep_chains < 2 (ensemble_concurrent)
  unsigned lockstep_chains = 4;   //!< >= 2: per device, lockstep sets of this many chains (ensemble_lockstep) ...
  unsigned lockstep_sets = 2;     //!< ... this many sets in flight
  /** How the per-device blocks of intervals meet.  RCCL: one ncclAllGather over xGMI -- the product path.
   *  HOST_STAGING: every device thread's block is copied into rank 0's buffer by the host.  It exists so that the
   *  G-thread logic (sharding, per-thread replicas, rendezvous, result order, medians, failure handling) can be
   *  rehearsed on a box with fewer cards than ranks -- RCCL refuses two ranks on one card -- and is never chosen by
   *  default. */
  enum Exchange { RCCL, HOST_STAGING } exchange = RCCL;
  double exchange_timeout_seconds = 120.0;   //!< a collective still pending after this long is aborted (fail fast)
  /** Which lock serialises set-up (allocation, uploads, launch plans, module loads), graph recording and tear-down of
   *  the device threads.
   *  PROCESS_WIDE (default): ONE lock for all cards -- a thread never allocates or synchronises a device while another
   *  thread records a graph or sets up, on whichever card.  No run on more than one card is on record yet, so the
   *  conservative lock is the default.
   *  PER_DEVICE: one lock per card for set-up and tear-down -- the cards' set-ups run side by side -- while GRAPH
   *  RECORDING stays exclusive for the whole process (RecordingGate: whoever records waits for every set-up section in
   *  the process to end, and new ones wait for the recording): only the paths that never record graphs are relaxed.
   *  With 16 experiments per card at 77 ms of set-up each, one lock for 8 cards serialises ~10 s of set-up against
   *  ~3.5 s of walking per card; this mode is what a real node is measured with (bench_cpp --per-device-locks, which
   *  bench.py's multi-GPU leg tries first).
   *  PER_RANK (tests): a lock per rank even when ranks share a card -- set-up on one host thread beside stepping and
   *  set-up of another on the SAME device, recording still exclusive: the concurrency PER_DEVICE allows between cards,
   *  exercised on a box with one. */
  enum Locking { PROCESS_WIDE, PER_DEVICE, PER_RANK } locking = PROCESS_WIDE;
  ErrorType error_type = ERROR_CONTOUR;      //!< fit.error_type: contour or projection intervals
  /** Called by every device thread (argument: its rank) when its experiments are done, before the rendezvous.
   *  May throw: the tests inject a failing rank with it. */
  std::function<void(size_t)> before_exchange;
};

struct MultiGpuEnsemble {
  std::vector<ExperimentResult> results;  //!< one per experiment, in experiment order (computed on its own GPU)
  std::vector<float> gathered;            //!< [nexperiments][nparameters][4] = point_estimate, lower, upper, coverage:
                                          //!< what rank 0 received through the RCCL all-gather, in experiment order
  std::vector<float> median_upper;        //!< per parameter: median over the experiments of the upper limit
  size_t nparameters = 0;
  int rccl_nranks = 0;                    //!< ncclCommCount of rank 0's communicator (0: host staging)
  std::vector<int> rccl_devices;          //!< ncclCommCuDevice of every rank's communicator
  std::vector<double> rank_seconds;       //!< per rank: replica set-up + its experiments, wall clock
  std::vector<double> rank_setup_seconds; //!< per rank: building its replica of the evaluators (upload + layout)
  /** per distinct device: the set-up lock of that card -- seconds its users waited for it (summed over the threads
   *  that asked), seconds it was held, acquisitions.  waited / (lanes x wall) is the serialised share of the run. */
  struct LockUse {
    int device;
    double waited_seconds, held_seconds;
    unsigned long long acquisitions;
  };
  std::vector<LockUse> setup_locks;
};

/** The ensemble of sxmc.cpp:44-145 over the GPUs of one node, driven from one process: a host thread per
 *  rank, rank r on devices[r]; every rank builds its own replica of the evaluators from the host tables
 *  (`tables[j]` = signal j's row-major samples, what build_pdfz takes); experiment k runs on rank k mod G, several in
 *  flight per device (lockstep sets or ensemble_concurrent); no data-path collective.  At the end each rank
 *  contributes its experiments' intervals to ONE RCCL all-gather (sxmc_comm_allgather_f32: interval.h:22-27 as
 *  4 floats per parameter, padded to equal blocks), and rank 0's copy gives the medians (sxmc.cpp:126-145,
 *  utils.h:76-90).  `signals`: name, dataset, source, nexpected of every signal (histogram ignored).
 *
 *  Failure: the ranks meet on the host before the collective (Rendezvous).  If any rank failed -- a bad table, an
 *  experiment that threw, an allocation -- NO rank enters the all-gather and the first error is rethrown.  A rank
 *  whose collective does not complete (a peer lost after the rendezvous, exchange_timeout_seconds, an asynchronous
 *  RCCL error) aborts its communicator (ncclCommAbort) instead of waiting: the call returns an error, never hangs.
 *  Locking: set-up, graph recording and tear-down are serialised by ONE lock for the process (MultiGpuOptions::locking;
 *  PER_DEVICE: the lock of their own card only). */
inline MultiGpuEnsemble ensemble_multi_gpu(const std::vector<int>& devices, unsigned nexperiments,
                                           unsigned long long base_seed, std::vector<Source>& sources,
                                           const std::vector<Signal>& signals,
                                           const std::vector<const std::vector<float>*>& tables, int nfields,
                                           std::vector<Systematic>& systematics, std::vector<Observable>& observables,
                                           unsigned nsteps, float burnin_fraction, const MultiGpuOptions& opt) {
  PoolScope pool;   // (see ensemble(): blocks are pooled per device)
  typedef std::chrono::steady_clock Clock;
  const size_t G = devices.size();
  if (G == 0 || tables.size() != signals.size()) throw pdfz::Error("ensemble_multi_gpu: bad arguments");
  size_t P = sources.size();
  for (const Systematic& s : systematics) P += s.npars;
  const size_t per = (nexperiments + G - 1) / G, block = per * P * 4;
  const bool rccl = opt.exchange == MultiGpuOptions::RCCL;
  std::vector<sxmc_comm_t> comms(G, nullptr);
  if (rccl && sxmc_comm_init_all(devices.data(), (int)G, comms.data()) != SXMC_OK) {
    throw pdfz::Error(std::string("ensemble_multi_gpu: RCCL communicators: ") + sxmc_comm_last_error());
  }
  MultiGpuEnsemble out;
  out.nparameters = P;
  out.results.resize(nexperiments);
  out.rank_seconds.assign(G, 0.0);
  out.rank_setup_seconds.assign(G, 0.0);
  if (rccl) {
    out.rccl_devices.assign(G, -1);
    for (size_t r = 0; r < G; r++) {
      int rk = -1, n = 0, dv = -1;
      if (sxmc_comm_query(comms[r], &rk, &n, &dv) == SXMC_OK) {
        out.rccl_devices[r] = dv;
        if (r == 0) out.rccl_nranks = n;
      }
    }
  }
  std::vector<float> rank0((size_t)G * block, std::numeric_limits<float>::quiet_NaN());
  std::vector<std::exception_ptr> errors(G);
  // one lock for the process, or one per card (ranks rehearsed on one card share theirs); key -1 = the process's
  std::map<int, SetupLock> locks;
  RecordingGate gate;   // recording anywhere excludes set-up everywhere (shared by the locks below; see MultiGpuOptions)
  auto lock_key = [&](size_t r) {
    return opt.locking == MultiGpuOptions::PER_DEVICE ? devices[r]
           : opt.locking == MultiGpuOptions::PER_RANK ? 1000 + (int)r : -1;
  };
  for (size_t r = 0; r < G; r++) {
    locks.emplace(std::piecewise_construct, std::forward_as_tuple(lock_key(r)),
                  std::forward_as_tuple(opt.locking == MultiGpuOptions::PROCESS_WIDE ? nullptr : &gate));
  }
  Rendezvous meet(G);
  std::atomic<bool> give_up{false};   // a rank abandoned the exchange: the others stop waiting for it
  std::vector<std::thread> threads;
  for (size_t r = 0; r < G; r++) {
    threads.emplace_back([&, r]() {
      std::vector<Signal> mine;
      float *d_send = nullptr, *d_recv = nullptr;
      sxmc_stream_t strm = nullptr;
      SetupLock& exclusive = locks.find(lock_key(r))->second;   // (find does not modify the map)
      std::vector<float> send(block, std::numeric_limits<float>::quiet_NaN());
      bool ok = true;
      const Clock::time_point t0 = Clock::now();
      try {
        check(sxmc_set_device(devices[r]));
        std::vector<Systematic> sys = systematics;
        std::vector<Observable> obs = observables;
        std::vector<Source> src = sources;
        {
          std::lock_guard<SetupLock> lock(exclusive);
          for (size_t j = 0; j < signals.size(); j++) {
            Signal s = signals[j];
            s.histogram = nullptr;
            s.par_arrays.clear();
            build_pdfz(s, *tables[j], nfields, obs, sys);
            mine.push_back(s);
          }
        }
        out.rank_setup_seconds[r] = std::chrono::duration<double>(Clock::now() - t0).count();
        std::vector<unsigned> ks;
        for (unsigned k = (unsigned)r; k < nexperiments; k += (unsigned)G) ks.push_back(k);
        // per device: experiments in flight either as lockstep sets (one pass over the tables per step and set)
        // or each with its own fill
        std::vector<ExperimentResult> res =
            opt.lockstep_chains >= 2
                ? ensemble_lockstep(ks, base_seed, src, mine, sys, obs, nsteps, burnin_fraction, opt.lockstep_chains,
                                    opt.lockstep_sets, opt.cl, opt.sync_interval,
                                    opt.graph_steps ? opt.graph_steps : 10, devices[r], &exclusive, opt.error_type)
                : ensemble_concurrent(ks, base_seed, src, mine, sys, obs, nsteps, burnin_fraction, opt.nconcurrent,
                                      opt.cl, opt.sync_interval, opt.graph_steps, devices[r], &exclusive, opt.error_type);
        for (size_t i = 0; i < res.size(); i++) {
          out.results[ks[i]] = res[i];
          for (size_t p = 0; p < P && p < res[i].intervals.size(); p++) {
            const Interval& iv = res[i].intervals[p];
            float* at = &send[(i * P + p) * 4];
            at[0] = iv.point_estimate;
            at[1] = iv.lower;
            at[2] = iv.upper;
            at[3] = iv.coverage;
          }
        }
        out.rank_seconds[r] = std::chrono::duration<double>(Clock::now() - t0).count();
        if (rccl) {
          std::lock_guard<SetupLock> lock(exclusive);
          check(sxmc_malloc((void**)&d_send, sizeof(float) * std::max<size_t>(block, 1)));
          check(sxmc_malloc((void**)&d_recv, sizeof(float) * std::max<size_t>(G * block, 1)));
          check(sxmc_stream_create_nonblocking(&strm));
          check(sxmc_memcpy_h2d(d_send, send.data(), sizeof(float) * block));
        }
        if (opt.before_exchange) opt.before_exchange(r);
      } catch (...) {
        errors[r] = std::current_exception();
        ok = false;
      }
      // ---- every rank arrives here, failed or not; the collective is entered by all ranks or by none
      const bool go = meet.arrive(ok);
      if (go) {
        try {
          if (rccl) {
            // the one exchange of the multi-GPU path
            if (sxmc_comm_allgather_f32(comms[r], d_send, d_recv, block, strm) != SXMC_OK) {
              throw pdfz::Error(std::string("all-gather of the intervals: ") + sxmc_comm_last_error());
            }
            // wait WITHOUT blocking inside the runtime: a peer that gave up, an asynchronous RCCL error or the
            // time limit end the wait
            const Clock::time_point w0 = Clock::now();
            for (;;) {
              int done = 0, failed = 0;
              check(sxmc_stream_query(strm, &done));
              if (done) break;
              if (give_up.load()) throw pdfz::Error("all-gather of the intervals: abandoned, another rank failed in it");
              if (sxmc_comm_async_error(comms[r], &failed) == SXMC_OK && failed) {
                throw pdfz::Error(std::string("all-gather of the intervals: ") + sxmc_comm_last_error());
              }
              if (std::chrono::duration<double>(Clock::now() - w0).count() > opt.exchange_timeout_seconds) {
                throw pdfz::Error("all-gather of the intervals: not complete after " +
                                  std::to_string(opt.exchange_timeout_seconds) + " s (rank " + std::to_string(r) + ")");
              }
              std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            if (r == 0) check(sxmc_memcpy_d2h(rank0.data(), d_recv, sizeof(float) * G * block));
          } else {
            std::copy(send.begin(), send.end(), rank0.begin() + (std::ptrdiff_t)(r * block));   // (disjoint blocks)
          }
        } catch (...) {
          errors[r] = std::current_exception();
          give_up.store(true);
          // end the pending collective on this rank's device BEFORE its stream and buffers are released (releasing
          // them would wait for a kernel that waits for a peer)
          if (comms[r]) {
            sxmc_comm_abort(comms[r]);
            comms[r] = nullptr;
          }
        }
      }
      std::lock_guard<SetupLock> lock(exclusive);
      for (Signal& s : mine) delete s.histogram;
      if (d_send) sxmc_free(d_send);
      if (d_recv) sxmc_free(d_recv);
      if (strm) sxmc_stream_destroy(strm);
    });
  }
  for (std::thread& th : threads) th.join();
  for (sxmc_comm_t c : comms)
    if (c) sxmc_comm_destroy(c);
  for (auto& kv : locks) {
    out.setup_locks.push_back({kv.first, kv.second.waited_seconds(), kv.second.held_seconds(), kv.second.count()});
  }
  for (size_t r = 0; r < G; r++) {
    if (!errors[r]) continue;
    // the first failing rank's error, said which rank
    try {
      std::rethrow_exception(errors[r]);
    } catch (const pdfz::Error& e) {
      throw pdfz::Error("ensemble_multi_gpu: rank " + std::to_string(r) + " (device " + std::to_string(devices[r]) +
                        "): " + e.msg);
    } catch (const std::exception& e) {
      throw pdfz::Error("ensemble_multi_gpu: rank " + std::to_string(r) + " (device " + std::to_string(devices[r]) +
                        "): " + e.what());
    } catch (...) {   // (a callback of the caller's may throw anything)
      throw pdfz::Error("ensemble_multi_gpu: rank " + std::to_string(r) + " (device " + std::to_string(devices[r]) +
                        "): an exception that is neither a pdfz::Error nor a std::exception");
    }
  }
  // rank r's block holds its experiments r, r + G, ... in that order
  out.gathered.assign((size_t)nexperiments * P * 4, 0.0f);
  for (unsigned k = 0; k < nexperiments; k++) {
    const size_t r = k % G, i = k / G;
    std::copy(rank0.begin() + (std::ptrdiff_t)(r * block + i * P * 4),
              rank0.begin() + (std::ptrdiff_t)(r * block + (i + 1) * P * 4),
              out.gathered.begin() + (std::ptrdiff_t)((size_t)k * P * 4));
  }
  for (size_t p = 0; p < P; p++) {
    std::vector<float> ups;
    for (unsigned k = 0; k < nexperiments; k++) ups.push_back(out.gathered[((size_t)k * P + p) * 4 + 2]);
    out.median_upper.push_back(ups.empty() ? 0.0f : median(ups));
  }
  return out;
}

/** The same with the options spelled out as arguments (round 2's signature). */
inline MultiGpuEnsemble ensemble_multi_gpu(const std::vector<int>& devices, unsigned nexperiments,
                                           unsigned long long base_seed, std::vector<Source>& sources,
                                           const std::vector<Signal>& signals,
                                           const std::vector<const std::vector<float>*>& tables, int nfields,
                                           std::vector<Systematic>& systematics, std::vector<Observable>& observables,
                                           unsigned nsteps, float burnin_fraction, unsigned nconcurrent,
                                           float cl = 0.9f, unsigned sync_interval = 10000, unsigned graph_steps = 0,
                                           unsigned lockstep_chains = 4, unsigned lockstep_sets = 2) {
  MultiGpuOptions opt;
  opt.cl = cl;
  opt.sync_interval = sync_interval;
  opt.graph_steps = graph_steps;
  opt.nconcurrent = nconcurrent;
  opt.lockstep_chains = lockstep_chains;
  opt.lockstep_sets = lockstep_sets;
  return ensemble_multi_gpu(devices, nexperiments, base_seed, sources, signals, tables, nfields, systematics,
                            observables, nsteps, burnin_fraction, opt);
}

}  // namespace sxmc
