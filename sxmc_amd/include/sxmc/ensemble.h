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
#include <exception>
#include <memory>
#include <mutex>
#include <random>
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

/** Contour::get_interval for every parameter of a chain (contour.cpp:30-69, likelihood.cpp:90-102). */
inline std::vector<Interval> contour_intervals(const Chain& chain, float cl = 0.9f) {
  const size_t ncol = chain.names.size(), P = ncol - 1, n = chain.nrows();
  float lmin = chain.at(0, P);
  for (size_t r = 1; r < n; r++) lmin = std::min(lmin, chain.at(r, P));
  const float delta = 0.5 * chisquare_quantile_1dof(cl);
  std::vector<size_t> contour;
  for (size_t r = 0; r < n; r++)
    if (chain.at(r, P) - lmin < delta) contour.push_back(r);
  // points near the maximum-likelihood point: widen until at least one is found
  std::vector<size_t> near;
  float dnll = 0.13f;
  do {
    near.clear();
    for (size_t r : contour)
      if (chain.at(r, P) - lmin < dnll) near.push_back(r);
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
    std::vector<unsigned> bins = dynamic_cast<pdfz::EvalHist*>(s.histogram)->GetBins();
    random_sample(rng, bins, observables, observed, s.dataset, events);
    if (observed_out) observed_out->push_back((unsigned)observed);
  }
  return events;
}

struct ExperimentResult {
  unsigned index = 0;
  std::vector<Interval> intervals;  //!< one per parameter
  size_t accepted = 0;
  size_t nevents = 0;
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

/** One iteration of sxmc.cpp:59-145: fake data -> MCMC -> intervals, on `stream` (null: default). */
inline ExperimentResult run_experiment(unsigned k, unsigned long long base_seed, std::vector<Source>& sources,
                                       std::vector<Signal>& signals, std::vector<Systematic>& systematics,
                                       std::vector<Observable>& observables, unsigned nsteps, float burnin_fraction,
                                       float cl, unsigned sync_interval, unsigned graph_steps = 0,
                                       sxmc_stream_t stream = nullptr, std::mutex* exclusive = nullptr) {
  const unsigned long long x = experiment_seed(base_seed, k);
  std::mt19937_64 rng(x);
  // `exclusive` (one chain per host thread): held over everything that allocates, copies through the
  // legacy stream or synchronises the device -- see MCMC::exclusive
  std::unique_lock<std::mutex> lock;
  if (exclusive) lock = std::unique_lock<std::mutex>(*exclusive);
  std::vector<float> data = make_fake_dataset(rng, signals, systematics, observables, true);
  std::unique_ptr<MCMC> mcmc(new MCMC(sources, signals, systematics, observables, x, stream));
  mcmc->graph_steps = graph_steps;
  mcmc->exclusive = exclusive;
  if (exclusive) lock.unlock();   // the walk takes it itself
  Chain chain = (*mcmc)(data, nsteps, burnin_fraction, false, sync_interval);
  if (exclusive) lock.lock();
  mcmc.reset();
  if (exclusive) lock.unlock();
  ExperimentResult r;
  r.index = k;
  r.intervals = contour_intervals(chain, cl);
  r.accepted = chain.accepted;
  r.nevents = data.size() / (observables.size() + 1);
  return r;
}

inline std::vector<ExperimentResult> ensemble(const std::vector<unsigned>& experiments, unsigned long long base_seed,
                                              std::vector<Source>& sources, std::vector<Signal>& signals,
                                              std::vector<Systematic>& systematics,
                                              std::vector<Observable>& observables, unsigned nsteps,
                                              float burnin_fraction, float cl = 0.9f, unsigned sync_interval = 10000,
                                              unsigned graph_steps = 0) {
  std::vector<ExperimentResult> out;
  for (unsigned k : experiments) {
    out.push_back(run_experiment(k, base_seed, sources, signals, systematics, observables, nsteps, burnin_fraction,
                                 cl, sync_interval, graph_steps));
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
                                                         unsigned sync_interval = 10000, unsigned graph_steps = 0) {
  const size_t lanes = std::max<size_t>(1, std::min<size_t>(nconcurrent, experiments.size()));
  std::vector<ExperimentResult> out(experiments.size());
  std::vector<std::exception_ptr> errors(lanes);
  std::mutex exclusive;  // set-up, graph recording and tear-down of the lanes, one at a time
  std::vector<std::thread> threads;
  for (size_t t = 0; t < lanes; t++) {
    threads.emplace_back([&, t]() {
      sxmc_stream_t strm = nullptr;
      std::vector<Signal> mine;
      try {
        {
          std::lock_guard<std::mutex> lock(exclusive);
          check(sxmc_stream_create_nonblocking(&strm));
          transfer_stream() = strm;
          for (const Signal& s : signals) mine.push_back(share_pdfz(s));
        }
        std::vector<Source> src = sources;
        std::vector<Systematic> sys = systematics;
        std::vector<Observable> obs = observables;
        for (size_t i = t; i < experiments.size(); i += lanes) {
          out[i] = run_experiment(experiments[i], base_seed, src, mine, sys, obs, nsteps, burnin_fraction, cl,
                                  sync_interval, graph_steps, strm, &exclusive);
        }
      } catch (...) {
        errors[t] = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lock(exclusive);
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

}  // namespace sxmc
