// bench_cpp.cpp -- the C++ host layer on BASELINE config 3's shape, no Python in the process: builds the 12 signals
// through sxmc::build_pdfz, walks one chain with sxmc::MCMC (the caller of the hot path, mcmc.cpp:143-387) and
// prints MCMC steps (= NLL evaluations) per second; optionally whole fake experiments (sxmc.cpp:59-145) as lockstep
// sets on one GPU, or sharded over the GPUs of the node with sxmc::ensemble_multi_gpu (BASELINE config 4: a host
// thread per GPU, ONE RCCL all-gather of the intervals).  One JSON line per leg on stdout.
//
//   bench_cpp [--scale 1.0] [--steps 2000] [--graph-steps 10] [--no-walk] [--reference-form | --sequential | --lookahead]
//             [--burnin 0.1] [--sync-interval 10000] [--walks lookahead=100000,sequential=100000,reference=3000]
//             [--experiments 0] [--exp-steps 2000] [--chains 4] [--sets 2] [--c4 8x100000]
//             [--devices G | --device-list 0,0] [--host-staging] [--per-device-locks] [--config fit.json] [--output-dir d]
//
// --reference-form: the walk issues mcmc.cpp:264-271 + 314-348 as written (S x EvalAsync, S x EvalFinished,
//   nll_event_chunks, finish_nll_jump_pick_combo; lookup table materialised, legacy default stream, no graph): what an
//   unchanged mcmc.cpp gets from this library.  --sequential: the batched step, one evaluation per step.  --lookahead:
//   the look-ahead pass (two evaluations per pass over the tables).  Neither: the walk decides ("auto": look-ahead where
//   the launch plan streams float columns, sequential where it streams codes -- MCMC::lookahead_auto).
// --c4 NxSTEPS: a SECOND ensemble leg on the current device after the first: N whole fake experiments of STEPS steps
//   each, N in flight with a fill each (ensemble_concurrent) -- BASELINE config 4's per-GPU share as written: "ensemble
//   of config (3)" = 1e5 steps per experiment (sxmc.cpp:59-145 with fit.nsteps).
// --devices G: the ensemble leg runs on devices 0..G-1 through ensemble_multi_gpu (needs --experiments).
// --host-staging: the blocks meet through host memory instead of RCCL (rehearsal of G ranks on fewer cards).
// --config: signals, observables, systematics, rates and sample tables come from a fit configuration
//   (sxmc::load_config, the reference's JSON schema: config.cpp:19-297) instead of the synthetic C3 generator.
// --output-dir: every experiment's chain is written to <d>/<output_prefix>_<k>.npz (one float32 column per parameter
//   + "likelihood": the "ls" ntuple of sxmc.cpp:130-141, readable by numpy.load and sxmc_amd/io.py).
// Synthetic inputs as SURVEY.md 8(d) C3 describes them (not bit-identical to bench.py's generator).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>

#include "../../sxmc_amd/include/sxmc/config.h"
#include "../../sxmc_amd/include/sxmc/ensemble.h"

namespace {
struct Options {
  double scale = 1.0;
  unsigned nsteps = 2000, graph_steps = 10;
  bool walk = true;
  bool reference_form = false;   // the walk issues the reference's own call sequence (mcmc.cpp:264-271, 314-348)
  bool sequential = false;       // ... or the batched step, one evaluation per step (no look-ahead)
  bool lookahead = false;        // ... or the look-ahead pass wherever the library offers it
  // --walks name=steps,...: several timed walks in ONE process over the same tables (names: auto, lookahead,
  // sequential, reference), each preceded by a short warm-up walk of its own form
  std::vector<std::pair<std::string, unsigned>> walks;
  float burnin = 0.1f;
  unsigned sync_interval = 10000;
  unsigned nexp = 0, esteps = 2000, L = 4, S = 2;
  unsigned c4_nexp = 0, c4_steps = 0;   // --c4 NxSTEPS: config 4's per-GPU share as written (a second ensemble leg)
  std::vector<int> devices;
  bool host_staging = false;
  bool per_device_locks = false;   // ensemble_multi_gpu: one set-up lock per card instead of one for the process
  std::string config;
  std::string output_dir;   // every experiment's chain as <output_dir>/<prefix>_<k>.npz (sxmc.cpp:130-141)
};

std::vector<int> parse_list(const char* s) {
  std::vector<int> v;
  for (const char* p = s; *p;) {
    v.push_back(std::atoi(p));
    while (*p && *p != ',') p++;
    if (*p == ',') p++;
  }
  return v;
}

Options parse(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
      return argv[++i];
    };
    if (a == "--scale") o.scale = std::atof(next());
    else if (a == "--steps") o.nsteps = (unsigned)std::atoi(next());
    else if (a == "--graph-steps") o.graph_steps = (unsigned)std::atoi(next());
    else if (a == "--no-walk") o.walk = false;
    else if (a == "--walks") {
      std::string v = next();
      size_t pos = 0;
      while (pos < v.size()) {
        const size_t comma = v.find(',', pos), end = comma == std::string::npos ? v.size() : comma;
        const std::string item = v.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq == std::string::npos) throw std::runtime_error("--walks wants name=steps[,name=steps...]");
        o.walks.emplace_back(item.substr(0, eq), (unsigned)std::atoi(item.c_str() + eq + 1));
        pos = end + 1;
      }
    }
    else if (a == "--reference-form") o.reference_form = true;
    else if (a == "--sequential") o.sequential = true;
    else if (a == "--lookahead") o.lookahead = true;
    else if (a == "--burnin") o.burnin = (float)std::atof(next());
    else if (a == "--sync-interval") o.sync_interval = (unsigned)std::atoi(next());
    else if (a == "--experiments") o.nexp = (unsigned)std::atoi(next());
    else if (a == "--exp-steps") o.esteps = (unsigned)std::atoi(next());
    else if (a == "--chains") o.L = (unsigned)std::atoi(next());
    else if (a == "--sets") o.S = (unsigned)std::atoi(next());
    else if (a == "--c4") {
      const std::string v = next();
      const size_t x = v.find('x');
      if (x == std::string::npos) throw std::runtime_error("--c4 wants <experiments>x<steps>");
      o.c4_nexp = (unsigned)std::atoi(v.c_str());
      o.c4_steps = (unsigned)std::atoi(v.c_str() + x + 1);
    }
    else if (a == "--devices") {
      const int g = std::atoi(next());
      for (int d = 0; d < g; d++) o.devices.push_back(d);
    } else if (a == "--device-list") o.devices = parse_list(next());
    else if (a == "--host-staging") o.host_staging = true;
    else if (a == "--per-device-locks") o.per_device_locks = true;
    else if (a == "--config") o.config = next();
    else if (a == "--output-dir") o.output_dir = next();
    else throw std::runtime_error("unknown argument " + a);
  }
  return o;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

static int run(int argc, char** argv);
// where the experiments' host time went, summed over the experiments (they overlap: lanes run side by side)
static std::string phases_json(const std::vector<sxmc::ExperimentResult>& res) {
  sxmc::ExperimentResult::Phases p;
  for (const sxmc::ExperimentResult& r : res) {
    p.data += r.phases.data;
    p.construct += r.phases.construct;
    p.walk_setup += r.phases.walk_setup;
    p.steps += r.phases.steps;
    p.walk_teardown += r.phases.walk_teardown;
    p.destroy += r.phases.destroy;
    p.intervals += r.phases.intervals;
  }
  char b[400];
  std::snprintf(b, sizeof b, "{\"data\": %.4f, \"construct\": %.4f, \"walk_setup\": %.4f, \"steps\": %.4f, "
                "\"walk_teardown\": %.4f, \"destroy\": %.4f, \"intervals\": %.4f}", p.data, p.construct, p.walk_setup,
                p.steps, p.walk_teardown, p.destroy, p.intervals);
  return b;
}

int main(int argc, char** argv) {
  try {
    return run(argc, argv);
  } catch (const pdfz::Error& e) {
    std::fprintf(stderr, "pdfz::Error: %s\n", e.msg.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
  }
  return 1;
}

static int run(int argc, char** argv) {
  const Options o = parse(argc, argv);
  std::vector<sxmc::Observable> observables;
  std::vector<sxmc::Systematic> systematics;
  std::vector<sxmc::Source> sources;
  std::vector<sxmc::Signal> signals;          // name, source, nexpected (histograms are built below)
  std::vector<std::vector<float>> tabs;       // row-major host tables, one per signal
  std::vector<float> data;                    // data events for the single walk: rows of D + 1 floats
  size_t F = 5;
  unsigned long long base_seed = 77;
  float burnin = 0.1f, cl = 0.9f;
  sxmc::ErrorType error_type = sxmc::ERROR_CONTOUR;
  std::string output_prefix = "lspace";
  Options run_opt = o;

  if (!o.config.empty()) {
    // ---- inputs from a fit configuration: the reference's control file + ROOT-free sample tables
    sxmc::FitConfig fc = sxmc::load_config(o.config, /*load_tables=*/false);
    if (fc.samples.empty()) fc = sxmc::load_config(o.config);
    if (!fc.samples.empty()) {
      // fit.samples (sxmc.cpp:84-94): a saved chain replaces the walk -- the intervals are taken from it and that is all
      std::vector<float> m;
      std::vector<std::string> fields;
      sxmc::read_table(sxmc::detail::join_path(fc.base_dir, fc.samples), m, fields);
      const sxmc::Chain chain = sxmc::chain_from_table(m, fields);
      const std::vector<sxmc::Interval> iv = sxmc::extract_intervals(
          chain, fc.confidence, fc.error_type == "projection" ? sxmc::ERROR_PROJECTION : sxmc::ERROR_CONTOUR);
      std::printf("{\"driver\": \"fit.samples (C++)\", \"samples\": \"%s\", \"rows\": %zu, \"error_type\": \"%s\", "
                  "\"intervals\": {", fc.samples.c_str(), chain.nrows(), fc.error_type.c_str());
      for (size_t p = 0; p < iv.size(); p++) {
        std::printf("%s\"%s\": [%.9g, %.9g, %.9g]", p ? ", " : "", chain.names[p].c_str(), (double)iv[p].point_estimate,
                    (double)iv[p].lower, (double)iv[p].upper);
      }
      std::printf("}}\n");
      // what sxmc.cpp:100-101 prints for the likelihood space: best fit + correlation matrix (stderr: stdout is JSON)
      sxmc::print_best_fit(std::cerr, chain, iv);
      sxmc::print_correlations(std::cerr, chain);
      return 0;
    }
    if (!sxmc::same_systematics_everywhere(fc)) {
      throw std::runtime_error("the batched drivers (one launch for all signals) need every signal to list every "
                               "systematic of the fit, in the same order");
    }
    if (!fc.data.empty()) {
      // configured data sets: experiment i fits file i of every data set instead of a fake one (sxmc.cpp:71-80)
      auto shared = std::make_shared<sxmc::FitConfig>();
      shared->data = fc.data;
      for (unsigned i = 0; i < fc.nexperiments; i++) {   // (every experiment has its file: fail now, not mid-run)
        std::vector<float> rows;
        sxmc::experiment_data(*shared, i, rows);
      }
      sxmc::data_source() = [shared](unsigned k, std::vector<float>& rows) { return sxmc::experiment_data(*shared, k, rows); };
    }
    observables = fc.observables;
    systematics = fc.systematics;
    sources = fc.sources;
    signals = fc.signals;
    const size_t rows_loaded = fc.rows_total();
    tabs = std::move(fc.tables);
    F = fc.nfields;
    burnin = fc.burnin_fraction;
    cl = fc.confidence;
    error_type = fc.error_type == "projection" ? sxmc::ERROR_PROJECTION : sxmc::ERROR_CONTOUR;
    output_prefix = fc.output_prefix;
    if (fc.seed) base_seed = fc.seed;
    if (run_opt.nexp == 0) run_opt.nexp = fc.nexperiments;
    run_opt.esteps = fc.nsteps;
    run_opt.walk = false;
    std::printf("{\"driver\": \"sxmc::load_config (C++)\", \"config\": \"%s\", \"signals\": %zu, \"observables\": %zu, "
                "\"systematics\": %zu, \"nfields\": %zu, \"rows_total\": %zu, \"experiments\": %u, \"steps\": %u}\n",
                o.config.c_str(), signals.size(), observables.size(), systematics.size(), F, rows_loaded,
                run_opt.nexp, run_opt.esteps);
  } else {
    const size_t S = 12, E = 100000;
    const size_t per_signal = (size_t)(1e8 * o.scale) / S;
    observables.resize(3);
    const float lo[3] = {0, 0, -1}, hi[3] = {10, 6, 1};
    for (size_t k = 0; k < 3; k++) {
      observables[k].field_index = k;
      observables[k].bins = 20;
      observables[k].lower = lo[k];
      observables[k].upper = hi[k];
    }
    systematics.resize(3);
    systematics[0].name = "r_shift"; systematics[0].type = pdfz::Systematic::SHIFT;
    systematics[0].observable_field_index = 1; systematics[0].sigmas = {0.05};
    systematics[1].name = "e_scale"; systematics[1].type = pdfz::Systematic::SCALE;
    systematics[1].observable_field_index = 0; systematics[1].sigmas = {0.01};
    systematics[2].name = "e_res"; systematics[2].type = pdfz::Systematic::RESOLUTION_SCALE;
    systematics[2].observable_field_index = 0; systematics[2].truth_field_index = 3; systematics[2].sigmas = {0.05};
    for (size_t q = 0; q < 3; q++) {
      systematics[q].means = {0.0};
      systematics[q].pidx = {(short)q};
    }
    // the tables are generated by one host thread per signal (10^8 samples are ~4 10^8 random draws)
    tabs.resize(S);
    {
      std::vector<std::thread> gen;
      for (size_t j = 0; j < S; j++) {
        gen.emplace_back([&, j]() {
          std::mt19937_64 rng(3 + 1000 * j);
          std::normal_distribution<float> gauss(0.0f, 1.0f);
          std::uniform_real_distribution<float> uni(0.0f, 1.0f);
          std::vector<float>& tab = tabs[j];
          tab.resize(per_signal * F);
          for (size_t i = 0; i < per_signal; i++) {
            const float e_true = 2.0f + 0.5f * j + 1.2f * gauss(rng);
            tab[i * F + 0] = e_true + 0.3f * gauss(rng);
            tab[i * F + 1] = 6.0f * std::cbrt(uni(rng));
            tab[i * F + 2] = 2.0f * uni(rng) - 1.0f;
            tab[i * F + 3] = e_true;
            tab[i * F + 4] = 0.0f;
          }
        });
      }
      for (std::thread& t : gen) t.join();
    }
    std::mt19937_64 rng(3);
    for (size_t j = 0; j < S; j++) {
      sxmc::Signal sig;
      sig.name = "signal" + std::to_string(j);
      sig.source = sxmc::Source("source" + std::to_string(j), j, 1.0f, 0.0f, false);
      sig.nexpected = (double)E / S;
      signals.push_back(sig);
      sources.push_back(sig.source);
      for (size_t e = 0; e < E / S; e++) {  // data events: samples of the mixture
        const size_t i = rng() % per_signal;
        data.push_back(tabs[j][i * F + 0]);
        data.push_back(tabs[j][i * F + 1]);
        data.push_back(tabs[j][i * F + 2]);
        data.push_back(0.0f);
      }
    }
  }
  const Options& opt = run_opt;
  const bool multi = !opt.devices.empty();
  if (!opt.output_dir.empty()) {
    const std::string stem = opt.output_dir + "/" + output_prefix + "_";
    sxmc::chain_sink() = [stem](unsigned k, const sxmc::Chain& chain) {
      sxmc::write_chain_npz(stem + std::to_string(k) + ".npz", chain.names, chain.rows);
    };
    // ... and beside each chain what sxmc.cpp:100-101 prints for the experiment: best fit + correlation matrix
    sxmc::report_sink() = [stem](unsigned k, const std::string& text) {
      std::ofstream f(stem + std::to_string(k) + ".txt");
      f << text;
    };
  }
  size_t rows_total = 0;
  for (const std::vector<float>& t : tabs) rows_total += t.size() / F;

  // ---- evaluators on the current device (the single walk and the one-GPU ensemble leg)
  if (opt.walk || ((opt.nexp > 0 || opt.c4_nexp > 0) && !multi)) {
    for (size_t j = 0; j < signals.size(); j++) sxmc::build_pdfz(signals[j], tabs[j], (int)F, observables, systematics);
  }

  if (opt.walk) {
    std::vector<std::pair<std::string, unsigned>> walks = opt.walks;
    if (walks.empty()) {
      walks.emplace_back(opt.reference_form ? "reference" : opt.sequential ? "sequential" : opt.lookahead ? "lookahead" : "auto",
                         opt.nsteps);
    }
    for (const auto& wk : walks) {
      const bool reference_form = wk.first == "reference", lookahead = wk.first == "lookahead", automatic = wk.first == "auto";
      if (!reference_form && !lookahead && !automatic && wk.first != "sequential") throw std::runtime_error("unknown walk " + wk.first);
      const unsigned nsteps = wk.second;
      // the reference's sequence launches its NLL kernels on the legacy default stream (mcmc.cpp:314-348), which is
      // what the evaluators' own streams order with; the batched forms walk on a stream of their own
      sxmc_stream_t strm = nullptr;
      if (!reference_form) sxmc::check(sxmc_stream_create_nonblocking(&strm));
      for (int pass = 0; pass < 2; pass++) {  // pass 0 warms up (clocks, launch plan); pass 1 is timed
        sxmc::MCMC mcmc(sources, signals, systematics, observables, 1234 + pass, strm);
        mcmc.graph_steps = reference_form ? 0 : opt.graph_steps;
        mcmc.lookahead = lookahead;
        mcmc.lookahead_auto = automatic;
        mcmc.reference_form = reference_form;
        unsigned long long l0 = 0, e0 = 0, l1 = 0, e1 = 0;
        sxmc::check(sxmc_deferred_eval_stats(&l0, &e0));
        const auto t0 = std::chrono::steady_clock::now();
        sxmc::Chain chain = mcmc(data, pass == 0 ? std::min(nsteps, 500u) : nsteps, opt.burnin, false, opt.sync_interval);
        const double sec = seconds_since(t0);
        sxmc::check(sxmc_deferred_eval_stats(&l1, &e1));
        if (pass == 1) {
          std::printf("{\"driver\": \"%s\", \"walk\": \"%s\", \"nsamples_total\": %zu, \"nsignals\": %zu, "
                      "\"nevents\": %zu, "
                      "\"steps\": %u, \"steps_per_graph\": %u, \"seconds\": %.4f, \"steps_per_sec\": %.1f, "
                      "\"setup_seconds\": %.4f, \"stepping_seconds\": %.4f, \"steps_per_sec_stepping\": %.1f, "
                      "\"burnin_fraction\": %.3f, \"sync_interval\": %u, "
                      "\"accepted\": %zu, \"rows_kept\": %zu, \"lookahead\": %s, \"passes\": %zu, "
                      "\"deferred_launches\": %llu, \"deferred_evaluations\": %llu, \"plan\": \"%s\"}\n",
                      reference_form ? "sxmc::MCMC, the reference's call sequence (C++)" : "sxmc::MCMC (C++)",
                      wk.first.c_str(), rows_total, signals.size(), data.size() / (observables.size() + 1), nsteps,
                      mcmc.graph_steps, sec, nsteps / sec, chain.setup_seconds, chain.steps_seconds,
                      nsteps / chain.steps_seconds, (double)opt.burnin, opt.sync_interval, chain.accepted, chain.nrows(),
                      mcmc.LookaheadPasses() ? "true" : "false", mcmc.LookaheadPasses(), l1 - l0, e1 - e0,
                      [&] {
                        std::string p = mcmc.LaunchPlan();
                        for (char& ch : p) ch = (ch == '\n' || ch == '"') ? ' ' : ch;
                        return p;
                      }().c_str());
          std::fflush(stdout);
        }
      }
      if (strm) sxmc_stream_destroy(strm);
    }
  }

  // ---- ensemble leg: whole fake experiments (fake data drawn on the device, walk with burn-in re-tuning, contour
  // intervals), in lockstep sets of L chains (S sets in flight per GPU); then, with --c4, config 4's per-GPU share
  auto ensemble_leg = [&](unsigned nexp, unsigned esteps, unsigned L, unsigned S, const char* tag) {
    std::vector<unsigned> ks;
    for (unsigned k = 0; k < nexp; k++) ks.push_back(k);
    if (opt.config.empty())
      for (sxmc::Signal& sg : signals) sg.nexpected = 8000.0;   // ~1e5 events per fake data set
    for (int pass = 0; pass < 2; pass++) {   // pass 0 (one round) compiles the lockstep kernel and warms up
      std::vector<unsigned> part(ks.begin(),
                                 pass == 0 ? ks.begin() + std::min<size_t>(ks.size(), std::max(1u, L) * S) : ks.end());
      sxmc::SetupLock lock;
      const auto t0 = std::chrono::steady_clock::now();
      // L < 2: a fill per chain, S experiments in flight (ensemble_concurrent) -- the faster form where the fill streams
      // codes; L = 2..4: lockstep sets, one pass over the tables per step for the chains of a set
      const unsigned esteps_now = pass == 0 ? std::min(300u, esteps) : esteps;
      // (the jump buffer is flushed every sync_interval steps, mcmc.cpp:351-377: the walk's own setting, capped by its length)
      const unsigned sync = std::min(esteps, std::max(opt.sync_interval, 1u));
      std::vector<sxmc::ExperimentResult> res =
          L < 2 ? sxmc::ensemble_concurrent(part, base_seed, sources, signals, systematics, observables, esteps_now,
                                            burnin, std::max(1u, S), cl, sync, opt.graph_steps, -1, &lock, error_type)
                : sxmc::ensemble_lockstep(part, base_seed, sources, signals, systematics, observables, esteps_now,
                                          burnin, L, S, cl, sync, opt.graph_steps, -1, &lock, error_type);
      const double sec = seconds_since(t0);
      if (pass == 1) {
        double steps_sec = 0;      // (the walks' stepping phases, summed over the experiments: they overlap)
        for (const sxmc::ExperimentResult& r : res) steps_sec += r.phases.steps;
        std::printf("{\"driver\": \"sxmc::%s (C++)\", \"leg\": \"%s\", \"experiments\": %zu, \"steps_each\": %u, "
                    "\"chains_per_fill\": %u, \"sets\": %u, \"sync_interval\": %u, \"seconds\": %.4f, "
                    "\"experiments_per_sec\": %.5f, "
                    "\"steps_per_sec_inside\": %.1f, \"nevents_first\": %zu, \"gathered_shape\": [%zu, %zu, 4], "
                    "\"phase_seconds_summed_over_experiments\": %s, "
                    "\"setup_lock\": {\"waited_seconds_summed_over_lanes\": %.4f, \"held_seconds\": %.4f, "
                    "\"acquisitions\": %llu, \"lanes\": %u}}\n",
                    L < 2 ? "ensemble_concurrent" : "ensemble_lockstep", tag, res.size(), esteps, L, S, sync, sec,
                    res.size() / sec, res.size() * (double)esteps / sec,
                    res.empty() ? (size_t)0 : res[0].nevents, res.size(), res.empty() ? (size_t)0 : res[0].intervals.size(),
                    phases_json(res).c_str(), lock.waited_seconds(), lock.held_seconds(), lock.count(),
                    std::max(1u, L) * S);
        std::fflush(stdout);
      }
    }
  };
  if (opt.nexp > 0 && !multi) ensemble_leg(opt.nexp, opt.esteps, opt.L, opt.S, "ensemble");
  if (opt.c4_nexp > 0 && !multi) ensemble_leg(opt.c4_nexp, opt.c4_steps, 1, opt.c4_nexp, "c4_per_gpu");
  for (sxmc::Signal& s : signals) {
    delete s.histogram;
    s.histogram = nullptr;
  }

  // ---- the same experiments sharded over the GPUs of the node: a host thread per device, its own replica of the
  // evaluators, experiment k on device k mod G, ONE RCCL all-gather of the intervals at the end
  if (opt.nexp > 0 && multi) {
    if (opt.config.empty())
      for (sxmc::Signal& sg : signals) sg.nexpected = 8000.0;
    std::vector<const std::vector<float>*> tp;
    for (const std::vector<float>& t : tabs) tp.push_back(&t);
    sxmc::MultiGpuOptions mo;
    mo.cl = cl;
    mo.sync_interval = opt.esteps;
    mo.graph_steps = opt.graph_steps;
    mo.lockstep_chains = opt.L;      // (< 2: ensemble_concurrent with --sets experiments in flight per device)
    mo.lockstep_sets = opt.S;
    mo.nconcurrent = opt.L < 2 ? std::max(1u, opt.S) : std::max(1u, opt.L * opt.S);
    mo.error_type = error_type;
    if (opt.host_staging) mo.exchange = sxmc::MultiGpuOptions::HOST_STAGING;
    if (opt.per_device_locks) mo.locking = sxmc::MultiGpuOptions::PER_DEVICE;
    const size_t G = opt.devices.size();
    for (int pass = 0; pass < 2; pass++) {   // pass 0: one round per device (kernel compilation, clocks)
      const unsigned n = pass == 0 ? (unsigned)std::min<size_t>(opt.nexp, G * std::max(1u, opt.L * opt.S)) : opt.nexp;
      const auto t0 = std::chrono::steady_clock::now();
      sxmc::MultiGpuEnsemble mg =
          sxmc::ensemble_multi_gpu(opt.devices, n, base_seed, sources, signals, tp, (int)F, systematics, observables,
                                   pass == 0 ? std::min(300u, opt.esteps) : opt.esteps, burnin, mo);
      const double sec = seconds_since(t0);
      if (pass == 0) continue;
      double setup_max = 0, rank_max = 0;
      for (double s : mg.rank_setup_seconds) setup_max = std::max(setup_max, s);
      for (double s : mg.rank_seconds) rank_max = std::max(rank_max, s);
      const double inside = std::max(rank_max - setup_max, 1e-9);
      std::string locks = "[", devs = "[", rdevs = "[";
      for (size_t i = 0; i < mg.setup_locks.size(); i++) {
        char b[256];
        std::snprintf(b, sizeof b, "%s{\"device\": %d, \"waited_seconds_summed_over_lanes\": %.4f, \"held_seconds\": %.4f, "
                      "\"acquisitions\": %llu}", i ? ", " : "", mg.setup_locks[i].device, mg.setup_locks[i].waited_seconds,
                      mg.setup_locks[i].held_seconds, mg.setup_locks[i].acquisitions);
        locks += b;
      }
      std::string nev = "[";   // events each experiment fitted
      for (size_t i = 0; i < mg.results.size(); i++) nev += (i ? ", " : "") + std::to_string(mg.results[i].nevents);
      for (size_t i = 0; i < G; i++) devs += (i ? ", " : "") + std::to_string(opt.devices[i]);
      for (size_t i = 0; i < mg.rccl_devices.size(); i++) rdevs += (i ? ", " : "") + std::to_string(mg.rccl_devices[i]);
      std::printf("{\"driver\": \"sxmc::ensemble_multi_gpu (C++)\", \"ranks\": %zu, \"devices\": %s], "
                  "\"exchange\": \"%s\", \"rccl_nranks\": %d, \"rccl_devices\": %s], \"experiments\": %u, "
                  "\"steps_each\": %u, \"chains_per_fill\": %u, \"sets\": %u, \"seconds\": %.4f, "
                  "\"replica_setup_seconds_max\": %.4f, \"experiments_per_sec\": %.4f, "
                  "\"experiments_per_sec_after_setup\": %.4f, \"steps_per_sec_inside\": %.1f, "
                  "\"median_upper_limit_source0\": %.6g, \"gathered_floats\": %zu, \"nevents\": %s], "
                  "\"data\": \"%s\", \"phase_seconds_summed_over_experiments\": %s, \"locking\": \"%s\", "
                  "\"setup_locks\": %s]}\n",
                  G, devs.c_str(), opt.host_staging ? "host staging (rehearsal)" : "ncclAllGather (RCCL)",
                  mg.rccl_nranks, rdevs.c_str(), n, opt.esteps, opt.L, opt.S, sec, setup_max, n / sec, n / inside,
                  n * (double)opt.esteps / inside, mg.median_upper.empty() ? 0.0 : (double)mg.median_upper[0],
                  mg.gathered.size(), nev.c_str(), sxmc::data_source() ? "configured data sets" : "fake",
                  phases_json(mg.results).c_str(), opt.per_device_locks ? "one lock per card" : "one lock for the process",
                  locks.c_str());
      std::fflush(stdout);
    }
  }
  return 0;
}
