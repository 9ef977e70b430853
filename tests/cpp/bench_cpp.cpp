// bench_cpp.cpp -- the C++ host layer on BASELINE config 3's shape: builds the 12 signals through
// sxmc::build_pdfz, walks one chain with sxmc::MCMC (the caller of the hot path, mcmc.cpp:143-387) and prints
// MCMC steps (= NLL evaluations) per second.  Usage: bench_cpp [scale=1.0] [nsteps=2000] [graph_steps=10]
// Synthetic inputs as SURVEY.md 8(d) C3 describes them (not bit-identical to bench.py's generator).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../sxmc_amd/include/sxmc/ensemble.h"

int main(int argc, char** argv) {
  const double scale = argc > 1 ? std::atof(argv[1]) : 1.0;
  const unsigned nsteps = argc > 2 ? (unsigned)std::atoi(argv[2]) : 2000;
  const unsigned graph_steps = argc > 3 ? (unsigned)std::atoi(argv[3]) : 10;
  const size_t S = 12, F = 5, E = 100000;
  const size_t per_signal = (size_t)(1e8 * scale) / S;

  std::vector<sxmc::Observable> observables(3);
  const float lo[3] = {0, 0, -1}, hi[3] = {10, 6, 1};
  for (size_t k = 0; k < 3; k++) {
    observables[k].field_index = k;
    observables[k].bins = 20;
    observables[k].lower = lo[k];
    observables[k].upper = hi[k];
  }
  std::vector<sxmc::Systematic> systematics(3);
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

  std::mt19937_64 rng(3);
  std::normal_distribution<float> gauss(0.0f, 1.0f);
  std::uniform_real_distribution<float> uni(0.0f, 1.0f);
  std::vector<sxmc::Source> sources;
  std::vector<sxmc::Signal> signals;
  std::vector<float> data;
  std::vector<float> tab(per_signal * F);
  for (size_t j = 0; j < S; j++) {
    for (size_t i = 0; i < per_signal; i++) {
      const float e_true = 2.0f + 0.5f * j + 1.2f * gauss(rng);
      tab[i * F + 0] = e_true + 0.3f * gauss(rng);
      tab[i * F + 1] = 6.0f * std::cbrt(uni(rng));
      tab[i * F + 2] = 2.0f * uni(rng) - 1.0f;
      tab[i * F + 3] = e_true;
      tab[i * F + 4] = 0.0f;
    }
    sxmc::Signal sig;
    sig.name = "signal" + std::to_string(j);
    sig.source = sxmc::Source("source" + std::to_string(j), j, 1.0f, 0.0f, false);
    sig.nexpected = (double)E / S;
    sxmc::build_pdfz(sig, tab, (int)F, observables, systematics);
    signals.push_back(sig);
    sources.push_back(sig.source);
    for (size_t e = 0; e < E / S; e++) {  // data events: samples of the mixture
      const size_t i = rng() % per_signal;
      data.push_back(tab[i * F + 0]);
      data.push_back(tab[i * F + 1]);
      data.push_back(tab[i * F + 2]);
      data.push_back(0.0f);
    }
    std::fprintf(stderr, "signal %zu of %zu built (%zu samples)\n", j + 1, S, per_signal);
  }
  tab.clear();
  tab.shrink_to_fit();

  sxmc_stream_t strm = nullptr;
  sxmc::check(sxmc_stream_create_nonblocking(&strm));
  for (int pass = 0; pass < 2; pass++) {  // pass 0 warms up (clocks, launch plan); pass 1 is timed
    sxmc::MCMC mcmc(sources, signals, systematics, observables, 1234 + pass, strm);
    mcmc.graph_steps = graph_steps;
    const auto t0 = std::chrono::steady_clock::now();
    sxmc::Chain chain = mcmc(data, pass == 0 ? std::min(nsteps, 500u) : nsteps, 0.1f, false, 10000);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (pass == 1) {
      std::printf("{\"driver\": \"sxmc::MCMC (C++)\", \"nsamples_total\": %zu, \"nsignals\": %zu, \"nevents\": %zu, "
                  "\"steps\": %u, \"steps_per_graph\": %u, \"seconds\": %.4f, \"steps_per_sec\": %.1f, "
                  "\"accepted\": %zu, \"rows_kept\": %zu}\n",
                  per_signal * S, S, data.size() / 4, nsteps, graph_steps, sec, nsteps / sec, chain.accepted,
                  chain.nrows());
    }
  }
  sxmc_stream_destroy(strm);
  for (sxmc::Signal& s : signals) delete s.histogram;
  return 0;
}
