// fit_types.h -- ROOT- and JSON-free forms of the reference's fit metadata structs, with the
// reference's field names, as far as the MCMC driver needs them:
//   Source      src/source.h:13-58        Observable  src/observable.h:22-42
//   Systematic  src/systematic.h:23-49    Signal      src/signal.h (name, dataset, source, nexpected,
//                                                       n_mc, histogram)
// Callers fill these structs directly, or sxmc::load_config (config.h: the reference's JSON schema, config.cpp:19-297,
// and ROOT-free sample tables in place of io/ttree_io.cpp) fills them from a fit configuration.
#pragma once

#include <cstddef>
#include <memory>
#include <string>
#include <vector>

#include "pdfz.h"

namespace sxmc {

struct Source {
  Source() {}
  Source(const std::string& _name, size_t _index, float _mean, float _sigma, bool _fixed)
      : name(_name), index(_index), mean(_mean), sigma(_sigma), fixed(_fixed) {}
  std::string name;
  size_t index = 0;    //!< Index in the list of sources
  float mean = 1.0f;   //!< Mean expectation (scaling, 1.0 is nominal)
  float sigma = 0.0f;  //!< Gaussian constraint (fractional)
  bool fixed = false;
};

struct Observable {
  std::string name;
  std::string field;
  size_t field_index = 0;  //!< Index in the sampled data for this field
  size_t bins = 0;
  float lower = 0;
  float upper = 0;
};

struct Systematic {
  std::string name;
  std::string title;
  std::string observable_field;  //!< Name of the field the systematic acts on (systematic.h)
  std::string truth_field;       //!< resolution_scale: name of the field holding the true value
  size_t observable_field_index = 0;
  size_t truth_field_index = 0;
  size_t npars = 1;            //!< Number of parameters in power series
  std::vector<double> means;   //!< Mean values (power series)
  std::vector<double> sigmas;  //!< Standard deviations
  std::vector<short> pidx;     //!< Global indices into the systematic block of the parameter vector
  pdfz::Systematic::Type type = pdfz::Systematic::SHIFT;
  bool fixed = false;
};

struct Signal {
  std::string name;
  std::string title;
  std::string filename;                       //!< where the samples came from (signal.h)
  std::vector<std::string> systematic_names;  //!< the systematics this signal lists in the configuration
  unsigned dataset = 0;
  Source source;
  double nexpected = 0;
  size_t n_mc = 0;  //!< number of MC samples BEFORE cuts (signal.cpp:28); build_pdfz fills it in when it is 0
  pdfz::Eval* histogram = nullptr;  //!< borrowed by the driver, as in the reference
  // keeps the parameter-index arrays alive (the reference leaks them, signal.cpp:139)
  std::vector<std::shared_ptr<pdfz::Array<short>>> par_arrays;
};

/** Signal::build_pdfz (signal.cpp:112-170): histogram evaluator of one signal with every systematic
 *  attached.  `samples` is the row-major [n][nfields] table, observables first. */
inline void build_pdfz(Signal& sig, const std::vector<float>& samples, int nfields,
                       const std::vector<Observable>& observables, std::vector<Systematic>& systematics) {
  std::vector<double> lower(observables.size()), upper(observables.size());
  std::vector<int> nbins(observables.size());
  for (const Observable& o : observables) {
    lower.at(o.field_index) = o.lower;
    upper.at(o.field_index) = o.upper;
    nbins.at(o.field_index) = (int)o.bins;
  }
  pdfz::EvalHist* h = new pdfz::EvalHist(samples, nfields, (int)observables.size(), lower, upper, nbins, sig.dataset);
  sig.histogram = h;
  // (a table loaded through load_config has been cut: n_mc is then the row count before the cuts, set by the loader)
  if (sig.n_mc == 0) sig.n_mc = samples.size() / (size_t)nfields;
  for (Systematic& s : systematics) {
    auto pars = std::make_shared<pdfz::Array<short>>(s.npars, true);
    for (size_t i = 0; i < s.pidx.size(); i++) pars->writeOnlyHostPtr()[i] = s.pidx[i];
    sig.par_arrays.push_back(pars);
    const int o = (int)s.observable_field_index, t = (int)s.truth_field_index;
    switch (s.type) {
      case pdfz::Systematic::SHIFT: h->AddSystematic(pdfz::ShiftSystematic(o, pars.get())); break;
      case pdfz::Systematic::SCALE: h->AddSystematic(pdfz::ScaleSystematic(o, pars.get())); break;
      case pdfz::Systematic::CTSCALE: h->AddSystematic(pdfz::CosThetaScaleSystematic(o, pars.get())); break;
      case pdfz::Systematic::RESOLUTION_SCALE:
        h->AddSystematic(pdfz::ResolutionScaleSystematic(o, t, pars.get()));
        break;
    }
  }
}

/** A copy of `base` whose evaluator shares base's sample table (pdfz::EvalHist::SharedSamples): what each
 *  additional concurrent chain on a GPU works with.  The caller deletes .histogram, as for build_pdfz. */
inline Signal share_pdfz(const Signal& base) {
  Signal s = base;
  s.histogram = new pdfz::EvalHist(*dynamic_cast<pdfz::EvalHist*>(base.histogram), pdfz::EvalHist::SharedSamples{});
  return s;
}

/** Signal::get_efficiency (signal.cpp:172-199): fraction of the MC samples inside the PDF domain with
 *  every systematic at its mean. */
inline double get_efficiency(Signal& sig, const std::vector<Systematic>& systematics) {
  size_t npars = 0;
  for (const Systematic& s : systematics) npars += s.npars;
  pdfz::Array<double> param_buffer(npars, true);
  size_t k = 0;
  for (const Systematic& s : systematics)
    for (size_t j = 0; j < s.npars; j++) param_buffer.writeOnlyHostPtr()[k++] = s.means[j];
  pdfz::Array<unsigned> norms_buffer(1, true);
  norms_buffer.writeOnlyHostPtr();
  sig.histogram->SetNormalizationBuffer(&norms_buffer);
  sig.histogram->SetParameterBuffer(&param_buffer);
  sig.histogram->EvalAsync(false);
  sig.histogram->EvalFinished();
  const double in_domain = norms_buffer.readOnlyHostPtr()[0];
  sig.histogram->ForgetBuffers();   // the two arrays above die with this call
  return 1.0 * in_domain / (double)sig.n_mc;
}

}  // namespace sxmc
