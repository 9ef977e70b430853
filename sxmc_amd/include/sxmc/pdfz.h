// pdfz.h -- the reference's pdfz interface (src/pdfz.h:87-627) on MI355X.
//
// Same names, constructor arguments, defaults, virtuals and throw behaviour as pdfz::Eval /
// pdfz::EvalHist, so that callers written against the reference (mcmc.cpp:233-242, 264-271,
// signal.cpp:131-146, 192-196, bench_sxmc.cpp:58-96) compile against this header.  Everything is a
// thin layer over the C ABI of libsxmc_hip.so (include/sxmc_hip.h); the arithmetic runs in
// hand-written gfx950 kernels.  Differences, all forced by what is absent here:
//   * ROOT-returning methods (CreateHistogram, CreateHistogramProjection, DefaultHistogram,
//     RandomSample) are replaced by plain-array accessors (GetBins, GetNormalizedHistogram);
//   * Optimize/OptimizeBin/OptimizeEval (brute-force launch autotuning, pdfz.cpp:622-814): the constructor's
//     `optimize` flag and Optimize() keep their meaning -- trial launches at the first evaluation with evaluation
//     points -- but the trials are those of the BATCH the library forms behind the per-evaluator calls
//     (sxmc_hist_set_optimize, include/sxmc_hip.h); the lookup's launch shape has nothing to choose;
//   * a systematic's parameter-index array is read when AddSystematic is called (the reference keeps
//     the pointer and reads it at every evaluation, pdfz.cpp:143).
#pragma once

#include <string>
#include <vector>

#include "device_array.h"

#ifndef SXMC_ARRAY_TEMPLATE
#define SXMC_ARRAY_TEMPLATE sxmc::DeviceArray
#endif

namespace pdfz {

template <typename T>
using Array = SXMC_ARRAY_TEMPLATE<T>;

/** pdfz::Error (pdfz.h:93-102): thrown by value. */
struct Error {
  Error(const std::string& _msg) { msg = _msg; }
  std::string msg;
};

/** pdfz::Systematic and its four kinds (pdfz.h:109-233). */
struct Systematic {
  enum Type { SHIFT, SCALE, RESOLUTION_SCALE, CTSCALE };
  Type type;
  Systematic(Type _type) : type(_type) {}
  virtual ~Systematic() {}
};

struct ShiftSystematic : public Systematic {  // x' = x + p
  ShiftSystematic(int _obs, Array<short>* _pars) : Systematic(SHIFT), obs(_obs), pars(_pars) {}
  int obs;
  Array<short>* pars;
};

struct ScaleSystematic : public Systematic {  // x' = x * (1 + p)
  ScaleSystematic(int _obs, Array<short>* _pars) : Systematic(SCALE), obs(_obs), pars(_pars) {}
  int obs;
  Array<short>* pars;
};

struct CosThetaScaleSystematic : public Systematic {  // x' = 1 + (x - 1) * (1 + p)
  CosThetaScaleSystematic(int _obs, Array<short>* _pars) : Systematic(CTSCALE), obs(_obs), pars(_pars) {}
  int obs;
  Array<short>* pars;
};

struct ResolutionScaleSystematic : public Systematic {  // x' = x + p * (x - x_true)
  ResolutionScaleSystematic(int _obs, int _true_obs, Array<short>* _pars)
      : Systematic(RESOLUTION_SCALE), obs(_obs), true_obs(_true_obs), pars(_pars) {}
  int obs;
  int true_obs;
  Array<short>* pars;
};

inline void throw_on(int rc) {
  if (rc == SXMC_OK) return;
  if (rc == SXMC_ERR_INVALID) throw Error(sxmc_last_error());
  throw sxmc::HipError(std::string("libsxmc_hip: ") + sxmc_last_error());
}

/** pdfz::Eval (pdfz.h:246-395): abstract evaluator interface. */
class Eval {
 public:
  /** Same arguments as the reference (pdfz.h:268-270).  The size validation of pdfz.cpp:64-82 is
   *  performed, in the same order and with the same messages, by sxmc_hist_create. */
  Eval(const std::vector<float>& /*samples*/, int _nfields, int _nobservables,
       const std::vector<double>& /*lower*/, const std::vector<double>& /*upper*/, unsigned _dataset = 0)
      : nfields(_nfields), nobservables(_nobservables), dataset(_dataset) {}
  virtual ~Eval() {}

 protected:
  Eval(int _nfields, int _nobservables, unsigned _dataset)
      : nfields(_nfields), nobservables(_nobservables), dataset(_dataset) {}

 public:
  virtual void SetEvalPoints(const std::vector<float>& points) = 0;
  virtual void SetPDFValueBuffer(Array<float>* output, int offset = 0, int stride = 1) {
    pdf_buffer = output;
    pdf_offset = offset;
    pdf_stride = stride;
  }
  virtual void SetNormalizationBuffer(Array<unsigned int>* norm, int offset = 0) {
    norm_buffer = norm;
    norm_offset = offset;
  }
  virtual void SetParameterBuffer(Array<double>* params, int offset = 0, int stride = 1) {
    param_buffer = params;
    param_offset = offset;
    param_stride = stride;
  }
  virtual void AddSystematic(const Systematic& syst) = 0;
  virtual void EvalAsync(bool do_eval_pdf = true) = 0;
  virtual void EvalFinished() = 0;
  /** The three buffers are BORROWED (pdfz.cpp:106-124 stores raw pointers, like this class): a caller whose
   *  buffers are about to die un-binds them, so that the next evaluation binds afresh or fails loudly instead of
   *  touching destroyed arrays.  (The reference leaves the dangling pointers in place.) */
  virtual void ForgetBuffers() {
    pdf_buffer = nullptr;
    norm_buffer = nullptr;
    param_buffer = nullptr;
  }

 protected:
  int nfields;
  int nobservables;
  unsigned dataset;
  Array<float>* pdf_buffer = nullptr;
  int pdf_offset = 0;
  int pdf_stride = 1;
  Array<unsigned int>* norm_buffer = nullptr;
  int norm_offset = 0;
  Array<double>* param_buffer = nullptr;
  int param_offset = 0;
  int param_stride = 1;
};

/** pdfz::EvalHist (pdfz.h:402-574): N-dimensional histogram PDF. */
class EvalHist : public Eval {
 public:
  EvalHist(const std::vector<float>& samples, int nfields, int nobservables, const std::vector<double>& lower,
           const std::vector<double>& upper, const std::vector<int>& nbins, unsigned dataset = 0,
           bool optimize = true)
      : Eval(samples, nfields, nobservables, lower, upper, dataset) {
    throw_on(sxmc_hist_create(samples.data(), samples.size(), 0, nfields, nobservables, lower.data(),
                              lower.size(), upper.data(), upper.size(), nbins.data(), nbins.size(), dataset,
                              &handle));
    throw_on(sxmc_hist_set_optimize(handle, optimize ? 1 : 0));   // needs_optimization(optimize), pdfz.cpp:188
  }
  /** A second evaluator over the SAME sample table as `base` (nothing copied; own histogram, event bins
   *  and bindings; systematics as attached to `base` so far): one per concurrent chain on a GPU. */
  struct SharedSamples {};
  EvalHist(const EvalHist& base, SharedSamples) : Eval(base.nfields, base.nobservables, base.dataset) {
    throw_on(sxmc_hist_create_shared(base.handle, &handle));
  }
  EvalHist(const EvalHist&) = delete;
  EvalHist& operator=(const EvalHist&) = delete;
  virtual ~EvalHist() { sxmc_hist_destroy(handle); }

  virtual void SetEvalPoints(const std::vector<float>& points) {
    throw_on(sxmc_hist_set_eval_points(handle, points.data(), points.size()));
  }

  virtual void AddSystematic(const Systematic& syst) {
    int obs = 0, extra = 0;
    Array<short>* pars = nullptr;
    if (syst.type == Systematic::SHIFT) {
      const ShiftSystematic& s = dynamic_cast<const ShiftSystematic&>(syst);
      obs = s.obs;
      pars = s.pars;
    } else if (syst.type == Systematic::SCALE) {
      const ScaleSystematic& s = dynamic_cast<const ScaleSystematic&>(syst);
      obs = s.obs;
      pars = s.pars;
    } else if (syst.type == Systematic::CTSCALE) {
      const CosThetaScaleSystematic& s = dynamic_cast<const CosThetaScaleSystematic&>(syst);
      obs = s.obs;
      pars = s.pars;
    } else if (syst.type == Systematic::RESOLUTION_SCALE) {
      const ResolutionScaleSystematic& s = dynamic_cast<const ResolutionScaleSystematic&>(syst);
      obs = s.obs;
      extra = s.true_obs;
      pars = s.pars;
    } else {
      throw Error("Unknown systematic type");
    }
    throw_on(sxmc_hist_add_systematic(handle, (int)syst.type, obs, extra, (int)pars->size(),
                                      pars->readOnlyHostPtr()));
  }

  /** Bind the three caller buffers (device side) and launch zero + fill (+ lookup) on this evaluator's
   *  stream; returns before completion (pdfz.cpp:441-488). */
  virtual void EvalAsync(bool do_eval_pdf = true) {
    Bind();
    throw_on(sxmc_hist_eval_async(handle, do_eval_pdf ? 1 : 0));
  }
  virtual void EvalFinished() { throw_on(sxmc_hist_eval_finished(handle)); }

  /** The accessor calls of pdfz.cpp:457-470, 484-487: outputs become device-valid (host copies are
   *  stale until read back), parameters are uploaded if the host side is newer. */
  void ForgetBuffers() override {
    Eval::ForgetBuffers();
    throw_on(sxmc_hist_set_pdf_value_buffer(handle, nullptr, 0, 1));
    throw_on(sxmc_hist_set_normalization_buffer(handle, nullptr, 0));
    throw_on(sxmc_hist_set_parameter_buffer(handle, nullptr, 0, 1));
  }

  void Bind() {
    if (pdf_buffer) throw_on(sxmc_hist_set_pdf_value_buffer(handle, pdf_buffer->writeOnlyPtr(), pdf_offset, pdf_stride));
    if (norm_buffer) throw_on(sxmc_hist_set_normalization_buffer(handle, norm_buffer->writeOnlyPtr(), norm_offset));
    if (param_buffer) throw_on(sxmc_hist_set_parameter_buffer(handle, param_buffer->readOnlyPtr(), param_offset, param_stride));
  }

  /** pdfz.cpp:622-628: trial launches choose the launch shape -- here at the next lookup evaluation of the batch this
   *  evaluator is evaluated in (they need an evaluation's bindings).  OptimizeBin (:630-727) is that choice for the fill;
   *  OptimizeEval (:729-814) tuned the lookup kernel, whose shape is one lane per evaluation point here: nothing to try. */
  virtual void Optimize() { throw_on(sxmc_hist_optimize(handle)); }
  virtual void OptimizeBin() { throw_on(sxmc_hist_optimize(handle)); }
  virtual void OptimizeEval() {}

  /** The launch plan of the batch this evaluator's evaluations run in + "tuned=.. trial_launches=.." (tests, logs). */
  std::string LaunchInfo() {
    std::vector<char> buf(16384);
    throw_on(sxmc_hist_launch_info(handle, buf.data(), buf.size()));
    return std::string(buf.data());
  }

  /** pdfz.h:542-556 */
  void GetSamples(std::vector<float>& sv) {
    size_t n = 0;
    throw_on(sxmc_hist_nsamples(handle, &n));
    const size_t oldsize = sv.size();
    sv.resize(oldsize + n * (size_t)(nobservables + 1));
    throw_on(sxmc_hist_get_samples(handle, sv.data() + oldsize, n * (size_t)(nobservables + 1)));
  }

  /** Bin contents of the last evaluation, row-major (what CreateHistogram reads, pdfz.cpp:511). */
  std::vector<unsigned> GetBins() {
    int b = 0;
    throw_on(sxmc_hist_total_nbins(handle, &b));
    std::vector<unsigned> out((size_t)b);
    throw_on(sxmc_hist_get_bins(handle, out.data(), out.size()));
    return out;
  }

  /** ROOT-free CreateHistogram (pdfz.cpp:498-594): fill only (EvalAsync(false)), then
   *  content = bins / bin_volume / norm, or 0 when norm == 0; row-major. */
  std::vector<double> GetNormalizedHistogram() {
    EvalAsync(false);
    EvalFinished();
    std::vector<unsigned> bins = GetBins();
    const unsigned norm = norm_buffer->readOnlyHostPtr()[norm_offset];
    double vol = 0;
    throw_on(sxmc_hist_bin_volume(handle, &vol));
    std::vector<double> out(bins.size(), 0.0);
    if (norm > 0)
      for (size_t i = 0; i < bins.size(); i++) out[i] = bins[i] / vol / norm;
    return out;
  }

  /** The sampling step of RandomSample (pdfz.cpp:843-918) on the device: appends `observed` events (rows of
   *  nobservables + 1 floats, last = dataset id) drawn from the histogram of the last evaluation
   *  (EvalAsync(false) first, as CreateHistogram does), redrawn while outside [lowers, uppers] when given. */
  void SampleEvents(std::vector<float>& events, size_t observed, unsigned long long seed,
                    const std::vector<float>& uppers = std::vector<float>(),
                    const std::vector<float>& lowers = std::vector<float>()) {
    const size_t row = (size_t)nobservables + 1, old = events.size();
    events.resize(old + observed * row);
    const bool cuts = !uppers.empty() && !lowers.empty();
    throw_on(sxmc_hist_random_sample(handle, observed, seed, cuts ? lowers.data() : nullptr,
                                     cuts ? uppers.data() : nullptr, events.data() + old));
  }

  sxmc_hist_t Handle() const { return handle; }

 protected:
  sxmc_hist_t handle = nullptr;
};

}  // namespace pdfz
