// test_cpp_api.cpp -- GPU tests of the C++ mirror of the reference's interface
// (sxmc_amd/include/sxmc/{device_array,pdfz,nll_kernels,mcmc}.h), written with the reference's own
// spelling (hemi::Array, HEMI_KERNEL_LAUNCH, pdfz::EvalHist) through sxmc_amd/include/spelling/.
//
// The pdfz cases are the reference's known-answer tests (test/test_pdfz.cpp, test_pdfz_2d.cpp,
// test_pdfz_syst.cpp) against the CURRENT reference API: evaluation points carry the dataset column
// and systematics take a parameter-index array.  Run by tests/test_gpu_cpp_api.py.
#include <sxmc/nll_kernels.h>  // spelling/: HEMI_KERNEL_LAUNCH, hemi::Array
#include <sxmc/pdfz.h>

#include "../../sxmc_amd/include/sxmc/ensemble.h"
// (pdfz::Error is the reference's exception type: a plain struct with `msg`, not a std::exception)
#define MINI_TEST_EXTRA_CATCH                                                              \
  }                                                                                        \
  catch (const pdfz::Error& e) {                                                           \
    failed++;                                                                              \
    std::printf("[ FAIL ] %s\n    pdfz::Error: %s\n", full.c_str(), e.msg.c_str());
#include "mini_test.h"

using std::isnan;

// ------------------------------------------------------------------ fixtures (test_pdfz_fixtures*.h)
static std::vector<float> with_dataset(const std::vector<float>& pts, int nobs, float dataset = 0) {
  std::vector<float> out;
  for (size_t i = 0; i < pts.size() / nobs; i++) {
    for (int k = 0; k < nobs; k++) out.push_back(pts[i * nobs + k]);
    out.push_back(dataset);
  }
  return out;
}

struct EvalHistConstructor {
  void SetUp() {
    nobservables = 1;
    nfields = 1;
    samples = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 1.1f, -0.1f};
    lower = {0.0};
    upper = {1.0};
    nbins = {2};
  }
  void TearDown() {}
  int nobservables, nfields;
  std::vector<float> samples;
  std::vector<double> lower, upper;
  std::vector<int> nbins;
};

struct EvalHistMethods : EvalHistConstructor {
  void SetUp() {
    EvalHistConstructor::SetUp();
    MakeEvaluator();
  }
  void MakeEvaluator() {
    evaluator = new pdfz::EvalHist(samples, nfields, nobservables, lower, upper, nbins);
    eval_points = with_dataset({-0.1f, 0.0f, 0.25f, 0.5f, 0.75f, 1.0f}, 1);
    pdf_values = new hemi::Array<float>(20, true);
    norm = new hemi::Array<unsigned int>(3, true);
    params = new hemi::Array<double>(5, true);
    params->writeOnlyHostPtr();
  }
  void TearDown() {
    delete evaluator;
    delete pdf_values;
    delete norm;
    delete params;
  }
  pdfz::EvalHist* evaluator;
  std::vector<float> eval_points;
  hemi::Array<float>* pdf_values;
  hemi::Array<unsigned int>* norm;
  hemi::Array<double>* params;
};

// ------------------------------------------------------------------ test_pdfz.cpp
TEST(PdfzError, Constructor) {
  pdfz::Error err("test");
  EXPECT_EQ(err.msg, std::string("test"));
}

TEST(SystematicObjects, Constructors) {
  hemi::Array<short> pars(1, true);
  pdfz::ShiftSystematic a(1, &pars);
  EXPECT_EQ(pdfz::Systematic::SHIFT, a.type);
  EXPECT_EQ(1, a.obs);
  pdfz::ScaleSystematic b(0, &pars);
  EXPECT_EQ(pdfz::Systematic::SCALE, b.type);
  pdfz::ResolutionScaleSystematic c(0, 2, &pars);
  EXPECT_EQ(pdfz::Systematic::RESOLUTION_SCALE, c.type);
  EXPECT_EQ(2, c.true_obs);
  // enum values are part of the descriptor format (pdfz.h:111-116)
  EXPECT_EQ(0, (int)pdfz::Systematic::SHIFT);
  EXPECT_EQ(1, (int)pdfz::Systematic::SCALE);
  EXPECT_EQ(2, (int)pdfz::Systematic::RESOLUTION_SCALE);
  EXPECT_EQ(3, (int)pdfz::Systematic::CTSCALE);
}

TEST_F(EvalHistConstructor, WrongSampleSize) {
  ASSERT_THROW(pdfz::EvalHist(samples, 2 /* nfields */, nobservables, lower, upper, nbins), pdfz::Error);
}
TEST_F(EvalHistConstructor, NobsLargerThanNfields) {
  ASSERT_THROW(pdfz::EvalHist(samples, nfields, 7 /* nobservables */, lower, upper, nbins), pdfz::Error);
}
TEST_F(EvalHistConstructor, WrongLowerSize) {
  lower.resize(2);
  ASSERT_THROW(pdfz::EvalHist(samples, nfields, nobservables, lower, upper, nbins), pdfz::Error);
}
TEST_F(EvalHistConstructor, WrongUpperSize) {
  upper.resize(2);
  ASSERT_THROW(pdfz::EvalHist(samples, nfields, nobservables, lower, upper, nbins), pdfz::Error);
}
TEST_F(EvalHistConstructor, WrongNbinsSize) {
  nbins.resize(2);
  ASSERT_THROW(pdfz::EvalHist(samples, nfields, nobservables, lower, upper, nbins), pdfz::Error);
}
TEST_F(EvalHistConstructor, ZeroBins) {
  nbins[0] = 0;
  ASSERT_THROW(pdfz::EvalHist(samples, nfields, nobservables, lower, upper, nbins), pdfz::Error);
}

static void expect_1d(float* r, int off, int st, double v1, double v2, double v3, double v4) {
  ASSERT_TRUE(isnan(r[off + 0 * st]));
  ASSERT_FLOAT_EQ(v1, r[off + 1 * st]);
  ASSERT_FLOAT_EQ(v2, r[off + 2 * st]);
  ASSERT_FLOAT_EQ(v3, r[off + 3 * st]);
  ASSERT_FLOAT_EQ(v4, r[off + 4 * st]);
  ASSERT_TRUE(isnan(r[off + 5 * st]));
}

TEST_F(EvalHistMethods, Evaluation) {
  evaluator->SetEvalPoints(eval_points);
  evaluator->SetPDFValueBuffer(pdf_values);
  evaluator->SetNormalizationBuffer(norm);
  evaluator->SetParameterBuffer(params);
  evaluator->EvalAsync();
  evaluator->EvalFinished();
  EXPECT_EQ((unsigned int)5, *norm->readOnlyHostPtr());
  expect_1d(pdf_values->hostPtr(), 0, 1, 1.6, 1.6, 0.4, 0.4);
}

TEST_F(EvalHistMethods, EvaluationOffsetStride) {
  evaluator->SetEvalPoints(eval_points);
  evaluator->SetPDFValueBuffer(pdf_values, 3, 2);
  evaluator->SetNormalizationBuffer(norm, 1);
  evaluator->SetParameterBuffer(params);
  norm->writeOnlyHostPtr()[0] = 77;  // detect incorrect writes
  norm->writeOnlyHostPtr()[1] = 88;
  norm->writeOnlyHostPtr()[2] = 99;
  norm->readOnlyDevicePtr();  // force flush to device
  evaluator->EvalAsync();
  evaluator->EvalFinished();
  EXPECT_EQ((unsigned int)77, norm->readOnlyHostPtr()[0]);
  EXPECT_EQ((unsigned int)5, norm->readOnlyHostPtr()[1]);
  EXPECT_EQ((unsigned int)99, norm->readOnlyHostPtr()[2]);
  expect_1d(pdf_values->hostPtr(), 3, 2, 1.6, 1.6, 0.4, 0.4);
}

TEST_F(EvalHistMethods, CreateHistogram1D) {
  // ROOT-free CreateHistogram: integral over the domain is 1 (test_pdfz.cpp:128-140)
  evaluator->SetNormalizationBuffer(norm);
  evaluator->SetParameterBuffer(params);
  std::vector<double> h = evaluator->GetNormalizedHistogram();
  EXPECT_EQ((size_t)2, h.size());
  ASSERT_FLOAT_EQ(1.0, (h[0] + h[1]) * 0.5);
  ASSERT_FLOAT_EQ(1.6, h[0]);
  ASSERT_FLOAT_EQ(0.4, h[1]);
}

// ------------------------------------------------------------------ test_pdfz_syst.cpp
struct EvalSystematics : EvalHistMethods {
  void SetUp() {
    EvalHistMethods::SetUp();
    evaluator->SetEvalPoints(eval_points);
    evaluator->SetPDFValueBuffer(pdf_values);
    evaluator->SetNormalizationBuffer(norm);
    evaluator->SetParameterBuffer(params);
    pars = new hemi::Array<short>(1, true);
    pars->writeOnlyHostPtr()[0] = 0;
  }
  void TearDown() {
    delete pars;
    EvalHistMethods::TearDown();
  }
  void Run(double p, unsigned expect_norm, double v1, double v2, double v3, double v4) {
    params->writeOnlyHostPtr()[0] = p;
    evaluator->EvalAsync();
    evaluator->EvalFinished();
    EXPECT_EQ(expect_norm, *norm->readOnlyHostPtr());
    expect_1d(pdf_values->hostPtr(), 0, 1, v1, v2, v3, v4);
  }
  hemi::Array<short>* pars;
};

TEST_F(EvalSystematics, Shift) {
  evaluator->AddSystematic(pdfz::ShiftSystematic(0, pars));
  Run(0.0, 5, 1.6, 1.6, 0.4, 0.4);
  Run(-0.25, 4, 1.5, 1.5, 0.5, 0.5);
  Run(0.25, 6, 1.0, 1.0, 1.0, 1.0);
}

TEST_F(EvalSystematics, Scale) {
  evaluator->AddSystematic(pdfz::ScaleSystematic(0, pars));
  Run(0.0, 5, 1.6, 1.6, 0.4, 0.4);
  Run(-0.1, 6, 5.0 / 3, 5.0 / 3, 1.0 / 3, 1.0 / 3);
  Run(1.0, 4, 1.0, 1.0, 1.0, 1.0);  // remember scale is 1 + 1.0
}

struct EvalResolution : EvalSystematics {
  void SetUp() {
    EvalHistConstructor::SetUp();
    nfields = 2;  // observable + a true-energy field fixed at 0.7 (test_pdfz_syst.cpp:168-176)
    samples = {0.1f, 0.7f, 0.2f, 0.7f, 0.3f, 0.7f, 0.4f, 0.7f, 0.5f, 0.7f, 1.1f, 0.7f, -0.1f, 0.7f};
    MakeEvaluator();
    evaluator->SetEvalPoints(eval_points);
    evaluator->SetPDFValueBuffer(pdf_values);
    evaluator->SetNormalizationBuffer(norm);
    evaluator->SetParameterBuffer(params);
    pars = new hemi::Array<short>(1, true);
    pars->writeOnlyHostPtr()[0] = 0;
    evaluator->AddSystematic(pdfz::ResolutionScaleSystematic(0, 1, pars));
  }
};

TEST_F(EvalResolution, ZeroNegPos) {
  Run(0.0, 5, 1.6, 1.6, 0.4, 0.4);
  Run(-0.30, 7, 2.0 * 5 / 7, 2.0 * 5 / 7, 2.0 * 2 / 7, 2.0 * 2 / 7);
  Run(0.30, 4, 2.0, 2.0, 0.0, 0.0);
}

// ------------------------------------------------------------------ test_pdfz_2d.cpp
struct EvalHist2D {
  void SetUp() {
    samples = {0.4f, 10.5f, 0.5f, 11.0f, 0.75f, 11.0f, 0.6f, 11.5f, 0.6f, 11.8f, 0.9f, 11.5f, 0.4f, 12.0f};
    lower = {0.0, 10.0};
    upper = {1.0, 12.0};
    nbins = {2, 3};
    evaluator = new pdfz::EvalHist(samples, 2, 2, lower, upper, nbins);
    eval_points = with_dataset({0.2f, 10.2f, 0.7f, 10.4f, 0.5f, 11.0f, 0.25f, 11.8f, 0.9f, 11.9f, 0.3f, 12.0f,
                                0.3f, 13.0f, 0.3f, 5.0f}, 2);
    pdf_values = new hemi::Array<float>(40, true);
    norm = new hemi::Array<unsigned int>(3, true);
    params = new hemi::Array<double>(5, true);
    params->writeOnlyHostPtr();
  }
  void TearDown() {
    delete evaluator;
    delete pdf_values;
    delete norm;
    delete params;
  }
  std::vector<float> samples, eval_points;
  std::vector<double> lower, upper;
  std::vector<int> nbins;
  pdfz::EvalHist* evaluator;
  hemi::Array<float>* pdf_values;
  hemi::Array<unsigned int>* norm;
  hemi::Array<double>* params;
};

TEST_F(EvalHist2D, EvaluationOffsetStride) {
  evaluator->SetEvalPoints(eval_points);
  evaluator->SetPDFValueBuffer(pdf_values, 3, 2);
  evaluator->SetNormalizationBuffer(norm, 1);
  evaluator->SetParameterBuffer(params);
  norm->writeOnlyHostPtr()[0] = 77;
  norm->writeOnlyHostPtr()[1] = 88;
  norm->writeOnlyHostPtr()[2] = 99;
  norm->readOnlyDevicePtr();
  evaluator->EvalAsync();
  evaluator->EvalFinished();
  EXPECT_EQ((unsigned int)77, norm->readOnlyHostPtr()[0]);
  EXPECT_EQ((unsigned int)6, norm->readOnlyHostPtr()[1]);
  EXPECT_EQ((unsigned int)99, norm->readOnlyHostPtr()[2]);
  const double n = 6 * (0.5 * (2.0 / 3.0));  // samples in boundary * bin area
  float* r = pdf_values->hostPtr();
  ASSERT_FLOAT_EQ(1 / n, r[3]);
  ASSERT_FLOAT_EQ(0 / n, r[5]);
  ASSERT_FLOAT_EQ(2 / n, r[7]);
  ASSERT_FLOAT_EQ(0 / n, r[9]);
  ASSERT_FLOAT_EQ(3 / n, r[11]);
  ASSERT_TRUE(isnan(r[13]));
  ASSERT_TRUE(isnan(r[15]));
  ASSERT_TRUE(isnan(r[17]));
}

TEST_F(EvalHist2D, CreateHistogram2D) {
  evaluator->SetNormalizationBuffer(norm);
  evaluator->SetParameterBuffer(params);
  std::vector<double> h = evaluator->GetNormalizedHistogram();  // row-major [x][y]
  const double n = 6 * (0.5 * (2.0 / 3.0));
  double integral = 0;
  for (double v : h) integral += v * 0.5 * (2.0 / 3.0);
  ASSERT_FLOAT_EQ(1.0, integral);
  ASSERT_FLOAT_EQ(1 / n, h[0 * 3 + 0]);
  ASSERT_FLOAT_EQ(0 / n, h[1 * 3 + 0]);
  ASSERT_FLOAT_EQ(2 / n, h[1 * 3 + 1]);
  ASSERT_FLOAT_EQ(3 / n, h[1 * 3 + 2]);
}

TEST_F(EvalHist2D, GetSamples) {
  std::vector<float> sv;
  evaluator->GetSamples(sv);
  EXPECT_EQ((size_t)21, sv.size());
  ASSERT_FLOAT_EQ(0.4, sv[0]);
  ASSERT_FLOAT_EQ(10.5, sv[1]);
  ASSERT_FLOAT_EQ(0.0, sv[2]);
  ASSERT_FLOAT_EQ(12.0, sv[19]);
}

// ------------------------------------------------------------------ device array semantics (Appendix B)
TEST(DeviceArray, LazyMirrorSemantics) {
  hemi::Array<int> a(4, true);
  EXPECT_EQ((size_t)4, a.size());
  int* h = a.writeOnlyHostPtr();
  for (int i = 0; i < 4; i++) h[i] = i + 1;
  const int* d = a.readOnlyDevicePtr();  // uploads
  EXPECT_TRUE(d != nullptr);
  EXPECT_EQ(3, a.readOnlyHostPtr()[2]);  // host still valid, no copy back needed
  a.ptr();                               // read-write device: host goes stale
  EXPECT_EQ(4, a.readOnlyHostPtr()[3]);  // copied back from device
  int src[2] = {7, 8};
  a.copyFromHost(src, 2);  // re-sizes
  EXPECT_EQ((size_t)2, a.size());
  a.readOnlyDevicePtr();
  EXPECT_EQ(8, a.hostPtr()[1]);
  hemi::Array<double> z(3, false);  // never written: reads as zeros on both sides
  z.readOnlyDevicePtr();
  EXPECT_EQ(0.0, z.readOnlyHostPtr()[1]);
}

// The block pool the ensemble runners hold (device_array.h): inside a PoolScope a released array's blocks come back for
// the next array of the same size class -- zeroed on the host side like a fresh one, a never-written array still reads
// as zeros on both sides -- and without a scope arrays allocate and free as before.
TEST(DeviceArray, BlockPoolRecyclesBlocksInsideAScope) {
  const int* first_dev = nullptr;
  const int* first_host = nullptr;
  {
    sxmc::PoolScope scope;
    EXPECT_TRUE(sxmc::BlockPool::instance().active());
    {
      hemi::Array<int> a(1000, true);
      int* h = a.writeOnlyHostPtr();
      for (int i = 0; i < 1000; i++) h[i] = i + 1;
      first_host = h;
      first_dev = a.readOnlyDevicePtr();
      EXPECT_EQ(1000, a.readOnlyHostPtr()[999]);
    }
    {
      hemi::Array<int> b(900, true);   // same size class (4096 bytes): the same blocks, and they read as zeros
      EXPECT_TRUE(b.readOnlyDevicePtr() == first_dev);
      EXPECT_TRUE(b.readOnlyHostPtr() == first_host);
      EXPECT_EQ(0, b.readOnlyHostPtr()[899]);
      b.ptr();                          // (device side valid, host stale: copied back on the next host read)
      EXPECT_EQ(0, b.readOnlyHostPtr()[5]);
      hemi::Array<int> c(5000, true);  // another class: other blocks
      EXPECT_TRUE(c.readOnlyDevicePtr() != first_dev);
      {
        sxmc::PoolScope inner;         // scopes nest: the pool lives until the outermost ends
      }
      EXPECT_TRUE(sxmc::BlockPool::instance().active());
    }
  }
  EXPECT_TRUE(!sxmc::BlockPool::instance().active());
  hemi::Array<int> d(1000, true);      // no scope: a plain allocation, freed for real
  d.writeOnlyHostPtr()[3] = 9;
  d.readOnlyDevicePtr();
  EXPECT_EQ(9, d.hostPtr()[3]);
}

// ------------------------------------------------------------------ NLL launch points + MCMC driver
static unsigned lcg(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
static float uni(unsigned& s) { return (lcg(s) >> 8) * (1.0f / 16777216.0f); }

struct SmallFit {
  // 3 signals, 2 observables + truth + dataset, shift + resolution systematics
  void SetUp() {
    unsigned s = 12345;
    observables.resize(2);
    observables[0].field_index = 0; observables[0].bins = 12; observables[0].lower = 0; observables[0].upper = 1;
    observables[1].field_index = 1; observables[1].bins = 9; observables[1].lower = 0; observables[1].upper = 2;
    systematics.resize(2);
    systematics[0].name = "shift"; systematics[0].type = pdfz::Systematic::SHIFT;
    systematics[0].observable_field_index = 1; systematics[0].means = {0.0}; systematics[0].sigmas = {0.05};
    systematics[0].pidx = {0};
    systematics[1].name = "res"; systematics[1].type = pdfz::Systematic::RESOLUTION_SCALE;
    systematics[1].observable_field_index = 0; systematics[1].truth_field_index = 2;
    systematics[1].means = {0.0}; systematics[1].sigmas = {0.1}; systematics[1].pidx = {1};
    for (int j = 0; j < 3; j++) {
      const size_t n = 20000 + 777 * j;
      std::vector<float> tab(n * 4);
      for (size_t i = 0; i < n; i++) {
        const float t = 0.2f + 0.25f * j + 0.3f * uni(s);
        tab[i * 4 + 0] = t + 0.2f * (uni(s) - 0.5f);
        tab[i * 4 + 1] = 2.2f * uni(s) - 0.1f;
        tab[i * 4 + 2] = t;
        tab[i * 4 + 3] = 0;
      }
      sxmc::Signal sig;
      sig.name = "sig" + std::to_string(j);
      sig.source = sxmc::Source("src" + std::to_string(j), j, 1.0f, 0.0f, false);
      sig.nexpected = 100 + 50 * j;
      sxmc::build_pdfz(sig, tab, 4, observables, systematics);
      signals.push_back(sig);
      tables.push_back(tab);
      sources.push_back(sig.source);
      for (int e = 0; e < 150; e++) {
        const size_t i = lcg(s) % n;
        data.push_back(tab[i * 4 + 0]);
        data.push_back(tab[i * 4 + 1]);
        data.push_back(0);
      }
    }
  }
  void TearDown() {
    for (sxmc::Signal& s : signals) delete s.histogram;
  }
  std::vector<sxmc::Source> sources;
  std::vector<sxmc::Signal> signals;
  std::vector<sxmc::Systematic> systematics;
  std::vector<sxmc::Observable> observables;
  std::vector<float> data;
  std::vector<std::vector<float>> tables;   // host copies of the sample tables (replicas on other GPUs)
};

TEST_F(SmallFit, EfficiencyIsInDomainFraction) {
  const double eff = sxmc::get_efficiency(signals[0], systematics);
  EXPECT_TRUE(eff > 0.5 && eff < 1.0);
  size_t inside = 0;
  std::vector<float> sv;
  dynamic_cast<pdfz::EvalHist*>(signals[0].histogram)->GetSamples(sv);
  for (size_t i = 0; i < sv.size() / 3; i++) {
    if (sv[3 * i] >= 0 && sv[3 * i] < 1 && sv[3 * i + 1] >= 0 && sv[3 * i + 1] < 2) inside++;
  }
  ASSERT_NEAR_REL(eff, (double)inside / signals[0].n_mc, 1e-12);
}

TEST_F(SmallFit, BatchedAndReferenceFormsWalkTheSameChain) {
  // debug_mode accepts every step, so the walk depends only on the proposal stream: the batched
  // form (one fill kernel for all signals, fused lookup + event sum) and the reference's own call
  // sequence must visit the same points and agree on the NLL to summation-order precision.
  sxmc::MCMC a(sources, signals, systematics, observables, 99);
  sxmc::Chain ca = a(data, 40, 0.0f, true, 16);
  sxmc::MCMC b(sources, signals, systematics, observables, 99);
  b.reference_form = true;
  sxmc::Chain cb = b(data, 40, 0.0f, true, 16);
  EXPECT_EQ((size_t)40, ca.nrows());
  EXPECT_EQ((size_t)40, cb.nrows());
  EXPECT_EQ((size_t)40, ca.accepted);
  EXPECT_EQ((size_t)6, ca.names.size());
  EXPECT_EQ(std::string("likelihood"), ca.names.back());
  for (size_t r = 0; r < 40; r++) {
    for (size_t c = 0; c < 5; c++) EXPECT_EQ(ca.at(r, c), cb.at(r, c));
    ASSERT_NEAR_REL(ca.at(r, 5), cb.at(r, 5), 1e-6);
    EXPECT_TRUE(std::isfinite(ca.at(r, 5)));
  }
}

TEST_F(SmallFit, UnchangedCallSequenceIsBatchedBehindTheApi) {
  // mcmc.cpp:264-271 as written -- EvalAsync on every evaluator, then EvalFinished on every evaluator -- with no
  // group call: the library defers the S evaluations and launches them as ONE group evaluation.  The chain is the one
  // S separate launch sequences walk (sxmc_set_deferred_eval(0)), bit for bit, and the counters show one launch
  // sequence per step.
  unsigned long long l0 = 0, e0 = 0, l1 = 0, e1 = 0, l2 = 0, e2 = 0;
  sxmc::check(sxmc_deferred_eval_stats(&l0, &e0));
  sxmc::MCMC a(sources, signals, systematics, observables, 5);
  a.reference_form = true;
  sxmc::Chain ca = a(data, 120, 0.1f, false, 50);
  sxmc::check(sxmc_deferred_eval_stats(&l1, &e1));
  sxmc::check(sxmc_set_deferred_eval(0));
  sxmc::MCMC b(sources, signals, systematics, observables, 5);
  b.reference_form = true;
  sxmc::Chain cb = b(data, 120, 0.1f, false, 50);
  sxmc::check(sxmc_deferred_eval_stats(&l2, &e2));
  sxmc::check(sxmc_set_deferred_eval(1));
  // 120 steps + the first evaluation of each of the S evaluators (mcmc.cpp:238-239: EvalAsync, EvalFinished one by one)
  const size_t S = signals.size();
  EXPECT_EQ((unsigned long long)(120 * S + S), e1 - e0);
  EXPECT_EQ((unsigned long long)(120 + S), l1 - l0);
  EXPECT_EQ(l1, l2);
  EXPECT_EQ(e1, e2);
  EXPECT_EQ(ca.nrows(), cb.nrows());
  EXPECT_EQ(ca.accepted, cb.accepted);
  EXPECT_TRUE(ca.accepted > 3 && ca.accepted < 120);
  bool same = ca.rows.size() == cb.rows.size();
  for (size_t k = 0; same && k < ca.rows.size(); k++) same = ca.rows[k] == cb.rows[k];
  EXPECT_TRUE(same);
}

TEST_F(SmallFit, MetropolisWalkWithBurnIn) {
  sxmc::MCMC m(sources, signals, systematics, observables, 7);
  sxmc::Chain c = m(data, 600, 0.2f, false, 100);
  // steps before 2 * burnin_steps are dropped when the widths are re-tuned (mcmc.cpp:307-310)
  EXPECT_EQ((size_t)(600 - 240), c.nrows());
  EXPECT_TRUE(c.accepted > 10 && c.accepted < 600);
  double mean_rate0 = 0;
  for (size_t r = 0; r < c.nrows(); r++) mean_rate0 += c.at(r, 0);
  mean_rate0 /= c.nrows();
  EXPECT_TRUE(mean_rate0 > 0.0 && mean_rate0 < 5.0);
}

TEST_F(SmallFit, GraphReplayedStepsWalkTheSameChain) {
  // HIP-graph capture of the per-step sequence: same launches, same arguments, so the same chain
  // bit for bit, across re-tuning points, jump-buffer flushes and a remainder shorter than the graph.
  sxmc::MCMC a(sources, signals, systematics, observables, 7);
  sxmc::Chain ca = a(data, 333, 0.2f, false, 100);
  sxmc::MCMC b(sources, signals, systematics, observables, 7);
  b.graph_steps = 8;
  sxmc::Chain cb = b(data, 333, 0.2f, false, 100);
  EXPECT_EQ(ca.nrows(), cb.nrows());
  EXPECT_EQ(ca.accepted, cb.accepted);
  EXPECT_TRUE(ca.accepted > 5);
  bool same = ca.rows.size() == cb.rows.size();
  for (size_t k = 0; same && k < ca.rows.size(); k++) same = ca.rows[k] == cb.rows[k];
  EXPECT_TRUE(same);
}

TEST_F(SmallFit, LookaheadWalkIsTheSequentialChain) {
  // the look-ahead walk: two evaluations per pass over the tables, one or two steps per pass -- the same chain bit
  // for bit, across re-tuning points (the look-ahead vector is formed anew with the new widths), jump-buffer
  // flushes (an exact stop at every one) and graph replays, in fewer passes than steps
  sxmc::MCMC a(sources, signals, systematics, observables, 7);
  sxmc::Chain ca = a(data, 333, 0.2f, false, 100);
  for (unsigned gs : {0u, 6u}) {
    sxmc::MCMC b(sources, signals, systematics, observables, 7);
    b.lookahead = true;
    b.graph_steps = gs;
    sxmc::Chain cb = b(data, 333, 0.2f, false, 100);
    EXPECT_EQ(ca.nrows(), cb.nrows());
    EXPECT_EQ(ca.accepted, cb.accepted);
    bool same = ca.rows.size() == cb.rows.size();
    for (size_t k = 0; same && k < ca.rows.size(); k++) same = ca.rows[k] == cb.rows[k];
    EXPECT_TRUE(same);
    EXPECT_TRUE(b.LookaheadPasses() > 0 && b.LookaheadPasses() < 333);
  }
}

TEST_F(SmallFit, BoxedFormOfTheFillGivesTheSameHistograms) {
  // this fit has the shape the boxed form is for (one observable resolution-scaled against a truth field, one shifted):
  // forced on (sxmc_group_set_boxes 1) and off, the same bins and norms for several parameter vectors; the plan of a
  // table this small has ONE form (boxes do not pay), so sxmc_group_adapt_fill_form reports 0 and changes nothing
  std::vector<sxmc_hist_t> handles;
  pdfz::Array<double> params(2, true);
  pdfz::Array<unsigned> norms(signals.size(), true);
  for (size_t j = 0; j < signals.size(); j++) {
    pdfz::EvalHist* h = dynamic_cast<pdfz::EvalHist*>(signals[j].histogram);
    h->SetNormalizationBuffer(&norms, (int)j);
    h->SetParameterBuffer(&params, 0);
    h->Bind();
    handles.push_back(h->Handle());
  }
  sxmc_group_t g = nullptr;
  sxmc::check(sxmc_group_create(handles.data(), (int)handles.size(), &g));
  int form = -1, changed = -1;
  sxmc::check(sxmc_group_adapt_fill_form(g, &form, &changed));
  EXPECT_EQ(form, 0);
  EXPECT_EQ(changed, 0);
  EXPECT_TRUE(sxmc_group_set_fill_form(g, 1) != SXMC_OK);
  const double sets[4][2] = {{0.0, 0.0}, {0.03, 0.08}, {-0.2, -0.5}, {0.01, 2.5}};
  for (const auto& p : sets) {
    params.writeOnlyHostPtr()[0] = p[0];
    params.writeOnlyHostPtr()[1] = p[1];
    (void)params.readOnlyPtr();
    std::vector<std::vector<unsigned>> got[2];
    std::vector<unsigned> nrm[2];
    for (int boxes = 1; boxes >= 0; boxes--) {
      sxmc::check(sxmc_group_set_boxes(g, boxes));
      char info[1024];
      sxmc::check(sxmc_group_launch_info(g, info, sizeof info));
      EXPECT_EQ(std::string(info).find("boxed+codes") != std::string::npos, boxes == 1);
      sxmc::check(sxmc_group_eval_async(g, 0, nullptr));
      sxmc::check(sxmc_group_synchronize(g));
      for (sxmc_hist_t h : handles) {
        int nb = 0;
        sxmc::check(sxmc_hist_total_nbins(h, &nb));
        std::vector<unsigned> bins((size_t)nb);
        sxmc::check(sxmc_hist_get_bins(h, bins.data(), bins.size()));
        got[boxes].push_back(bins);
      }
      const unsigned* n = norms.readOnlyHostPtr();
      nrm[boxes].assign(n, n + signals.size());
    }
    EXPECT_TRUE(got[0] == got[1]);
    EXPECT_TRUE(nrm[0] == nrm[1]);
    EXPECT_TRUE(nrm[1][0] > 0u);
  }
  sxmc::check(sxmc_group_destroy(g));
}

TEST_F(SmallFit, AutoWalkTakesTheLookaheadPassOnlyWhereThePlanStreamsFloatColumns) {
  // MCMC::lookahead_auto: the walk decides -- the look-ahead pass where the fill streams float columns (this small
  // fit), one evaluation per step where it streams codes; the chain is the sequential one either way
  sxmc::MCMC a(sources, signals, systematics, observables, 7);
  sxmc::Chain ca = a(data, 200, 0.2f, false, 100);
  sxmc::MCMC b(sources, signals, systematics, observables, 7);
  b.lookahead_auto = true;
  sxmc::Chain cb = b(data, 200, 0.2f, false, 100);
  EXPECT_EQ(ca.accepted, cb.accepted);
  bool same = ca.rows.size() == cb.rows.size();
  for (size_t k = 0; same && k < ca.rows.size(); k++) same = ca.rows[k] == cb.rows[k];
  EXPECT_TRUE(same);
  // what the plan of this fit's evaluators streams decides which walk it was
  std::vector<sxmc_hist_t> handles;
  for (const sxmc::Signal& s : signals) handles.push_back(dynamic_cast<pdfz::EvalHist*>(s.histogram)->Handle());
  sxmc_group_t g = nullptr;
  sxmc::check(sxmc_group_create(handles.data(), (int)handles.size(), &g));
  int members = 0, supported = 0;
  unsigned long long rows = 0, exact_rows = 0, never_rows = 0;
  sxmc::check(sxmc_group_set_lut_output(g, 0));
  sxmc::check(sxmc_group_codes_info(g, &members, &rows, &exact_rows, &never_rows));
  sxmc::check(sxmc_group_lookahead_supported(g, &supported));
  sxmc::check(sxmc_group_destroy(g));
  EXPECT_EQ(b.LookaheadPasses() > 0, members == 0 && supported != 0);
}

TEST(LaneBarrier, RoundsOfUnequalSizeAndABrokenBarrier) {
  // the meeting point of ensemble_concurrent's lanes: every round's participants leave together, a last round with
  // fewer lanes works, and a lane that fails releases everybody for good
  sxmc::LaneBarrier meet;
  const size_t lanes = 4, experiments = 10;     // rounds of 4, 4 and 2
  std::atomic<int> inside{0}, worst{0};
  std::vector<std::thread> threads;
  for (size_t t = 0; t < lanes; t++) {
    threads.emplace_back([&, t]() {
      for (size_t i = t; i < experiments; i += lanes) {
        const size_t round_lanes = std::min(lanes, experiments - (i - t));
        meet.arrive_and_wait(round_lanes);      // "all set up"
        const int now = ++inside;
        int seen = worst.load();
        while (now > seen && !worst.compare_exchange_weak(seen, now)) {
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2 + (int)t));
        --inside;
        meet.arrive_and_wait(round_lanes);      // "all stepped"
      }
    });
  }
  for (std::thread& th : threads) th.join();
  EXPECT_TRUE(worst.load() >= 2 && worst.load() <= 4);
  // a waiter is released by break_all, and nobody waits afterwards
  std::atomic<bool> released{false};
  std::thread waiter([&]() {
    meet.arrive_and_wait(2);
    released = true;
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  EXPECT_TRUE(!released.load());
  meet.break_all();
  waiter.join();
  EXPECT_TRUE(released.load());
  meet.arrive_and_wait(5);                       // returns at once
}

TEST(NllLaunch, ReferenceSpelling) {
  // the launch macro with the reference's argument order (mcmc.cpp:396-414)
  const size_t ne = 5, ns = 2, np = 2;
  hemi::Array<float> lut(ne * ns, true);
  for (size_t i = 0; i < ne * ns; i++) lut.writeOnlyHostPtr()[i] = 0.5f + 0.1f * i;
  hemi::Array<double> pars(np, true), means(np, true), sigmas(np, true), nexp(ns, true);
  hemi::Array<unsigned> n_mc(ns, true), norms(ns, true);
  hemi::Array<short> sid(ns, true);
  for (size_t j = 0; j < ns; j++) {
    pars.writeOnlyHostPtr()[j] = 1.0 + j;
    means.writeOnlyHostPtr()[j] = 1.0;
    sigmas.writeOnlyHostPtr()[j] = 0.0;
    nexp.writeOnlyHostPtr()[j] = 10.0 * (j + 1);
    n_mc.writeOnlyHostPtr()[j] = 1000;
    norms.writeOnlyHostPtr()[j] = 500;
    sid.writeOnlyHostPtr()[j] = (short)j;
  }
  hemi::Array<double> sums(64 * 256, true), total(1, true), nll(1, true);
  sums.writeOnlyHostPtr();
  HEMI_KERNEL_LAUNCH(nll_event_chunks, 64, 256, 0, 0, lut.readOnlyPtr(), pars.readOnlyPtr(), ne, ns,
                     nexp.readOnlyPtr(), n_mc.readOnlyPtr(), sid.readOnlyPtr(), norms.readOnlyPtr(), sums.ptr());
  HEMI_KERNEL_LAUNCH(nll_event_reduce, 1, 128, 128 * sizeof(double), 0, (size_t)(64 * 256), sums.ptr(),
                     total.ptr());
  HEMI_KERNEL_LAUNCH(nll_total, 1, 1, 0, 0, np, pars.readOnlyPtr(), ns, np, means.readOnlyPtr(),
                     sigmas.readOnlyPtr(), total.ptr(), nexp.readOnlyPtr(), n_mc.readOnlyPtr(),
                     sid.readOnlyPtr(), norms.readOnlyPtr(), nll.ptr());
  double ev = 0;
  for (size_t i = 0; i < ne; i++) {
    double s = 0;
    for (size_t j = 0; j < ns; j++) s += (1.0 + j) * 10.0 * (j + 1) * 0.5f * (0.5f + 0.1f * (j * ne + i));
    ev += std::log(s);
  }
  const double want = -ev + 1.0 * 10.0 * 500 / 1000 + 2.0 * 20.0 * 500 / 1000;
  ASSERT_NEAR_REL(total.readOnlyHostPtr()[0], ev, 1e-6);
  ASSERT_NEAR_REL(nll.readOnlyHostPtr()[0], want, 1e-6);
}

// ------------------------------------------------------------------ ensemble layer (sxmc.cpp:44-145)
TEST(Ensemble, ChisquareQuantileAndMedian) {
  ASSERT_NEAR_REL(sxmc::chisquare_quantile_1dof(0.9), 2.705543454, 1e-8);
  ASSERT_NEAR_REL(sxmc::chisquare_quantile_1dof(0.95), 3.841458821, 1e-8);
  EXPECT_EQ(2.0f, sxmc::median(std::vector<float>{3, 1, 2}));
  EXPECT_EQ(2.5f, sxmc::median(std::vector<float>{4, 1, 3, 2}));
}

TEST(Ensemble, ContourIntervalOnAParabola) {
  sxmc::Chain c;
  c.names = {"x", "likelihood"};
  for (int i = 0; i <= 6000; i++) {
    const float x = 0.001f * i;
    c.rows.push_back(x);
    c.rows.push_back(0.5f * ((x - 3.0f) / 0.5f) * ((x - 3.0f) / 0.5f) + 10.0f);
  }
  std::vector<sxmc::Interval> iv = sxmc::contour_intervals(c, 0.9f);
  const double half = 0.5 * std::sqrt(2.705543454);
  EXPECT_TRUE(std::fabs(iv[0].point_estimate - 3.0) < 2e-3);
  EXPECT_TRUE(std::fabs(iv[0].lower - (3.0 - half)) < 2e-3);
  EXPECT_TRUE(std::fabs(iv[0].upper - (3.0 + half)) < 2e-3);
  EXPECT_EQ(-999.0f, iv[0].coverage);
}

static sxmc::Chain chain_of(std::initializer_list<std::pair<float, float>> rows) {
  sxmc::Chain c;
  c.names = {"x", "likelihood"};
  for (const auto& r : rows) {
    c.rows.push_back(r.first);
    c.rows.push_back(r.second);
  }
  return c;
}

TEST(Ensemble, ContourKnownAnswers) {
  // (the same three cases as tests/test_ensemble_cpu.py, against contour.cpp:17-69 / likelihood.cpp:90-102)
  // 1. the threshold is delta = 0.5 * chi2_1(0.9) = 1.35277 and strict
  std::vector<sxmc::Interval> iv =
      sxmc::contour_intervals(chain_of({{1.0f, 10.0f}, {2.0f, 10.0f + 1.35277f - 1e-3f},
                                        {3.0f, 10.0f + 1.35277f + 1e-3f}, {0.5f, 10.05f}}), 0.9f);
  EXPECT_EQ(0.5f, iv[0].lower);
  EXPECT_EQ(2.0f, iv[0].upper);
  EXPECT_EQ(0.75f, iv[0].point_estimate);
  EXPECT_EQ(-999.0f, iv[0].coverage);
  // 2. -lmin goes into the selection as text with 6 significant digits: 348086.3125 -> "348086"
  EXPECT_EQ(348086.0, sxmc::as_printed(348086.3125f));
  const float l2 = -348086.3125f;
  iv = sxmc::contour_intervals(chain_of({{0.0f, l2}, {5.0f, l2 + 1.5625f}, {7.0f, l2 + 1.6875f}}), 0.9f);
  EXPECT_EQ(0.0f, iv[0].lower);
  EXPECT_EQ(5.0f, iv[0].upper);
  EXPECT_EQ(0.0f, iv[0].point_estimate);
  // 3. the 0.13 * 5^k widening: with lmin = -348085.6875 the first pass finds nothing
  const float l3 = -348085.6875f;
  iv = sxmc::contour_intervals(chain_of({{0.0f, l3}, {4.0f, l3 + 0.25f}, {9.0f, l3 + 0.9375f}, {20.0f, l3 + 1.0625f}}),
                               0.9f);
  EXPECT_EQ(2.0f, iv[0].point_estimate);
  EXPECT_EQ(0.0f, iv[0].lower);
  EXPECT_EQ(9.0f, iv[0].upper);
}

TEST(Ensemble, GausFitAndProjectionKnownAnswers) {
  // bin contents that ARE a Gaussian at the bin centres: chi2 = 0 at (A, mu, sigma), which the fit must find
  std::vector<double> x, y;
  for (int i = 0; i < 120; i++) x.push_back(-2.95 + 0.1 * i);
  for (double v : x) y.push_back(1000.0 * std::exp(-0.5 * ((v - 3.2) / 0.9) * ((v - 3.2) / 0.9)));
  double a = 0, mu = 0, sigma = 0;
  EXPECT_TRUE(sxmc::gaus_fit(x, y, a, mu, sigma));
  EXPECT_TRUE(std::fabs(a - 1000.0) < 1e-3 && std::fabs(mu - 3.2) < 1e-8 && std::fabs(sigma - 0.9) < 1e-8);
  EXPECT_TRUE(!sxmc::gaus_fit({0.0, 1.0}, {5.0, 5.0}, a, mu, sigma));
  // samples of N(5, 1): central interval around the fitted mean (projection.cpp:47-66)
  std::mt19937_64 rng(1);
  std::normal_distribution<float> gauss(5.0f, 1.0f);
  std::vector<float> v;
  for (int i = 0; i < 200000; i++) v.push_back(gauss(rng));
  sxmc::Interval iv = sxmc::projection_interval(v, 0.9f);
  EXPECT_TRUE(!iv.one_sided && std::fabs(iv.point_estimate - 5.0f) < 0.02f);
  EXPECT_TRUE(std::fabs(iv.lower - (5.0f - 1.645f)) < 0.2f && iv.upper > 5.0f + 1.5f && iv.upper < 5.0f + 2.1f);
  EXPECT_TRUE(iv.coverage >= 0.9f && iv.coverage < 0.96f);
  for (float& t : v) t = std::fabs(t - 5.0f);   // piled up at its lower boundary: one-sided (projection.cpp:36-45)
  iv = sxmc::projection_interval(v, 0.9f);
  EXPECT_TRUE(iv.one_sided && iv.lower <= 1e-3f && iv.coverage >= 0.9f && std::fabs(iv.upper - 1.645f) < 0.1f);
}

TEST_F(SmallFit, FakeDatasetAndWholeExperiments) {
  std::mt19937_64 rng(5);
  std::vector<unsigned> observed;
  std::vector<float> ev = sxmc::make_fake_dataset(rng, signals, systematics, observables, false, &observed);
  size_t total = 0;
  for (size_t j = 0; j < signals.size(); j++) {
    const double eff = sxmc::get_efficiency(signals[j], systematics);
    EXPECT_EQ((unsigned)std::floor(signals[j].nexpected * eff + 0.5), observed[j]);
    total += observed[j];
  }
  EXPECT_EQ(total * 3, ev.size());
  for (size_t i = 0; i < total; i++) {
    EXPECT_TRUE(ev[3 * i] >= 0 && ev[3 * i] < 1 && ev[3 * i + 1] >= 0 && ev[3 * i + 1] < 2);
    EXPECT_EQ(0.0f, ev[3 * i + 2]);
  }
  // two whole fake experiments (this rank's share of an ensemble): fake data -> MCMC -> contour intervals
  std::vector<sxmc::ExperimentResult> res =
      sxmc::ensemble({0u, 3u}, 11, sources, signals, systematics, observables, 800, 0.2f, 0.9f, 200);
  EXPECT_EQ((size_t)2, res.size());
  EXPECT_EQ(3u, res[1].index);
  for (const sxmc::ExperimentResult& r : res) {
    EXPECT_EQ((size_t)5, r.intervals.size());
    EXPECT_TRUE(r.accepted > 5 && r.nevents > 100);
    for (const sxmc::Interval& iv : r.intervals) {
      EXPECT_TRUE(iv.lower <= iv.point_estimate && iv.point_estimate <= iv.upper);
      EXPECT_TRUE(std::isfinite(iv.lower) && std::isfinite(iv.upper));
    }
    EXPECT_TRUE(r.intervals[0].lower >= 0);  // negative rates are rejected by the NLL
  }
  EXPECT_TRUE(res[0].intervals[0].upper != res[1].intervals[0].upper);  // different data sets
}

TEST_F(SmallFit, ConcurrentExperimentsMatchSequential) {
  // three experiments in flight (a host thread, a non-blocking stream and evaluators sharing the sample
  // tables each; steps replayed from HIP graphs): the intervals of the one-at-a-time loop, exactly
  const std::vector<unsigned> ks = {0u, 3u, 5u, 6u, 9u};
  std::vector<sxmc::ExperimentResult> seq =
      sxmc::ensemble(ks, 11, sources, signals, systematics, observables, 400, 0.2f, 0.9f, 100);
  std::vector<sxmc::ExperimentResult> par = sxmc::ensemble_concurrent(ks, 11, sources, signals, systematics,
                                                                      observables, 400, 0.2f, 3, 0.9f, 100, 8);
  EXPECT_EQ(seq.size(), par.size());
  for (size_t i = 0; i < seq.size(); i++) {
    EXPECT_EQ(seq[i].index, par[i].index);
    EXPECT_EQ(seq[i].accepted, par[i].accepted);
    EXPECT_EQ(seq[i].nevents, par[i].nevents);
    for (size_t p = 0; p < seq[i].intervals.size(); p++) {
      EXPECT_EQ(seq[i].intervals[p].lower, par[i].intervals[p].lower);
      EXPECT_EQ(seq[i].intervals[p].upper, par[i].intervals[p].upper);
      EXPECT_EQ(seq[i].intervals[p].point_estimate, par[i].intervals[p].point_estimate);
    }
    // every experiment says where its host time went; the pool the runners hold is gone when they return
    EXPECT_TRUE(par[i].phases.steps > 0 && par[i].phases.data > 0 && par[i].phases.walk_setup > 0);
    EXPECT_TRUE(par[i].phases.walk_teardown >= 0 && par[i].phases.intervals >= 0);
  }
  EXPECT_TRUE(!sxmc::BlockPool::instance().active());
}

TEST_F(SmallFit, LockstepExperimentsMatchSequential) {
  // nine experiments: two rounds of 2 sets x 2 chains advanced together (one pass over the sample tables per step
  // and set, sxmc_multigroup_step_async) + one left over: the intervals of the one-at-a-time loop, exactly
  const std::vector<unsigned> ks = {0u, 1u, 2u, 3u, 4u, 5u, 6u, 7u, 8u};
  std::vector<sxmc::ExperimentResult> seq =
      sxmc::ensemble(ks, 31, sources, signals, systematics, observables, 300, 0.2f, 0.9f, 100);
  std::vector<sxmc::ExperimentResult> par =
      sxmc::ensemble_lockstep(ks, 31, sources, signals, systematics, observables, 300, 0.2f, 2, 2, 0.9f, 100, 8);
  EXPECT_EQ(seq.size(), par.size());
  for (size_t i = 0; i < seq.size(); i++) {
    EXPECT_EQ(seq[i].index, par[i].index);
    EXPECT_EQ(seq[i].accepted, par[i].accepted);
    EXPECT_EQ(seq[i].nevents, par[i].nevents);
    for (size_t p = 0; p < seq[i].intervals.size(); p++) {
      EXPECT_EQ(seq[i].intervals[p].lower, par[i].intervals[p].lower);
      EXPECT_EQ(seq[i].intervals[p].upper, par[i].intervals[p].upper);
      EXPECT_EQ(seq[i].intervals[p].point_estimate, par[i].intervals[p].point_estimate);
    }
  }
}

TEST_F(SmallFit, MultiGpuRunnerGathersTheIntervalsThroughRccl) {
  // sxmc::ensemble_multi_gpu on the GPUs of this box (one here): a host thread per device with its own replica
  // of the evaluators, experiment k on device k mod G, ONE RCCL all-gather of the intervals at the end.  What
  // comes back through the all-gather must be the intervals of the one-at-a-time loop, bit for bit.
  int ndev = 0;
  ASSERT_EQ(SXMC_OK, sxmc_device_count(&ndev));
  std::vector<int> devices;
  for (int d = 0; d < ndev && d < 8; d++) devices.push_back(d);
  const unsigned N = 9;   // per device: one round of 2 lockstep sets x 4 chains + one experiment left over
  std::vector<unsigned> ks;
  for (unsigned k = 0; k < N; k++) ks.push_back(k);
  std::vector<sxmc::ExperimentResult> seq =
      sxmc::ensemble(ks, 21, sources, signals, systematics, observables, 400, 0.2f, 0.9f, 100);
  std::vector<const std::vector<float>*> tabs;
  for (const std::vector<float>& t : tables) tabs.push_back(&t);
  sxmc::MultiGpuEnsemble mg = sxmc::ensemble_multi_gpu(devices, N, 21, sources, signals, tabs, 4, systematics,
                                                       observables, 400, 0.2f, 2, 0.9f, 100, 8);
  EXPECT_EQ((size_t)5, mg.nparameters);
  EXPECT_EQ((size_t)N * 5 * 4, mg.gathered.size());
  for (unsigned k = 0; k < N; k++) {
    EXPECT_EQ(k, mg.results[k].index);
    EXPECT_EQ(seq[k].accepted, mg.results[k].accepted);
    for (size_t p = 0; p < 5; p++) {
      const float* g = &mg.gathered[((size_t)k * 5 + p) * 4];
      EXPECT_EQ(seq[k].intervals[p].point_estimate, g[0]);
      EXPECT_EQ(seq[k].intervals[p].lower, g[1]);
      EXPECT_EQ(seq[k].intervals[p].upper, g[2]);
      EXPECT_EQ(seq[k].intervals[p].coverage, g[3]);
    }
  }
  std::vector<float> ups;
  for (unsigned k = 0; k < N; k++) ups.push_back(seq[k].intervals[0].upper);
  EXPECT_EQ(sxmc::median(ups), mg.median_upper[0]);
  EXPECT_EQ(2.5f, sxmc::median(std::vector<float>{1.0f, 4.0f, 2.0f, 3.0f}));   // utils.h:76-90: mean of the two middle ones
}

TEST_F(SmallFit, MultiGpuRunnerWithTwoLogicalRanksOnOneCard) {
  // G = 2 device threads, both on card 0, through everything but the RCCL call (RCCL refuses two ranks on one card;
  // MultiGpuOptions::HOST_STAGING stands in for the all-gather): sharding k mod G, a replica of the evaluators per
  // thread, the rendezvous, the order of the results, the medians.  Three ranks as well (uneven shares, padded blocks).
  // ... and EIGHT (the node's shape, BASELINE config 4: every rank with two or three experiments), with one lock for the
  // process (the default) and with one per card.
  // -8: one lock per card (one card here: one lock); -108: one lock PER RANK -- eight host threads setting up, stepping
  // and tearing down side by side on the same device, only graph recording exclusive (what PER_DEVICE allows between
  // cards, exercised on one).
  for (int G : {2, 3, 8, -8, -108}) {
    const bool per_device = G == -8, per_rank = G == -108;
    G = G < 0 ? 8 : G;
    const unsigned N = G == 8 ? 19 : 7;
    std::vector<unsigned> ks;
    for (unsigned k = 0; k < N; k++) ks.push_back(k);
    std::vector<sxmc::ExperimentResult> seq =
        sxmc::ensemble(ks, 21, sources, signals, systematics, observables, 300, 0.2f, 0.9f, 100);
    std::vector<const std::vector<float>*> tabs;
    for (const std::vector<float>& t : tables) tabs.push_back(&t);
    sxmc::MultiGpuOptions opt;
    opt.sync_interval = 100;
    opt.graph_steps = 8;
    opt.lockstep_chains = 2;
    opt.lockstep_sets = 1;
    opt.exchange = sxmc::MultiGpuOptions::HOST_STAGING;
    if (per_device) opt.locking = sxmc::MultiGpuOptions::PER_DEVICE;
    if (per_rank) opt.locking = sxmc::MultiGpuOptions::PER_RANK;
    sxmc::MultiGpuEnsemble mg = sxmc::ensemble_multi_gpu(std::vector<int>((size_t)G, 0), N, 21, sources, signals, tabs, 4,
                                                         systematics, observables, 300, 0.2f, opt);
    EXPECT_EQ(0, mg.rccl_nranks);
    EXPECT_EQ(per_rank ? (size_t)G : (size_t)1, mg.setup_locks.size());   // one card: one lock shared by the ranks
    if (!per_rank) EXPECT_EQ(per_device ? 0 : -1, mg.setup_locks[0].device);   // (-1: the process's lock)
    EXPECT_TRUE(mg.setup_locks[0].acquisitions > 0 && mg.setup_locks[0].held_seconds > 0);
    EXPECT_EQ((size_t)G, mg.rank_seconds.size());
    for (unsigned k = 0; k < N; k++) {
      EXPECT_EQ(k, mg.results[k].index);
      EXPECT_EQ(seq[k].accepted, mg.results[k].accepted);
      for (size_t p = 0; p < 5; p++) {
        const float* g = &mg.gathered[((size_t)k * 5 + p) * 4];
        EXPECT_EQ(seq[k].intervals[p].point_estimate, g[0]);
        EXPECT_EQ(seq[k].intervals[p].lower, g[1]);
        EXPECT_EQ(seq[k].intervals[p].upper, g[2]);
        EXPECT_EQ(seq[k].intervals[p].coverage, g[3]);
      }
    }
    for (size_t p = 0; p < 5; p++) {
      std::vector<float> ups;
      for (unsigned k = 0; k < N; k++) ups.push_back(seq[k].intervals[p].upper);
      EXPECT_EQ(sxmc::median(ups), mg.median_upper[p]);   // utils.h:76-90
    }
  }
}

TEST_F(SmallFit, MultiGpuRunnerFailsFastWhenOneRankFails) {
  // a rank that fails before the exchange must not leave the others waiting in it: the ranks meet on the host first,
  // nobody enters the collective, the failing rank's error comes back -- with host staging (2 ranks on one card) and
  // on the real RCCL path (the ranks this box has)
  std::vector<const std::vector<float>*> tabs;
  for (const std::vector<float>& t : tables) tabs.push_back(&t);
  int ndev = 0;
  ASSERT_EQ(SXMC_OK, sxmc_device_count(&ndev));
  for (int pass = 0; pass < 3; pass++) {   // (pass 2: eight logical ranks on card 0, a rank in the middle fails)
    sxmc::MultiGpuOptions opt;
    opt.sync_interval = 100;
    opt.graph_steps = 8;
    opt.lockstep_chains = 0;
    opt.nconcurrent = 2;
    opt.exchange = pass == 1 ? sxmc::MultiGpuOptions::RCCL : sxmc::MultiGpuOptions::HOST_STAGING;
    std::vector<int> devices;
    if (pass == 0) devices = {0, 0};
    else if (pass == 2) devices.assign(8, 0);
    else for (int d = 0; d < ndev && d < 8; d++) devices.push_back(d);
    const size_t bad = pass == 2 ? 5 : devices.size() - 1;
    opt.before_exchange = [bad](size_t r) {
      if (r == bad) throw pdfz::Error("injected failure");
    };
    const auto t0 = std::chrono::steady_clock::now();
    bool threw = false;
    std::string msg;
    try {
      sxmc::ensemble_multi_gpu(devices, pass == 2 ? 10 : 4, 21, sources, signals, tabs, 4, systematics, observables, 200,
                               0.2f, opt);
    } catch (const pdfz::Error& e) {
      threw = true;
      msg = e.msg;
    }
    EXPECT_TRUE(threw);
    EXPECT_TRUE(msg.find("injected failure") != std::string::npos);
    EXPECT_TRUE(msg.find("rank " + std::to_string(bad)) != std::string::npos);
    EXPECT_TRUE(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 60.0);
  }
  // a failure while the replicas are built (a table that is not rows of nfields floats): same outcome
  {
    std::vector<float> broken(tables[0].begin(), tables[0].begin() + 4 * 100 + 1);
    std::vector<const std::vector<float>*> tabs2 = tabs;
    tabs2[1] = &broken;
    sxmc::MultiGpuOptions opt;
    opt.exchange = sxmc::MultiGpuOptions::HOST_STAGING;
    ASSERT_THROW(sxmc::ensemble_multi_gpu({0, 0}, 4, 21, sources, signals, tabs2, 4, systematics, observables, 200, 0.2f,
                                          opt),
                 pdfz::Error);
  }
  // and the runner still works afterwards (nothing was left locked, recording or half torn down)
  sxmc::MultiGpuOptions opt;
  opt.sync_interval = 100;
  opt.exchange = sxmc::MultiGpuOptions::HOST_STAGING;
  sxmc::MultiGpuEnsemble mg =
      sxmc::ensemble_multi_gpu({0, 0}, 2, 21, sources, signals, tabs, 4, systematics, observables, 200, 0.2f, opt);
  EXPECT_EQ((size_t)2, mg.results.size());
  EXPECT_EQ(1u, mg.results[1].index);
}

int main(int argc, char** argv) {
  int ndev = 0;
  if (sxmc_device_count(&ndev) != SXMC_OK || ndev < 1) {
    std::printf("no GPU: %s\n", sxmc_last_error());
    return 2;
  }
  return mini::run_all(argc > 1 ? argv[1] : nullptr);
}
