// sxmc_evaluator.cpp -- the evaluator behind pdfz::EvalHist (sxmc_hist_*: pdfz.cpp:179-302, 441-495) and the deferred batches that turn an
// unchanged caller's S x EvalAsync, S x EvalFinished into one launch sequence per step.
#include "sxmc_host.h"

using namespace sxhost;

namespace sxhost {

// ------------------------------------------------------------------------------ deferred evaluations
// The reference's caller evaluates its S signals one by one: EvalAsync on every evaluator, then EvalFinished on every
// evaluator (mcmc.cpp:264-271, bench_sxmc.cpp:193-200) -- S x (zero, fill, lookup) on S streams.  Launched as asked,
// that is S full-grid fills with S fixed costs.  sxmc_hist_eval_async therefore DEFERS: the evaluator joins the calling
// thread's batch, and the batch goes to the device as ONE group launch (zero, ONE fill over all members, lookup: what
// sxmc_group_eval_async does) either when the last sibling of a batch seen before arrives -- so the device works while
// the caller moves on to its EvalFinished calls, as with the reference -- or at the first call that could observe
// the difference: sxmc_hist_eval_finished, any copy, launch, synchronisation or change of a member.  An unchanged
// caller of the reference's sequence gets the batched fill; results are those of the separate launches bit for bit
// (integer counters; the lookup is per evaluator either way).
std::mutex g_auto_mutex;
std::vector<AutoGroup> g_auto_groups;     // groups made for batches of two or more (a single evaluator has h->self)
std::atomic<int> g_defer{-1};             // -1: ask the environment (SXMC_DEFER_EVAL=0 switches deferral off)
thread_local std::shared_ptr<DeferredBatch> t_deferred;
thread_local bool t_flushing = false;

bool deferral_enabled() {
  int v = g_defer.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("SXMC_DEFER_EVAL");
    v = (e && e[0] == '0') ? 0 : 1;
    g_defer.store(v, std::memory_order_relaxed);
  }
  return v > 0;
}

std::atomic<unsigned long long> g_deferred_launches{0}, g_deferred_evaluations{0};

// Where a batch is launched.  The reference's evaluators each launch on a stream of their own, and its caller's NLL
// kernels go to the legacy default stream (HEMI_KERNEL_LAUNCH(..., 0, 0, ...), mcmc.cpp:314-348), which orders with
// those streams implicitly.  A batch is ONE launch sequence, so there is nothing for separate streams to overlap, and
// every hop between an evaluator's stream and the legacy stream costs the runtime a cross-queue dependency per step.
// A batch of two or more therefore goes to the legacy default stream itself -- ordered with everything the reference's
// streams would be ordered with, and with the caller's NLL kernels by plain queue order.  SXMC_DEFER_STREAM=own
// launches on the first member's stream instead (measurement).
// (While ANY host thread of the process records a graph the runtime refuses work on the legacy stream -- "operation
// would make the legacy stream depend on a capturing blocking stream" -- so a batch that falls into such a moment goes
// to the first member's stream and is waited for there: slower, never wrong.)
// The gate between the two: a recording holds it SHARED from sxmc_graph_begin_capture to sxmc_graph_end_capture (several
// threads may record at once), a batch that goes to the legacy stream holds it EXCLUSIVE while it is being launched --
// so no recording can begin between the decision "the legacy stream is free" and the launches that rely on it (round 4
// read a counter and launched afterwards: ADVICE r4).  A batch that cannot have the gate at once goes to its first
// member's stream.
std::shared_mutex g_capture_gate;
hipStream_t batch_stream(const std::vector<sxmc_hist*>& m, std::unique_lock<std::shared_mutex>& legacy_hold) {
  static const bool own = [] {
    const char* e = measure_env("SXMC_DEFER_STREAM");
    return e && std::string(e) == "own";
  }();
  if (m.size() < 2 || own || t_capturing) return m[0]->stream;
  legacy_hold = std::unique_lock<std::shared_mutex>(g_capture_gate, std::try_to_lock);
  return legacy_hold.owns_lock() ? nullptr : m[0]->stream;
}

// EvalFinished's wait: hipStreamSynchronize.  (An MCMC step waits once per evaluation, mcmc.cpp:268-270, so polling
// hipStreamQuery first was tried -- SXMC_FINISH_SPIN_US microseconds of it, default 0 -- and measured no faster at
// BASELINE config 3, 4 450-4 480 against 4 530-4 570 steps/s: the runtime's own wait already spins.)
hipError_t wait_for_stream(hipStream_t s) {
  static const long spin_us = [] {
    const char* e = measure_env("SXMC_FINISH_SPIN_US");
    return e ? std::atol(e) : 0L;
  }();
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) return hipSuccess;
      if (e != hipErrorNotReady) return e;
      (void)hipGetLastError();
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
  }
  return hipStreamSynchronize(s);
}

// (A batch as a GRAPH -- its zero, fill and lookup recorded once and replayed into the legacy stream -- was built and
// measured against the three launches one by one, alternating on one box: 8 490-8 540 steps/s of the unchanged caller's
// walk at config 3 against 8 910-8 940.  hipGraphLaunch into the legacy stream costs more than the two kernel boundaries
// it saves; the batch goes out launch by launch.  profiles/r05_dropin_ab.log.)
// Launches the calling thread's batch, if any.  Every entry point that could observe a deferred evaluation calls this
// first (SX_FLUSH).
int flush_deferred() {
  if (!t_deferred || t_deferred->n.load(std::memory_order_relaxed) == 0 || t_flushing) return SXMC_OK;
  std::vector<sxmc_hist*> m;
  int do_eval_pdf = 0;
  {
    std::lock_guard<std::mutex> lock(t_deferred->m);
    m.swap(t_deferred->members);
    t_deferred->n.store(0, std::memory_order_relaxed);
    do_eval_pdf = t_deferred->do_eval_pdf;
    for (sxmc_hist* h : m) h->deferred.reset();
  }
  if (m.empty()) return SXMC_OK;
  sxmc_group* g = nullptr;
  if (m.size() == 1) {
    if (!m[0]->self) {
      int rc = sxmc_group_create(&m[0], 1, &m[0]->self);
      if (rc) return rc;
    }
    g = m[0]->self;
  } else {
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    for (AutoGroup& a : g_auto_groups)
      if (a.members == m) g = a.g;
    if (!g) {
      int rc = sxmc_group_create(m.data(), (int)m.size(), &g);
      if (rc) return rc;
      g_auto_groups.push_back(AutoGroup{m, g});
    }
  }
  auto fl = std::make_shared<BatchInFlight>();
  std::unique_lock<std::shared_mutex> legacy_hold;     // (held, when the batch goes to the legacy stream, until it is launched)
  fl->stream = batch_stream(m, legacy_hold);
  for (sxmc_hist* h : m) fl->host_visible = fl->host_visible || h->outputs_host_visible;
  t_flushing = true;    // (the group calls below are themselves flush points)
  int rc = SXMC_OK;
  // EvalHist's `optimize` (pdfz.cpp:188, 441-448, 622-628): the reference's evaluator runs its launch-shape trials
  // inside its first EvalAsync, once it has evaluation points, unless it was constructed with optimize = false -- and
  // never while making a histogram (pdfz.cpp:503-504: CreateHistogram switches it off around its EvalAsync(false)).
  // Here the trials are the BATCH's (sxmc_group_optimize: a few timed fills choose lanes per CU, teams and codes for
  // this box; only a long pure stream has anything to choose, it returns at once otherwise): at the batch's first
  // lookup evaluation, when every member asks for it.  The evaluation proper follows and zeroes what the trials counted.
  // A plan with a boxed and an ordered form of the fill (sxmc_group_adapt_fill_form): the unchanged caller has no flush
  // to ask at, so its batches ask themselves -- the first one and every 256th after it wait for the batch's stream and
  // read the parameters back (a few hundred microseconds per 256 steps of ~100 us each).
  if (m.size() >= 2 && g->cfg_box < 0 && (g->adapt_tick++ & 255u) == 0u) {
    int form = 0, changed = 0;
    g->last_stream = fl->stream;
    rc = sxmc_group_adapt_fill_form(g, &form, &changed);
  }
  // (... before the trial launches, which then time the form that runs)
  bool want = do_eval_pdf != 0;
  for (sxmc_hist* h : m) want = want && h->want_optimize && h->has_points;
  if (m.size() == 1 || m[0]->cfg_threads > 0 || m[0]->cfg_bpc > 0) {
    g->cfg_threads = m[0]->cfg_threads;   // (a launch shape set by hand on the evaluators)
    g->cfg_bpc = m[0]->cfg_bpc;
  } else if (rc == SXMC_OK && m.size() >= 2 && !g->tuned && want) {
    g->tuned = true;
    rc = sxmc_group_optimize(g, fl->stream, nullptr);
  }
  if (rc == SXMC_OK) rc = sxmc_group_eval_async(g, do_eval_pdf, fl->stream);
  if (rc != SXMC_OK && g->built) {
    // trial fills (or a fill whose lookup then failed) have counted into the members' histograms and normalisations:
    // a failed evaluation leaves them zeroed, not half-counted (the error code is what the caller gets)
    const std::string why = g_last_error;
    (void)sx_launch_zero(g->d_descs, (int)g->members.size(), g->max_bins, g->d_ticket, fl->stream);
    (void)hipGetLastError();
    g_last_error = why;
  }
  t_flushing = false;
  for (sxmc_hist* h : m) h->inflight = rc == SXMC_OK ? fl : nullptr;
  if (rc == SXMC_OK) {
    g_deferred_launches.fetch_add(1, std::memory_order_relaxed);
    g_deferred_evaluations.fetch_add(m.size(), std::memory_order_relaxed);
  }
  return rc;
}

// Is the thread's batch complete -- a batch of two or more seen before, and no known batch goes on beyond it?
bool deferred_batch_complete() {
  std::lock_guard<std::mutex> own(t_deferred->m);           // (another thread may be taking a dying evaluator out)
  const std::vector<sxmc_hist*>& m = t_deferred->members;
  if (m.size() < 2) return false;
  std::lock_guard<std::mutex> lock(g_auto_mutex);
  bool exact = false;
  for (const AutoGroup& a : g_auto_groups) {
    if (a.members.size() < m.size() || !std::equal(m.begin(), m.end(), a.members.begin())) continue;
    if (a.members.size() > m.size()) return false;
    exact = true;
  }
  return exact;
}

// An evaluator dies: it leaves the batches, and the groups made for batches it was part of go with it.
void forget_evaluator(sxmc_hist* h) {
  std::vector<sxmc_group*> doomed;
  {
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    for (size_t i = 0; i < g_auto_groups.size();) {
      AutoGroup& a = g_auto_groups[i];
      if (std::find(a.members.begin(), a.members.end(), h) != a.members.end()) {
        doomed.push_back(a.g);
        g_auto_groups.erase(g_auto_groups.begin() + (long)i);
      } else {
        i++;
      }
    }
  }
  for (sxmc_group* g : doomed) sxmc_group_destroy(g);
}

}  // namespace sxhost

extern "C" {

// ------------------------------------------------------------------------------ evaluator
int sxmc_hist_create(const float* samples, size_t nsamples_floats, int samples_on_device, int nfields,
                     int nobservables, const double* lower, size_t n_lower, const double* upper,
                     size_t n_upper, const int* nbins, size_t n_nbins, unsigned dataset, sxmc_hist_t* out) {
  SX_REQUIRE(out, "null argument");
  *out = nullptr;
  // Eval::Eval validation, pdfz.cpp:64-82 (same order, same messages)
  SX_REQUIRE(nfields > 0 && nsamples_floats % (size_t)nfields == 0,
             "Length of samples array is not divisible by number of fields.");
  SX_REQUIRE(nobservables != 0, "Number of observables in PDF is zero.");
  SX_REQUIRE(nobservables > 0 && nobservables <= nfields,
             "Number of observables cannot be greater than number of fields.");
  SX_REQUIRE((int)n_upper == nobservables, "Number of upper bounds must be same as number of observables.");
  SX_REQUIRE((int)n_lower == nobservables, "Number of lower bounds must be same as number of observables.");
  // EvalHist::EvalHist validation, pdfz.cpp:189-195
  SX_REQUIRE((int)n_nbins == nobservables, "Size of nbins array must be same as number of observables.");
  SX_REQUIRE(nfields <= SXMC_MAX_NFIELDS,
             "Exceeded maximum number of fields per sample. Edit MAX_NFIELDS in pdfz.cpp to fix this!");
  SX_REQUIRE(nsamples_floats == 0 || samples, "null samples");
  SX_REQUIRE(lower && upper && nbins, "null argument");

  sxmc_hist* h = new sxmc_hist;
  h->nfields = nfields;
  h->nobs = nobservables;
  h->dataset = dataset;
  h->nsamples = nsamples_floats / (size_t)nfields;
  h->lower.assign(lower, lower + nobservables);
  h->upper.assign(upper, upper + nobservables);
  h->nbins.assign(nbins, nbins + nobservables);
  h->stride.assign((size_t)nobservables, 0);
  h->scale.assign((size_t)nobservables, 0.0);

  // bin volume, row-major strides, total bins: pdfz.cpp:200-219
  double vol = 1.0f;
  bool bad = false;
  for (int i = 0; i < nobservables; i++) {
    if (nbins[i] < 0) bad = true;
    vol *= (upper[i] - lower[i]) / nbins[i];
  }
  long long total = 1;
  h->stride[(size_t)nobservables - 1] = 1;
  for (int i = nobservables - 2; i >= 0; i--) {
    long long st = (long long)nbins[i + 1] * h->stride[(size_t)i + 1];
    if (st > INT_MAX) bad = true;
    h->stride[(size_t)i] = (int)std::min<long long>(st, INT_MAX);
  }
  total = (long long)h->stride[0] * nbins[0];
  if (bad || total > INT_MAX) {
    delete h;
    return fail(SXMC_ERR_INVALID, "Histogram too large or negative bin count (total bins must fit in int).");
  }
  if (total == 0) {
    delete h;
    return fail(SXMC_ERR_INVALID, "Cannot make histogram with zero bins.");
  }
  for (int i = 0; i < nobservables; i++) {
    if (!(upper[i] > lower[i])) {
      delete h;
      return fail(SXMC_ERR_INVALID, "Upper bound must be greater than lower bound.");
    }
    h->scale[(size_t)i] = nbins[i] / (upper[i] - lower[i]);  // pdfz.cpp:366-368, in host double
  }
  h->total_nbins = (int)total;
  h->bin_volume = vol;

  h->nvec = (h->nsamples + SXMC_VEC - 1) / SXMC_VEC;
  h->pitch = std::max<size_t>(64, (h->nvec * SXMC_VEC + 63) / 64 * 64);

  auto cleanup = [&](int code) {
    if (h->d_bins) (void)hipFree(h->d_bins);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return code;
  };
#define SX_HIP_H(expr)                                                                              \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return cleanup(fail(SXMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)));        \
  } while (0)

  SX_HIP_H(hipStreamCreate(&h->stream));
  h->store = std::make_shared<SampleStore>();
  SX_HIP_H(hipMalloc((void**)&h->store->d_cols, sizeof(float) * h->pitch * (size_t)nfields));
  SX_HIP_H(hipMalloc((void**)&h->d_bins, sizeof(unsigned) * (size_t)h->total_nbins));
  SX_HIP_H(hipMemset(h->d_bins, 0, sizeof(unsigned) * (size_t)h->total_nbins));
  if (h->nsamples) {
    const float* d_aos = samples;
    float* staging = nullptr;
    if (!samples_on_device) {
      SX_HIP_H(hipMalloc((void**)&staging, sizeof(float) * nsamples_floats));
      hipError_t e = hipMemcpy(staging, samples, sizeof(float) * nsamples_floats, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(staging);
        return cleanup(fail(SXMC_ERR_HIP, std::string("hipMemcpy samples: ") + hipGetErrorString(e)));
      }
      d_aos = staging;
    }
    hipError_t e = sx_launch_transpose(d_aos, h->store->d_cols, h->nsamples, nfields, h->pitch, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (staging) (void)hipFree(staging);
    if (e != hipSuccess) return cleanup(fail(SXMC_ERR_HIP, std::string("transpose: ") + hipGetErrorString(e)));
  }
#undef SX_HIP_H
  *out = h;
  return SXMC_OK;
}

int sxmc_hist_create_shared(sxmc_hist_t base, sxmc_hist_t* out) {
  SX_REQUIRE(base && out, "null argument");
  *out = nullptr;
  sxmc_hist* h = new sxmc_hist;
  h->store = base->store;  // one copy of the MC table for every evaluator that shares it
  h->nfields = base->nfields;
  h->nobs = base->nobs;
  h->nsamples = base->nsamples;
  h->nvec = base->nvec;
  h->pitch = base->pitch;
  h->dataset = base->dataset;
  h->lower = base->lower;
  h->upper = base->upper;
  h->scale = base->scale;
  h->nbins = base->nbins;
  h->stride = base->stride;
  h->total_nbins = base->total_nbins;
  h->bin_volume = base->bin_volume;
  h->systs = base->systs;
  h->cfg_threads = base->cfg_threads;
  h->cfg_bpc = base->cfg_bpc;
  h->want_optimize = base->want_optimize;
  hipError_t e = hipStreamCreate(&h->stream);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_bins, sizeof(unsigned) * (size_t)h->total_nbins);
  if (e == hipSuccess) e = hipMemset(h->d_bins, 0, sizeof(unsigned) * (size_t)h->total_nbins);
  if (e != hipSuccess) {
    if (h->d_bins) (void)hipFree(h->d_bins);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return fail(SXMC_ERR_HIP, std::string("create_shared: ") + hipGetErrorString(e));
  }
  *out = h;
  return SXMC_OK;
}

int sxmc_hist_destroy(sxmc_hist_t h) {
  if (!h) return SXMC_OK;
  if (std::shared_ptr<DeferredBatch> b = h->deferred) {
    // (an evaluation asked for and never waited for does not outlive its evaluator -- whichever thread asked)
    std::lock_guard<std::mutex> lock(b->m);
    b->members.erase(std::remove(b->members.begin(), b->members.end(), h), b->members.end());
    b->n.store(b->members.size(), std::memory_order_relaxed);
    h->deferred.reset();
  }
  (void)flush_deferred();
  if (h->inflight && !h->inflight->done.load()) (void)hipStreamSynchronize(h->inflight->stream);
  (void)hipStreamSynchronize(h->stream);
  forget_evaluator(h);
  if (h->self) sxmc_group_destroy(h->self);
  if (h->d_bins) (void)hipFree(h->d_bins);
  if (h->d_read_bins) (void)hipFree(h->d_read_bins);
  for (void* p : h->retired) (void)hipFree(p);
  if (h->d_cdf) (void)hipFree(h->d_cdf);
  if (h->d_sample) (void)hipFree(h->d_sample);
  free_sparse(h);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return SXMC_OK;
}

int sxmc_hist_add_systematic(sxmc_hist_t h, int type, int obs, int extra_field, int npars, const short* pars) {
  SX_REQUIRE(h, "null evaluator");
  SX_REQUIRE(type == SXMC_SYST_SHIFT || type == SXMC_SYST_SCALE || type == SXMC_SYST_RESOLUTION_SCALE ||
                 type == SXMC_SYST_CTSCALE,
             "Unknown systematic type");  // pdfz.cpp:169-171
  SX_REQUIRE(obs >= 0 && obs < h->nfields, "Systematic observable index out of range");
  if (type == SXMC_SYST_RESOLUTION_SCALE) {
    SX_REQUIRE(extra_field >= 0 && extra_field < h->nfields, "Systematic truth field index out of range");
  }
  SX_REQUIRE(npars >= 0 && npars <= SXMC_MAX_SYST_PARS, "Too many parameters for one systematic");
  SX_REQUIRE(npars == 0 || pars, "null parameter index list");
  SX_REQUIRE((int)h->systs.size() < SXMC_MAX_SYST, "Too many systematics on one evaluator");
  if (h->deferred) SX_FLUSH();
  HostSyst s;
  s.type = type;
  s.obs = obs;
  s.extra_field = type == SXMC_SYST_RESOLUTION_SCALE ? extra_field : 0;
  s.pars.assign(pars, pars + npars);
  // check the slot budget (observables + distinct referenced extra fields)
  h->systs.push_back(s);
  std::vector<int> slots;
  member_slots(h, slots);
  if ((int)slots.size() > SXMC_MAX_NFIELDS) {
    h->systs.pop_back();
    return fail(SXMC_ERR_INVALID, "Too many fields referenced");
  }
  h->version++;
  return SXMC_OK;
}

int sxmc_hist_set_eval_points(sxmc_hist_t h, const float* points, size_t npoints_floats) {
  SX_FLUSH();
  TraceRange trace("sxmc: SetEvalPoints");
  SX_REQUIRE(h, "null evaluator");
  const size_t row = (size_t)h->nobs + 1;
  SX_REQUIRE(npoints_floats % row == 0,
             "Number of entries in evaluation points array not divisible by number of observables.");
  SX_REQUIRE(npoints_floats == 0 || points, "null points");
  const size_t n = npoints_floats / row;
  SX_REQUIRE(n <= (size_t)INT_MAX, "too many evaluation points");
  // pdfz.cpp:264-301: the bin of each point never changes between evaluations, so it is
  // resolved once on the host (NaN coordinates count as outside the domain).
  std::vector<int> rb;
  sxplan::eval_point_bins(points, n, h->nobs, h->lower.data(), h->upper.data(), h->scale.data(), h->stride.data(),
                          h->total_nbins, h->dataset, rb);
  // The caller has finished with the evaluator's previous evaluations (pdfz.h:354-357), so the table is
  // replaced in place: no device-wide synchronisation, and no hipFree / hipMalloc (both stall every stream of
  // the device) unless the new data set outgrows the buffer -- other chains on the GPU keep running.
  bool moved = !h->has_points;
  if (n > h->read_bins_cap) {
    if (h->d_read_bins) h->retired.push_back(h->d_read_bins);
    h->d_read_bins = nullptr;
    const size_t cap = std::max<size_t>(n + n / 4, 1024);
    SX_HIP(hipMalloc((void**)&h->d_read_bins, sizeof(int) * cap));
    h->read_bins_cap = cap;
    moved = true;
  }
  if (n) SX_HIP(hipMemcpy(h->d_read_bins, rb.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  h->npoints = n;
  h->has_points = true;
  h->points_version++;
  h->h_read_bins = rb;
  if (h->total_nbins > kLdsMaxBins) {
    SX_HIP(hipDeviceSynchronize());   // (the sparse structures are rebuilt: in-flight lookups may read the old ones)
    int rc = build_sparse(h, rb);
    if (rc) return rc;
    moved = true;
  }
  if (moved) h->version++;   // descriptors hold the pointers: a full re-plan; otherwise only the points changed
  return SXMC_OK;
}

int sxmc_hist_set_pdf_value_buffer(sxmc_hist_t h, float* d_output, int offset, int stride) {
  SX_REQUIRE(h, "null evaluator");
  if (h->pdf == d_output && h->pdf_off == offset && h->pdf_stride == stride) return SXMC_OK;
  if (h->deferred) SX_FLUSH();   // (the evaluation asked for runs with the bindings it was asked with)
  h->pdf = d_output;
  h->pdf_off = offset;
  h->pdf_stride = stride;
  h->outputs_host_visible = host_can_read(h->pdf) || host_can_read(h->norm);
  h->version++;
  return SXMC_OK;
}
int sxmc_hist_set_normalization_buffer(sxmc_hist_t h, unsigned* d_norm, int offset) {
  SX_REQUIRE(h, "null evaluator");
  if (h->norm == d_norm && h->norm_off == offset) return SXMC_OK;
  if (h->deferred) SX_FLUSH();
  h->norm = d_norm;
  h->norm_off = offset;
  h->outputs_host_visible = host_can_read(h->pdf) || host_can_read(h->norm);
  h->version++;
  return SXMC_OK;
}
int sxmc_hist_set_parameter_buffer(sxmc_hist_t h, const double* d_params, int offset, int stride) {
  SX_REQUIRE(h, "null evaluator");
  if (h->params == d_params && h->par_off == offset && h->par_stride == stride) return SXMC_OK;
  if (h->deferred) SX_FLUSH();
  h->params = d_params;
  h->par_off = offset;
  h->par_stride = stride;
  h->version++;
  return SXMC_OK;
}

int sxmc_hist_eval_async(sxmc_hist_t h, int do_eval_pdf) {
  SX_REQUIRE(h, "null evaluator");
  do_eval_pdf = do_eval_pdf ? 1 : 0;
  if (t_capturing || !deferral_enabled()) {
    // launched as asked, on the evaluator's own stream (a recording takes what is launched, when it is launched)
    SX_FLUSH();
    if (!h->self) {
      int rc = sxmc_group_create(&h, 1, &h->self);
      if (rc) return rc;
    }
    h->self->cfg_threads = h->cfg_threads;
    h->self->cfg_bpc = h->cfg_bpc;
    h->inflight.reset();
    return sxmc_group_eval_async(h->self, do_eval_pdf, h->stream);
  }
  // the checks an immediate launch would make: a caller that forgot a binding hears of it here, not at EvalFinished
  if (!h->norm) return fail(SXMC_ERR_STATE, "evaluation before SetNormalizationBuffer");
  if (!h->systs.empty() && !h->params) return fail(SXMC_ERR_STATE, "evaluation before SetParameterBuffer");
  if (do_eval_pdf && h->has_points && !h->pdf) return fail(SXMC_ERR_STATE, "evaluation before SetPDFValueBuffer");
  if (!t_deferred) {
    t_deferred = std::make_shared<DeferredBatch>();
    t_deferred->owner = std::this_thread::get_id();
  }
  if (h->deferred && h->deferred != t_deferred) {
    return fail(SXMC_ERR_STATE, "EvalAsync of an evaluator whose previous EvalAsync, on another host thread, has not "
                                "been followed by EvalFinished there");
  }
  // a second evaluation of the same evaluator, or another kind of evaluation, starts a new batch
  if (h->deferred || (t_deferred->n.load(std::memory_order_relaxed) != 0 && t_deferred->do_eval_pdf != do_eval_pdf)) {
    SX_FLUSH();
  }
  {
    std::lock_guard<std::mutex> lock(t_deferred->m);
    t_deferred->members.push_back(h);
    t_deferred->n.store(t_deferred->members.size(), std::memory_order_relaxed);
    t_deferred->do_eval_pdf = do_eval_pdf;
    h->deferred = t_deferred;
  }
  h->inflight.reset();
  if (deferred_batch_complete()) return flush_deferred();   // the last sibling: the device starts now
  return SXMC_OK;
}

int sxmc_hist_eval_finished(sxmc_hist_t h) {
  SX_REQUIRE(h, "null evaluator");
  if (h->deferred && h->deferred != t_deferred) {
    return fail(SXMC_ERR_STATE, "EvalFinished on another host thread than the evaluator's EvalAsync");
  }
  SX_FLUSH();
  if (std::shared_ptr<BatchInFlight> fl = h->inflight) {
    // one wait per batch: the first sibling's EvalFinished waits, the others find it done
    if (!fl->done.load(std::memory_order_acquire)) {
      if (fl->stream == nullptr && lazy_finish_enabled() && !t_capturing && !fl->host_visible) {
        // (ordered on the device; the host waits when it could first tell: see settle_for)
        g_unsettled[current_device_slot()].store(true, std::memory_order_release);
      } else {
        SX_HIP(wait_for_stream(fl->stream));
      }
      fl->done.store(true, std::memory_order_release);
    }
    h->inflight.reset();
    return SXMC_OK;
  }
  SX_HIP(hipStreamSynchronize(h->stream));
  return SXMC_OK;
}

int sxmc_set_deferred_eval(int enable) {
  SX_FLUSH();
  g_defer.store(enable ? 1 : 0, std::memory_order_relaxed);
  return SXMC_OK;
}

int sxmc_set_lazy_finish(int enable) {
  SX_FLUSH();
  if (int rc = settle()) return rc;
  g_lazy_finish.store(enable ? 1 : 0, std::memory_order_relaxed);
  return SXMC_OK;
}

int sxmc_deferred_eval_stats(unsigned long long* launches, unsigned long long* evaluations) {
  SX_REQUIRE(launches && evaluations, "null argument");
  *launches = g_deferred_launches.load();
  *evaluations = g_deferred_evaluations.load();
  return SXMC_OK;
}

int sxmc_hist_total_nbins(sxmc_hist_t h, int* v) {
  SX_REQUIRE(h && v, "null argument");
  *v = h->total_nbins;
  return SXMC_OK;
}
int sxmc_hist_bin_volume(sxmc_hist_t h, double* v) {
  SX_REQUIRE(h && v, "null argument");
  *v = h->bin_volume;
  return SXMC_OK;
}
int sxmc_hist_nsamples(sxmc_hist_t h, size_t* v) {
  SX_REQUIRE(h && v, "null argument");
  *v = h->nsamples;
  return SXMC_OK;
}
int sxmc_hist_npoints(sxmc_hist_t h, size_t* v) {
  SX_REQUIRE(h && v, "null argument");
  *v = h->has_points ? h->npoints : 0;
  return SXMC_OK;
}
int sxmc_hist_get_bins(sxmc_hist_t h, unsigned* out, size_t n) {
  SX_FLUSH();
  SX_REQUIRE(h && out, "null argument");
  SX_REQUIRE(n == (size_t)h->total_nbins, "bins buffer size mismatch");
  if (!h->bins_valid) {
    return fail(SXMC_ERR_STATE,
                "the histogram is not filled (the last evaluation counted only the event bins, or "
                "sxmc_group_finish_step_async cleared it): evaluate with do_eval_pdf = 0");
  }
  SX_HIP(hipMemcpy(out, h->d_bins, sizeof(unsigned) * n, hipMemcpyDeviceToHost));
  return SXMC_OK;
}
int sxmc_hist_get_read_bins(sxmc_hist_t h, int* out, size_t n) {
  SX_REQUIRE(h && (out || n == 0), "null argument");
  SX_REQUIRE(h->has_points && n == h->npoints, "read_bins buffer size mismatch");
  if (n) SX_HIP(hipMemcpy(out, h->d_read_bins, sizeof(int) * n, hipMemcpyDeviceToHost));
  return SXMC_OK;
}
int sxmc_hist_get_samples(sxmc_hist_t h, float* out, size_t n) {
  SX_FLUSH();
  SX_REQUIRE(h && (out || n == 0), "null argument");
  SX_REQUIRE(n == h->nsamples * ((size_t)h->nobs + 1), "samples buffer size mismatch");
  if (!n) return SXMC_OK;
  float* tmp = nullptr;
  SX_HIP(hipMalloc((void**)&tmp, sizeof(float) * n));
  hipError_t e = sx_launch_untranspose_obs(h->store->d_cols, tmp, h->nsamples, h->nobs, h->pitch, (float)h->dataset,
                                           h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(out, tmp, sizeof(float) * n, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(SXMC_ERR_HIP, std::string("get_samples: ") + hipGetErrorString(e));
  return SXMC_OK;
}
int sxmc_hist_random_sample(sxmc_hist_t h, size_t nobserved, unsigned long long seed, const float* lowers,
                            const float* uppers, float* h_events) {
  SX_FLUSH();
  SX_REQUIRE(h && (h_events || nobserved == 0), "null argument");
  SX_REQUIRE(h->nobs <= 3, "Cannot EvalHist::CreateHistogram for dimensions greater than 3!");   // pdfz.cpp:499-501
  SX_REQUIRE((lowers == nullptr) == (uppers == nullptr), "give both cut arrays or neither");
  if (!h->bins_valid) {
    return fail(SXMC_ERR_STATE, "the histogram is not filled: evaluate with do_eval_pdf = 0 first (CreateHistogram)");
  }
  if (nobserved == 0) return SXMC_OK;
  if (!h->d_cdf) SX_HIP(hipMalloc((void**)&h->d_cdf, sizeof(unsigned) * (size_t)h->total_nbins));
  SX_HIP(sx_hist_cdf(h->d_bins, h->d_cdf, h->total_nbins, h->stream));
  unsigned total = 0;
  SX_HIP(hipMemcpy(&total, h->d_cdf + (h->total_nbins - 1), sizeof(unsigned), hipMemcpyDeviceToHost));
  SX_REQUIRE(total > 0, "cannot sample an empty histogram");
  const size_t row = (size_t)h->nobs + 1;
  // (the evaluator's own grow-only buffer: an allocation and a hipFree per draw would each wait for the device to
  //  drain, i.e. for whatever other chains have queued)
  const size_t need = sizeof(float) * nobserved * row + sizeof(unsigned);   // + the count of points never accepted
  if (need > h->cap_sample) {
    if (h->d_sample) SX_HIP(hipFree(h->d_sample));
    h->d_sample = nullptr;
    h->cap_sample = 0;
    SX_HIP(hipMalloc((void**)&h->d_sample, need + need / 4));
    h->cap_sample = need + need / 4;
  }
  float* const d_rows = h->d_sample;
  unsigned* d_exhausted = reinterpret_cast<unsigned*>(d_rows + nobserved * row);
  SX_HIP(hipMemsetAsync(d_exhausted, 0, sizeof(unsigned), h->stream));
  SX_HIP(sx_random_sample(h->d_cdf, h->total_nbins, h->nobs, h->nbins.data(), h->lower.data(), h->upper.data(), lowers,
                          uppers, seed, nobserved, (float)h->dataset, d_rows, d_exhausted, h->stream));
  SX_HIP(hipStreamSynchronize(h->stream));
  unsigned exhausted = 0;
  SX_HIP(hipMemcpy(&exhausted, d_exhausted, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (exhausted) {
    return fail(SXMC_ERR_STATE, std::to_string(exhausted) + " of " + std::to_string(nobserved) +
                                    " events could not be drawn inside the cuts in 1024 attempts each (the reference "
                                    "would redraw for ever, pdfz.cpp:838-905): the cuts leave (almost) none of the "
                                    "histogram's content");
  }
  SX_HIP(hipMemcpy(h_events, d_rows, sizeof(float) * nobserved * row, hipMemcpyDeviceToHost));
  return SXMC_OK;
}

int sxmc_hist_get_stream(sxmc_hist_t h, sxmc_stream_t* s) {
  SX_FLUSH();
  SX_REQUIRE(h && s, "null argument");
  *s = h->stream;
  return SXMC_OK;
}
int sxmc_hist_set_optimize(sxmc_hist_t h, int enable) {
  SX_REQUIRE(h, "null evaluator");
  h->want_optimize = enable != 0;
  return SXMC_OK;
}

// EvalHist::Optimize (pdfz.cpp:622-628), called by hand: the trials run (again) at the next lookup evaluation of every
// batch this evaluator is part of -- they need the bindings of an evaluation, which an evaluator has then.
int sxmc_hist_optimize(sxmc_hist_t h) {
  SX_REQUIRE(h, "null evaluator");
  if (h->deferred) SX_FLUSH();
  if (!h->has_points) return SXMC_OK;        // (pdfz.cpp:623: nothing without evaluation points)
  h->want_optimize = true;
  std::lock_guard<std::mutex> lock(g_auto_mutex);
  for (AutoGroup& a : g_auto_groups) {
    if (std::find(a.members.begin(), a.members.end(), h) != a.members.end()) {
      a.g->tuned = false;
      a.g->cfg_threads = a.g->cfg_bpc = a.g->cfg_teams = 0;   // (what earlier trials chose)
      a.g->cfg_codes = -1;
    }
  }
  return SXMC_OK;
}

// The launch plan of the batch the evaluator's last deferred evaluation went into (or of its own launches), as
// sxmc_group_launch_info prints it, + "tuned=<0|1> trial_launches=<n>" for the group.
int sxmc_hist_launch_info(sxmc_hist_t h, char* out, size_t n) {
  SX_REQUIRE(h && out && n > 0, "null argument");
  SX_FLUSH();
  sxmc_group* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    for (AutoGroup& a : g_auto_groups)
      if (std::find(a.members.begin(), a.members.end(), h) != a.members.end()) g = a.g;
  }
  if (!g) g = h->self;
  if (!g) {
    std::snprintf(out, n, "%s", "");
    return SXMC_OK;
  }
  std::vector<char> buf(8192);
  int rc = sxmc_group_launch_info(g, buf.data(), buf.size());
  if (rc) return rc;
  std::snprintf(out, n, "%stuned=%d trial_launches=%d\n", buf.data(), g->tuned ? 1 : 0, g->trial_launches);
  return SXMC_OK;
}

int sxmc_hist_set_launch_config(sxmc_hist_t h, int bin_threads, int bin_blocks_per_cu) {
  SX_REQUIRE(h, "null evaluator");
  SX_REQUIRE(bin_threads == 0 || (bin_threads >= 64 && bin_threads <= 1024 && bin_threads % 64 == 0),
             "bin_threads must be 0 or a multiple of 64 up to 1024");
  SX_REQUIRE(bin_blocks_per_cu >= 0 && bin_blocks_per_cu <= 16, "bin_blocks_per_cu out of range");
  if (h->deferred) SX_FLUSH();
  h->cfg_threads = bin_threads;
  h->cfg_bpc = bin_blocks_per_cu;
  return SXMC_OK;
}

}  // extern "C"
