// mcmc.h -- ROOT-free MCMC driver in the shape of the reference's MCMC class (src/mcmc.{h,cpp}): the
// caller of the hot path.  Same constructor arguments and call operator; the chain comes back as a
// plain table (sxmc::Chain) instead of a TNtuple wrapped in a LikelihoodSpace.
//
// What it does per step (mcmc.cpp:261-348): re-evaluate every signal's PDF at the proposed vector
// (when systematics float), event log-sum, then the fused reduce + nll_total + Metropolis + next
// proposal.  Here the S evaluators are stepped as ONE batched launch sequence through
// sxmc_group_eval_nll_async (zero; fill of all signals in one kernel; lookup fused with the event sum)
// followed by the fused step end -- 4 launches per step instead of the reference's 3*S + 2.  `reference_form = true` issues the
// reference's own sequence of entry points instead (per-evaluator EvalAsync/EvalFinished,
// nll_event_chunks, finish_nll_jump_pick_combo); both forms give the same numbers up to the
// summation order of the event partial sums.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "fit_types.h"
#include "nll_kernels.h"

namespace sxmc {

/** The sampled likelihood space: one row per kept step = parameters..., likelihood (mcmc.cpp:100-114). */
struct Chain {
  std::vector<std::string> names;  //!< parameter names, then "likelihood"
  std::vector<float> rows;         //!< row-major [nrows][names.size()]
  size_t accepted = 0;             //!< accepted proposals over the whole walk
  double setup_seconds = 0;        //!< of the walk that made it: entry to the first step (buffers, SetEvalPoints, first
                                   //!< evaluation, launch-shape trials), host clock
  double steps_seconds = 0;        //!< ... and the steps themselves, re-tunings, flushes and graph recording included
  size_t nrows() const { return names.empty() ? 0 : rows.size() / names.size(); }
  float at(size_t row, size_t col) const { return rows[row * names.size() + col]; }
};

/** The lock that serialises set-up, graph recording and tear-down of the chains that share a DEVICE (allocation,
 *  uploads through the legacy stream, launch-plan rebuilds and device-wide synchronisation are calls the runtime
 *  refuses, or stalls, beside another thread's recording on that device).  A std::mutex that also keeps the time its
 *  users spent waiting for it and holding it: with G devices x lanes host threads that is the Amdahl term of an
 *  ensemble, and a run reports it instead of guessing (bench_cpp: "setup_lock").  One per device: chains on
 *  different devices never contend. */
/** Shared by the SetupLocks of several cards (ensemble_multi_gpu with one lock per card): GRAPH RECORDING anywhere in the
 *  process excludes set-up -- allocation, uploads, launch plans, module loads, device-wide synchronisation --
 *  everywhere in the process, while the set-ups of different cards run side by side.  Holding a SetupLock holds the gate
 *  shared; begin_recording() trades that for the exclusive side until end_recording(). */
struct RecordingGate {
  std::shared_timed_mutex m;
};

class SetupLock {
 public:
  SetupLock() = default;
  explicit SetupLock(RecordingGate* gate_) : gate(gate_) {}
  void lock() {
    const auto t0 = std::chrono::steady_clock::now();
    m.lock();
    if (gate) gate->m.lock_shared();
    since = std::chrono::steady_clock::now();
    waited_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(since - t0).count(),
                        std::memory_order_relaxed);
    acquisitions.fetch_add(1, std::memory_order_relaxed);
  }
  void unlock() {
    held_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::steady_clock::now() - since).count(),
                      std::memory_order_relaxed);
    if (gate) gate->m.unlock_shared();
    m.unlock();
  }
  /** The holder is about to record a HIP graph: nobody in the process may be inside a set-up section meanwhile (the
   *  other holders of the gate finish theirs first; new ones wait).  Only the holder of this lock calls these. */
  void begin_recording() {
    if (!gate) return;
    gate->m.unlock_shared();
    gate->m.lock();
  }
  void end_recording() {
    if (!gate) return;
    gate->m.unlock();
    gate->m.lock_shared();
  }
  double waited_seconds() const { return 1e-9 * (double)waited_ns.load(); }   //!< summed over all the threads that asked
  double held_seconds() const { return 1e-9 * (double)held_ns.load(); }
  unsigned long long count() const { return acquisitions.load(); }

 private:
  std::mutex m;
  RecordingGate* gate = nullptr;
  std::chrono::steady_clock::time_point since;   // (written and read by the holder only)
  std::atomic<unsigned long long> waited_ns{0}, held_ns{0}, acquisitions{0};
};

/** begin_recording / end_recording of a held SetupLock (may be null or not held: nothing to do then), scope-bound. */
struct RecordingScope {
  SetupLock* lock;
  explicit RecordingScope(SetupLock* l, bool held) : lock(held ? l : nullptr) {
    if (lock) lock->begin_recording();
  }
  ~RecordingScope() {
    if (lock) lock->end_recording();
  }
  RecordingScope(const RecordingScope&) = delete;
  RecordingScope& operator=(const RecordingScope&) = delete;
};

/** Chains advanced TOGETHER (sxmc_multigroup_step_async): one fill pass over the shared sample tables per step for
 *  all of them, then every chain's own step end.  Shared by the MCMC objects of one lockstep set, each walking on
 *  its own host thread and on the set's ONE stream: a chain that is ready for its next run of steps leaves its
 *  arguments here; the last one to arrive launches the run for all -- replays of a HIP graph of `graph_steps`
 *  recorded lockstep steps, the remainder step by step -- so the host threads meet once per run (a handful of
 *  times per walk), not once per step.  Every chain of the set must ask for the same runs. */
class LockstepSet {
 public:
  /** exclusive: the mutex that serialises set-up, graph recording and tear-down in this process (may be null). */
  LockstepSet(size_t nchains, sxmc_stream_t stream_, SetupLock* exclusive_ = nullptr)
      : stream(stream_), exclusive(exclusive_), groups(nchains, nullptr), args(nchains) {}
  ~LockstepSet() {
    if (graph) sxmc_graph_destroy(graph);
    if (mg) sxmc_multigroup_destroy(mg);
  }
  LockstepSet(const LockstepSet&) = delete;
  LockstepSet& operator=(const LockstepSet&) = delete;

  /** Chain `index` is ready to take `nsteps` steps with `a`: returns when those steps of ALL chains are launched. */
  void advance(size_t index, sxmc_group_t group, const sxmc_step_args& a, unsigned nsteps, unsigned graph_steps) {
    std::unique_lock<std::mutex> lock(m);
    if (broken) throw std::runtime_error("lockstep set: " + why);
    if (groups[index] != group || std::memcmp(&args[index], &a, sizeof a) != 0) dirty = true;
    groups[index] = group;
    args[index] = a;
    const unsigned long long gen = generation;
    if (++arrived == groups.size()) {
      int rc = launch(nsteps, graph_steps);
      arrived = 0;
      generation++;
      if (rc != SXMC_OK) {
        broken = true;
        why = sxmc_last_error();
      }
      cv.notify_all();
      if (broken) throw std::runtime_error("lockstep set: " + why);
    } else {
      cv.wait(lock, [&] { return generation != gen || broken; });
      if (broken) throw std::runtime_error("lockstep set: " + why);
    }
  }
  /** The chain's group is about to be destroyed: multigroup and graph are rebuilt at the next run. */
  void leave(size_t index) {
    std::lock_guard<std::mutex> lock(m);
    groups[index] = nullptr;
    dirty = true;
  }
  /** A chain of the set failed: its partners must not wait for it. */
  void abandon(const std::string& reason) {
    std::lock_guard<std::mutex> lock(m);
    broken = true;
    why = reason;
    cv.notify_all();
  }
  sxmc_stream_t stream;

 private:
  int launch(unsigned nsteps, unsigned graph_steps) {
    if (dirty) {
      if (graph) sxmc_graph_destroy(graph);
      graph = nullptr;
      if (mg) sxmc_multigroup_destroy(mg);
      mg = nullptr;
      int rc = sxmc_multigroup_create(groups.data(), (int)groups.size(), &mg);
      if (rc) return rc;
      dirty = false;
      stepped = false;
    }
    if (graph_steps > 0 && stepped && nsteps >= graph_steps) {
      if (!graph || recorded != graph_steps) {
        // recording does not tolerate another thread's allocations: under the process's set-up mutex
        std::unique_lock<SetupLock> excl;
        if (exclusive) excl = std::unique_lock<SetupLock>(*exclusive);
        RecordingScope recording(exclusive, exclusive != nullptr);
        if (graph) sxmc_graph_destroy(graph);
        graph = nullptr;
        int rc = sxmc_graph_begin_capture(stream);
        if (rc) return rc;
        for (unsigned k = 0; k < graph_steps && rc == SXMC_OK; k++) rc = sxmc_multigroup_step_async(mg, stream, args.data());
        const std::string err = rc ? sxmc_last_error() : "";
        int rc2 = sxmc_graph_end_capture(stream, &graph);
        if (rc) {
          if (graph) sxmc_graph_destroy(graph);
          graph = nullptr;
          (void)err;
          return rc;
        }
        if (rc2) return rc2;
        recorded = graph_steps;
      }
      int rc = sxmc_graph_launch(graph, stream, (int)(nsteps / graph_steps));
      if (rc) return rc;
      nsteps %= graph_steps;
    }
    for (unsigned k = 0; k < nsteps; k++) {
      // the first step of a new set of chains builds launch plans (allocations, a device-wide synchronisation),
      // which another set's recording does not tolerate: under the process's set-up mutex, like the recording
      std::unique_lock<SetupLock> excl;
      if (!stepped && exclusive) excl = std::unique_lock<SetupLock>(*exclusive);
      int rc = sxmc_multigroup_step_async(mg, stream, args.data());
      if (rc) return rc;
      stepped = true;   // (the launch plans are in place once a step has been launched: recording may follow)
    }
    return SXMC_OK;
  }

  std::mutex m;
  std::condition_variable cv;
  SetupLock* exclusive;
  std::vector<sxmc_group_t> groups;
  std::vector<sxmc_step_args> args;
  sxmc_multigroup_t mg = nullptr;
  sxmc_graph_t graph = nullptr;
  unsigned recorded = 0;
  size_t arrived = 0;
  unsigned long long generation = 0;
  bool dirty = true, broken = false, stepped = false;
  std::string why;
};

class MCMC {
 public:
  LockstepSet* lockstep = nullptr;  //!< set: this chain steps together with the other chains of the set (same
  size_t lockstep_index = 0;        //!< sample tables, same systematics, the set's stream); every run of steps is
                                    //!< launched (graph replays of graph_steps steps) by the chain that arrives last
  bool reference_form = false;  //!< launch the reference's own kernel sequence instead of the batched one
  bool verbose = false;
  unsigned graph_steps = 0;     //!< > 0: replay the batched step from a HIP graph of this many recorded steps
  bool optimize = true;         //!< EvalHist's `optimize` constructor flag for the batched launch: a few trial
                                //!< launches at the start of a walk pick the lane count per CU (sxmc_group_optimize)
  bool consume = true;          //!< batched form: the step end also clears histograms and normalisations for the
                                //!< next step (sxmc_group_step_async: 2 launches per step; nothing reads them
                                //!< between the steps of a walk)
  SetupLock* exclusive = nullptr;   //!< with one chain per host thread: the lock this walk holds while it
                                    //!< allocates, uploads, rebuilds launch plans, records its graph and frees
                                    //!< (calls the runtime refuses beside another thread's recording); it is
                                    //!< released while the walk only launches and waits on its own stream
  bool lookahead = false;       //!< batched, consuming form, one chain: the LOOK-AHEAD WALK.  Every pass over the tables
                                //!< evaluates the step's proposal AND the vector the next step proposes after a rejection
                                //!< (a second set of evaluators over the same tables), and the step end decides one or two
                                //!< steps (sxmc_multigroup_lookahead_step_async).  Same chain bit for bit; +25 % steps
                                //!< per second where the fill streams float columns (it is bound by the stream, and the
                                //!< pass streams once for two evaluations), SLOWER where it streams 16-bit codes (the
                                //!< pass is then bound by its arithmetic: BASELINE config 3, 9 380 against 10 290).
  unsigned adapt_interval = 1000;  //!< a plan with two forms of the fill: steps between the flushes at which the walk asks
                                   //!< which one to take (sxmc_group_adapt_fill_form); 0: only at the reference's flushes
  bool lookahead_auto = false;  //!< let the walk decide: the look-ahead pass where the launch plan streams float columns,
                                //!< one evaluation per step where it streams codes (what the measurements above say)
  /** Hooks for a driver that runs several walks side by side (sxmc::ensemble_concurrent): called once the walk's
   *  set-up is over -- buffers, first evaluation, launch plan, recorded graph; the set-up lock released -- and BEFORE
   *  its first long run of steps is queued; and once its last step has finished, before its tear-down.  An ensemble's
   *  lanes meet at both: set-up and tear-down synchronise the whole device (allocation, launch plans), and beside a
   *  chain that has a second of replays queued every one of those calls waits that second out -- measured at 1e5
   *  steps per experiment: eight lanes walked ONE AT A TIME, 10 500 steps/s for the card (profiles/r05_c4_lanes.log). */
  std::function<void()> on_setup_done, on_steps_done;
  bool lut_output = false;      //!< materialise the lookup table in the batched step (nothing reads it; when
                                //!< false the event sum runs over distinct event-bin tuples, see sxmc_hip.h)
  unsigned long long seed = 0;  //!< gRandom->GetSeed() in the reference (mcmc.cpp:125)

  MCMC(const std::vector<Source>& sources, const std::vector<Signal>& signals,
       const std::vector<Systematic>& systematics, const std::vector<Observable>& observables,
       unsigned long long _seed = 1, sxmc_stream_t _stream = nullptr)
      : seed(_seed),
        stream(_stream),
        nsources(sources.size()),
        nsignals(signals.size()),
        nsystematics(systematics.size()),
        nobservables(observables.size()) {
    // mcmc.cpp:37-45: launch shapes of the NLL kernels
    nnllblocks = 64;
    nllblocksize = 256;
    nnllthreads = nnllblocks * nllblocksize;
    nreducethreads = 128;

    size_t npars = 0;
    for (const Systematic& s : systematics) npars += s.npars;
    nparameters = nsources + npars;
    parameter_means.reset(new pdfz::Array<double>(nparameters, true));
    parameter_sigma.reset(new pdfz::Array<double>(nparameters, true));
    parameter_fixed.resize(nparameters);
    nfloat = 0;
    for (size_t i = 0; i < nsources; i++) {
      parameter_means->writeOnlyHostPtr()[i] = sources[i].mean;
      parameter_sigma->writeOnlyHostPtr()[i] = sources[i].sigma;
      parameter_fixed[i] = sources[i].fixed;
      nfloat += sources[i].fixed ? 0 : 1;
      parameter_names.push_back(sources[i].name);
    }
    systematics_fixed = true;
    size_t k = nsources;
    for (const Systematic& s : systematics) {
      if (!s.fixed) systematics_fixed = false;
      for (size_t j = 0; j < s.npars; j++) {
        parameter_means->writeOnlyHostPtr()[k] = s.means[j];
        parameter_sigma->writeOnlyHostPtr()[k] = s.sigmas[j];
        parameter_fixed[k] = s.fixed;
        nfloat += s.fixed ? 0 : 1;
        parameter_names.push_back(s.name + "_" + std::to_string(j));
        k++;
      }
    }
    parameter_names.push_back("likelihood");

    nexpected.reset(new pdfz::Array<double>(nsignals, true));
    n_mc.reset(new pdfz::Array<unsigned>(nsignals, true));
    source_id.reset(new pdfz::Array<short>(nsignals, true));
    for (size_t i = 0; i < nsignals; i++) {
      pdfs.push_back(signals[i].histogram);
      nexpected->writeOnlyHostPtr()[i] = signals[i].nexpected;
      n_mc->writeOnlyHostPtr()[i] = (unsigned)signals[i].n_mc;
      source_id->writeOnlyHostPtr()[i] = (short)signals[i].source.index;
    }

    rngs.reset(new pdfz::Array<RNGState>(nparameters, true));
    const int bs = 128;
    const int nb = (int)(nparameters / bs + 1);
    SXMC_KERNEL_LAUNCH(init_device_rngs, nb, bs, 0, stream, (int)nparameters, seed, rngs->writeOnlyPtr());

    // the batched form needs every evaluator to be a histogram evaluator of this library
    std::vector<sxmc_hist_t> handles;
    for (pdfz::Eval* p : pdfs) {
      pdfz::EvalHist* h = dynamic_cast<pdfz::EvalHist*>(p);
      if (!h) {
        handles.clear();
        break;
      }
      handles.push_back(h->Handle());
    }
    if (!handles.empty()) check(sxmc_group_create(handles.data(), (int)handles.size(), &group));
  }

  ~MCMC() {
    if (group) sxmc_group_destroy(group);
  }
  MCMC(const MCMC&) = delete;
  MCMC& operator=(const MCMC&) = delete;

  /** Initial proposal widths, mcmc.cpp:198-228 (including its `i < nsignals` test at :217). */
  std::vector<float> initial_jump_widths() const {
    std::vector<float> w(nparameters);
    const float scale_factor = 2.4 * 2.4 / nfloat;  // Haario, 2001
    for (size_t i = 0; i < nparameters; i++) {
      if (parameter_fixed[i]) {
        w[i] = -1;
        continue;
      }
      const float mean = parameter_means->readOnlyHostPtr()[i];
      const float sigma = parameter_sigma->readOnlyHostPtr()[i];
      float width = 0.1;
      if (sigma > 0) {
        width = sigma;
      } else if (i < nsignals) {
        const float m = std::max(mean, (float)10);
        width = std::sqrt(m) / m;
      } else {
        width = std::sqrt(std::max(mean, (float)1));
      }
      w[i] = 0.1 * width * scale_factor;
    }
    return w;
  }

  /** MCMC::operator() (mcmc.cpp:143-387).  data: rows of nobservables+1 floats (last = dataset id). */
  Chain operator()(std::vector<float>& data, unsigned nsteps, float burnin_fraction,
                   const bool debug_mode = false, unsigned sync_interval = 10000) {
    const std::chrono::steady_clock::time_point walk_t0 = std::chrono::steady_clock::now();
    std::unique_lock<SetupLock> excl;  // (first local: released last, after the arrays below are freed)
    if (exclusive) excl = std::unique_lock<SetupLock>(*exclusive);
    // array transfers of this walk are ordered on the chain's stream (a blocking copy through the legacy
    // default stream would neither wait for a non-blocking stream nor leave other chains alone)
    struct TransferGuard {
      sxmc_stream_t prev;
      explicit TransferGuard(sxmc_stream_t s) : prev(transfer_stream()) {
        if (s) transfer_stream() = s;
      }
      ~TransferGuard() { transfer_stream() = prev; }
    } transfer_guard(stream);
    const unsigned burnin_steps = nsteps * burnin_fraction;
    Chain chain;
    chain.names = parameter_names;
    const size_t ncol = nparameters + 1;

    pdfz::Array<double> current_vector(nparameters, true), proposed_vector(nparameters, true);
    for (size_t i = 0; i < nparameters; i++)
      current_vector.writeOnlyHostPtr()[i] = parameter_means->readOnlyHostPtr()[i];
    proposed_vector.writeOnlyHostPtr();
    pdfz::Array<unsigned> normalizations(nsignals, true);
    normalizations.writeOnlyHostPtr();
    pdfz::Array<double> event_partial_sums(std::max<size_t>(nnllthreads, 1024), true);
    event_partial_sums.writeOnlyHostPtr();
    pdfz::Array<double> event_total_sum(1, true);
    event_total_sum.writeOnlyHostPtr();
    pdfz::Array<int> jump_counter(1, true), accept_counter(1, true);
    jump_counter.writeOnlyHostPtr()[0] = 0;
    accept_counter.writeOnlyHostPtr()[0] = 0;
    pdfz::Array<float> jump_buffer((size_t)sync_interval * ncol, true);
    pdfz::Array<double> current_nll(1, true), proposed_nll(1, true);
    current_nll.writeOnlyHostPtr();
    proposed_nll.writeOnlyHostPtr();

    pdfz::Array<float> jump_width(nparameters, true);
    {
      const std::vector<float> w = initial_jump_widths();
      for (size_t i = 0; i < nparameters; i++) jump_width.writeOnlyHostPtr()[i] = w[i];
    }
    const float scale_factor = 2.4 * 2.4 / nfloat;

    // mcmc.cpp:230-242: bind, first evaluation at the current vector, then re-point at the proposal
    const size_t nevents = data.size() / (nobservables + 1);
    pdfz::Array<float> lut(nevents * nsignals, true);
    for (size_t i = 0; i < pdfs.size(); i++) {
      pdfz::Eval* p = pdfs[i];
      p->SetEvalPoints(data);
      p->SetPDFValueBuffer(&lut, (int)(i * nevents), 1);
      p->SetNormalizationBuffer(&normalizations, (int)i);
      p->SetParameterBuffer(&current_vector, (int)nsources);
      p->EvalAsync();
      p->EvalFinished();
      p->SetParameterBuffer(&proposed_vector, (int)nsources);
    }

    nll(lut.readOnlyPtr(), nevents, current_vector.readOnlyPtr(), current_nll.writeOnlyPtr(),
        normalizations.readOnlyPtr(), event_partial_sums.ptr(), event_total_sum.ptr());
    SXMC_KERNEL_LAUNCH(pick_new_vector, 1, 64, 0, stream, (int)nparameters, rngs->ptr(), jump_width.readOnlyPtr(),
                       current_vector.readOnlyPtr(), proposed_vector.writeOnlyPtr());

    const bool batched = group != nullptr && !reference_form;
    if (batched) {
      // bindings must be current before the group reads them (proposal vector as parameter buffer)
      for (pdfz::Eval* p : pdfs) dynamic_cast<pdfz::EvalHist*>(p)->Bind();
      check(sxmc_group_set_lut_output(group, lut_output ? 1 : 0));
      if (optimize && !optimized) {
        check(sxmc_group_optimize(group, stream, nullptr));
        optimized = true;
      }
    }
    const bool reevaluate = nsystematics > 0 && !systematics_fixed;

    // Recorded steps need a created stream (blocking: it still orders with the copies of the array
    // accessors, which go through the legacy default stream) and the batched form.
    const bool in_lockstep = lockstep && batched && reevaluate && consume;
    const unsigned gsteps = (batched && reevaluate && !in_lockstep) ? graph_steps : 0;

    // ---- look-ahead walk: a shadow set of evaluators over the same tables, bound to the look-ahead vector
    bool ahead = (lookahead || lookahead_auto) && batched && reevaluate && consume && !in_lockstep && !lut_output &&
                 nparameters <= 256;
    if (ahead && lookahead_auto && !lookahead) {
      int members = 0;
      unsigned long long rows = 0, exact_rows = 0, never_rows = 0;
      check(sxmc_group_codes_info(group, &members, &rows, &exact_rows, &never_rows));
      ahead = members == 0;     // (a plan over codes walks sequentially)
    }
    if (ahead) {
      // not every shape is offered the look-ahead pass (histograms beyond LDS; problems so small that the sequential
      // step ends in the one-workgroup form, whose event sum is partitioned differently): those walk sequentially
      int ok = 0;
      check(sxmc_group_lookahead_supported(group, &ok));
      ahead = ok != 0;
    }
    std::vector<std::unique_ptr<pdfz::EvalHist>> shadow;
    sxmc_group_t shadow_group = nullptr;
    sxmc_multigroup_t pair = nullptr;
    pdfz::Array<double> ahead_vector(nparameters, true);
    pdfz::Array<unsigned> ahead_norms(nsignals, true);
    pdfz::Array<float> ahead_lut(ahead ? nevents * nsignals : 1, true);
    pdfz::Array<int> ahead_stop(1, true);
    struct AheadGuard {   // (the multigroup goes before its groups, the group before its evaluators)
      sxmc_multigroup_t* pair;
      sxmc_group_t* group;
      ~AheadGuard() {
        if (*pair) sxmc_multigroup_destroy(*pair);
        if (*group) sxmc_group_destroy(*group);
      }
    } ahead_guard{&pair, &shadow_group};
    if (ahead) {
      ahead_vector.writeOnlyHostPtr();
      ahead_norms.writeOnlyHostPtr();
      ahead_lut.writeOnlyHostPtr();
      std::vector<sxmc_hist_t> handles;
      for (size_t i = 0; i < pdfs.size(); i++) {
        pdfz::EvalHist* base = dynamic_cast<pdfz::EvalHist*>(pdfs[i]);
        shadow.emplace_back(new pdfz::EvalHist(*base, pdfz::EvalHist::SharedSamples{}));
        pdfz::EvalHist* p = shadow.back().get();
        p->SetEvalPoints(data);
        p->SetPDFValueBuffer(&ahead_lut, (int)(i * nevents), 1);
        p->SetNormalizationBuffer(&ahead_norms, (int)i);
        p->SetParameterBuffer(&ahead_vector, (int)nsources);
        p->Bind();
        handles.push_back(p->Handle());
      }
      check(sxmc_group_create(handles.data(), (int)handles.size(), &shadow_group));
      check(sxmc_group_set_lut_output(shadow_group, 0));
      // a pass of two evaluations is bound by vector issue and needs more registers than 1024 lanes leave each
      // (spills inside the stream loop drain the loads in flight): 768 lanes, the kernel compiled for that bound
      check(sxmc_group_set_launch_config(group, 768, 1));
      check(sxmc_group_set_launch_config(shadow_group, 768, 1));
      sxmc_group_t both[2] = {group, shadow_group};
      check(sxmc_multigroup_create(both, 2, &pair));
    }
    bool ahead_planned = false;   // the look-ahead pair has launched once: its plans exist
    bool setup_announced = false; // on_setup_done has been called
    sxmc_stream_t strm = stream;
    sxmc_graph_t graph = nullptr;
    const bool own_stream = gsteps > 0 && !strm;
    if (own_stream) check(sxmc_stream_create(&strm));

    // Device pointers of one run of steps, resolved once per run: the accessors may copy (after the
    // host wrote a counter or the widths), which must not happen while a graph is being recorded.
    struct {
      const float* lut;
      const double *means, *sigmas, *nexpected;
      const unsigned* n_mc;
      const short* source_id;
      const float* jump_width;
      double *proposed, *current, *sums, *nll_current, *nll_proposed;
      unsigned* norms;
      int *accepted, *counter;
      float* jump_buffer;
      RNGState* rng;
    } d;
    auto resolve = [&]() {
      d.lut = lut.readOnlyPtr();
      d.means = parameter_means->readOnlyPtr();
      d.sigmas = parameter_sigma->readOnlyPtr();
      d.nexpected = nexpected->readOnlyPtr();
      d.n_mc = n_mc->readOnlyPtr();
      d.source_id = source_id->readOnlyPtr();
      d.jump_width = jump_width.readOnlyPtr();
      d.proposed = proposed_vector.ptr();
      d.current = current_vector.ptr();
      d.sums = event_partial_sums.ptr();
      d.nll_current = current_nll.ptr();
      d.nll_proposed = proposed_nll.ptr();
      d.norms = normalizations.ptr();
      d.accepted = accept_counter.ptr();
      d.counter = jump_counter.ptr();
      d.jump_buffer = jump_buffer.writeOnlyPtr();
      d.rng = rngs->ptr();
    };
    auto one_step = [&]() {
      int npartial = (int)nnllthreads;
      if (batched && reevaluate && consume) {
        // two launches: fill of all signals; lookup + event sum + step end + clearing for the next step
        check(sxmc_group_step_async(group, strm, d.means, d.sigmas, d.rng, d.nll_current, d.nll_proposed, d.current,
                                    d.proposed, d.accepted, d.counter, d.jump_buffer, (int)nparameters, nsources,
                                    d.jump_width, d.nexpected, d.n_mc, d.source_id, d.norms, debug_mode ? 1 : 0));
        return;
      }
      if (batched && reevaluate) {
        // zero, fill of all signals in one kernel, lookup fused with the event partial sums
        check(sxmc_group_eval_nll_async(group, strm, d.proposed, d.nexpected, d.n_mc, d.source_id, d.norms, d.sums,
                                        &npartial));
      } else {
        if (reevaluate) {
          // mcmc.cpp:264-271 as written.  (The evaluators launch on their own streams, which order with the legacy
          // default stream -- where the reference launches its NLL kernels -- and with nothing else: a walk that was
          // given its own stream waits for its step end before the evaluators read the new proposal.)
          if (strm) check(sxmc_stream_synchronize(strm));
          for (pdfz::Eval* p : pdfs) p->EvalAsync();
          for (pdfz::Eval* p : pdfs) p->EvalFinished();
        }
        SXMC_KERNEL_LAUNCH(nll_event_chunks, nnllblocks, nllblocksize, 0, strm, d.lut, d.proposed, nevents, nsignals,
                           d.nexpected, d.n_mc, d.source_id, d.norms, d.sums);
      }
      SXMC_KERNEL_LAUNCH(finish_nll_jump_pick_combo, 1, nreducethreads, nreducethreads * sizeof(double), strm,
                         (size_t)npartial, d.sums, nsignals, nsources, d.means, d.sigmas, d.rng, d.nll_current,
                         d.nll_proposed, d.current, d.proposed, d.accepted, d.counter, d.jump_buffer,
                         (int)nparameters, d.jump_width, d.nexpected, d.n_mc, d.source_id, d.norms, debug_mode);
    };
    // Steps after which the jump buffer is read back (mcmc.cpp:351-377)
    // (a plan with a boxed and an ordered form of the fill is asked for its form at every flush, from the parameters at
    // that moment, and a chain moves -- config 3's resolution parameter by ~0.05 in 5 000 steps: flushes every
    // adapt_interval steps bound how stale the choice gets; the chain does not depend on where the flushes fall)
    bool two_forms = false;
    auto flush_due = [&](unsigned i) {
      return i % sync_interval == 0 || i == nsteps - 1 || i == burnin_steps - 1 || i == 2 * burnin_steps - 1 ||
             (two_forms && adapt_interval > 0 && i % adapt_interval == adapt_interval - 1);
    };

    unsigned i = 0;
    const std::chrono::steady_clock::time_point steps_t0 = std::chrono::steady_clock::now();
    chain.setup_seconds = std::chrono::duration<double>(steps_t0 - walk_t0).count();
    while (i < nsteps) {
      // Re-tune the proposal from the burn-in samples (mcmc.cpp:274-311); the width becomes
      // scale_factor x the standard deviation of the parameter over the steps kept so far
      if (i == burnin_steps || i == 2 * burnin_steps) {
        for (size_t j = 0; j < nparameters; j++) {
          if (parameter_fixed[j]) continue;
          const double sd = column_stddev(chain, j);
          const double fit_width = sd > 0 ? sd : jump_width.readOnlyHostPtr()[j];
          jump_width.hostPtr()[j] = scale_factor * fit_width;
        }
        if (!debug_mode) chain.rows.clear();
      }

      // steps i..f need the host only before the first and after the last
      unsigned f = i;
      while (!flush_due(f)) f++;
      unsigned n = f - i + 1;
      resolve();
      // a plan with a boxed and an ordered form of the fill: the form of the steps up to the next flush, from the
      // parameters the device holds now (sxmc_group_adapt_fill_form); recorded steps replay the old form, so they are
      // recorded again further down
      if (batched && reevaluate && !in_lockstep && !ahead) {
        int form = 0, changed = 0;
        check(sxmc_group_adapt_fill_form(group, &form, &changed));
        two_forms = form != 0 && adapt_interval < sync_interval;
        if (changed && graph) {
          check(sxmc_graph_destroy(graph));
          graph = nullptr;
        }
      }
      const bool replay = !ahead && gsteps > 0 && i > 0 && n >= gsteps;
      if (replay && !graph) {  // record gsteps steps once; the launch plan is current after the eager step 0
        RecordingScope recording(exclusive, excl.owns_lock());
        check(sxmc_graph_begin_capture(strm));
        try {
          for (unsigned k = 0; k < gsteps; k++) one_step();
        } catch (...) {
          sxmc_graph_end_capture(strm, &graph);
          throw;
        }
        check(sxmc_graph_end_capture(strm, &graph));
      }
      // set-up is over once the first run of steps after step 0 has its graph (or needs none); a lockstep chain
      // must not hold the lock while it waits for its partners, who need it for their own set-up.  The look-ahead
      // walk builds its plans in its first pass and records its graph further down: it keeps the lock until then.
      // The lock goes BEFORE the run is queued: a run is up to sync_interval steps -- a second of device work at
      // config 3 -- and whoever holds the lock while that is queued keeps every other chain's set-up waiting.
      if ((i > 0 || in_lockstep) && !ahead && excl.owns_lock()) excl.unlock();
      if (i > 0 && !ahead && !setup_announced) {
        setup_announced = true;
        if (on_setup_done) on_setup_done();     // (an ensemble's lanes meet here: see MCMC::on_setup_done)
      }
      if (replay) {
        check(sxmc_graph_launch(graph, strm, (int)(n / gsteps)));
        n %= gsteps;
      }
      if (in_lockstep) {
        // this run of steps together with the other chains of the set (recorded and replayed there)
        sxmc_step_args a;
        std::memset(&a, 0, sizeof a);   // (padding too: the set compares argument blocks bytewise)
        a.d_means = d.means;
        a.d_sigmas = d.sigmas;
        a.d_rng = reinterpret_cast<sxmc_rng_state*>(d.rng);
        a.d_nll_current = d.nll_current;
        a.d_nll_proposed = d.nll_proposed;
        a.d_v_current = d.current;
        a.d_v_proposed = d.proposed;
        a.d_accepted = d.accepted;
        a.d_counter = d.counter;
        a.d_jump_buffer = d.jump_buffer;
        a.nparameters = (int)nparameters;
        a.nsources = nsources;
        a.d_jump_width = d.jump_width;
        a.d_nexpected = d.nexpected;
        a.d_n_mc = d.n_mc;
        a.d_source_id = d.source_id;
        a.d_norms = d.norms;
        a.debug_mode = debug_mode ? 1 : 0;
        lockstep->advance(lockstep_index, group, a, n, graph_steps);
        n = 0;
      }
      if (ahead && n > 0) {
        // exactly n more steps, taken one or two per pass: passes in rounds of about what is still needed, the step
        // counter read back after each round; a pass beyond the stop does nothing (the counter was 0 at the flush)
        sxmc_step_args a;
        std::memset(&a, 0, sizeof a);
        a.d_means = d.means;
        a.d_sigmas = d.sigmas;
        a.d_rng = reinterpret_cast<sxmc_rng_state*>(d.rng);
        a.d_nll_current = d.nll_current;
        a.d_nll_proposed = d.nll_proposed;
        a.d_v_current = d.current;
        a.d_v_proposed = d.proposed;
        a.d_accepted = d.accepted;
        a.d_counter = d.counter;
        a.d_jump_buffer = d.jump_buffer;
        a.nparameters = (int)nparameters;
        a.nsources = nsources;
        a.d_jump_width = d.jump_width;
        a.d_nexpected = d.nexpected;
        a.d_n_mc = d.n_mc;
        a.d_source_id = d.source_id;
        a.d_norms = d.norms;
        a.debug_mode = debug_mode ? 1 : 0;
        ahead_stop.writeOnlyHostPtr()[0] = (int)n;
        const int* d_stop = ahead_stop.readOnlyPtr();
        double* d_ahead = ahead_vector.ptr();
        const unsigned* d_anorms = ahead_norms.ptr();
        // the look-ahead vector for the chain as it stands (new widths after a re-tuning included)
        check(sxmc_lookahead_begin(strm, (int)nparameters, reinterpret_cast<const sxmc_rng_state*>(d.rng), d.jump_width,
                                   d.current, d_ahead));
        auto one_pass = [&]() {
          check(sxmc_multigroup_lookahead_step_async(pair, strm, &a, d_ahead, d_anorms, d_stop));
          ahead_passes++;
        };
        unsigned done = 0;
        while (done < n) {
          const unsigned need = n - done;
          const double rate = ahead_passes_seen >= 16 ? std::min(2.0, 1.03 * ahead_steps_seen / ahead_passes_seen) : 1.75;
          unsigned k = std::max(1u, (unsigned)(need / rate));
          const size_t p0 = ahead_passes;
          // The pair's first pass builds its launch plans (allocations, a module load, a device-wide synchronisation)
          // and the recording must not meet another thread's allocation: both under the set-up lock, like the
          // sequential walk's recording.  Every other round only launches and waits on this chain's stream.
          const bool records = gsteps > 0 && k > gsteps && !graph;
          if (exclusive) {
            if (!ahead_planned || records) {
              if (!excl.owns_lock()) excl.lock();
            } else if (i > 0 && excl.owns_lock()) {
              excl.unlock();
            }
          }
          if (gsteps > 0 && k > gsteps) {
            if (!graph) {
              one_pass();   // (plans in place before recording)
              ahead_planned = true;
              k--;
              RecordingScope recording(exclusive, excl.owns_lock());
              check(sxmc_graph_begin_capture(strm));
              try {
                for (unsigned q = 0; q < gsteps; q++) one_pass();
              } catch (...) {
                sxmc_graph_end_capture(strm, &graph);
                throw;
              }
              check(sxmc_graph_end_capture(strm, &graph));
              ahead_passes -= gsteps;
              if (i > 0 && excl.owns_lock()) excl.unlock();
            }
            check(sxmc_graph_launch(graph, strm, (int)(k / gsteps)));
            ahead_passes += (size_t)(k / gsteps) * gsteps;
            k %= gsteps;
          }
          for (unsigned q = 0; q < k; q++) {
            one_pass();
            if (!ahead_planned) {
              ahead_planned = true;
              if (i > 0 && excl.owns_lock()) excl.unlock();
            }
          }
          (void)jump_counter.ptr();   // (the device side changed behind the mirror's back: the host copy is stale)
          const unsigned now = (unsigned)jump_counter.readOnlyHostPtr()[0];   // (a blocking copy on the chain's stream)
          if (now <= done && now < n) throw pdfz::Error("look-ahead walk: the chain did not advance");
          ahead_steps_seen += now - done;
          ahead_passes_seen += ahead_passes - p0;
          done = now;
        }
        n = 0;
      }
      for (unsigned k = 0; k < n; k++) one_step();

      // Flush the jump buffer (mcmc.cpp:351-377); the host reads go through blocking copies
      const int njumps = jump_counter.readOnlyHostPtr()[0];
      const int naccepted = accept_counter.readOnlyHostPtr()[0];
      if (verbose) {
        std::printf("MCMC: Step %u/%u (%d in buffer, %d accepted)\n", f + 1, nsteps, njumps, naccepted);
      }
      const float* jb = jump_buffer.readOnlyHostPtr();
      chain.rows.insert(chain.rows.end(), jb, jb + (size_t)njumps * ncol);
      chain.accepted += (size_t)naccepted;
      jump_counter.writeOnlyHostPtr()[0] = 0;
      accept_counter.writeOnlyHostPtr()[0] = 0;
      if (batched) {
        // the cooperative step end waits inside its kernel, with a bound: a wait that ran into it left the steps of
        // this run invalid -- never seen on a healthy device, and not to be passed on silently if it ever happens
        unsigned timeouts = 0;
        check(sxmc_group_step_end_timeouts(group, strm, &timeouts));
        if (timeouts) {
          throw pdfz::Error("MCMC: " + std::to_string(timeouts) + " workgroup(s) of the cooperative step end gave up "
                            "waiting (sxmc_group_step_end_timeouts): the chain is not valid");
        }
      }
      i = f + 1;
    }
    if (strm) check(sxmc_stream_synchronize(strm));
    chain.steps_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - steps_t0).count();
    if (on_steps_done) on_steps_done();
    if (lockstep) lockstep->leave(lockstep_index);
    // the evaluators borrowed this walk's arrays (lookup table, normalisations, parameter vectors): un-bind them
    // before they die, or the next evaluation of an evaluator would touch destroyed arrays
    for (pdfz::Eval* p : pdfs) p->ForgetBuffers();
    for (auto& p : shadow) p->ForgetBuffers();
    if (exclusive && !excl.owns_lock()) excl.lock();  // tear-down frees device memory
    if (graph) check(sxmc_graph_destroy(graph));
    if (own_stream) check(sxmc_stream_destroy(strm));
    if (stream) {
      check(sxmc_stream_synchronize(stream));  // this chain only: others may be running beside it
    } else {
      check(sxmc_device_synchronize());
    }
    return chain;
  }

  size_t NumParameters() const { return nparameters; }
  /** Look-ahead walk: passes over the tables launched so far (each evaluates twice and takes one or two steps). */
  size_t LookaheadPasses() const { return ahead_passes; }

  /** The launch plan of the chain's group as the library describes it (sxmc_group_launch_info): which table form and
   *  kernel each launch of the fill takes.  Empty when the walk is not batched. */
  std::string LaunchPlan() const {
    if (!group) return std::string();
    char buf[2048];
    buf[0] = 0;
    if (sxmc_group_launch_info(group, buf, sizeof buf) != SXMC_OK) return std::string();
    return std::string(buf);
  }

 protected:
  /** MCMC::nll (mcmc.cpp:390-415): three launches over an evaluated lookup table. */
  void nll(const float* lut, size_t nevents, const double* v, double* out, const unsigned* norms,
           double* event_partial_sums, double* event_total_sum) {
    SXMC_KERNEL_LAUNCH(nll_event_chunks, nnllblocks, nllblocksize, 0, stream, lut, v, nevents, nsignals,
                       nexpected->readOnlyPtr(), n_mc->readOnlyPtr(), source_id->readOnlyPtr(), norms,
                       event_partial_sums);
    SXMC_KERNEL_LAUNCH(nll_event_reduce, 1, nreducethreads, nreducethreads * sizeof(double), stream,
                       (size_t)nnllthreads, event_partial_sums, event_total_sum);
    SXMC_KERNEL_LAUNCH(nll_total, 1, 1, 0, stream, nparameters, v, nsignals, nsources,
                       parameter_means->readOnlyPtr(), parameter_sigma->readOnlyPtr(), event_total_sum,
                       nexpected->readOnlyPtr(), n_mc->readOnlyPtr(), source_id->readOnlyPtr(), norms, out);
  }

  static double column_stddev(const Chain& c, size_t col) {
    const size_t n = c.nrows();
    if (n < 2) return 0.0;
    double s = 0, s2 = 0;
    for (size_t r = 0; r < n; r++) {
      const double x = c.at(r, col);
      s += x;
      s2 += x * x;
    }
    const double var = s2 / n - (s / n) * (s / n);
    return var > 0 ? std::sqrt(var) : 0.0;
  }

 private:
  sxmc_stream_t stream;  //!< every launch of this chain goes here (null: the legacy default stream, as the
                         //!< reference; one non-blocking stream per chain when several run on one GPU)
  size_t nsources, nsignals, nsystematics, nobservables;
  size_t nparameters = 0, nfloat = 0;
  bool systematics_fixed = true;
  unsigned nnllblocks, nllblocksize, nnllthreads, nreducethreads;
  std::unique_ptr<pdfz::Array<double>> parameter_means, parameter_sigma, nexpected;
  std::unique_ptr<pdfz::Array<unsigned>> n_mc;
  std::unique_ptr<pdfz::Array<short>> source_id;
  std::unique_ptr<pdfz::Array<RNGState>> rngs;
  std::vector<std::string> parameter_names;
  std::vector<bool> parameter_fixed;
  std::vector<pdfz::Eval*> pdfs;
  sxmc_group_t group = nullptr;
  bool optimized = false;
  size_t ahead_passes = 0, ahead_passes_seen = 0;   //!< look-ahead walk: passes launched / counted in the rate below
  double ahead_steps_seen = 0;
};

}  // namespace sxmc
