// sxmc_host.h -- what the translation units of libsxmc_hip.so's host side share: the handle structures behind the C ABI
// (include/sxmc_hip.h), the error and ordering macros every entry point starts with, and the functions one unit offers
// the others.  Internal.  The units, by concern:
//   sxmc_runtime.cpp      errors, tracing, device / memory / stream / graph / event entry points, the lazy EvalFinished's
//                         bookkeeping (settle), the capture gate
//   sxmc_launch_plan.cpp  a group's launch plan: tables (bucketed copies, codes, sparse structures), launch classes,
//                         partitions, refresh; event classes; the fill launches (group_fill)
//   sxmc_evaluator.cpp    the evaluator (sxmc_hist_*) and the deferred batches behind its EvalAsync / EvalFinished
//   sxmc_group.cpp        the group entry points: configuration, autotune, evaluation, the step and its step-end forms
//   sxmc_multigroup.cpp   lockstep chains and the look-ahead pass
//   sxmc_nll_api.cpp      the NLL launch points with the reference's argument lists; the measurement build's test hooks
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <rocprofiler-sdk-roctx/roctx.h>

#include "sxmc_device.h"
#include "sxmc_plan.h"

extern "C" {
hipError_t sx_nll_init_rngs(int, int, hipStream_t, int, unsigned long long, sxmc_rng_state*);
hipError_t sx_nll_philox_dump(hipStream_t, sxmc_rng_state*, unsigned*, int);
hipError_t sx_nll_pick_new_vector(int, int, hipStream_t, int, sxmc_rng_state*, const float*, const double*,
                                  double*);
hipError_t sx_nll_jump_decider(int, int, hipStream_t, sxmc_rng_state*, double*, const double*, double*,
                               const double*, unsigned, int*, int*, float*);
hipError_t sx_nll_event_chunks(int, int, hipStream_t, const float*, const double*, size_t, size_t,
                               const double*, const unsigned*, const short*, const unsigned*, double*);
hipError_t sx_nll_event_reduce(int, hipStream_t, size_t, const double*, double*);
hipError_t sx_nll_total(hipStream_t, size_t, const double*, size_t, size_t, const double*, const double*,
                        const double*, const double*, const unsigned*, const short*, const unsigned*, double*);
hipError_t sx_nll_finish_combo(int, hipStream_t, size_t, const double*, size_t, size_t, const double*,
                               const double*, sxmc_rng_state*, double*, double*, double*, double*, int*, int*,
                               float*, int, const float*, const double*, const unsigned*, const short*,
                               const unsigned*, bool);
}


// MEASUREMENT BUILD (make VARIANT=_measure EXTRA=-DSXMC_MEASURE=1): the kernels' `dbg` hooks, the entry points that set
// them and the environment switches that exist for A/B runs only.  The product library has none of them: measure_env()
// is the constant nullptr there, and what remains readable from the environment are documented defaults that the ABI
// can set too (SXMC_ROCTX, SXMC_CODES, SXMC_DEFER_EVAL, SXMC_LAZY_FINISH, SXMC_COOP_STEP_END, SXMC_FUSED_STEP).
#ifndef SXMC_MEASURE
#define SXMC_MEASURE 0
#endif

struct sxmc_group;
struct sxmc_hist;

namespace sxhost {

// ---- errors, tracing (sxmc_runtime.cpp)
extern thread_local std::string g_last_error;
extern thread_local bool t_capturing;                       // this thread is recording a HIP graph (sxmc_graph_begin_capture)
extern thread_local unsigned long long t_capture_epoch;     // one per recording
extern thread_local std::vector<sxmc_group*> t_capture_groups;   // groups launched in the current recording
int fail(int code, const std::string& msg);
inline const char* measure_env(const char* name) {
#if SXMC_MEASURE
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
bool tracing();
struct TraceRange {
  bool on;
  explicit TraceRange(const char* name) : on(tracing()) {
    if (on) roctxRangePushA(name);
  }
  ~TraceRange() {
    if (on) roctxRangePop();
  }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};


}  // namespace sxhost

#define SX_HIP(expr)                                                                                                \
  do {                                                                                                              \
    hipError_t _e = (expr);                                                                                         \
    if (_e != hipSuccess) {                                                                                         \
      return ::sxhost::fail(SXMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ ":" +    \
                                              std::to_string(__LINE__) + ")");                                       \
    }                                                                                                               \
  } while (0)

#define SX_REQUIRE(cond, msg)                                  \
  do {                                                         \
    if (!(cond)) return ::sxhost::fail(SXMC_ERR_INVALID, msg); \
  } while (0)

namespace sxhost {
constexpr int kLdsMaxBins = 40960 - 128;  // 160 KiB of LDS per workgroup minus header and trash words

struct HostSyst {
  int type, obs, extra_field;
  std::vector<short> pars;
};

struct DeviceProps {
  int cus = 0;
  int lds_per_cu = 0;
  bool valid = false;
};

int get_props(DeviceProps& p);
}  // namespace sxhost

// The immutable part of an evaluator: the column-major sample table and (built on demand) the
// pre-binned column.  Shared by evaluators created with sxmc_hist_create_shared, so that several
// chains / experiments on one GPU read ONE copy of the MC tables.
struct SampleStore {
  float* d_cols = nullptr;
  // pre-binned columns built so far, one per (observable mask, width); kept until the table dies because
  // descriptors of other groups may point at them
  struct PreColumn {
    unsigned mask;
    int width;
    void* ptr;
  };
  std::vector<PreColumn> pre;
  std::mutex pre_mutex;  // evaluators sharing the table may be set up from different host threads
  void* find_pre(unsigned mask, int width) const {
    for (const PreColumn& p : pre)
      if (p.mask == mask && p.width == width) return p.ptr;
    return nullptr;
  }
  // Bucketing (layout_kernels.hip).  The SORT of the rows by the untouched observables' bin indices is done once
  // per set of untouched observables (`mask`) and kept; `rejected`: it was tried and the padding of the granules
  // would have outweighed the columns saved (tiny tables with many buckets).
  struct BucketSort {
    unsigned mask = 0;
    int ordered = -1;               // observable whose raw value orders the rows inside every bucket, or -1
    int box_truth = -1;             // >= 0: `ordered` is a BOXED observable (fill_boxed_kernel) and this the truth field
                                    // its resolution scale reads: rows of a bucket go by (stratum of x - t, x)
    bool rejected = true;
    unsigned* d_rows = nullptr;     // [nsamples] row numbers in sorted order (rows outside the domain last)
    size_t nkept = 0;               // rows inside the domain of every untouched observable
    unsigned nkeys_total = 0;       // size of the key space (= the key of "outside")
    unsigned radix[SXMC_MAX_NFIELDS] = {0};
    std::vector<unsigned> keys;     // bucket keys present, ascending
    std::vector<unsigned> key_pre;  // bin offset of each of them
    std::vector<unsigned> lsrc, lvalid, lwhich;  // logical granules: first sorted position, rows, index into keys
  };
  // ... and the COPIES laid out from it, one per (mask, streamed fields, runs); kept until the table dies because
  // descriptors of other groups may point at them.
  struct Bucketed {
    unsigned mask = 0;
    std::vector<int> fields;        // streamed fields, in the order of the compacted slots
    int runs = 1;                   // granule order transposed for this many runs (1 = sorted order)
    bool complete = false;
    const BucketSort* sort = nullptr;
    float* d_cols = nullptr;        // [fields.size()][pitch]
    size_t pitch = 0;
    unsigned* d_gpre = nullptr;     // [ngranules] bin offset of the granule
    unsigned* d_gkp = nullptr;      // [ngranules] pairs {bucket key, bin offset}
    float* d_gedge = nullptr;       // [ngranules] pairs {first, last} value of the sort's ordered observable
    float* d_gbox = nullptr;        // [ngranules] {xmin, xmax, tmin, tmax} of the sort's boxed observable and its truth field
    double box_dx = 0, box_dt = 0;  // ... their mean extents over the granules with finite boxes
    bool q16 = false;               // d_qcol holds ONE 16-bit code per row of field 0 (boxed tables)
    size_t ngranules = 0;           // physical granules (with the runs' padding)
    size_t nkept = 0;               // samples in the copy
    // CODES (fill_ordered_body): the streamed fields once more as 16-bit codes, two per word; built on demand
    unsigned* d_qcol = nullptr;     // [(nq + 1) / 2][pitch]
    int nq = 0;                     // fields coded (the streamed ones: all but the ordered observable's)
    bool codes_tried = false;
    double qbase[SXMC_MAX_QSLOTS] = {0}, qstep[SXMC_MAX_QSLOTS] = {0};
    unsigned long long q_exact_rows = 0, q_never_rows = 0;   // rows marked "ask the exact columns" / "never counted"
  };
  std::vector<std::unique_ptr<BucketSort>> sorts;
  std::vector<std::unique_ptr<Bucketed>> bucketed;
  BucketSort* find_sort(unsigned mask, int ordered, int box_truth = -1) const {
    for (const auto& b : sorts)
      if (b->mask == mask && b->ordered == ordered && b->box_truth == box_truth) return b.get();
    return nullptr;
  }
  Bucketed* find_bucketed(const BucketSort* sort, const std::vector<int>& fields, int runs) const {
    for (const auto& b : bucketed)
      if (b->sort == sort && b->fields == fields && b->runs == runs) return b.get();
    return nullptr;
  }
  ~SampleStore() {
    if (d_cols) (void)hipFree(d_cols);
    for (PreColumn& p : pre) (void)hipFree(p.ptr);
    for (auto& b : bucketed) {
      if (b->d_cols) (void)hipFree(b->d_cols);
      if (b->d_gpre) (void)hipFree(b->d_gpre);
      if (b->d_gkp) (void)hipFree(b->d_gkp);
      if (b->d_gedge) (void)hipFree(b->d_gedge);
      if (b->d_gbox) (void)hipFree(b->d_gbox);
      if (b->d_qcol) (void)hipFree(b->d_qcol);
    }
    for (auto& b : sorts)
      if (b->d_rows) (void)hipFree(b->d_rows);
  }
};

struct sxmc_hist {
  std::shared_ptr<SampleStore> store;
  int nfields = 0, nobs = 0;
  size_t nsamples = 0, nvec = 0, pitch = 0;
  unsigned dataset = 0;
  std::vector<double> lower, upper, scale;
  std::vector<int> nbins, stride;
  int total_nbins = 0;
  double bin_volume = 0;
  unsigned* d_bins = nullptr;
  int* d_read_bins = nullptr;
  size_t read_bins_cap = 0;        // (grow-only: a new data set of about the same size re-uses the buffer)
  std::vector<void*> retired;      // outgrown device buffers, freed with the evaluator
  unsigned* d_cdf = nullptr;       // prefix sums of the histogram, for sxmc_hist_random_sample
  float* d_sample = nullptr;       // ... and the rows it draws (grow-only: a fake experiment per walk draws about as many)
  size_t cap_sample = 0;           // bytes
  bool has_points = false;
  size_t npoints = 0;
  float* pdf = nullptr;
  int pdf_off = 0, pdf_stride = 1;
  unsigned* norm = nullptr;
  int norm_off = 0;
  const double* params = nullptr;
  int par_off = 0, par_stride = 1;
  std::vector<sxhost::HostSyst> systs;
  hipStream_t stream = nullptr;
  unsigned long long version = 1;
  sxmc_group* self = nullptr;
  int cfg_threads = 0, cfg_bpc = 0;
  bool outputs_host_visible = false;   // a bound output (norm, pdf values) lies in memory the host reads directly
  bool want_optimize = true;       // EvalHist's `optimize` (pdfz.cpp:188, 441-448): trial launches at the first evaluation
  // sparse counting (histogram too large for LDS): one counter per distinct event bin
  unsigned* d_cnt = nullptr;       // [ntargets]
  int* d_read_slot = nullptr;      // [npoints]: counter slot of each event, or -1 / -2
  unsigned* d_filter = nullptr;
  unsigned* d_table = nullptr;
  unsigned* d_coarse = nullptr;    // coarse filter staged in LDS by the fill kernel
  int ntargets = 0, filter_shift = 0, table_shift = 0, coarse_shift = 0;
  // the event bins grouped by bucket, for sparse counting over a bucketed table walked in runs (fill_sparse_kernel)
  unsigned* d_bdir = nullptr;      // [nkeys + 1] pairs {first table entry, log2 size | flag}
  unsigned* d_btkeys = nullptr;
  unsigned* d_btslot = nullptr;
  unsigned btab_mask = 0;
  unsigned long long btab_points_version = 0;
  bool btab_valid = false;
  std::vector<unsigned> targets;   // sorted distinct event bins (host copy: members with equal sets share tables)
  std::vector<int> h_read_bins;    // host copies of d_read_bins / d_read_slot: the group forms event classes from them
  std::vector<int> h_read_slot;
  unsigned long long points_version = 0;
  bool bins_valid = true;          // false after a sparse evaluation: the dense histogram was not filled
  const sxmc_group* cleared_by = nullptr;  // the group whose finish_step cleared this histogram and nothing has
                                           // counted into it since (any group's fill resets it)
  // sxmc_hist_eval_async defers (see "deferred evaluations" below)
  std::shared_ptr<struct DeferredBatch> deferred;  // the host thread's batch of evaluations not launched yet it sits in
  std::shared_ptr<struct BatchInFlight> inflight;  // the batched launch the last sxmc_hist_eval_async went into
};

// A host thread's evaluations asked for and not launched yet (sxmc_hist_eval_async).  Filled and launched by its
// thread only; another thread may only take an evaluator OUT of it (sxmc_hist_destroy), under the mutex.
struct DeferredBatch {
  std::mutex m;
  std::vector<sxmc_hist*> members;
  std::atomic<size_t> n{0};                // members.size(), readable without the mutex
  int do_eval_pdf = 0;
  std::thread::id owner;
};

// One batched launch of deferred evaluations: the stream it went to, and whether some member's EvalFinished has
// already waited for it (the siblings' EvalFinished then return at once: S - 1 runtime calls saved per step).
struct BatchInFlight {
  hipStream_t stream = nullptr;
  std::atomic<bool> done{false};
  bool host_visible = false;   // some member writes its results where the host reads them directly: EvalFinished waits
};


namespace sxhost {
struct LaunchClass {
  SxLaunchShape shape;
  std::vector<unsigned> prog;  // one word per systematic when every one has a single coefficient
  bool prog_simple = false;
  unsigned pre_mask = 0;       // observables no systematic writes, streamed as one pre-binned column
  std::vector<int> member_idx;
  SxSignalDesc* d_descs = nullptr;
  SxSignalDesc* d_descs_sparse = nullptr;  // same members, sparse flavour (global-histogram classes)
  SxSegment* d_segs = nullptr;
  unsigned* d_blk_off = nullptr;
  unsigned long long total_vec = 0;
  int partition = 0;  // 1 sliced, 2 interleaved (what build_partition chose)
  bool light = false; // a pure stream: runs best with few waves per CU (see group_rebuild)
  bool runs_mode = false;  // bucketed tables laid out in per-wave runs; the sparse flavour runs fill_sparse_kernel
  int teams = 1;           // teams of workgroups per member over a bucketed table (sxplan::interleaved_segments)
  bool codes = false;      // ordered tables: the streamed columns go as 16-bit codes (fill_ordered_body's CODES)
  bool dual = false;       // boxed tables with an ordered twin plan beside them (sxmc_group::twin, sxmc_group_adapt_fill_form)
  float box_dx = 0, box_dt = 0;   // ... mean extents of the members' granule boxes, what the choice is made from
  int box_obs = -1, box_truth = -1;   // ... the boxed observable and its truth field, as slots of the members' full descriptors
  unsigned padded_rstride = 0;  // ... with the LDS histogram in the padded form: words between its replicas (0: not)
  unsigned plain_rstride = 0;   // ordered tables: words between the replicas of the LDS histogram in its swizzled form
};

void free_class(LaunchClass& c);
}  // namespace sxhost

struct sxmc_group {
  std::vector<sxmc_hist*> members;
  std::vector<unsigned long long> seen;
  std::vector<unsigned long long> seen_points;   // members' points_version at the last (re)plan
  std::vector<SxSignalDesc> h_descs;
  SxSignalDesc* d_descs = nullptr;  // member order: zero / eval kernels
  SxSignalDesc* d_descs_sparse = nullptr;  // member order, sparse flavour where a member supports it
  bool sparse_ready = false;        // some member has sparse structures and all of those have points
  int max_bins_sparse = 0;
  int cfg_sparse = 1;               // count only the event bins when evaluating for lookup
  std::vector<sxhost::LaunchClass> classes;
  int cfg_threads = 0, cfg_bpc = 0;
  int cfg_seen_threads = -1, cfg_seen_bpc = -1;
  int cfg_partition = 0, cfg_seen_partition = -1;  // 0 auto, 1 sliced, 2 interleaved
  int cfg_teams = 0, cfg_seen_teams = -1;          // teams per member over a bucketed table (0 = 1, the default)
  int cfg_prebin = 1, cfg_seen_prebin = -1;        // pre-bin the observables no systematic writes
  int cfg_bucket = 1, cfg_seen_bucket = -1;        // stream a bucketed copy of the table where that pays
  bool order_blocked = false;                      // a plan with ordered tables beyond LDS could not be laid out in runs
  int cfg_order = 1, cfg_seen_order = -1;          // ... with the rows of a bucket ordered by a monotonically written observable
  int cfg_box = -1, cfg_seen_box = -2;             // ... or grouped into boxes of a two-field observable (-1: where it pays, 1: wherever it applies)
  bool box_blocked = false;                        // a plan with boxed tables found no room for its codes / LDS form: planned again without
  sxmc_group* twin = nullptr;                      // boxed plan, cfg_box < 0: the same members planned in the ORDERED form (owned);
                                                   // group_fill launches the one or the other (fill_form)
  bool is_twin = false;
  int fill_form = 2;                               // 1: the boxed plan's launches, 2: the twin's (where there is a twin)
  double* h_pin = nullptr;                         // pinned host words sxmc_group_adapt_fill_form reads the parameters into
  unsigned adapt_tick = 0;                         // deferred batches launched for this group (every 256th asks for the form)
  float box_limit = 0.12f;                         // sxmc_group_adapt_fill_form: boxed while the image of a mean box is narrower (bins)
  int cfg_rtc = 1, cfg_seen_rtc = -1;              // specialise the fill kernel at run time for programs not built in
  int cfg_codes = -1, cfg_seen_codes = -2;         // ordered tables streamed as 16-bit codes (-1: SXMC_CODES, default on)
  int cfg_queue_log = 0, cfg_seen_queue_log = -1;  // ... cap on the queues of ambiguous rows, log2(entries) (0: what fits)
  std::string rtc_note;                            // why a run-time specialisation could not be had (last failure)
  std::string plan_note;                           // why a launch of the plan took a slower general path (for launch_info)
  int cfg_tail = 1;                                // sxmc_group_step_async: the one-workgroup step end where it fits
  bool tuned = false;                              // a deferred batch's group: the launch-shape trials have run
  int trial_launches = 0;                          // fills sxmc_group_optimize has launched for this group (all calls)
  int cfg_coop = -1;                               // ... and the cooperative one-launch step end (-1: SXMC_COOP_STEP_END, default on)
  int coop_fits = -1;                              // ... decided at the group's first step: do the step ends of all chains
                                                   //     stepping in this process fit the device many times over?
  int last_step_launches = 0;                      // kernels the last sxmc_group_step_async launched
  unsigned long long plan_generation = 0;          // counts launch plans built (a multigroup re-validates on change)
  std::vector<const SampleStore::Bucketed*> member_bucket;  // per member: the copy its fill streams, or null
  int debug_mode = 0;
  // measurement build only (the gated step): the stream this group's NEXT step puts its fill on, beside the step end of
  // the step before (null: the step is launched on one stream, as always), and the step's index inside its recording
  hipStream_t split_fill_stream = nullptr;
  int split_node = 0;
  std::vector<hipEvent_t> split_events;
  int max_bins = 0;
  unsigned long long max_points = 0;
  bool same_points = false;
  hipStream_t last_stream = nullptr;
  bool built = false;
  // Events grouped by their tuple of bins over the members (used when the lookup table is not wanted,
  // cfg_lut == 0): one row per distinct tuple, weighted by how many events share it.  One set per descriptor
  // flavour (dense bins / sparse counter slots).
  struct EventClasses {
    int* d_rb = nullptr;             // [nmembers][K]
    unsigned* d_weight = nullptr;    // [K]
    SxSignalDesc* d_descs = nullptr; // member descriptors reading the class tables, no lookup-table output
    size_t K = 0, cap_rb = 0, cap_weight = 0;
    bool tables_valid = false, descs_valid = false;
    std::vector<unsigned long long> seen_points;
  };
  EventClasses ec[2];               // [0] dense, [1] sparse flavour
  std::vector<SxSignalDesc> h_descs_sparse;
  int cfg_lut = 1;
  // sxmc_group_finish_step_async zeroes histograms and normalisations for the next evaluation: that
  // evaluation then skips its zero kernel.  0 = nothing pre-zeroed, 1 = dense flavour, 2 = sparse flavour.
  int prezeroed = 0;
  bool last_sparse = false;
  unsigned long long capture_epoch = 0;  // last recording this group launched in
  unsigned* d_ticket = nullptr;  // arrival counter of the fused step end, zeroed by the zero kernel
  double* d_step_sums = nullptr; // 1024 partial sums of the fused step
  std::vector<char> h_tail;          // fused step (fill_step_kernel): the step end's arguments, passed to the kernel by value
  int cfg_seen_fused = -2;
  int cfg_fused = -1;                // the whole step in ONE launch where the fill has that form (-1: SXMC_FUSED_STEP, default OFF: measured slower)
  unsigned long long* d_coop_slots = nullptr;  // cooperative step end: one hand-over slot per worker (step_end_kernel)
  double* d_coop_last = nullptr;               // ... and the last partial of each that was not NaN
  // profiling of the fill kernel
  bool prof = false;
  std::vector<hipEvent_t> ev0, ev1;
  int prof_n = 0;
};


struct sxmc_multigroup {
  std::vector<sxmc_group*> groups;
  std::vector<void*> fill_fn;        // per launch of the plan: the lockstep kernel (hipFunction_t)
  std::vector<size_t> lds_bytes;
  std::vector<unsigned> fill_w;          // the kernels' layout argument: words per histogram, or the ordered fill's replica layout
  std::vector<unsigned long long> seen;  // the groups' plan generations the kernels were chosen for
  std::string why_not;               // set when the chains cannot be stepped together
  bool joint_ends = true;            // the chains' step ends share two launches (sx_launch_chain_ends)
};


namespace sxhost {

// ---- the launch plan (sxmc_launch_plan.cpp)
using sxplan::ordered_queue_bytes;
constexpr unsigned kMinQueueLog = 9;     // the smallest queues of ambiguous rows a fill over codes works with: 2^9 entries
void member_slots(const sxmc_hist* h, std::vector<int>& slot_col);
void free_sparse(sxmc_hist* h);
int build_sparse(sxmc_hist* h, const std::vector<int>& rb);
unsigned ordered_queue_log(size_t room, int cap = 0);
bool fused_step_requested(const sxmc_group* g);
int group_refresh(sxmc_group* g);
int group_check_bound(sxmc_group* g, bool need_pdf);
void free_event_classes(sxmc_group::EventClasses& ec);
int ensure_event_classes(sxmc_group* g, bool sparse);
int group_prepare_fill(sxmc_group* g, hipStream_t s, bool sparse);
int group_fill(sxmc_group* g, hipStream_t s, bool sparse = false);

// ---- deferred evaluations and the lazy EvalFinished (sxmc_evaluator.cpp, sxmc_runtime.cpp)
struct AutoGroup {
  std::vector<sxmc_hist*> members;
  sxmc_group* g = nullptr;
};
extern std::mutex g_auto_mutex;
extern std::vector<AutoGroup> g_auto_groups;     // groups made for batches of two or more (a single evaluator has h->self)
extern thread_local std::shared_ptr<DeferredBatch> t_deferred;
extern std::shared_mutex g_capture_gate;         // shared: a recording in progress; exclusive: a batch on its way to the legacy stream
int flush_deferred();
// (the lazy EvalFinished, sxmc_runtime.cpp: a batch on the legacy stream whose EvalFinished did not wait, per device)
constexpr int kMaxDevices = 64;
extern std::atomic<int> g_lazy_finish;
extern std::atomic<bool> g_unsettled[kMaxDevices];
int current_device_slot();
bool lazy_finish_enabled();
bool host_can_read(const void* p);
int settle();
int settle_for(hipStream_t s);

// ---- the step and its end (sxmc_group.cpp)
extern std::atomic<int> g_stepping_groups;       // groups of this process that have stepped a chain (see note_stepping)
constexpr int kCoopMaxWorkers = 128;
bool step_end_takes_tail(const sxmc_group* g, bool sparse, unsigned long long ne);
int step_sum_blocks(unsigned long long ne);
void note_stepping(sxmc_group* g);
bool step_end_is_cooperative(const sxmc_group* g, unsigned long long ne);
int group_step_tail(sxmc_group* g, hipStream_t st, bool sparse, const SxSignalDesc* descs, unsigned long long ne,
                    const unsigned* weight, const SxStepArgs& a);

}  // namespace sxhost

#define SX_ORDER(s)                                  \
  do {                                               \
    int _rc = ::sxhost::settle_for((hipStream_t)(s));          \
    if (_rc) return _rc;                             \
  } while (0)

#define SX_FLUSH()                                                          \
  do {                                                                      \
    if (::sxhost::t_deferred && ::sxhost::t_deferred->n.load(std::memory_order_relaxed) != 0) { \
      int _rc = ::sxhost::flush_deferred();                                           \
      if (_rc) return _rc;                                                  \
    }                                                                       \
  } while (0)


// For the library's other translation units that are not part of the host side proper (sxmc_comm.cpp)
int sx_flush_and_order(hipStream_t s);
