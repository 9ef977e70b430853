// sxmc_hip.cpp -- host side of libsxmc_hip.so: the C ABI declared in include/sxmc_hip.h.
// Handles, validation, device-resident layout, launch sizing.  The arithmetic lives in
// pdfz_kernels.hip / nll_kernels.hip.  There is no CPU fallback: every evaluation entry point
// launches gfx950 kernels or fails with an error code.
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

namespace {

thread_local std::string g_last_error;
thread_local bool t_capturing = false;  // this thread is recording a HIP graph (sxmc_graph_begin_capture)
thread_local unsigned long long t_capture_epoch = 0;   // one per recording
thread_local std::vector<struct sxmc_group*> t_capture_groups;  // groups launched in the current recording

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// MEASUREMENT BUILD (make VARIANT=_measure EXTRA=-DSXMC_MEASURE=1): the kernels' `dbg` hooks, the entry points that set
// them and the environment switches that exist for A/B runs only.  The product library has none of them: measure_env()
// is the constant nullptr there, and what remains readable from the environment are documented defaults that the ABI
// can set too (SXMC_ROCTX, SXMC_CODES, SXMC_DEFER_EVAL, SXMC_LAZY_FINISH, SXMC_COOP_STEP_END, SXMC_FUSED_STEP).
#ifndef SXMC_MEASURE
#define SXMC_MEASURE 0
#endif
inline const char* measure_env(const char* name) {
#if SXMC_MEASURE
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Host-side roctx ranges around the phases of a step (tracing): what rocprofv3 --marker-trace
// shows beside the kernel trace.  Off unless SXMC_ROCTX=1 is in the environment or sxmc_set_tracing(1) was called: a
// step makes three of them, and config 2's step is 21 us.
std::atomic<int> g_tracing{-1};
inline bool tracing() {
  int t = g_tracing.load(std::memory_order_relaxed);
  if (t < 0) {
    const char* e = std::getenv("SXMC_ROCTX");
    t = (e && e[0] && e[0] != '0') ? 1 : 0;
    g_tracing.store(t, std::memory_order_relaxed);
  }
  return t > 0;
}
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

#define SX_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      return fail(SXMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (sxmc_hip.cpp:" +    \
                                    std::to_string(__LINE__) + ")");                          \
    }                                                                                       \
  } while (0)

#define SX_REQUIRE(cond, msg) \
  do {                        \
    if (!(cond)) return fail(SXMC_ERR_INVALID, msg); \
  } while (0)

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

int get_props(DeviceProps& p) {
  static thread_local DeviceProps cache;
  static thread_local int cache_dev = -1;
  int dev = 0;
  SX_HIP(hipGetDevice(&dev));
  if (!cache.valid || cache_dev != dev) {
    hipDeviceProp_t prop;
    SX_HIP(hipGetDeviceProperties(&prop, dev));
    cache.cus = prop.multiProcessorCount;
    cache.lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (cache.lds_per_cu <= 0) cache.lds_per_cu = 160 * 1024;
    cache.valid = true;
    cache_dev = dev;
  }
  p = cache;
  return SXMC_OK;
}

}  // namespace

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
  BucketSort* find_sort(unsigned mask, int ordered) const {
    for (const auto& b : sorts)
      if (b->mask == mask && b->ordered == ordered) return b.get();
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
  std::vector<HostSyst> systs;
  hipStream_t stream = nullptr;
  unsigned long long version = 1;
  sxmc_group* self = nullptr;
  int cfg_threads = 0, cfg_bpc = 0;
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
};

namespace {
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
  unsigned padded_rstride = 0;  // ... with the LDS histogram in the padded form: words between its replicas (0: not)
  unsigned plain_rstride = 0;   // ordered tables: words between the replicas of the LDS histogram in its swizzled form
};

void free_class(LaunchClass& c) {
  if (c.d_descs) (void)hipFree(c.d_descs);
  if (c.d_descs_sparse) (void)hipFree(c.d_descs_sparse);
  c.d_descs_sparse = nullptr;
  if (c.d_segs) (void)hipFree(c.d_segs);
  if (c.d_blk_off) (void)hipFree(c.d_blk_off);
  c.d_descs = nullptr;
  c.d_segs = nullptr;
  c.d_blk_off = nullptr;
}

// Who reads which units: sxmc_plan.h (apportion_workgroups, interleaved_segments, build_partition), here as thin
// adapters from descriptors to their unit counts.
using sxplan::apportion_workgroups;
std::vector<unsigned long long> unit_counts(const std::vector<SxSignalDesc>& descs) {
  std::vector<unsigned long long> nvec;
  for (const SxSignalDesc& d : descs) nvec.push_back(d.nvec);
  return nvec;
}
void interleaved_segments(const std::vector<SxSignalDesc>& descs, const std::vector<int>& K, int threads, int grid,
                          std::vector<SxSegment>& segs, std::vector<unsigned>& blk_off) {
  sxplan::interleaved_segments(unit_counts(descs), K, threads, grid, segs, blk_off);
}
void build_partition(const std::vector<SxSignalDesc>& descs, int grid, int threads, int want_mode,
                     std::vector<SxSegment>& segs, std::vector<unsigned>& blk_off, int& mode_out,
                     unsigned long long align = 1, int groups = 1) {
  sxplan::build_partition(unit_counts(descs), grid, threads, want_mode, segs, blk_off, mode_out, align, groups);
}
}  // namespace

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
  std::vector<LaunchClass> classes;
  int cfg_threads = 0, cfg_bpc = 0;
  int cfg_seen_threads = -1, cfg_seen_bpc = -1;
  int cfg_partition = 0, cfg_seen_partition = -1;  // 0 auto, 1 sliced, 2 interleaved
  int cfg_teams = 0, cfg_seen_teams = -1;          // teams per member over a bucketed table (0 = 1, the default)
  int cfg_prebin = 1, cfg_seen_prebin = -1;        // pre-bin the observables no systematic writes
  int cfg_bucket = 1, cfg_seen_bucket = -1;        // stream a bucketed copy of the table where that pays
  bool order_blocked = false;                      // a plan with ordered tables beyond LDS could not be laid out in runs
  int cfg_order = 1, cfg_seen_order = -1;          // ... with the rows of a bucket ordered by a monotonically written observable
  int cfg_rtc = 1, cfg_seen_rtc = -1;              // specialise the fill kernel at run time for programs not built in
  int cfg_codes = -1, cfg_seen_codes = -2;         // ordered tables streamed as 16-bit codes (-1: SXMC_CODES, default on)
  int cfg_queue_log = 0, cfg_seen_queue_log = -1;  // ... cap on the queues of ambiguous rows, log2(entries) (0: what fits)
  std::string rtc_note;                            // why a run-time specialisation could not be had (last failure)
  std::string plan_note;                           // why a launch of the plan took a slower general path (for launch_info)
  int cfg_tail = 1;                                // sxmc_group_step_async: the one-workgroup step end where it fits
  bool tuned = false;                              // a deferred batch's group: the launch-shape trials have run
  int trial_launches = 0;                          // fills sxmc_group_optimize has launched for this group (all calls)
  int cfg_coop = -1;                               // ... and the cooperative one-launch step end (-1: SXMC_COOP_STEP_END, default on)
  int last_step_launches = 0;                      // kernels the last sxmc_group_step_async launched
  unsigned long long plan_generation = 0;          // counts launch plans built (a multigroup re-validates on change)
  std::vector<const SampleStore::Bucketed*> member_bucket;  // per member: the copy its fill streams, or null
  int debug_mode = 0;
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

namespace {

// Slot assignment: observables first (slot k = field k), then every other field a systematic
// references, ascending.
void member_slots(const sxmc_hist* h, std::vector<int>& slot_col) {
  slot_col.clear();
  for (int k = 0; k < h->nobs; k++) slot_col.push_back(k);
  std::vector<int> extra;
  for (const HostSyst& s : h->systs) {
    if (s.obs >= h->nobs) extra.push_back(s.obs);
    if (s.type == SXMC_SYST_RESOLUTION_SCALE && s.extra_field >= h->nobs) extra.push_back(s.extra_field);
  }
  std::sort(extra.begin(), extra.end());
  extra.erase(std::unique(extra.begin(), extra.end()), extra.end());
  for (int c : extra) slot_col.push_back(c);
}

int slot_of(const std::vector<int>& slot_col, int field) {
  for (size_t k = 0; k < slot_col.size(); k++)
    if (slot_col[k] == field) return (int)k;
  return 0;
}

void free_bucket_tables(sxmc_hist* h) {
  if (h->d_bdir) (void)hipFree(h->d_bdir);
  if (h->d_btkeys) (void)hipFree(h->d_btkeys);
  if (h->d_btslot) (void)hipFree(h->d_btslot);
  h->d_bdir = h->d_btkeys = h->d_btslot = nullptr;
  h->btab_valid = false;
}

void free_sparse(sxmc_hist* h) {
  free_bucket_tables(h);
  if (h->d_cnt) (void)hipFree(h->d_cnt);
  if (h->d_read_slot) (void)hipFree(h->d_read_slot);
  if (h->d_filter) (void)hipFree(h->d_filter);
  if (h->d_table) (void)hipFree(h->d_table);
  if (h->d_coarse) (void)hipFree(h->d_coarse);
  h->d_coarse = nullptr;
  h->d_cnt = nullptr;
  h->d_read_slot = nullptr;
  h->d_filter = nullptr;
  h->d_table = nullptr;
  h->ntargets = 0;
  h->targets.clear();
  h->h_read_slot.clear();
}

using sxplan::ceil_log2;

// Sparse-counting structures of one evaluator from its event bins (host side, once per SetEvalPoints).
int build_sparse(sxmc_hist* h, const std::vector<int>& rb) {
  free_sparse(h);
  sxplan::SparseTables st;
  sxplan::build_sparse_tables(rb, st);   // (sxmc_plan.h: targets, slots, filters, table)
  const std::vector<unsigned>&targets = st.targets, &filter = st.filter, &table = st.table, &coarse = st.coarse;
  const std::vector<int>& slot = st.slot;
  const size_t T = targets.size();
  const int cbits = st.cbits, fbits = st.fbits, tbits = st.tbits;
  SX_HIP(hipMalloc((void**)&h->d_cnt, sizeof(unsigned) * std::max<size_t>(T, 4)));
  SX_HIP(hipMemset(h->d_cnt, 0, sizeof(unsigned) * std::max<size_t>(T, 4)));
  SX_HIP(hipMalloc((void**)&h->d_read_slot, sizeof(int) * std::max<size_t>(slot.size(), 1)));
  if (!slot.empty()) SX_HIP(hipMemcpy(h->d_read_slot, slot.data(), sizeof(int) * slot.size(), hipMemcpyHostToDevice));
  SX_HIP(hipMalloc((void**)&h->d_filter, sizeof(unsigned) * filter.size()));
  SX_HIP(hipMemcpy(h->d_filter, filter.data(), sizeof(unsigned) * filter.size(), hipMemcpyHostToDevice));
  SX_HIP(hipMalloc((void**)&h->d_table, sizeof(unsigned) * table.size()));
  SX_HIP(hipMemcpy(h->d_table, table.data(), sizeof(unsigned) * table.size(), hipMemcpyHostToDevice));
  SX_HIP(hipMalloc((void**)&h->d_coarse, sizeof(unsigned) * coarse.size()));
  SX_HIP(hipMemcpy(h->d_coarse, coarse.data(), sizeof(unsigned) * coarse.size(), hipMemcpyHostToDevice));
  h->h_read_slot = slot;
  h->coarse_shift = 32 - cbits;
  h->ntargets = (int)T;
  h->targets = targets;
  h->filter_shift = 32 - fbits;
  h->table_shift = 32 - tbits;
  return SXMC_OK;
}

// The sparse flavour of a member's descriptor: counters instead of the histogram, slots instead of bins.
void make_sparse_desc(const sxmc_hist* h, SxSignalDesc& d) {
  d.sparse_real_nbins = h->total_nbins;
  d.bins = h->d_cnt;
  d.total_nbins = std::max(h->ntargets, 1);
  d.read_bins = h->d_read_slot;
  d.sparse_filter = h->d_filter;
  d.sparse_table = h->d_table;
  d.sparse_filter_shift = h->filter_shift;
  d.sparse_table_shift = h->table_shift;
  d.sparse_coarse = h->d_coarse;
  d.sparse_coarse_shift = h->coarse_shift;
}

int fill_desc(const sxmc_hist* h, SxSignalDesc& d) {
  std::memset(&d, 0, sizeof(d));
  std::vector<int> slot_col;
  member_slots(h, slot_col);
  d.cols = h->store->d_cols;
  d.col_pitch = h->pitch;
  d.nsamples = h->nsamples;
  d.nvec = h->nvec;
  d.bins = h->d_bins;
  d.norm = h->norm ? h->norm + h->norm_off : nullptr;
  d.total_nbins = h->total_nbins;
  d.nobs = h->nobs;
  d.nslot = (int)slot_col.size();
  d.nsyst = (int)h->systs.size();
  d.param_stride = h->par_stride;
  d.params = h->params ? h->params + h->par_off : nullptr;
  for (int k = 0; k < d.nslot; k++) d.slot_col[k] = slot_col[k];
  for (int k = 0; k < h->nobs; k++) {
    d.nbins[k] = h->nbins[(size_t)k];
    d.bin_stride[k] = h->stride[k];
    d.lower[k] = h->lower[k];
    d.upper[k] = h->upper[k];
    d.scale[k] = h->scale[k];
  }
  int ncoef = 0;
  for (int s = 0; s < d.nsyst; s++) {
    const HostSyst& hs = h->systs[s];
    SxSystOp& op = d.syst[s];
    op.type = (short)hs.type;
    op.obs_slot = (short)slot_of(slot_col, hs.obs);
    op.extra_slot = (short)(hs.type == SXMC_SYST_RESOLUTION_SCALE ? slot_of(slot_col, hs.extra_field) : 0);
    op.npars = (short)hs.pars.size();
    op.coef_start = (short)ncoef;
    for (size_t i = 0; i < hs.pars.size(); i++) {
      op.pars[i] = hs.pars[i];
      if (ncoef < 64) d.coef_par[ncoef] = hs.pars[i];
      ncoef++;
    }
  }
  d.ncoef = ncoef;
  d.read_bins = h->has_points ? h->d_read_bins : nullptr;
  d.npoints = h->has_points ? h->npoints : 0;
  d.pdf_out = h->pdf ? h->pdf + h->pdf_off : nullptr;
  d.pdf_stride = h->pdf_stride;
  d.bin_volume = h->bin_volume;
  return SXMC_OK;
}

struct DevBuf {  // device temporary, freed on scope exit
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

// The rows of member `h`'s table sorted by their bin indices in the observables of `mask` (those no systematic
// writes) and cut into 256-row granules, bucket by bucket.  Fetched from the table's cache or built; *out =
// nullptr when bucketing does not pay for this table.  d_full_desc: the member's descriptor on the device.
// `ordered` >= 0: inside every bucket the rows are in ascending order of that observable's raw value (NaN last).
int get_bucket_sort(sxmc_hist* h, const SxSignalDesc* d_full_desc, unsigned mask, int ordered,
                    const SampleStore::BucketSort** out) {
  *out = nullptr;
  SampleStore& st = *h->store;
  std::lock_guard<std::mutex> lock(st.pre_mutex);
  if (SampleStore::BucketSort* have = st.find_sort(mask, ordered)) {
    *out = have->rejected ? nullptr : have;
    return SXMC_OK;
  }
  st.sorts.push_back(std::make_unique<SampleStore::BucketSort>());
  SampleStore::BucketSort* b = st.sorts.back().get();
  b->mask = mask;
  b->ordered = ordered;
  const size_t n = h->nsamples;
  if (n == 0 || n > 0x7FFFFF00ull) return SXMC_OK;

  // key space: mixed radix over the untouched observables, last one fastest, bases nbins + 1 (an index can
  // come out as nbins one ulp below the upper edge; such samples form buckets of their own)
  unsigned long long nkeys = 1;
  for (int k = h->nobs - 1; k >= 0; k--) {
    if (!((mask >> k) & 1u)) continue;
    b->radix[k] = (unsigned)nkeys;
    nkeys *= (unsigned long long)h->nbins[(size_t)k] + 1ull;
    if (nkeys > (1ull << 20)) return SXMC_OK;
  }
  const unsigned outside = (unsigned)nkeys;
  const int bits = ceil_log2((size_t)nkeys + 1);

  DevBuf keys0, keys1, rows0, dfirst;
  SX_HIP(keys0.alloc(n * 4));
  SX_HIP(keys1.alloc(n * 4));
  SX_HIP(rows0.alloc(n * 4));
  SX_HIP(hipMalloc((void**)&b->d_rows, n * 4));
  SX_HIP(dfirst.alloc((nkeys + 1) * 4));
  SX_HIP(hipMemset(dfirst.p, 0xFF, (nkeys + 1) * 4));
  const unsigned* order_rows = nullptr;
  DevBuf rows1;
  if (ordered >= 0) {
    // rows by the ordered observable's value first; the stable sort by bucket below keeps that order inside a bucket
    SX_HIP(rows1.alloc(n * 4));
    SX_HIP(sx_order_keys(st.d_cols + (size_t)ordered * h->pitch, n, keys0.as<unsigned>(),
                         rows0.as<unsigned>(), nullptr));
    SX_HIP(sx_bucket_sort(keys0.as<unsigned>(), keys1.as<unsigned>(), rows0.as<unsigned>(), rows1.as<unsigned>(), n, 32,
                          nullptr));
    order_rows = rows1.as<unsigned>();
  }
  SX_HIP(sx_bucket_keys(d_full_desc, n, mask, b->radix, outside, order_rows, keys0.as<unsigned>(), rows0.as<unsigned>(),
                        nullptr));
  SX_HIP(sx_bucket_sort(keys0.as<unsigned>(), keys1.as<unsigned>(), rows0.as<unsigned>(), b->d_rows, n, bits, nullptr));
  SX_HIP(sx_bucket_first(keys1.as<unsigned>(), n, dfirst.as<unsigned>(), nullptr));
  std::vector<unsigned> first((size_t)nkeys + 1);
  SX_HIP(hipMemcpy(first.data(), dfirst.p, first.size() * 4, hipMemcpyDeviceToHost));

  // logical granules: bucket by bucket, each bucket padded to whole granules (sxmc_plan.h)
  sxplan::GranulePlan gp;
  sxplan::bucket_granules(first, outside, n, gp);
  if (!gp.worth_it) {  // mostly padding: not worth it
    (void)hipFree(b->d_rows);
    b->d_rows = nullptr;
    return SXMC_OK;
  }
  b->lsrc = gp.lsrc;
  b->lvalid = gp.lvalid;
  b->lwhich = gp.lwhich;
  b->keys = gp.present;
  b->nkeys_total = outside;
  b->nkept = gp.kept;
  sxplan::bucket_key_offsets(gp.present, mask, b->radix, h->nbins.data(), h->stride.data(), h->nobs, b->key_pre);
  b->rejected = false;
  *out = b;
  return SXMC_OK;
}

// The bucketed COPY of the table for a sort: the columns `fields`, granule order transposed for `runs` runs --
// run r holds logical granules [r * T, (r + 1) * T), and the runs are interleaved granule by granule (physical
// p = t * runs + r), so that `runs` consumers that each walk one run read neighbouring addresses at the same
// time.  runs = 1: the sorted order itself.
int get_bucketed(sxmc_hist* h, const SampleStore::BucketSort* bs, const std::vector<int>& fields, int runs,
                 const SampleStore::Bucketed** out) {
  *out = nullptr;
  SampleStore& st = *h->store;
  std::lock_guard<std::mutex> lock(st.pre_mutex);
  if (SampleStore::Bucketed* have = st.find_bucketed(bs, fields, runs)) {
    if (!have->complete) return fail(SXMC_ERR_HIP, "an earlier attempt to lay this table out failed");
    *out = have;
    return SXMC_OK;
  }
  SX_REQUIRE(!fields.empty() && runs >= 1, "bad bucketed layout request");
  const bool pack_rows = bs->ordered >= 0 && h->total_nbins <= kLdsMaxBins;   // (fill_ordered_body, histogram in LDS)
  st.bucketed.push_back(std::make_unique<SampleStore::Bucketed>());
  SampleStore::Bucketed* b = st.bucketed.back().get();
  b->mask = bs->mask;
  b->fields = fields;
  b->runs = runs;
  b->sort = bs;
  sxplan::BucketedLayout lay;   // physical granule order for `runs` runs (sxmc_plan.h)
  sxplan::bucketed_layout(bs->lsrc, bs->lvalid, bs->lwhich, bs->keys, bs->key_pre, bs->nkeys_total, runs, pack_rows, lay);
  const size_t P = lay.P, A = lay.A;
  const std::vector<unsigned>&psrc = lay.psrc, &pvalid = lay.pvalid, &ppre = lay.ppre, &pkp = lay.pkp;
  b->ngranules = P;
  b->nkept = bs->nkept;
  b->pitch = std::max<size_t>(64, P * 256);
  DevBuf dsrc, dvalid;
  SX_HIP(hipMalloc((void**)&b->d_cols, sizeof(float) * b->pitch * fields.size()));
  SX_HIP(hipMalloc((void**)&b->d_gpre, sizeof(unsigned) * A));
  SX_HIP(hipMalloc((void**)&b->d_gkp, sizeof(unsigned) * 2 * A));
  SX_HIP(hipMemcpy(b->d_gpre, ppre.data(), sizeof(unsigned) * A, hipMemcpyHostToDevice));
  SX_HIP(hipMemcpy(b->d_gkp, pkp.data(), sizeof(unsigned) * 2 * A, hipMemcpyHostToDevice));
  SX_HIP(dsrc.alloc(A * 4));
  SX_HIP(dvalid.alloc(A * 4));
  SX_HIP(hipMemcpy(dsrc.p, psrc.data(), A * 4, hipMemcpyHostToDevice));
  SX_HIP(hipMemcpy(dvalid.p, pvalid.data(), A * 4, hipMemcpyHostToDevice));
  SX_HIP(sx_bucket_gather(st.d_cols, h->pitch, (int)fields.size(), fields.data(), bs->d_rows, dsrc.as<unsigned>(),
                          dvalid.as<unsigned>(), P, b->d_cols, b->pitch, nullptr));
  if (bs->ordered >= 0) {
    // the ordered observable's column is the last of `fields` (group_rebuild)
    SX_HIP(hipMalloc((void**)&b->d_gedge, sizeof(float) * 2 * A));
    SX_HIP(hipMemset(b->d_gedge, 0, sizeof(float) * 2 * A));
    SX_HIP(sx_bucket_edges(b->d_cols + (fields.size() - 1) * b->pitch, dvalid.as<unsigned>(), P, b->d_gedge, nullptr));
  }
  SX_HIP(hipDeviceSynchronize());
  b->complete = true;
  *out = b;
  return SXMC_OK;
}

// CODES (fill_ordered_body): the streamed fields of a bucketed copy with an ordered observable once more, as 16-bit
// codes inside a window per field, two fields to a word.  `cd`: the member as its fill sees it -- slots 0 .. nobs-1 are
// observables (their domains centre the windows), the others fields only read.  The window of an observable is its
// finite range in the table cut to the domain widened by its own width on either side (a value further out needs a
// scale or shift of the order of the whole domain to come back in: its row is marked "ask the exact columns"
// instead); the window of a field that is only read is its finite range, cut to three such widths around the
// observables' windows when they overlap at all.  Built once per copy; left out (d_qcol stays null) when more than
// 2 % of the rows would ask the exact columns: such a table gains nothing.
bool codes_enabled(const sxmc_group* g) {
  if (g->cfg_codes >= 0) return g->cfg_codes != 0;
  static const bool on = [] {
    const char* e = std::getenv("SXMC_CODES");
    return !e || std::atoi(e) != 0;
  }();
  return on;
}

int get_bucket_codes(sxmc_hist* h, const SampleStore::Bucketed* bkc, const SxSignalDesc& cd) {
  SampleStore& st = *h->store;
  std::lock_guard<std::mutex> lock(st.pre_mutex);
  SampleStore::Bucketed* b = const_cast<SampleStore::Bucketed*>(bkc);
  if (b->codes_tried) return SXMC_OK;
  b->codes_tried = true;
  // (an ordered copy: every field but the ordered observable's, the last; an unordered one -- the sparse counting over
  // runs -- : every field)
  const int nq = (int)b->fields.size() - ((b->sort && b->sort->ordered >= 0) ? 1 : 0);
  // (below 2^22 granules a unit's byte offset into a column of codes fits 32 bits, and a unit number 28: what the
  // ordered kernel's addressing and its queue entries assume)
  if (nq < 2 || nq > SXMC_MAX_QSLOTS || b->ngranules == 0 || b->ngranules >= ((size_t)1 << 22) || !b->sort) {
    return SXMC_OK;
  }
  const unsigned long long n = (unsigned long long)b->ngranules * 256ull;
  float mm[2 * SXMC_MAX_NFIELDS];
  SX_HIP(sx_column_minmax(b->d_cols, b->pitch, nq, n, mm, nullptr));
  sxplan::CodeWindows cw;   // (sxmc_plan.h: pure, tested without a device)
  sxplan::code_windows(mm, nq, cd.nobs, cd.lower, cd.upper, cw);
  for (int m = 0; m < nq; m++) {
    b->qbase[m] = cw.base[(size_t)m];
    b->qstep[m] = cw.step[(size_t)m];
  }
  // Do they pay?  A sample is ambiguous when a bin coordinate lies within ~half a code step (in bins) of an integer:
  // about sum_k nbins_k * step_k / (upper_k - lower_k) of the samples for systematics near their means.  Beyond 2 in
  // 10^3, four 256-sample units in ten hold an ambiguous sample and the queues' traffic eats what the codes save
  // (measured at 200 bins per observable: slower than the float stream): such tables keep their float stream.
  double ambiguous = 0;
  for (int m = 0; m < nq && m < cd.nobs; m++) ambiguous += (double)cd.nbins[m] * cw.step[(size_t)m] / (cd.upper[m] - cd.lower[m]);
  static const bool gate_lifted = [] {   // (SXMC_CODES_GATE=1, measurement: tools/codes_gate_probe.py)
    const char* e = measure_env("SXMC_CODES_GATE");
    return e && e[0] == '1';
  }();
  if (!(ambiguous <= 2e-3) && !gate_lifted) return SXMC_OK;
  // (the codes are an extra: a table they do not fit beside -- +4 bytes per row and pair of fields -- keeps its float stream)
  if (hipMalloc((void**)&b->d_qcol, sizeof(unsigned) * b->pitch * (size_t)((nq + 1) / 2)) != hipSuccess) {
    (void)hipGetLastError();
    b->d_qcol = nullptr;
    return SXMC_OK;
  }
  unsigned long long tally[2] = {0, 0};
  hipError_t e = sx_column_codes(b->d_cols, b->pitch, nq, b->qbase, b->qstep, n, b->d_qcol, tally, nullptr);
  if (e != hipSuccess || (double)tally[0] > 0.02 * (double)std::max<size_t>(b->nkept, 1)) {
    (void)hipFree(b->d_qcol);
    b->d_qcol = nullptr;
    if (e != hipSuccess) return fail(SXMC_ERR_HIP, std::string("codes of a bucketed table: ") + hipGetErrorString(e));
    return SXMC_OK;
  }
  b->nq = nq;
  b->q_exact_rows = tally[0];
  b->q_never_rows = tally[1];
  return SXMC_OK;
}

using sxplan::ordered_rstride_padded;
using sxplan::ordered_queue_bytes;
// the largest set of queues (512 .. 2048 entries: every wave of the workgroup owns an equal slice) that fits `room`
// bytes, as log2(entries); 0: none, the launch then streams the float columns
constexpr unsigned kMinQueueLog = 9;
unsigned ordered_queue_log(size_t room, int cap = 0) {
  unsigned qlog = 11;
  // (sxmc_group_set_codes_queue_log caps the queues at 2^9 .. 2^11 entries -- smaller queues fill up and are emptied in
  // the middle of the stream, and whole granules are handed to the float columns; the results do not depend on it)
  if (cap > 0) qlog = (unsigned)std::min(std::max(cap, (int)kMinQueueLog), 11);
  while (qlog >= kMinQueueLog && ordered_queue_bytes(qlog) > room) qlog--;
  return qlog >= kMinQueueLog ? qlog : 0;
}

// The evaluator's event bins grouped by the buckets of a sort (fill_sparse_kernel): per bucket key an
// open-addressing table keyed by the event bin's index contribution of the WRITTEN observables (flat index minus
// the bucket's offset, canonical decomposition), value = the event bin's counter slot (its rank among the sorted
// distinct event bins, as in build_sparse).  Rebuilt when the evaluation points or the untouched set change.
int build_bucket_tables(sxmc_hist* h, const SampleStore::BucketSort* bs) {
  if (h->btab_valid && h->btab_mask == bs->mask && h->btab_points_version == h->points_version) return SXMC_OK;
  free_bucket_tables(h);
  sxplan::BucketTables bt;   // directory + per-bucket tables of the event bins (sxmc_plan.h)
  sxplan::bucket_tables(bs->nkeys_total, bs->mask, bs->radix, h->nbins.data(), h->stride.data(), h->nobs, h->targets, bt);
  const std::vector<unsigned>&dir = bt.dir, &tkeys = bt.tkeys, &tslot = bt.tslot;
  SX_HIP(hipMalloc((void**)&h->d_bdir, sizeof(unsigned) * dir.size()));
  SX_HIP(hipMemcpy(h->d_bdir, dir.data(), sizeof(unsigned) * dir.size(), hipMemcpyHostToDevice));
  SX_HIP(hipMalloc((void**)&h->d_btkeys, sizeof(unsigned) * std::max<size_t>(tkeys.size(), 4)));
  SX_HIP(hipMalloc((void**)&h->d_btslot, sizeof(unsigned) * std::max<size_t>(tkeys.size(), 4)));
  if (!tkeys.empty()) {
    SX_HIP(hipMemcpy(h->d_btkeys, tkeys.data(), sizeof(unsigned) * tkeys.size(), hipMemcpyHostToDevice));
    SX_HIP(hipMemcpy(h->d_btslot, tslot.data(), sizeof(unsigned) * tslot.size(), hipMemcpyHostToDevice));
  }
  h->btab_mask = bs->mask;
  h->btab_points_version = h->points_version;
  h->btab_valid = true;
  return SXMC_OK;
}

// LDS of fill_ordered_body: per chain 2^rlog replicas of the histogram, each padded to whole 64-word blocks (the
// swizzle permutes inside a block) + 16 words (replicas of a bin in different banks), + header and trash words.
unsigned ordered_rstride(int max_bins) { return (((unsigned)max_bins + 63u) & ~63u) + 16u; }
size_t ordered_lds_bytes(int max_bins, int nchain, unsigned rlog) {
  return (4 + ((size_t)nchain * ordered_rstride(max_bins) << rlog) + 64) * 4;
}

// The member's problem as the fill sees it once its table is bucketed: only the observables some systematic
// writes (+ the extra fields), slots renumbered, columns = the bucketed copy.  `keep`: full slot -> new slot or -1.
// `ordered` >= 0: that observable rides in the last slot and its geometry at index nobs2 (fill_ordered_kernel).
void compact_desc(const SxSignalDesc& full, const std::vector<int>& keep, int nobs2, SxSignalDesc& cd, int ordered = -1) {
  cd = full;
  if (ordered >= 0) {
    cd.bin_stride[nobs2] = full.bin_stride[ordered];
    cd.nbins[nobs2] = full.nbins[ordered];
    cd.lower[nobs2] = full.lower[ordered];
    cd.upper[nobs2] = full.upper[ordered];
    cd.scale[nobs2] = full.scale[ordered];
  }
  int nslot = 0;
  for (int k = 0; k < full.nslot; k++) {
    if (keep[(size_t)k] < 0) continue;
    const int q = keep[(size_t)k];
    cd.slot_col[q] = q;  // the copy holds exactly the streamed fields, in slot order
    if (q < nobs2) {     // an observable the fill still bins (the others it keeps are read-only inputs)
      cd.bin_stride[q] = full.bin_stride[k];
      cd.nbins[q] = full.nbins[k];
      cd.lower[q] = full.lower[k];
      cd.upper[q] = full.upper[k];
      cd.scale[q] = full.scale[k];
    }
    nslot++;
  }
  cd.nobs = nobs2;
  cd.nslot = nslot;
  for (int q = 0; q < full.nsyst; q++) {
    cd.syst[q].obs_slot = (short)keep[(size_t)full.syst[q].obs_slot];
    cd.syst[q].extra_slot =
        (short)(full.syst[q].type == SXMC_SYST_RESOLUTION_SCALE ? keep[(size_t)full.syst[q].extra_slot] : 0);
  }
}

// The whole step in one launch (fill_step_kernel), when asked for: its role workgroups carry the fill's LDS allotment
// and need 16 KB of their own beside it, which the plan of an ordered fill then leaves free.
bool fused_step_requested(const sxmc_group* g) {
  static const int env_default = [] {
    const char* e = std::getenv("SXMC_FUSED_STEP");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return (g->cfg_fused < 0 ? env_default : g->cfg_fused) != 0;
}

int group_rebuild(sxmc_group* g) {
  TraceRange trace("sxmc: launch plan (tables, partitions, kernels)");
  // Descriptors may still be read by kernels in flight on another stream: rebuilds are rare
  // (bindings change only during setup), so a device-wide sync is the simple safe choice.
  SX_HIP(hipDeviceSynchronize());
  DeviceProps props;
  int rc = get_props(props);
  if (rc) return rc;

  const int n = (int)g->members.size();
  g->h_descs.assign((size_t)n, SxSignalDesc{});
  for (LaunchClass& c : g->classes) free_class(c);
  g->classes.clear();
  g->max_bins = 0;
  g->max_points = 0;
  g->same_points = n > 0;
  g->plan_note.clear();

  int threads = g->cfg_threads > 0 ? g->cfg_threads : 512;
  if (threads < 64 || threads > 1024 || threads % 64) threads = 512;

  for (int i = 0; i < n; i++) fill_desc(g->members[i], g->h_descs[i]);
  if (!g->d_descs) SX_HIP(hipMalloc((void**)&g->d_descs, sizeof(SxSignalDesc) * std::max(n, 1)));
  if (n) SX_HIP(hipMemcpy(g->d_descs, g->h_descs.data(), sizeof(SxSignalDesc) * n, hipMemcpyHostToDevice));

  std::vector<SxSignalDesc> fill_descs((size_t)n);  // each member as its fill launch sees it
  struct BucketPlan {                               // bucketed members: what to lay out once the class's shape is known
    const SampleStore::BucketSort* sort = nullptr;
    std::vector<int> fields;
  };
  std::vector<BucketPlan> plans((size_t)n);
  g->member_bucket.assign((size_t)n, nullptr);
  for (int i = 0; i < n; i++) {
    sxmc_hist* h = g->members[i];
    const SxSignalDesc& d = g->h_descs[i];
    fill_descs[(size_t)i] = d;
    g->max_bins = std::max(g->max_bins, h->total_nbins);
    g->max_points = std::max<unsigned long long>(g->max_points, d.npoints);
    if (!h->has_points || h->npoints != g->members[0]->npoints) g->same_points = false;

    const int lds_hist = h->total_nbins <= kLdsMaxBins ? 1 : 0;
    int key_nobs = 0, key_nslot = 0, static_prog = -1, pre_width = 0;
    unsigned pre_mask = 0;
    bool runs_mode = false;
    std::vector<unsigned> prog;
    bool prog_simple = false;

    // A program every systematic of which is a short polynomial can run as straight-line code: from the table of
    // kernels built into the library, or specialised now through hiprtc (sxmc_rtc.cpp).
    int prog_ncoef = 0;
    bool specialisable = d.nsyst <= 8;
    for (int q = 0; q < d.nsyst; q++) {
      prog_ncoef += d.syst[q].npars;
      specialisable = specialisable && d.syst[q].npars >= 1 && d.syst[q].npars <= SXMC_MAX_SYST_PARS;
    }
    specialisable = specialisable && prog_ncoef <= 16;
    auto prog_words = [](const SxSignalDesc& x) {
      std::vector<unsigned> w;
      for (int q = 0; q < x.nsyst; q++) {
        w.push_back((unsigned)x.syst[q].type | ((unsigned)x.syst[q].obs_slot << 4) |
                    ((unsigned)x.syst[q].extra_slot << 8) | (x.syst[q].npars > 1 ? (unsigned)x.syst[q].npars << 12 : 0u));
      }
      return w;
    };
    // is there a kernel for this specialisation?  *fn: the run-time one, or null for a built-in one
    auto have_kernel = [&](int nobs_, int nslot_, int prew, int runs, const std::vector<unsigned>& words, int sp,
                           void** fn) {
      *fn = nullptr;
      if (prew == 5) {   // (sp: index into the ordered programs built in: histograms in LDS, no runs)
        if (sp >= 0 && lds_hist && !runs) return true;
      } else if (runs ? sx_fill_static_supports_sparse_runs(sp) : sx_fill_static_supports(sp, lds_hist, prew)) {
        return true;
      }
      if (!g->cfg_rtc) return false;
      SxRtcSpec k{};
      k.nobs = nobs_;
      k.nslot = nslot_;
      k.lds_hist = lds_hist;
      k.pre_width = prew;
      k.sparse_runs = runs;
      k.nops = (int)words.size();
      for (size_t q = 0; q < words.size(); q++) k.ops[q] = words[q];
      std::string err;
      *fn = sx_rtc_get(k, &err);
      if (!*fn) g->rtc_note = err;
      return *fn != nullptr;
    };
    void *rtc_fill = nullptr, *rtc_sparse = nullptr;

    // ---- bucketed table: the observables no systematic writes become a per-granule bin offset and the
    // fill sees the lower-dimensional problem of the ones that are written.  With an ORDERED observable (written
    // only by monotone one-coefficient systematics, read by nothing): that one too is a per-granule constant,
    // worked out per evaluation from the granule's end values, except in the granules that straddle a bin edge.
    bool bucketed = false;
    auto try_bucket = [&](int ordered) -> int {
      unsigned touched = 0, read = 0;
      for (int q = 0; q < d.nsyst; q++) {
        touched |= 1u << d.syst[q].obs_slot;
        if (d.syst[q].type == SXMC_SYST_RESOLUTION_SCALE) read |= 1u << d.syst[q].extra_slot;
      }
      // compacted slots: the observables that are written (still binned by the fill), then everything that is only
      // read -- the extra fields, and an untouched observable that serves as some systematic's truth field (its own
      // bin index is the bucket's; its VALUE is still an input) --, then the ordered observable
      unsigned mask = 0;
      std::vector<int> keep((size_t)d.nslot, -1), fields;
      int nobs2 = 0;
      for (int k = 0; k < d.nobs; k++) {
        if (k == ordered) continue;
        if ((touched >> k) & 1u) {
          keep[(size_t)k] = (int)fields.size();
          fields.push_back(d.slot_col[k]);
          nobs2++;
        } else {
          mask |= 1u << k;
        }
      }
      for (int k = 0; k < d.nslot; k++) {
        if (keep[(size_t)k] >= 0 || k == ordered) continue;
        if (k >= d.nobs || ((read >> k) & 1u)) {
          keep[(size_t)k] = (int)fields.size();
          fields.push_back(d.slot_col[k]);
        }
      }
      if (ordered >= 0) {
        keep[(size_t)ordered] = (int)fields.size();
        fields.push_back(d.slot_col[ordered]);
      }
      // (beyond LDS the granule word has no room for the row count: something must be binned per sample)
      const bool shape_ok = ordered >= 0 ? (nobs2 <= 5 && fields.size() <= 7 && (lds_hist || nobs2 >= 1))
                                         : (mask && nobs2 >= 1 && sx_fill_has_specialization(nobs2, (int)fields.size()));
      if (!shape_ok) return SXMC_OK;
      SxSignalDesc cd;
      compact_desc(d, keep, nobs2, cd, ordered);
      const std::vector<unsigned> prog2 = prog_words(cd);
      const int prew = ordered >= 0 ? 5 : 3;
      const int sp = ordered >= 0 ? sx_fill_find_ordered_program(cd.nobs, cd.nslot, (int)prog2.size(), prog2.data())
                                  : sx_fill_find_static_program(cd.nobs, cd.nslot, (int)prog2.size(), prog2.data());
      if (!have_kernel(cd.nobs, cd.nslot, prew, 0, prog2, sp, &rtc_fill)) return SXMC_OK;
      const SampleStore::BucketSort* bs = nullptr;
      int rc2 = get_bucket_sort(h, g->d_descs + i, mask, ordered, &bs);
      if (rc2) return rc2;
      if (!bs) {
        rtc_fill = nullptr;
        return SXMC_OK;
      }
      fill_descs[(size_t)i] = cd;     // (columns, unit count and granule table: once the layout is chosen)
      plans[(size_t)i].sort = bs;
      plans[(size_t)i].fields = fields;
      bucketed = true;
      // histograms beyond LDS, evaluated at data events: per-wave runs + event bins grouped by bucket
      bool narrow = true;   // (the runs kernel forms idx * stride + bin with ONE signed 24-bit multiply-add)
      for (int k = 0; k < h->nobs; k++) {
        narrow = narrow && h->nbins[(size_t)k] < (1 << 23) && h->stride[(size_t)k] < (1 << 23);
      }
      runs_mode = !lds_hist && narrow && h->has_points && h->d_table &&
                  have_kernel(cd.nobs, cd.nslot, prew, 1, prog2, sp, &rtc_sparse);
      if (!lds_hist && !narrow && h->has_points && h->d_table) {
        // (a regression on very large histograms must be visible: sxmc_group_launch_info prints this)
        g->plan_note = "histogram with a bin count or stride of 2^23 or more: the sparse counting over runs (one signed "
                       "24-bit multiply-add per index) does not apply, the general sparse path runs instead";
      }
      if (runs_mode) {
        rc2 = build_bucket_tables(h, bs);
        if (rc2) return rc2;
      }
      key_nobs = cd.nobs;
      key_nslot = cd.nslot;
      prog = prog2;
      prog_simple = true;
      static_prog = rtc_fill ? -1 : sp;
      pre_mask = mask | (ordered >= 0 ? 1u << (16 + ordered) : 0u);
      pre_width = prew;
      return SXMC_OK;
    };
    if (g->cfg_bucket && d.nsyst > 0 && specialisable) {
      // the ordered observable: written by one-coefficient shift / scale / cos-theta scale only and read by
      // nothing; of several, the one with the fewest bins (fewest granules that straddle an edge)
      int ordered = -1;
      // (a histogram beyond LDS: only with the event-bin counters over runs; the round-1 filter path of a table
      // left in sorted order has no ordered form)
      bool narrow_o = true;
      for (int k = 0; k < h->nobs; k++) {
        narrow_o = narrow_o && h->nbins[(size_t)k] < (1 << 23) && h->stride[(size_t)k] < (1 << 23);
      }
      const bool beyond_ok = !lds_hist && !g->order_blocked && narrow_o && h->has_points && h->d_table && n <= props.cus;
      if (g->cfg_order && ((lds_hist && h->total_nbins < (1 << 24)) || beyond_ok)) {
        for (int k = 0; k < d.nobs; k++) {
          bool written = false, ok = true;
          for (int q = 0; q < d.nsyst; q++) {
            const SxSystOp& op = d.syst[q];
            if (op.obs_slot == k) {
              written = true;
              ok = ok && op.npars == 1 &&
                   (op.type == SXMC_SYST_SHIFT || op.type == SXMC_SYST_SCALE || op.type == SXMC_SYST_CTSCALE);
            }
            if (op.type == SXMC_SYST_RESOLUTION_SCALE && op.extra_slot == k) ok = false;
          }
          if (written && ok && (ordered < 0 || h->nbins[(size_t)k] < h->nbins[(size_t)ordered])) ordered = k;
        }
        // Does it pay?  Up to nbins + 1 granules per bucket straddle an edge and stream everything; with fewer
        // than twice that many granules in all, most do (BASELINE config 5: 61 granules per bucket against 200
        // bins of r) and the ordered form only adds work.  cfg_order == 2 (tests): wherever it applies.
        if (ordered >= 0 && g->cfg_order == 1) {
          double buckets = 1.0;
          for (int k = 0; k < d.nobs; k++) {
            bool written = false;
            for (int q = 0; q < d.nsyst; q++) written = written || d.syst[q].obs_slot == k;
            if (!written) buckets *= (double)h->nbins[(size_t)k];
          }
          const double straddling = buckets * ((double)h->nbins[(size_t)ordered] + 1.0);
          if ((double)h->nsamples / 256.0 < 2.0 * straddling) ordered = -1;
        }
      }
      if (ordered >= 0) {
        rc = try_bucket(ordered);
        if (rc) return rc;
        if (bucketed && !lds_hist && !runs_mode) {   // (no kernel for the runs: the unordered layout has the filter path)
          bucketed = false;
          rtc_fill = rtc_sparse = nullptr;
          fill_descs[(size_t)i] = d;
          plans[(size_t)i] = BucketPlan{};
        }
      }
      if (!bucketed) {
        rc = try_bucket(-1);
        if (rc) return rc;
      }
    }
    if (!bucketed) {
      const bool spec = sx_fill_has_specialization(d.nobs, d.nslot) && d.ncoef <= 64;
      key_nobs = spec ? d.nobs : 0;
      key_nslot = spec ? d.nslot : 0;
      prog = prog_words(d);
      const int sp = (spec && specialisable)
                         ? sx_fill_find_static_program(key_nobs, key_nslot, (int)prog.size(), prog.data())
                         : -1;
      prog_simple = spec && specialisable && have_kernel(key_nobs, key_nslot, 0, 0, prog, sp, &rtc_fill);
      static_prog = (prog_simple && !rtc_fill) ? sp : -1;
      // pre-binning: observables that no systematic writes (built-in programs only; bucketing covers the rest)
      if (g->cfg_prebin && sx_fill_static_supports(static_prog, lds_hist, 1)) {
        unsigned touched = 0;
        for (int q = 0; q < d.nsyst; q++) touched |= 1u << d.syst[q].obs_slot;
        long long bound = 0;  // largest value the partial index can take (index == nbins included)
        for (int k = 0; k < d.nobs; k++) {
          if (!((touched >> k) & 1u)) {
            pre_mask |= 1u << k;
            bound += (long long)h->nbins[(size_t)k] * h->stride[(size_t)k];
          }
        }
        pre_width = bound < 0xFF ? 1 : 2;
        if (!pre_mask || bound >= 0xFFFF) {
          pre_mask = 0;
          pre_width = 0;
        }
      }
    }
    LaunchClass* cls = nullptr;
    for (LaunchClass& c : g->classes) {
      if (c.shape.nobs == key_nobs && c.shape.nslot == key_nslot && c.shape.lds_hist == lds_hist &&
          c.prog_simple == prog_simple && (!prog_simple || c.prog == prog) && c.pre_mask == pre_mask &&
          c.shape.pre_width == pre_width && c.runs_mode == runs_mode && c.shape.rtc_fill == rtc_fill &&
          c.shape.rtc_sparse == rtc_sparse) {
        cls = &c;
      }
    }
    if (!cls) {
      g->classes.push_back(LaunchClass{});
      cls = &g->classes.back();
      cls->shape.nobs = key_nobs;
      cls->shape.nslot = key_nslot;
      cls->shape.lds_hist = lds_hist;
      cls->shape.threads = threads;
      cls->shape.debug_mode = 0;
      cls->prog = prog;
      cls->prog_simple = prog_simple;
      cls->shape.static_prog = static_prog;
      cls->shape.pre_width = pre_width;
      cls->pre_mask = pre_mask;
      cls->runs_mode = runs_mode;
      cls->shape.rtc_fill = rtc_fill;
      cls->shape.rtc_sparse = rtc_sparse;
    }
    cls->member_idx.push_back(i);
  }

  // sparse flavour: members whose histogram exceeds LDS count into per-event-bin counters
  std::vector<SxSignalDesc> sparse_descs = g->h_descs;
  g->sparse_ready = false;
  g->max_bins_sparse = 0;
  bool sparse_ok = true;
  for (int i = 0; i < n; i++) {
    sxmc_hist* h = g->members[i];
    if (h->total_nbins > kLdsMaxBins) {
      if (h->has_points && h->d_table) {
        make_sparse_desc(h, sparse_descs[(size_t)i]);
        // members that look up the same set of bins (the usual case: one data set, one binning) share ONE
        // filter and table, so the probes of all signals hit the same few cache lines
        for (int k = 0; k < i; k++) {
          const sxmc_hist* o = g->members[k];
          if (o->d_table && o->total_nbins == h->total_nbins && o->targets == h->targets) {
            sparse_descs[(size_t)i].sparse_filter = o->d_filter;
            sparse_descs[(size_t)i].sparse_table = o->d_table;
            sparse_descs[(size_t)i].sparse_coarse = o->d_coarse;
            break;
          }
        }
        g->sparse_ready = true;
      } else {
        sparse_ok = false;
      }
    }
    g->max_bins_sparse = std::max(g->max_bins_sparse, sparse_descs[(size_t)i].total_nbins);
  }
  g->sparse_ready = g->sparse_ready && sparse_ok;
  g->h_descs_sparse = sparse_descs;
  g->ec[0].descs_valid = g->ec[1].descs_valid = false;
  g->prezeroed = 0;
  if (!g->d_descs_sparse) SX_HIP(hipMalloc((void**)&g->d_descs_sparse, sizeof(SxSignalDesc) * std::max(n, 1)));
  if (n) SX_HIP(hipMemcpy(g->d_descs_sparse, sparse_descs.data(), sizeof(SxSignalDesc) * n, hipMemcpyHostToDevice));

  for (LaunchClass& c : g->classes) {
    const bool bucketed = c.shape.pre_width == 3 || c.shape.pre_width == 5;
    const bool ordered = c.shape.pre_width == 5;
    // ---- threads per workgroup, LDS
    int cls_max_bins = 0, cls_nsyst = 0;
    for (int idx : c.member_idx) {
      cls_max_bins = std::max(cls_max_bins, g->h_descs[(size_t)idx].total_nbins);
      cls_nsyst = std::max(cls_nsyst, g->h_descs[(size_t)idx].nsyst);
    }
    c.shape.lds_bytes = c.shape.lds_hist ? ((size_t)cls_max_bins + 4 + 64) * 4 : 64;
    if (ordered && c.shape.lds_hist) c.shape.lds_bytes = ordered_lds_bytes(cls_max_bins, 1, 0);   // (replicas: below)
    c.shape.sparse_runs = 0;
    c.shape.sparse_lds_bytes = 0;
    std::vector<int> K;   // runs mode: workgroups per member
    if (c.runs_mode) {
      // every wave owns 2 x 512 words of LDS (table keys + counts) and walks its own run of consecutive granules:
      // member j gets K_j workgroups (in proportion to its granules) = K_j x waves runs.  Three workgroups of
      // 512 per CU measured best at BASELINE config 5 (2.14 ms; one of 1024: 2.29 ms; thread counts that are not
      // powers of two 2.4 ms); the kernel is bound by vector-instruction issue and HBM together, and 24 waves
      // per CU is what its registers allow.
      const int rthreads = g->cfg_threads > 0 ? c.shape.threads : 512;
      const size_t need = (size_t)(rthreads / 64) * 2u * ((size_t)4 << SXMC_SPARSE_SMAX_LOG2);
      std::vector<unsigned long long> sizes;
      for (int idx : c.member_idx) sizes.push_back((unsigned long long)plans[(size_t)idx].sort->lsrc.size() * 64ull);
      const int rbpc = std::min(g->cfg_bpc > 0 ? g->cfg_bpc : std::max(1, 1536 / rthreads),
                                std::max(1, (int)((size_t)props.lds_per_cu / need)));
      if (need > (size_t)props.lds_per_cu || !apportion_workgroups(sizes, props.cus * rbpc, rthreads, K)) {
        if (ordered && !g->order_blocked) {   // the ordered layout needs the runs: plan again without it
          g->order_blocked = true;
          return group_rebuild(g);
        }
        c.runs_mode = false;   // (more such members than workgroups: the table stays in sorted order)
      } else {
        c.shape.threads = rthreads;
        c.shape.sparse_lds_bytes = need;
      }
    }
    if (!c.shape.lds_hist && g->sparse_ready && !c.runs_mode) {
      int cshift = 32;
      for (int idx : c.member_idx) cshift = std::min(cshift, g->members[idx]->coarse_shift);
      c.shape.lds_bytes = ((size_t)4 + ((size_t)1 << (32 - cshift - 5))) * 4;   // header + largest coarse filter
      // a filter of more than half the LDS leaves room for one workgroup per CU: make it a full one
      if (g->cfg_threads <= 0 && c.shape.lds_bytes * 2 > (size_t)props.lds_per_cu) c.shape.threads = 1024;
    }
    if (g->cfg_threads <= 0 && g->cfg_bpc <= 0 && c.shape.lds_hist && !bucketed && !c.runs_mode) {
      // A SHORT launch with its histograms in LDS (BASELINE config 2: 80 MB, ~10 units per lane): the launch's fixed
      // cost is most of it, and a large part of that is the flush -- every workgroup sends its private histogram
      // to HBM with memory-side atomics.  ONE workgroup of 1024 per CU instead of two of 512 keeps the lanes and
      // halves the histograms to flush: config 2, same box, 17.6 us against 21.0 (35 700 against 33 200 evals/s;
      // 768 x 1: 18.0, 512 x 1: 21.5, 256 x 4: 27.5, 1024 x 2: 19.6; profiles/r03_c2_sweep_policy_x_shape.log).
      double bytes = 0;
      for (int idx : c.member_idx) bytes += (double)g->h_descs[(size_t)idx].nvec * SXMC_VEC * 4.0 * std::max(1, c.shape.nslot);
      if (bytes < 2.0e8) {
        c.shape.threads = 1024;
      }
    }
    int threads = c.shape.threads;  // (shadows the group-wide default above)

    // ---- the members' descriptors; bucketed members: lay the table out now that the shape is known
    std::vector<SxSignalDesc> descs;
    unsigned long long prefix = 0;
    for (size_t q = 0; q < c.member_idx.size(); q++) {
      const int idx = c.member_idx[q];
      SxSignalDesc d = fill_descs[(size_t)idx];
      sxmc_hist* h = g->members[idx];
      if (c.shape.pre_width == 1 || c.shape.pre_width == 2) {
        SampleStore& st = *h->store;
        std::lock_guard<std::mutex> lock(st.pre_mutex);
        void* pre = st.find_pre(c.pre_mask, c.shape.pre_width);
        if (!pre) {
          const size_t npad = h->nvec * SXMC_VEC;
          SX_HIP(hipMalloc(&pre, std::max<size_t>(npad * (size_t)c.shape.pre_width, 16)));
          st.pre.push_back({c.pre_mask, c.shape.pre_width, pre});
          SX_HIP(sx_launch_prebin(g->d_descs + idx, npad, c.pre_mask, c.shape.pre_width, pre, nullptr));
          SX_HIP(hipDeviceSynchronize());
        }
        d.pre = pre;
      }
      if (bucketed) {
        const int runs = c.runs_mode ? std::max(1, K[q]) * (threads / 64) : 1;
        const SampleStore::Bucketed* bk = nullptr;
        rc = get_bucketed(h, plans[(size_t)idx].sort, plans[(size_t)idx].fields, runs, &bk);
        if (rc) return rc;
        d.cols = bk->d_cols;
        d.col_pitch = bk->pitch;
        d.nsamples = bk->ngranules * 256;
        d.nvec = bk->ngranules * 64;
        d.pre = bk->d_gpre;
        d.edges = bk->d_gedge;
        g->member_bucket[(size_t)idx] = bk;
        // CODES: ordered table, histogram in LDS, 2 to 4 streamed fields, every systematic on them affine (one
        // coefficient) -- the conditions fill_ordered_body's kCodes states at compile time
        // (Histograms beyond LDS, counted at the event bins over run-walked tables -- BASELINE config 5 -- were given
        // codes too and measured: bit-identical, and SLOWER, 4.0 ms against 2.1-2.6.  With 200 bins per written
        // observable a code step is 1/220 of a bin, 0.8 % of the samples are ambiguous and 87 % of the 256-sample
        // units hold one; the fix-up then runs almost everywhere.  Codes pay where bins are coarse against 2^-16 of
        // the window: not offered there.)
        bool affine = ordered && c.shape.lds_hist && c.shape.nobs >= 1 && c.shape.nslot - 1 >= 2 &&
                      c.shape.nslot - 1 <= SXMC_MAX_QSLOTS && !c.runs_mode && codes_enabled(g);
        for (unsigned w : c.prog) affine = affine && ((int)((w >> 4) & 15u) == c.shape.nslot - 1 || ((w >> 12) & 15u) == 0u);
        if (affine) {
          rc = get_bucket_codes(h, bk, d);
          if (rc) return rc;
          if (bk->d_qcol) {
            d.qcol = bk->d_qcol;
            for (int m = 0; m < bk->nq; m++) {
              d.qbase[m] = bk->qbase[m];
              d.qstep[m] = bk->qstep[m];
            }
            c.codes = true;
          }
        }
      }
      d.vec_start = prefix;
      prefix += d.nvec;
      descs.push_back(d);
    }
    c.total_vec = prefix;
    // Over codes, where nothing was asked for: TWO workgroups of 512 lanes per CU, each with half the replicas of the
    // LDS histogram.  One of 768 or 1024 with all four replicas is as fast alone (config 3, alternating on one box:
    // 81.0-81.6 us against 81.3-81.4 and 79.6-82.8), but with other chains' launches in flight -- the fake experiments
    // of an ensemble -- two workgroups per CU let one launch's tail run under the next one's start: 13 280 chain-steps/s
    // against 12 430 and 12 840 (profiles/r04b_codes_shapes_ab.log).  A histogram too large for two workgroups' LDS:
    // one of 768.  sxmc_group_optimize times these shapes on the box it runs on.
    const bool codes_auto = c.codes && g->cfg_threads <= 0 && g->cfg_bpc <= 0;
    bool codes_two = false;
    if (codes_auto) {
      const size_t one = std::max(c.shape.lds_bytes, ordered_lds_bytes(cls_max_bins, 1, 0)) + ordered_queue_bytes(kMinQueueLog);
      codes_two = 2 * (one + 2048) <= (size_t)props.lds_per_cu && c.shape.nobs == 1;   // (+ the padded form's guard rows)
      threads = c.shape.threads = codes_two ? 512 : 768;
    }
    // Waves per CU.  The fill is a stream: HBM delivers most with about 32 KiB of loads in flight per CU,
    // which is 512 lanes with one unit (3-4 columns x 16 bytes) each; more waves only queue up (measured
    // -8 % at BASELINE config 3).  Members whose per-sample arithmetic is long (a run-time decoded program
    // of two or more systematics, the shape-agnostic kernel) or that probe L2 per sample (histograms
    // beyond LDS) need the second set of waves to hide it.
    const double stream_bytes = (double)c.total_vec * SXMC_VEC * 4.0 * std::max(1, c.shape.nslot - (ordered ? 1 : 0));
    const bool light = c.shape.lds_hist && (c.shape.nobs > 0 || ordered) &&
                       (c.shape.static_prog >= 0 || c.shape.rtc_fill || cls_nsyst <= 1) &&
                       stream_bytes >= 2.0e8;  // (short launches are ramp-bound: they take all the waves)
    c.light = light;
    int bpc = g->cfg_bpc > 0 ? g->cfg_bpc : codes_auto ? (codes_two ? 2 : 1) : std::max(1, (light ? 512 : 1024) / threads);
    const size_t lds_need = std::max(c.shape.lds_bytes, c.shape.sparse_lds_bytes);
    const int lds_limit = std::max(1, (int)((size_t)props.lds_per_cu / std::max<size_t>(lds_need, 1)));
    bpc = std::min(bpc, lds_limit);
    c.shape.lds_layout = 0;
    if (ordered && c.shape.lds_hist) {
      // replicas of the LDS histogram (fill_ordered_body): as many as the workgroup's share of LDS holds, up to 4
      unsigned rlog = 0;
      const size_t share = (size_t)props.lds_per_cu / (size_t)std::max(1, bpc) - (fused_step_requested(g) ? 16 * 1024 : 0);
      const size_t qreserve = c.codes ? ordered_queue_bytes(kMinQueueLog) : 0;   // (room for the smallest queues)
      // (SXMC_ORDERED_REPLICAS_LOG2, measurement: fewer replicas leave LDS for a second workgroup per CU -- of another
      // chain's launch, say)
      static const unsigned rlog_max = [] {
        const char* e = measure_env("SXMC_ORDERED_REPLICAS_LOG2");
        return e ? (unsigned)std::min(std::max(std::atoi(e), 0), 2) : 2u;
      }();
      while (rlog < rlog_max && ordered_lds_bytes(cls_max_bins, 1, rlog + 1) + qreserve <= share) rlog++;
      c.shape.lds_layout = ordered_rstride(cls_max_bins) | (rlog << 24);
      c.shape.lds_bytes = ordered_lds_bytes(cls_max_bins, 1, rlog);
      c.plain_rstride = ordered_rstride(cls_max_bins);
      if (c.codes) {
        // the padded form of the histogram where every member qualifies (one observable binned per sample, the
        // outermost dimension) and it fits with room for the smallest queue; then the queues of ambiguous rows, in
        // what the replicas leave of the workgroup's share
        c.padded_rstride = 0;
        if (c.shape.nobs == 1) {
          unsigned rs = 0;
          bool all = true;
          for (size_t q = 0; q < c.member_idx.size(); q++) {
            const sxmc_hist* h = g->members[(size_t)c.member_idx[q]];
            const long long S = descs[q].bin_stride[0], nb = descs[q].nbins[0];   // (slot 0 of the compacted problem)
            all = all && S >= 1 && nb >= 1 && S * nb == (long long)h->total_nbins && S * (nb + 2) < (1ll << 22);
            if (all) rs = std::max(rs, ordered_rstride_padded(h->total_nbins, (int)nb));
          }
          if (all && rs) {
            unsigned prl = 0;
            auto bytes = [&](unsigned rl) { return (4 + ((size_t)rs << rl) + 64) * 4 + ordered_queue_bytes(kMinQueueLog); };
            if (bytes(0) <= share) {
              while (prl < rlog_max && bytes(prl + 1) <= share) prl++;
              c.padded_rstride = rs;
              c.shape.lds_layout = rs | (prl << 24) | (1u << 27);
              c.shape.lds_bytes = (4 + ((size_t)rs << prl) + 64) * 4;
            }
          }
        }
        const unsigned qlog = share > c.shape.lds_bytes ? ordered_queue_log(share - c.shape.lds_bytes, g->cfg_queue_log) : 0;
        c.shape.lds_layout |= qlog << 28;
        c.shape.lds_bytes += ordered_queue_bytes(qlog);
        if (!qlog) c.codes = false;   // (no room for queues: the kernel streams the float columns)
      }
    }
    unsigned long long grid = (unsigned long long)props.cus * bpc;
    const unsigned long long want = (c.total_vec + threads - 1) / threads;  // >= 1 unit per lane
    grid = std::max<unsigned long long>(1, std::min(grid, want));
    c.shape.grid = c.total_vec ? (int)grid : 0;
    if (c.runs_mode) {
      int used = 0;
      for (int k : K) used += k;
      c.shape.grid = used;
      c.shape.sparse_runs = 1;
    }
    SX_HIP(hipMalloc((void**)&c.d_descs, sizeof(SxSignalDesc) * descs.size()));
    SX_HIP(hipMemcpy(c.d_descs, descs.data(), sizeof(SxSignalDesc) * descs.size(), hipMemcpyHostToDevice));
    if (g->sparse_ready && !c.shape.lds_hist) {
      std::vector<SxSignalDesc> sd = descs;
      for (size_t q = 0; q < sd.size(); q++) {
        sxmc_hist* h = g->members[c.member_idx[q]];
        make_sparse_desc(h, sd[q]);
        sd[q].sparse_filter = sparse_descs[(size_t)c.member_idx[q]].sparse_filter;  // shared tables
        sd[q].sparse_table = sparse_descs[(size_t)c.member_idx[q]].sparse_table;
        sd[q].sparse_coarse = sparse_descs[(size_t)c.member_idx[q]].sparse_coarse;
        if (c.runs_mode) {
          // members that look up the same set of bins share ONE set of bucket tables (one data set, one binning)
          const sxmc_hist* owner = h;
          for (size_t k = 0; k < q; k++) {
            const sxmc_hist* o = g->members[c.member_idx[k]];
            if (o->btab_valid && o->btab_mask == h->btab_mask && o->total_nbins == h->total_nbins &&
                o->targets == h->targets) {
              owner = o;
              break;
            }
          }
          sd[q].sparse_dir = owner->d_bdir;
          sd[q].sparse_tkeys = owner->d_btkeys;
          sd[q].sparse_tslot = owner->d_btslot;
          sd[q].pre = g->member_bucket[(size_t)c.member_idx[q]]->d_gkp;   // {bucket key, bin offset} per granule
        }
      }
      SX_HIP(hipMalloc((void**)&c.d_descs_sparse, sizeof(SxSignalDesc) * sd.size()));
      SX_HIP(hipMemcpy(c.d_descs_sparse, sd.data(), sizeof(SxSignalDesc) * sd.size(), hipMemcpyHostToDevice));
    }
    if (c.shape.grid > 0) {
      std::vector<SxSegment> segs;
      std::vector<unsigned> blk_off;
      if (c.runs_mode) {
        interleaved_segments(descs, K, threads, c.shape.grid, segs, blk_off);
        c.partition = 2;
      } else {
        // Bucketed tables are sorted by bin, so a member's workgroups can work as TEAMS over contiguous parts of it
        // (sxplan::interleaved_segments): a workgroup of a team of 7 sees a third of the histogram's bins, and the
        // flush -- one memory-side atomic per non-zero bin of every workgroup, 1.3 M per launch at config 3 -- sends
        // a third of the atomics, against a coarser interleaving of the stream.  Which wins depends on the BOX
        // (tools/part_groups_sweep.sh, profiles/r03_c3_teams_sweep.log: 3 teams 129.4 us against 133.4-134.4 on one,
        // 128.4 against 124.9 on another, each consistently over alternating runs), so the default is one team and
        // sxmc_group_optimize tries three on the box it runs on (SXMC_PART_GROUPS forces a count for A/B runs).
        static const int forced_groups = [] {
          const char* e = measure_env("SXMC_PART_GROUPS");
          return e ? std::atoi(e) : 0;
        }();
        const int groups = (bucketed && c.shape.lds_hist)
                               ? (forced_groups > 0 ? forced_groups : std::max(1, g->cfg_teams)) : 1;
        c.teams = groups;
        build_partition(descs, c.shape.grid, threads, g->cfg_partition, segs, blk_off, c.partition, bucketed ? 64 : 1,
                        groups);
      }
      SX_HIP(hipMalloc((void**)&c.d_segs, sizeof(SxSegment) * std::max<size_t>(segs.size(), 1)));
      SX_HIP(hipMalloc((void**)&c.d_blk_off, sizeof(unsigned) * blk_off.size()));
      if (!segs.empty()) {
        SX_HIP(hipMemcpy(c.d_segs, segs.data(), sizeof(SxSegment) * segs.size(), hipMemcpyHostToDevice));
      }
      SX_HIP(hipMemcpy(c.d_blk_off, blk_off.data(), sizeof(unsigned) * blk_off.size(), hipMemcpyHostToDevice));
    }
  }

  g->seen.resize((size_t)n);
  g->seen_points.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    g->seen[i] = g->members[i]->version;
    g->seen_points[i] = g->members[i]->points_version;
  }
  g->cfg_seen_threads = g->cfg_threads;
  g->cfg_seen_bpc = g->cfg_bpc;
  g->cfg_seen_partition = g->cfg_partition;
  g->cfg_seen_teams = g->cfg_teams;
  g->cfg_seen_prebin = g->cfg_prebin;
  g->cfg_seen_bucket = g->cfg_bucket;
  g->cfg_seen_order = g->cfg_order;
  g->cfg_seen_rtc = g->cfg_rtc;
  g->cfg_seen_codes = g->cfg_codes;
  g->cfg_seen_queue_log = g->cfg_queue_log;
  g->cfg_seen_fused = g->cfg_fused;
  g->plan_generation++;
  g->built = true;
  return SXMC_OK;
}

// New evaluation points of the same size class (same buffers, sxmc_hist_set_eval_points) change two fields of
// the members' descriptors and nothing else: they are patched and re-uploaded, with no device-wide
// synchronisation and no re-planning, so a fake experiment's set-up does not stall the other chains on the GPU.
int group_update_points(sxmc_group* g) {
  const int n = (int)g->members.size();
  g->max_points = 0;
  g->same_points = n > 0;
  for (int i = 0; i < n; i++) {
    sxmc_hist* h = g->members[i];
    const unsigned long long np = h->has_points ? h->npoints : 0;
    for (std::vector<SxSignalDesc>* set : {&g->h_descs, &g->h_descs_sparse}) {
      (*set)[(size_t)i].read_bins = h->has_points ? h->d_read_bins : nullptr;
      (*set)[(size_t)i].npoints = np;
    }
    g->max_points = std::max(g->max_points, np);
    if (!h->has_points || h->npoints != g->members[0]->npoints) g->same_points = false;
    g->seen_points[(size_t)i] = h->points_version;
  }
  if (n) {
    SX_HIP(hipMemcpy(g->d_descs, g->h_descs.data(), sizeof(SxSignalDesc) * n, hipMemcpyHostToDevice));
    SX_HIP(hipMemcpy(g->d_descs_sparse, g->h_descs_sparse.data(), sizeof(SxSignalDesc) * n, hipMemcpyHostToDevice));
  }
  g->ec[0].descs_valid = g->ec[1].descs_valid = false;
  return SXMC_OK;
}

int group_refresh(sxmc_group* g) {
  bool stale = !g->built || g->cfg_seen_threads != g->cfg_threads || g->cfg_seen_bpc != g->cfg_bpc ||
               g->cfg_seen_partition != g->cfg_partition || g->cfg_seen_teams != g->cfg_teams ||
               g->cfg_seen_prebin != g->cfg_prebin ||
               g->cfg_seen_bucket != g->cfg_bucket || g->cfg_seen_rtc != g->cfg_rtc ||
               g->cfg_seen_order != g->cfg_order || g->cfg_seen_codes != g->cfg_codes ||
               g->cfg_seen_queue_log != g->cfg_queue_log || g->cfg_seen_fused != g->cfg_fused;
  bool points = false;
  for (size_t i = 0; !stale && i < g->members.size(); i++) {
    if (g->seen[i] != g->members[i]->version) stale = true;
    if (g->seen_points[i] != g->members[i]->points_version) points = true;
  }
  if ((stale || points) && t_capturing) {
    return fail(SXMC_ERR_STATE, "the group's launch plan is out of date: evaluate once before recording a graph");
  }
  if (stale) return group_rebuild(g);
  return points ? group_update_points(g) : SXMC_OK;
}

int group_check_bound(sxmc_group* g, bool need_pdf) {
  for (sxmc_hist* h : g->members) {
    if (!h->norm) return fail(SXMC_ERR_STATE, "evaluation before SetNormalizationBuffer");
    if (!h->systs.empty() && !h->params) return fail(SXMC_ERR_STATE, "evaluation before SetParameterBuffer");
    if (need_pdf && h->has_points && !h->pdf) return fail(SXMC_ERR_STATE, "evaluation before SetPDFValueBuffer");
  }
  return SXMC_OK;
}

void free_event_classes(sxmc_group::EventClasses& ec) {
  if (ec.d_rb) (void)hipFree(ec.d_rb);
  if (ec.d_weight) (void)hipFree(ec.d_weight);
  if (ec.d_descs) (void)hipFree(ec.d_descs);
  ec = sxmc_group::EventClasses{};
}

// Event classes of one descriptor flavour, built on the host from the members' event-bin tables:
// events with the same bin (counter slot) in every member contribute the same term to the event sum,
// so the sum runs over the distinct tuples, each weighted by its multiplicity.  Tables are rebuilt when
// evaluation points change, the descriptor copies whenever the group is rebuilt.
int ensure_event_classes(sxmc_group* g, bool sparse) {
  sxmc_group::EventClasses& ec = g->ec[sparse ? 1 : 0];
  const size_t S = g->members.size();
  bool tables_ok = ec.tables_valid && ec.seen_points.size() == S;
  for (size_t j = 0; tables_ok && j < S; j++) tables_ok = ec.seen_points[j] == g->members[j]->points_version;
  if (tables_ok && ec.descs_valid) return SXMC_OK;
  if (t_capturing) {
    return fail(SXMC_ERR_STATE, "the event classes are out of date: evaluate once before recording a graph");
  }
  // (the caller is done with the group's previous evaluations; buffers are re-used, so nothing stalls the device)
  const std::vector<SxSignalDesc>& flavour = sparse ? g->h_descs_sparse : g->h_descs;
  if (!tables_ok) {
    const size_t E = g->members[0]->npoints;
    // the table each member's descriptor of this flavour reads
    std::vector<const std::vector<int>*> arr(S);
    for (size_t j = 0; j < S; j++) {
      const sxmc_hist* h = g->members[j];
      const bool slots = sparse && flavour[j].read_bins == h->d_read_slot && h->d_read_slot != nullptr;
      arr[j] = slots ? &h->h_read_slot : &h->h_read_bins;
      if (arr[j]->size() != E) return fail(SXMC_ERR_STATE, "event-bin table of a member is missing");
    }
    sxplan::EventClasses cls;   // distinct tuples of event bins + multiplicities (sxmc_plan.h)
    sxplan::event_classes(arr, E, cls);
    const size_t K = cls.K;
    const std::vector<int>& tables = cls.tables;
    const std::vector<unsigned>& weight = cls.weight;
    if (tables.size() > ec.cap_rb) {
      if (ec.d_rb) SX_HIP(hipFree(ec.d_rb));
      ec.d_rb = nullptr;
      ec.cap_rb = tables.size() + tables.size() / 4;
      SX_HIP(hipMalloc((void**)&ec.d_rb, sizeof(int) * ec.cap_rb));
    }
    if (std::max<size_t>(K, 1) > ec.cap_weight) {
      if (ec.d_weight) SX_HIP(hipFree(ec.d_weight));
      ec.d_weight = nullptr;
      ec.cap_weight = K + K / 4 + 16;
      SX_HIP(hipMalloc((void**)&ec.d_weight, sizeof(unsigned) * ec.cap_weight));
    }
    SX_HIP(hipMemcpy(ec.d_rb, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
    if (K) SX_HIP(hipMemcpy(ec.d_weight, weight.data(), sizeof(unsigned) * K, hipMemcpyHostToDevice));
    ec.K = K;
    ec.seen_points.resize(S);
    for (size_t j = 0; j < S; j++) ec.seen_points[j] = g->members[j]->points_version;
    ec.tables_valid = true;
    ec.descs_valid = false;
  }
  std::vector<SxSignalDesc> descs = flavour;
  for (size_t j = 0; j < S; j++) {
    descs[j].read_bins = ec.d_rb + j * ec.K;
    descs[j].npoints = ec.K;
    descs[j].pdf_out = nullptr;
  }
  if (!ec.d_descs) SX_HIP(hipMalloc((void**)&ec.d_descs, sizeof(SxSignalDesc) * std::max<size_t>(S, 1)));
  SX_HIP(hipMemcpy(ec.d_descs, descs.data(), sizeof(SxSignalDesc) * S, hipMemcpyHostToDevice));
  ec.descs_valid = true;
  return SXMC_OK;
}

// What precedes a group's fill launches: the zeroing (unless the last step end already cleared for this
// evaluation) and the bookkeeping of who cleared what.
int group_prepare_fill(sxmc_group* g, hipStream_t s, bool sparse) {
  // Recorded launches do not run now, so what a recording "pre-zeroed" is not zero yet: the first
  // evaluation of every recording zeroes explicitly, and sxmc_graph_end_capture drops the flag.
  bool first_in_recording = false;
  if (t_capturing && g->capture_epoch != t_capture_epoch) {
    g->capture_epoch = t_capture_epoch;
    t_capture_groups.push_back(g);
    first_in_recording = true;
  }
  bool skip_zero = g->prezeroed == (sparse ? 2 : 1) && !first_in_recording;
  for (sxmc_hist* h : g->members) {  // (an evaluator may also be evaluated alone or through another group)
    skip_zero = skip_zero && h->cleared_by == g;
    h->cleared_by = nullptr;
  }
  g->prezeroed = 0;
  g->last_sparse = sparse;
  if (!skip_zero) {
    SX_HIP(sx_launch_zero(sparse ? g->d_descs_sparse : g->d_descs, (int)g->members.size(),
                          sparse ? g->max_bins_sparse : g->max_bins, g->d_ticket, s));
  }
  for (size_t i = 0; i < g->members.size(); i++) {
    g->members[i]->bins_valid = g->members[i]->total_nbins > kLdsMaxBins ? !sparse : true;
  }
  return SXMC_OK;
}

int group_fill(sxmc_group* g, hipStream_t s, bool sparse = false) {
  TraceRange trace("sxmc: fill (EvalHist of all signals)");
  sparse = sparse && g->sparse_ready && g->cfg_sparse;
  int rc = group_prepare_fill(g, s, sparse);
  if (rc) return rc;
  for (LaunchClass& c : g->classes) {
    // profiled launches (sxmc_group_profile) carry two events stamped with the dispatch's own begin and end
    const bool rec = g->prof && !t_capturing && g->prof_n < (int)g->ev0.size() && c.shape.grid > 0;
    c.shape.ev_start = rec ? (void*)g->ev0[g->prof_n] : nullptr;
    c.shape.ev_stop = rec ? (void*)g->ev1[g->prof_n] : nullptr;
    c.shape.debug_mode = g->debug_mode;
    if (sparse && c.d_descs_sparse && c.shape.sparse_runs) {
      SX_HIP(sx_launch_fill_sparse_runs(c.shape, c.d_descs_sparse, c.d_segs, c.d_blk_off, s));
    } else {
      SX_HIP(sx_launch_fill(c.shape, (sparse && c.d_descs_sparse) ? c.d_descs_sparse : c.d_descs, c.d_segs,
                            c.d_blk_off, s));
    }
    c.shape.ev_start = c.shape.ev_stop = nullptr;
    if (rec) g->prof_n++;
  }
  return SXMC_OK;
}

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
struct AutoGroup {
  std::vector<sxmc_hist*> members;
  sxmc_group* g = nullptr;
};
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
std::atomic<int> g_captures_in_progress{0};
hipStream_t batch_stream(const std::vector<sxmc_hist*>& m) {
  static const bool own = [] {
    const char* e = measure_env("SXMC_DEFER_STREAM");
    return e && std::string(e) == "own";
  }();
  const bool legacy_ok = g_captures_in_progress.load(std::memory_order_acquire) == 0;
  return (m.size() >= 2 && !own && legacy_ok) ? nullptr : m[0]->stream;
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
  fl->stream = batch_stream(m);
  t_flushing = true;    // (the group calls below are themselves flush points)
  int rc = SXMC_OK;
  // EvalHist's `optimize` (pdfz.cpp:188, 441-448, 622-628): the reference's evaluator runs its launch-shape trials
  // inside its first EvalAsync, once it has evaluation points, unless it was constructed with optimize = false -- and
  // never while making a histogram (pdfz.cpp:503-504: CreateHistogram switches it off around its EvalAsync(false)).
  // Here the trials are the BATCH's (sxmc_group_optimize: a few timed fills choose lanes per CU, teams and codes for
  // this box; only a long pure stream has anything to choose, it returns at once otherwise): at the batch's first
  // lookup evaluation, when every member asks for it.  The evaluation proper follows and zeroes what the trials counted.
  bool want = do_eval_pdf != 0;
  for (sxmc_hist* h : m) want = want && h->want_optimize && h->has_points;
  if (m.size() == 1 || m[0]->cfg_threads > 0 || m[0]->cfg_bpc > 0) {
    g->cfg_threads = m[0]->cfg_threads;   // (a launch shape set by hand on the evaluators)
    g->cfg_bpc = m[0]->cfg_bpc;
  } else if (m.size() >= 2 && !g->tuned && want) {
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

// LAZY EvalFinished.  A batch launched on the legacy default stream is ordered, on the device, before everything the
// caller does next through this ABI on that stream or on any blocking stream -- its NLL kernels (mcmc.cpp:314-348),
// blocking copies to the host, the next evaluation.  So sxmc_hist_eval_finished of such a batch does not stop the host:
// it notes that the thread has an unwaited batch, and the wait happens at the first call that could tell -- one that
// names a stream which does NOT order with the legacy stream (created non-blocking), or that synchronises.  The host
// then runs ahead of the device like a caller of the group API does, instead of idling the device once per step while
// it wakes up and launches the rest of the step (measured at BASELINE config 3: ~25 us of ~190).  What this cannot
// cover is device work the caller issues OUTSIDE this ABI on a non-blocking stream of its own right after
// EvalFinished; sxmc_set_lazy_finish(0) (SXMC_LAZY_FINISH=0) restores the blocking wait for such callers.
std::atomic<int> g_lazy_finish{-1};
thread_local bool t_unsettled = false;   // this thread returned from an EvalFinished without waiting
bool lazy_finish_enabled() {
  int v = g_lazy_finish.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("SXMC_LAZY_FINISH");
    v = (e && e[0] == '0') ? 0 : 1;
    g_lazy_finish.store(v, std::memory_order_relaxed);
  }
  return v > 0;
}
int settle() {
  if (!t_unsettled) return SXMC_OK;
  t_unsettled = false;
  SX_HIP(hipStreamSynchronize(nullptr));
  return SXMC_OK;
}
// Before work is put on stream `s`: does `s` order with the legacy stream by itself?
int settle_for(hipStream_t s) {
  if (!t_unsettled || s == nullptr) return SXMC_OK;
  unsigned flags = 0;
  if (hipStreamGetFlags(s, &flags) == hipSuccess && !(flags & hipStreamNonBlocking)) return SXMC_OK;
  (void)hipGetLastError();
  return settle();
}

#define SX_ORDER(s)                                  \
  do {                                               \
    if (t_unsettled) {                               \
      int _rc = settle_for((hipStream_t)(s));        \
      if (_rc) return _rc;                           \
    }                                                \
  } while (0)

#define SX_FLUSH()                                                          \
  do {                                                                      \
    if (t_deferred && t_deferred->n.load(std::memory_order_relaxed) != 0) { \
      int _rc = flush_deferred();                                           \
      if (_rc) return _rc;                                                  \
    }                                                                       \
  } while (0)

}  // namespace

extern "C" {

const char* sxmc_last_error(void) { return g_last_error.c_str(); }
const char* sxmc_version(void) { return "sxmc_hip 0.1 (gfx950)"; }

int sxmc_device_count(int* count) {
  SX_REQUIRE(count, "null argument");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SXMC_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return SXMC_OK;
}

int sxmc_set_device(int device) {
  SX_HIP(hipSetDevice(device));
  return SXMC_OK;
}

int sxmc_get_device(int* device) {
  SX_REQUIRE(device, "null argument");
  SX_HIP(hipGetDevice(device));
  return SXMC_OK;
}

int sxmc_device_info(int device, char* name, int* compute_units, size_t* hbm_bytes, int* lds_bytes_per_cu,
                     int* clock_khz) {
  hipDeviceProp_t prop;
  SX_HIP(hipGetDeviceProperties(&prop, device));
  if (name) {
    std::snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (clock_khz) *clock_khz = prop.clockRate;
  return SXMC_OK;
}

int sxmc_set_tracing(int enable) {
  g_tracing.store(enable ? 1 : 0, std::memory_order_relaxed);
  return SXMC_OK;
}

int sxmc_device_synchronize(void) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}

int sxmc_device_pci_bus_id(int device, char* out, size_t out_bytes) {
  SX_REQUIRE(out && out_bytes >= 16, "buffer of at least 16 bytes");
  SX_HIP(hipDeviceGetPCIBusId(out, (int)out_bytes, device));
  return SXMC_OK;
}

int sxmc_mem_info(size_t* free_bytes, size_t* total_bytes) {
  SX_REQUIRE(free_bytes && total_bytes, "null argument");
  SX_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return SXMC_OK;
}

int sxmc_malloc(void** d_ptr, size_t bytes) {
  SX_REQUIRE(d_ptr, "null argument");
  SX_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
  return SXMC_OK;
}
int sxmc_free(void* d_ptr) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (d_ptr) SX_HIP(hipFree(d_ptr));
  return SXMC_OK;
}
int sxmc_host_alloc(void** h_ptr, size_t bytes) {
  SX_REQUIRE(h_ptr, "null argument");
  SX_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 4, hipHostMallocDefault));
  return SXMC_OK;
}
int sxmc_host_free(void* h_ptr) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (h_ptr) SX_HIP(hipHostFree(h_ptr));
  return SXMC_OK;
}
int sxmc_memcpy_h2d(void* d, const void* h, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
  return SXMC_OK;
}
int sxmc_memcpy_d2h(void* h, const void* d, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
  return SXMC_OK;
}
int sxmc_memcpy_d2d(void* d, const void* s, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice));
  return SXMC_OK;
}
int sxmc_memcpy_h2d_async(void* d, const void* h, size_t n, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  if (n) SX_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_memcpy_d2h_async(void* h, const void* d, size_t n, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  if (n) SX_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_memset(void* d, int v, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemset(d, v, n));
  return SXMC_OK;
}

int sxmc_stream_create(sxmc_stream_t* s) {
  SX_REQUIRE(s, "null argument");
  hipStream_t st;
  SX_HIP(hipStreamCreate(&st));
  *s = st;
  return SXMC_OK;
}
int sxmc_stream_create_nonblocking(sxmc_stream_t* s) {
  SX_REQUIRE(s, "null argument");
  hipStream_t st;
  SX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = st;
  return SXMC_OK;
}
int sxmc_stream_destroy(sxmc_stream_t s) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (s) SX_HIP(hipStreamDestroy((hipStream_t)s));
  return SXMC_OK;
}
int sxmc_stream_synchronize(sxmc_stream_t s) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_HIP(hipStreamSynchronize((hipStream_t)s));
  return SXMC_OK;
}
int sxmc_stream_query(sxmc_stream_t s, int* done) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_REQUIRE(done, "null argument");
  const hipError_t e = hipStreamQuery((hipStream_t)s);
  if (e == hipSuccess) {
    *done = 1;
    return SXMC_OK;
  }
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();   // (not an error: clear the sticky code)
    *done = 0;
    return SXMC_OK;
  }
  SX_HIP(e);
  return SXMC_OK;
}

int sxmc_graph_begin_capture(sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(s, "the legacy default stream cannot be captured: pass a created stream");
  SX_REQUIRE(!t_capturing, "a capture is already in progress on this thread");
  SX_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
  g_captures_in_progress.fetch_add(1, std::memory_order_acq_rel);
  t_capturing = true;
  t_capture_epoch++;
  t_capture_groups.clear();
  return SXMC_OK;
}
int sxmc_graph_end_capture(sxmc_stream_t s, sxmc_graph_t* out) {
  SX_REQUIRE(s && out, "null argument");
  SX_REQUIRE(t_capturing, "no capture in progress on this thread");
  t_capturing = false;
  g_captures_in_progress.fetch_sub(1, std::memory_order_acq_rel);
  for (sxmc_group* g : t_capture_groups) g->prezeroed = 0;  // nothing recorded has run yet
  t_capture_groups.clear();
  hipGraph_t graph = nullptr;
  SX_HIP(hipStreamEndCapture((hipStream_t)s, &graph));
  if (!graph) return fail(SXMC_ERR_HIP, "hipStreamEndCapture returned no graph (an error ended the capture)");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(SXMC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  *out = exec;
  return SXMC_OK;
}
int sxmc_graph_launch(sxmc_graph_t graph, sxmc_stream_t s, int times) {
  SX_FLUSH();
  SX_ORDER(s);
  TraceRange trace("sxmc: graph replay");
  SX_REQUIRE(graph, "null graph");
  SX_REQUIRE(times >= 0, "negative repeat count");
  for (int i = 0; i < times; i++) SX_HIP(hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_graph_destroy(sxmc_graph_t graph) {
  if (graph) SX_HIP(hipGraphExecDestroy((hipGraphExec_t)graph));
  return SXMC_OK;
}

int sxmc_event_create(sxmc_event_t* e) {
  SX_REQUIRE(e, "null argument");
  hipEvent_t ev;
  SX_HIP(hipEventCreate(&ev));
  *e = ev;
  return SXMC_OK;
}
int sxmc_event_destroy(sxmc_event_t e) {
  if (e) SX_HIP(hipEventDestroy((hipEvent_t)e));
  return SXMC_OK;
}
int sxmc_event_record(sxmc_event_t e, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_HIP(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_event_synchronize(sxmc_event_t e) {
  SX_HIP(hipEventSynchronize((hipEvent_t)e));
  return SXMC_OK;
}
int sxmc_event_elapsed_ms(sxmc_event_t a, sxmc_event_t b, float* ms) {
  SX_REQUIRE(ms, "null argument");
  SX_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return SXMC_OK;
}

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
  h->version++;
  return SXMC_OK;
}
int sxmc_hist_set_normalization_buffer(sxmc_hist_t h, unsigned* d_norm, int offset) {
  SX_REQUIRE(h, "null evaluator");
  if (h->norm == d_norm && h->norm_off == offset) return SXMC_OK;
  if (h->deferred) SX_FLUSH();
  h->norm = d_norm;
  h->norm_off = offset;
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
      if (fl->stream == nullptr && lazy_finish_enabled() && !t_capturing) {
        t_unsettled = true;      // (ordered on the device; the host waits when it could first tell: see settle_for)
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

// ------------------------------------------------------------------------------ group
int sxmc_group_create(const sxmc_hist_t* members, int nmembers, sxmc_group_t* out) {
  SX_REQUIRE(out && nmembers >= 0 && (members || nmembers == 0), "bad arguments");
  for (int i = 0; i < nmembers; i++) SX_REQUIRE(members[i], "null member");
  sxmc_group* g = new sxmc_group;
  g->members.assign(members, members + nmembers);
  hipError_t e = hipMalloc((void**)&g->d_ticket, 256);
  if (e == hipSuccess) e = hipMemset(g->d_ticket, 0, 256);
  if (e == hipSuccess) e = hipMalloc((void**)&g->d_step_sums, 1024 * sizeof(double));
  // (the cooperative step end's hand-over slots, 128 workers at most: see step_end_is_cooperative)
  if (e == hipSuccess) e = hipMalloc((void**)&g->d_coop_slots, sizeof(unsigned long long) * 128);
  if (e == hipSuccess) e = hipMalloc((void**)&g->d_coop_last, sizeof(double) * 128);
  if (e == hipSuccess) e = sx_step_end_slots_init(g->d_coop_slots, g->d_coop_last, 128);
  if (e != hipSuccess) {
    if (g->d_ticket) (void)hipFree(g->d_ticket);
    if (g->d_step_sums) (void)hipFree(g->d_step_sums);
    if (g->d_coop_slots) (void)hipFree(g->d_coop_slots);
    if (g->d_coop_last) (void)hipFree(g->d_coop_last);
    delete g;
    return fail(SXMC_ERR_HIP, std::string("group allocation: ") + hipGetErrorString(e));
  }
  *out = g;
  return SXMC_OK;
}

int sxmc_group_destroy(sxmc_group_t g) {
  SX_FLUSH();
  if (!g) return SXMC_OK;
  (void)hipDeviceSynchronize();
  for (LaunchClass& c : g->classes) free_class(c);
  if (g->d_descs) (void)hipFree(g->d_descs);
  if (g->d_descs_sparse) (void)hipFree(g->d_descs_sparse);
  if (g->d_ticket) (void)hipFree(g->d_ticket);
  if (g->d_step_sums) (void)hipFree(g->d_step_sums);
  if (g->d_coop_slots) (void)hipFree(g->d_coop_slots);
  if (g->d_coop_last) (void)hipFree(g->d_coop_last);
  free_event_classes(g->ec[0]);
  free_event_classes(g->ec[1]);
  for (hipEvent_t e : g->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : g->ev1) (void)hipEventDestroy(e);
  delete g;
  return SXMC_OK;
}

int sxmc_group_set_launch_config(sxmc_group_t g, int bin_threads, int bin_blocks_per_cu) {
  SX_REQUIRE(g, "null group");
  SX_REQUIRE(bin_threads == 0 || (bin_threads >= 64 && bin_threads <= 1024 && bin_threads % 64 == 0),
             "bin_threads must be 0 or a multiple of 64 up to 1024");
  SX_REQUIRE(bin_blocks_per_cu >= 0 && bin_blocks_per_cu <= 16, "bin_blocks_per_cu out of range");
  g->cfg_threads = bin_threads;
  g->cfg_bpc = bin_blocks_per_cu;
  return SXMC_OK;
}

int sxmc_group_optimize(sxmc_group_t g, sxmc_stream_t s, int* chosen_threads) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g, "null group");
  SX_REQUIRE(!t_capturing, "not while recording a graph");
  if (chosen_threads) *chosen_threads = 0;
  int rc = group_refresh(g);
  if (rc) return rc;
  rc = group_check_bound(g, false);
  if (rc) return rc;
  // only the pure-stream launches have anything to choose: how many lanes per CU keep HBM busiest differs
  // by a few per cent from one box to the next
  if (g->classes.empty() || g->cfg_threads > 0 || g->cfg_bpc > 0) return SXMC_OK;
  for (const LaunchClass& c : g->classes)
    if (!c.light) return SXMC_OK;
  hipStream_t st = (hipStream_t)s;
  hipEvent_t e0, e1;
  SX_HIP(hipEventCreate(&e0));
  SX_HIP(hipEventCreate(&e1));
  // (over codes a lane has half the bytes per unit in flight: the larger shapes are the candidates there)
  bool has_codes = false;
  for (const LaunchClass& c : g->classes) has_codes = has_codes || c.codes;
  // (threads, workgroups per CU; the first is what group_rebuild takes where nothing is asked for)
  typedef std::pair<int, int> Shape;
  const std::vector<Shape> candidates =
      has_codes ? std::vector<Shape>{{512, 2}, {768, 1}, {1024, 1}, {896, 1}, {640, 1}}
                : std::vector<Shape>{{512, 1}, {448, 1}, {576, 1}, {640, 1}, {768, 1}};
  int best_threads = 0, best_bpc = 0;
  float best_ms = 0;
  int failure = SXMC_OK;
  // one candidate's time: eight fills of the current plan, the first warms up, the minimum of the other seven counts
  auto timed_fill = [&]() -> float {
    float ms = 1e30f;
    for (int rep = 0; rep < 8 && failure == SXMC_OK; rep++) {
      hipError_t e = hipEventRecord(e0, st);
      g->trial_launches++;
      if (e == hipSuccess) failure = group_fill(g, st, false);
      if (failure == SXMC_OK && e == hipSuccess) e = hipEventRecord(e1, st);
      if (failure == SXMC_OK && e == hipSuccess) e = hipEventSynchronize(e1);
      float t = 0;
      if (failure == SXMC_OK && e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
      if (failure == SXMC_OK && e != hipSuccess) failure = fail(SXMC_ERR_HIP, std::string("optimize: ") + hipGetErrorString(e));
      if (rep > 0 && t < ms) ms = t;
    }
    return ms;
  };
  for (const Shape& shape : candidates) {
    const int cand = shape.first;
    g->cfg_threads = cand;
    g->cfg_bpc = shape.second;
    if ((failure = group_refresh(g)) != SXMC_OK) break;
    const float ms = timed_fill();
    if (failure != SXMC_OK) break;
    if (best_threads == 0 || ms < best_ms) {
      best_threads = cand;
      best_bpc = shape.second;
      best_ms = ms;
    }
  }
  // the default shape is what 0, 0 means: keep the configuration "automatic" when it won
  const bool is_auto = best_threads == candidates[0].first && best_bpc == candidates[0].second;
  g->cfg_threads = (failure == SXMC_OK && !is_auto) ? best_threads : 0;
  g->cfg_bpc = (failure == SXMC_OK && !is_auto) ? best_bpc : 0;
  // second choice, for bucketed tables: one team of workgroups per member or three (see group_rebuild: which is
  // faster differs from box to box by ~3 % either way); three must win by 1.5 % to be taken
  bool has_bucketed = false;
  for (const LaunchClass& c : g->classes) has_bucketed = has_bucketed || ((c.shape.pre_width == 3 || c.shape.pre_width == 5) && c.shape.lds_hist);
  if (failure == SXMC_OK && has_bucketed && g->cfg_teams == 0) {
    float ms_of[2] = {best_ms, 1e30f};
    for (int pass = 0; pass < 2 && failure == SXMC_OK; pass++) {
      g->cfg_teams = pass == 0 ? 0 : 3;
      if ((failure = group_refresh(g)) != SXMC_OK) break;
      const float ms = timed_fill();
      ms_of[pass] = ms;
    }
    g->cfg_teams = (failure == SXMC_OK && ms_of[1] < 0.985f * ms_of[0]) ? 3 : 0;
  }
  // third choice, where the plan streams codes by default: the codes against the float columns, at the parameters
  // that are bound now.  Whether they pay was estimated from the binning (get_bucket_codes); this is the measurement:
  // the float stream is taken if it wins by 3 %.
  if (failure == SXMC_OK && has_codes && g->cfg_codes < 0) {
    float ms_of[2] = {1e30f, 1e30f};
    for (int pass = 0; pass < 2 && failure == SXMC_OK; pass++) {
      g->cfg_codes = pass == 0 ? -1 : 0;
      if ((failure = group_refresh(g)) != SXMC_OK) break;
      const float ms = timed_fill();
      ms_of[pass] = ms;
    }
    g->cfg_codes = (failure == SXMC_OK && ms_of[1] < 0.97f * ms_of[0]) ? 0 : -1;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (failure != SXMC_OK) return failure;
  if (chosen_threads) *chosen_threads = best_threads;
  return group_refresh(g);
}

int sxmc_group_set_partition_teams(sxmc_group_t g, int teams) {
  SX_REQUIRE(g, "null group");
  SX_REQUIRE(teams >= 0 && teams <= 64, "teams must be 0 (default: one) to 64");
  g->cfg_teams = teams;
  return SXMC_OK;
}

int sxmc_group_set_partition(sxmc_group_t g, int mode) {
  SX_REQUIRE(g, "null group");
  SX_REQUIRE(mode >= 0 && mode <= 2, "partition mode must be 0 (auto), 1 (sliced) or 2 (interleaved)");
  g->cfg_partition = mode;
  return SXMC_OK;
}

int sxmc_group_set_sparse(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_sparse = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_prebinning(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_prebin = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_bucketing(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_bucket = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_ordering(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_order = enable == 2 ? 2 : enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_codes(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_codes = enable < 0 ? -1 : enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_codes_queue_log(sxmc_group_t g, int log2_entries) {
  SX_REQUIRE(g, "null group");
  SX_REQUIRE(log2_entries == 0 || (log2_entries >= (int)kMinQueueLog && log2_entries <= 11),
             "the queues of ambiguous rows hold 2^9 .. 2^11 entries (0: as many as fit)");
  g->cfg_queue_log = log2_entries;
  return SXMC_OK;
}

int sxmc_group_codes_info(sxmc_group_t g, int* members, unsigned long long* rows, unsigned long long* exact_rows,
                          unsigned long long* never_rows) {
  SX_REQUIRE(g && members && rows && exact_rows && never_rows, "null argument");
  int rc = group_refresh(g);
  if (rc) return rc;
  *members = 0;
  *rows = *exact_rows = *never_rows = 0;
  for (const LaunchClass& c : g->classes) {
    if (!c.codes) continue;
    for (int idx : c.member_idx) {
      const SampleStore::Bucketed* bk = g->member_bucket[(size_t)idx];
      if (!bk || !bk->d_qcol) continue;
      *members += 1;
      *rows += (unsigned long long)bk->ngranules * 256ull;
      *exact_rows += bk->q_exact_rows;
      *never_rows += bk->q_never_rows;
    }
  }
  return SXMC_OK;
}

int sxmc_group_codes_windows(sxmc_group_t g, int member, int* nfields, double* base, double* step) {
  SX_REQUIRE(g && nfields && base && step, "null argument");
  SX_REQUIRE(member >= 0 && member < (int)g->members.size(), "no such member");
  int rc = group_refresh(g);
  if (rc) return rc;
  *nfields = 0;
  for (const LaunchClass& c : g->classes) {
    if (!c.codes) continue;
    for (int idx : c.member_idx) {
      const SampleStore::Bucketed* bk = g->member_bucket[(size_t)idx];
      if (idx != member || !bk || !bk->d_qcol) continue;
      *nfields = bk->nq;
      for (int m = 0; m < bk->nq; m++) {
        base[m] = bk->qbase[m];
        step[m] = bk->qstep[m];
      }
    }
  }
  return SXMC_OK;
}

int sxmc_group_set_runtime_kernels(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_rtc = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_launch_info(sxmc_group_t g, char* out, size_t n) {
  SX_REQUIRE(g && out && n > 0, "null argument");
  int rc = group_refresh(g);
  if (rc) return rc;
  std::string text;
  for (size_t i = 0; i < g->classes.size(); i++) {
    const LaunchClass& c = g->classes[i];
    char line[512];
    const char* kind = c.shape.rtc_fill ? "runtime" : c.shape.static_prog >= 0 ? "builtin" : c.shape.nobs ? "decoded" : "generic";
    std::snprintf(line, sizeof line,
                  "launch %zu: members=%zu nobs=%d nslot=%d hist=%s program=%s table=%s%s threads=%d grid=%d partition=%d teams=%d\n",
                  i, c.member_idx.size(), c.shape.nobs, c.shape.nslot, c.shape.lds_hist ? "lds" : "global", kind,
                  c.shape.pre_width == 5 ? (c.codes ? "ordered+codes" : "ordered") : c.shape.pre_width == 3 ? "bucketed" : c.shape.pre_width ? "prebinned" : "rows",
                  c.runs_mode ? (c.shape.rtc_sparse ? "+runs(runtime)" : "+runs(builtin)") : "", c.shape.threads,
                  c.shape.grid, c.partition, c.teams);
    text += line;
  }
  if (!g->rtc_note.empty()) text += "runtime specialisation failed: " + g->rtc_note.substr(0, 300) + "\n";
  if (!g->plan_note.empty()) text += "note: " + g->plan_note + "\n";
  std::snprintf(out, n, "%s", text.c_str());
  return SXMC_OK;
}

int sxmc_group_set_lut_output(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_lut = enable ? 1 : 0;
  return SXMC_OK;
}

#if SXMC_MEASURE
// measurement build only: the kernels' hooks (fill_kernels.inc.h: SXMC_MEASURE).  RESULTS ARE WRONG with a mode set.
int sxmc_group_set_debug_mode(sxmc_group_t g, int mode) {
  SX_REQUIRE(g, "null group");
  g->debug_mode = mode;
  return SXMC_OK;
}
#endif

int sxmc_group_eval_async(sxmc_group_t g, int do_eval_pdf, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g, "null group");
  int rc = group_refresh(g);
  if (rc) return rc;
  rc = group_check_bound(g, do_eval_pdf != 0);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)s;
  g->last_stream = st;
  // lookup-only evaluation of histograms beyond LDS capacity counts just the event bins
  const bool sparse = do_eval_pdf && g->sparse_ready && g->cfg_sparse;
  rc = group_fill(g, st, sparse);
  if (rc) return rc;
  // pdfz.cpp:474-476: no lookup without evaluation points or when do_eval_pdf is false
  if (do_eval_pdf && g->max_points > 0) {
    SX_HIP(sx_launch_eval_pdf(sparse ? g->d_descs_sparse : g->d_descs, (int)g->members.size(), g->max_points, st));
  }
  return SXMC_OK;
}

int sxmc_group_eval_nll_async(sxmc_group_t g, sxmc_stream_t s, const double* d_pars, const double* d_nexpected,
                              const unsigned* d_n_mc, const short* d_source_id, const unsigned* d_norms,
                              double* d_sums, int* npartial_out) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g && d_pars && d_nexpected && d_n_mc && d_source_id && d_norms && d_sums && npartial_out,
             "null argument");
  int rc = group_refresh(g);
  if (rc) return rc;
  rc = group_check_bound(g, true);
  if (rc) return rc;
  if (!g->same_points) return fail(SXMC_ERR_STATE, "members do not share one set of evaluation points");
  hipStream_t st = (hipStream_t)s;
  g->last_stream = st;
  const bool sparse = g->sparse_ready && g->cfg_sparse;
  if (!g->cfg_lut) {
    rc = ensure_event_classes(g, sparse);  // (may upload tables: before anything is launched)
    if (rc) return rc;
  }
  rc = group_fill(g, st, sparse);
  if (rc) return rc;
  SX_REQUIRE(g->members.size() <= 1024, "too many members for the fused evaluation");
  // lookup table wanted: one pass over the events in order; otherwise over the distinct bin tuples
  unsigned long long ne = g->members[0]->npoints;
  const SxSignalDesc* descs = sparse ? g->d_descs_sparse : g->d_descs;
  const unsigned* weight = nullptr;
  if (!g->cfg_lut) {
    rc = ensure_event_classes(g, sparse);
    if (rc) return rc;
    const sxmc_group::EventClasses& ec = g->ec[sparse ? 1 : 0];
    ne = ec.K;
    descs = ec.d_descs;
    weight = ec.d_weight;
  }
  const int block = 128;
  const int grid = (int)std::min<unsigned long long>(1024, std::max<unsigned long long>(1, (ne + block - 1) / block));
  SX_HIP(sx_launch_eval_nll(descs, (int)g->members.size(), ne, weight, d_pars, d_nexpected, d_n_mc, d_source_id,
                            d_norms, d_sums, grid, block, st));
  *npartial_out = grid;
  return SXMC_OK;
}

int sxmc_group_mcmc_step_async(sxmc_group_t g, sxmc_stream_t s, const double* d_means, const double* d_sigmas,
                               sxmc_rng_state* d_rng, double* d_nll_current, double* d_nll_proposed,
                               double* d_v_current, double* d_v_proposed, int* d_accepted, int* d_counter,
                               float* d_jump_buffer, int nparameters, size_t nsources, const float* d_jump_width,
                               const double* d_nexpected, const unsigned* d_n_mc, const short* d_source_id,
                               const unsigned* d_norms, int debug_mode) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g && d_means && d_sigmas && d_rng && d_nll_current && d_nll_proposed && d_v_current && d_v_proposed &&
                 d_accepted && d_counter && d_jump_buffer && d_jump_width && d_nexpected && d_n_mc && d_source_id &&
                 d_norms,
             "null argument");
  SX_REQUIRE(nparameters > 0, "nparameters must be positive");
  int rc = group_refresh(g);
  if (rc) return rc;
  rc = group_check_bound(g, true);
  if (rc) return rc;
  if (!g->same_points) return fail(SXMC_ERR_STATE, "members do not share one set of evaluation points");
  SX_REQUIRE(g->members.size() <= 1024, "too many members for the fused step");
  hipStream_t st = (hipStream_t)s;
  g->last_stream = st;
  const bool sparse = g->sparse_ready && g->cfg_sparse;
  unsigned long long ne = g->members[0]->npoints;
  const SxSignalDesc* descs = sparse ? g->d_descs_sparse : g->d_descs;
  const unsigned* weight = nullptr;
  if (!g->cfg_lut) {
    rc = ensure_event_classes(g, sparse);
    if (rc) return rc;
    const sxmc_group::EventClasses& ec = g->ec[sparse ? 1 : 0];
    ne = ec.K;
    descs = ec.d_descs;
    weight = ec.d_weight;
  }
  rc = group_fill(g, st, sparse);  // zero (also clears the ticket) + fill
  if (rc) return rc;
  const int block = 128;
  const int grid = (int)std::min<unsigned long long>(1024, std::max<unsigned long long>(1, (ne + block - 1) / block));
  SxStepArgs a;
  a.nsignals = g->members.size();
  a.nsources = nsources;
  a.means = d_means;
  a.sigmas = d_sigmas;
  a.rng = d_rng;
  a.nll_current = d_nll_current;
  a.nll_proposed = d_nll_proposed;
  a.v_current = d_v_current;
  a.v_proposed = d_v_proposed;
  a.accepted = d_accepted;
  a.counter = d_counter;
  a.jump_buffer = d_jump_buffer;
  a.nparameters = nparameters;
  a.debug_mode = debug_mode;
  a.jump_width = d_jump_width;
  a.nexpected = d_nexpected;
  a.n_mc = d_n_mc;
  a.source_id = d_source_id;
  a.norms = d_norms;
  SX_HIP(sx_launch_eval_nll_finish(descs, (int)g->members.size(), ne, weight, g->d_step_sums, g->d_ticket, a, grid,
                                   block, st));
  return SXMC_OK;
}

int sxmc_group_finish_step_async(sxmc_group_t g, sxmc_stream_t s, size_t npartial_sums, const double* d_sums,
                                 const double* d_means, const double* d_sigmas, sxmc_rng_state* d_rng,
                                 double* d_nll_current, double* d_nll_proposed, double* d_v_current,
                                 double* d_v_proposed, int* d_accepted, int* d_counter, float* d_jump_buffer,
                                 int nparameters, size_t nsources, const float* d_jump_width,
                                 const double* d_nexpected, const unsigned* d_n_mc, const short* d_source_id,
                                 const unsigned* d_norms, int debug_mode) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g && d_sums && d_means && d_sigmas && d_rng && d_nll_current && d_nll_proposed && d_v_current &&
                 d_v_proposed && d_accepted && d_counter && d_jump_buffer && d_jump_width && d_nexpected &&
                 d_n_mc && d_source_id && d_norms,
             "null argument");
  SX_REQUIRE(nparameters > 0, "nparameters must be positive");
  if (!g->built) return fail(SXMC_ERR_STATE, "finish_step before any evaluation of the group");
  SxStepArgs a;
  a.nsignals = g->members.size();
  a.nsources = nsources;
  a.means = d_means;
  a.sigmas = d_sigmas;
  a.rng = d_rng;
  a.nll_current = d_nll_current;
  a.nll_proposed = d_nll_proposed;
  a.v_current = d_v_current;
  a.v_proposed = d_v_proposed;
  a.accepted = d_accepted;
  a.counter = d_counter;
  a.jump_buffer = d_jump_buffer;
  a.nparameters = nparameters;
  a.debug_mode = debug_mode;
  a.jump_width = d_jump_width;
  a.nexpected = d_nexpected;
  a.n_mc = d_n_mc;
  a.source_id = d_source_id;
  a.norms = d_norms;
  const bool sparse = g->last_sparse;
  SX_HIP(sx_launch_finish_zero(sparse ? g->d_descs_sparse : g->d_descs, (int)g->members.size(),
                               sparse ? g->max_bins_sparse : g->max_bins, npartial_sums, d_sums, g->d_ticket, a, 128,
                               (hipStream_t)s));
  g->prezeroed = sparse ? 2 : 1;
  for (sxmc_hist* h : g->members) {
    h->bins_valid = false;
    h->cleared_by = g;
  }
  return SXMC_OK;
}

}  // extern "C" (helpers below are internal)

namespace {
// The end of a step after the fill: lookup + event sum + finish_nll_jump_pick_combo + the clearing for the next
// evaluation -- one workgroup in one launch where that is small, two launches otherwise (see sxmc_group_step_async).
// Does the end of a step over `ne` rows take the one-workgroup form (tail_step_kernel)?  One rule, asked by the
// sequential step and by the look-ahead walk (which must partition its event sum exactly like the sequential step,
// or the two could round the NLL differently and part ways at an accept boundary).
bool step_end_takes_tail(const sxmc_group* g, bool sparse, unsigned long long ne) {
  unsigned long long words = 0;
  const std::vector<SxSignalDesc>& flavour = sparse ? g->h_descs_sparse : g->h_descs;
  for (const SxSignalDesc& d : flavour) words += (unsigned long long)d.total_nbins;
  const unsigned long long gathers = ne * g->members.size();
  return gathers <= 256ull && words <= (1ull << 16) && g->cfg_tail != 0;
}
// workgroups of 128 rows the event sum of a step is cut into (eval_nll_kernel; eval_nll2_kernel per candidate)
int step_sum_blocks(unsigned long long ne) {
  return (int)std::min<unsigned long long>(1024, std::max<unsigned long long>(1, (ne + 127) / 128));
}
// Does the step end run as ONE cooperative launch (step_end_kernel: workgroups that wait for each other inside the
// kernel)?  By default (sxmc_group_set_cooperative_step_end / SXMC_COOP_STEP_END=0 switch it off), where the whole grid
// is small enough to be resident many times over -- at most 128 workers of 128 lanes: up to 16 384 rows, BASELINE
// configs 2 and 3 with event classes -- so that several chains' step ends, each waiting for its own workgroups, always
// fit the device together.  Measured (profiles/r04_step_end_ab_*): with the partials handed over through per-worker
// slots (no fences, no counters) one kernel of 14.2 us replaces 9.5 + 6.8 us at BASELINE config 3 (+1.4 % evaluations
// per second) and 9.3 replaces 6.1 + 5.3 us at config 2 (+10 %); the first form -- release fence, arrival counter,
// acquire -- was SLOWER than the two launches (17.8 us): inside a replayed graph a kernel boundary costs 0-1.3 us.
constexpr int kCoopMaxWorkers = 128;
bool step_end_is_cooperative(const sxmc_group* g, unsigned long long ne) {
  static const int env_default = [] {
    const char* e = std::getenv("SXMC_COOP_STEP_END");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  const int on = g->cfg_coop < 0 ? env_default : g->cfg_coop;
  return on != 0 && g->cfg_tail != 0 && step_sum_blocks(ne) <= kCoopMaxWorkers;
}

// Does a step of this group run as ONE launch (fill_step_kernel)?  Only on request (sxmc_group_set_fused_step /
// SXMC_FUSED_STEP=1): built, bit-identical, and MEASURED SLOWER than the fill followed by the cooperative step end --
// config 3: 6 640-6 970 against 6 870-7 260 evals/s, config 2: 43.5 k against 51.5 k (profiles/r04_step_forms_ab_*).
// In-kernel stamps (profiles/r04_fused_step_study.log) say why: the roles are resident 10-15 us before the fill ends
// and see its end 2.0-2.3 us after the last fill workgroup has left, but then need 12.2 us for what the separate
// step_end_kernel does in 14.2 us INCLUDING its launch -- waiting for the fill's flush to be acknowledged and the
// count to travel costs what the kernel boundary costs (0.9 us gap + ~2.5 us ramp), and the look-ups are no faster
// for their tables having been fetched early.  Where offered: the step end is cooperative anyway, the plan is one
// fill launch that has the fused form (the built-in ordered programs: BASELINE config 3; the empty program over a
// pre-binned column: config 2), the vectors are short enough for the finisher's staged form, and the fill's LDS leaves
// room for the roles' own.
bool step_is_fused(const sxmc_group* g, bool sparse, unsigned long long ne, int nparameters) {
  const int on = fused_step_requested(g) ? 1 : 0;
  if (!on || sparse || g->classes.size() != 1 || (g->debug_mode & ~8)) return false;   // (8: the fused launch without its roles)
  if (step_end_takes_tail(g, sparse, ne) || !step_end_is_cooperative(g, ne)) return false;
  if (nparameters > 256 || g->members.size() > 256) return false;
  const LaunchClass& c = g->classes[0];
  if (!sx_fill_has_step_form(c.shape) || c.shape.threads < 128) return false;
  DeviceProps props;
  if (get_props(props)) return false;
  const size_t staging = 16 * sizeof(double) + ((g->members.size() + 15) / 16 * 16) * 48;   // eval_nll_block_part's
  return c.shape.lds_bytes >= staging && c.shape.lds_bytes + 16 * 1024 <= (size_t)props.lds_per_cu;
}

int group_step_tail(sxmc_group* g, hipStream_t st, bool sparse, const SxSignalDesc* descs, unsigned long long ne,
                    const unsigned* weight, const SxStepArgs& a) {
  TraceRange trace("sxmc: step end (lookup + event sum, finish_nll_jump_pick_combo + clearing)");
  // How much the end of the step has to touch decides its shape.  One workgroup doing all of it in one launch
  // saves a launch and a boundary, but a row costs S double divisions and a log: measured at BASELINE config 3
  // (~8000 rows x 12 members) one CU needs 59 us for what ~60 workgroups + the step-end launch do in 15.6 us, and
  // at config 2 (~2500 x 6) 16 us against 9, at the bench_pdfz shape (1000 x 1) 8 us against 7; config 1 (10 x 2)
  // gains 0.6 us of 14.7.  Only the smallest problems take it.
  if (step_end_takes_tail(g, sparse, ne)) {
    SX_HIP(sx_launch_tail_step(descs, (int)g->members.size(), ne, weight, a, st));
    g->last_step_launches += 1;
  } else if (step_end_is_cooperative(g, ne)) {
    // ONE launch: the event sum's workgroups + a finisher that waits for them inside the kernel (step_end_kernel)
    const int nvb = step_sum_blocks(ne);
    SX_HIP(sx_launch_step_end(descs, sparse ? g->d_descs_sparse : g->d_descs, (int)g->members.size(),
                              sparse ? g->max_bins_sparse : g->max_bins, ne, weight, g->d_coop_slots, g->d_coop_last,
                              g->d_ticket, nvb, a, st));
    g->last_step_launches += 1;
  } else {
    g->last_step_launches += 2;
    const int block = 128;
    const int grid = step_sum_blocks(ne);
    SX_HIP(sx_launch_eval_nll(descs, (int)g->members.size(), ne, weight, a.v_proposed, a.nexpected, a.n_mc, a.source_id,
                              a.norms, g->d_step_sums, grid, block, st));
    SX_HIP(sx_launch_finish_zero(sparse ? g->d_descs_sparse : g->d_descs, (int)g->members.size(),
                                 sparse ? g->max_bins_sparse : g->max_bins, (size_t)grid, g->d_step_sums, g->d_ticket, a,
                                 128, st));
  }
  g->prezeroed = sparse ? 2 : 1;
  for (sxmc_hist* h : g->members) {
    h->bins_valid = false;
    h->cleared_by = g;
  }
  return SXMC_OK;
}
}  // namespace

extern "C" {

int sxmc_group_step_async(sxmc_group_t g, sxmc_stream_t s, const double* d_means, const double* d_sigmas,
                          sxmc_rng_state* d_rng, double* d_nll_current, double* d_nll_proposed, double* d_v_current,
                          double* d_v_proposed, int* d_accepted, int* d_counter, float* d_jump_buffer, int nparameters,
                          size_t nsources, const float* d_jump_width, const double* d_nexpected, const unsigned* d_n_mc,
                          const short* d_source_id, const unsigned* d_norms, int debug_mode) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(g && d_means && d_sigmas && d_rng && d_nll_current && d_nll_proposed && d_v_current && d_v_proposed &&
                 d_accepted && d_counter && d_jump_buffer && d_jump_width && d_nexpected && d_n_mc && d_source_id &&
                 d_norms,
             "null argument");
  SX_REQUIRE(nparameters > 0, "nparameters must be positive");
  int rc = group_refresh(g);
  if (rc) return rc;
  rc = group_check_bound(g, true);
  if (rc) return rc;
  if (!g->same_points) return fail(SXMC_ERR_STATE, "members do not share one set of evaluation points");
  SX_REQUIRE(g->members.size() <= 1024, "too many members for the fused step");
  hipStream_t st = (hipStream_t)s;
  g->last_stream = st;
  const bool sparse = g->sparse_ready && g->cfg_sparse;
  unsigned long long ne = g->members[0]->npoints;
  const SxSignalDesc* descs = sparse ? g->d_descs_sparse : g->d_descs;
  const unsigned* weight = nullptr;
  if (!g->cfg_lut) {
    rc = ensure_event_classes(g, sparse);  // (may upload tables: before anything is launched)
    if (rc) return rc;
    const sxmc_group::EventClasses& ec = g->ec[sparse ? 1 : 0];
    ne = ec.K;
    descs = ec.d_descs;
    weight = ec.d_weight;
  }
  const bool zero_launched = g->prezeroed != (sparse ? 2 : 1);   // (group_fill decides the same way, plus bookkeeping)
  SxStepArgs a;
  a.nsignals = g->members.size();
  a.nsources = nsources;
  a.means = d_means;
  a.sigmas = d_sigmas;
  a.rng = d_rng;
  a.nll_current = d_nll_current;
  a.nll_proposed = d_nll_proposed;
  a.v_current = d_v_current;
  a.v_proposed = d_v_proposed;
  a.accepted = d_accepted;
  a.counter = d_counter;
  a.jump_buffer = d_jump_buffer;
  a.nparameters = nparameters;
  a.debug_mode = debug_mode;
  a.jump_width = d_jump_width;
  a.nexpected = d_nexpected;
  a.n_mc = d_n_mc;
  a.source_id = d_source_id;
  a.norms = d_norms;
  // THE WHOLE STEP IN ONE LAUNCH where the fill has that form (fill_step_kernel: the fill's workgroups, then a finisher
  // and the event sum's workers as later blocks of the same grid) -- not for a launch whose fill is being timed
  // (sxmc_group_profile: the fill alone is what the roofline figures are about, so profiled steps stay two launches)
  const bool profiled = g->prof && !t_capturing && g->prof_n < (int)g->ev0.size();
  const bool fused = !profiled && step_is_fused(g, sparse, ne, nparameters);
  if (fused) {
    LaunchClass& c = g->classes[0];
    const int nvb = step_sum_blocks(ne);
    g->h_tail.resize(sx_tail_args_bytes());   // (passed to the kernel by value: frozen at capture like every argument)
    sx_tail_args_fill(g->h_tail.data(), descs, g->d_descs, (int)g->members.size(), g->max_bins, ne, weight,
                      g->d_coop_slots, g->d_coop_last, g->d_ticket, nvb, a);
    c.shape.tail = g->h_tail.data();
    c.shape.tail_blocks = nvb + 1;
  }
  rc = group_fill(g, st, sparse);
  if (fused) {
    g->classes[0].shape.tail = nullptr;
    g->classes[0].shape.tail_blocks = 0;
  }
  if (rc) return rc;
  g->last_step_launches = (int)g->classes.size() + (zero_launched ? 1 : 0);
  if (fused) {
    g->prezeroed = 1;
    for (sxmc_hist* h : g->members) {
      h->bins_valid = false;
      h->cleared_by = g;
    }
    return SXMC_OK;
  }
  return group_step_tail(g, st, sparse, descs, ne, weight, a);
}

}  // extern "C"

struct sxmc_multigroup {
  std::vector<sxmc_group*> groups;
  std::vector<void*> fill_fn;        // per launch of the plan: the lockstep kernel (hipFunction_t)
  std::vector<size_t> lds_bytes;
  std::vector<unsigned> fill_w;          // the kernels' layout argument: words per histogram, or the ordered fill's replica layout
  std::vector<unsigned long long> seen;  // the groups' plan generations the kernels were chosen for
  std::string why_not;               // set when the chains cannot be stepped together
  bool joint_ends = true;            // the chains' step ends share two launches (sx_launch_chain_ends)
};

namespace {
// Chains can share a fill pass when their launch plans are the same plan over the same tables.
bool multigroup_prepare(sxmc_multigroup* mg) {
  const size_t C = mg->groups.size();
  sxmc_group* g0 = mg->groups[0];
  mg->why_not.clear();
  for (size_t c = 1; c < C; c++) {
    sxmc_group* g = mg->groups[c];
    if (g->members.size() != g0->members.size() || g->classes.size() != g0->classes.size()) {
      mg->why_not = "the chains' groups differ in members or launches";
      return false;
    }
    for (size_t j = 0; j < g->members.size(); j++) {
      if (g->members[j]->store != g0->members[j]->store || g->members[j]->nbins != g0->members[j]->nbins ||
          g->members[j]->lower != g0->members[j]->lower || g->members[j]->upper != g0->members[j]->upper) {
        mg->why_not = "the chains' evaluators do not share their sample tables (sxmc_hist_create_shared) or geometry";
        return false;
      }
    }
  }
  DeviceProps props;
  if (get_props(props)) return false;
  mg->fill_fn.assign(g0->classes.size(), nullptr);
  mg->lds_bytes.assign(g0->classes.size(), 0);
  mg->fill_w.assign(g0->classes.size(), 0u);
  for (size_t i = 0; i < g0->classes.size(); i++) {
    const LaunchClass& c0 = g0->classes[i];
    if (!c0.shape.lds_hist || !c0.prog_simple ||
        !(c0.shape.pre_width == 0 || c0.shape.pre_width == 3 || c0.shape.pre_width == 5) ||
        (c0.shape.nobs == 0 && c0.shape.pre_width != 5)) {
      mg->why_not = "a launch of the plan has its histogram beyond LDS, a run-time decoded program or a pre-binned column";
      return false;
    }
    for (size_t c = 1; c < C; c++) {
      const LaunchClass& cc = mg->groups[c]->classes[i];
      if (cc.shape.nobs != c0.shape.nobs || cc.shape.nslot != c0.shape.nslot || cc.shape.lds_hist != c0.shape.lds_hist ||
          cc.shape.pre_width != c0.shape.pre_width || cc.prog != c0.prog || cc.prog_simple != c0.prog_simple ||
          cc.shape.grid != c0.shape.grid || cc.shape.threads != c0.shape.threads || cc.member_idx != c0.member_idx ||
          cc.partition != c0.partition) {
        mg->why_not = "the chains' launch plans differ (systematics, launch configuration)";
        return false;
      }
    }
    size_t hist_words = c0.shape.lds_bytes / 4 - 4 - 64;
    size_t lds = (4 + C * hist_words + 64) * 4;
    if (c0.shape.pre_width == 5) {
      // ordered fill: the kernel argument is the replica layout; as many replicas as fit beside the other chains'
      // (codes: the padded form of the histograms if the chains' histograms fit that way with the smallest queues; a plan
      // with two workgroups per CU leaves each of them half the CU's LDS)
      const size_t per_cu = (size_t)std::max(1, c0.shape.grid / std::max(1, props.cus));
      const size_t lds_share = (size_t)props.lds_per_cu / std::min<size_t>(per_cu, 2);
      size_t rstride = c0.plain_rstride;
      bool padded = false;
      if (c0.codes && c0.padded_rstride &&
          (4 + C * (size_t)c0.padded_rstride + 64) * 4 + ordered_queue_bytes(kMinQueueLog) <= lds_share) {
        rstride = c0.padded_rstride;
        padded = true;
      }
      const size_t reserve = c0.codes ? ordered_queue_bytes(kMinQueueLog) : 0;
      unsigned rlog = 0;
      while (rlog < 2 && (4 + (C * rstride << (rlog + 1)) + 64) * 4 + reserve <= lds_share) rlog++;
      lds = (4 + (C * rstride << rlog) + 64) * 4;
      hist_words = rstride | ((size_t)rlog << 24) | (padded ? (size_t)1 << 27 : 0);
      if (c0.codes) {   // the queues of ambiguous rows (fill_ordered_body's CODES), shared by the chains
        const unsigned qlog = lds_share > lds ? ordered_queue_log(lds_share - lds, g0->cfg_queue_log) : 0;
        lds += ordered_queue_bytes(qlog);
        hist_words |= (size_t)qlog << 28;
      }
    }
    if (lds > (size_t)props.lds_per_cu) {
      mg->why_not = "the chains' histograms do not fit LDS together";
      return false;
    }
    SxRtcSpec k{};
    k.nobs = c0.shape.nobs;
    k.nslot = c0.shape.nslot;
    k.lds_hist = 1;
    k.pre_width = c0.shape.pre_width;
    k.nchain = (int)C;
    // (ordered tables: the kernel is compiled for the workgroup size of the plan -- 512, 768 or 1024 lanes)
    k.max_threads = c0.shape.pre_width == 5 ? (c0.shape.threads <= 512 ? 512 : c0.shape.threads <= 768 ? 768 : 1024) : 0;
    k.nops = (int)c0.prog.size();
    for (size_t q = 0; q < c0.prog.size(); q++) k.ops[q] = c0.prog[q];
    std::string err;
    mg->fill_fn[i] = sx_rtc_get(k, &err);
    if (!mg->fill_fn[i]) {
      mg->why_not = "the lockstep kernel could not be compiled: " + err.substr(0, 300);
      return false;
    }
    mg->lds_bytes[i] = lds;
    mg->fill_w[i] = (unsigned)hist_words;
  }
  return true;
}
}  // namespace

extern "C" {

int sxmc_multigroup_create(const sxmc_group_t* groups, int ngroups, sxmc_multigroup_t* out) {
  SX_REQUIRE(groups && out && ngroups >= 2 && ngroups <= 4, "a multigroup steps 2 to 4 chains together");
  for (int i = 0; i < ngroups; i++) SX_REQUIRE(groups[i], "null group");
  sxmc_multigroup* mg = new sxmc_multigroup;
  mg->groups.assign(groups, groups + ngroups);
  if (const char* e = measure_env("SXMC_JOINT_STEP_END")) mg->joint_ends = std::atoi(e) != 0;   // (A/B runs of whole programs)
  *out = mg;
  return SXMC_OK;
}

int sxmc_multigroup_destroy(sxmc_multigroup_t mg) {
  delete mg;
  return SXMC_OK;
}

int sxmc_multigroup_step_async(sxmc_multigroup_t mg, sxmc_stream_t s, const sxmc_step_args* args) {
  SX_FLUSH();
  SX_ORDER(s);
  TraceRange trace("sxmc: lockstep step (one fill pass for the set's chains + their step ends)");
  SX_REQUIRE(mg && args, "null argument");
  hipStream_t st = (hipStream_t)s;
  const size_t C = mg->groups.size();
  // every chain's own plan first (may upload: before anything is launched)
  bool replan = mg->seen.size() != C;
  for (size_t c = 0; c < C; c++) {
    sxmc_group* g = mg->groups[c];
    const sxmc_step_args& a = args[c];
    SX_REQUIRE(a.d_means && a.d_sigmas && a.d_rng && a.d_nll_current && a.d_nll_proposed && a.d_v_current &&
                   a.d_v_proposed && a.d_accepted && a.d_counter && a.d_jump_buffer && a.d_jump_width &&
                   a.d_nexpected && a.d_n_mc && a.d_source_id && a.d_norms && a.nparameters > 0,
               "null argument");
    int rc = group_refresh(g);
    if (rc) return rc;
    if (!replan && mg->seen[c] != g->plan_generation) replan = true;
    rc = group_check_bound(g, true);
    if (rc) return rc;
    if (!g->same_points) return fail(SXMC_ERR_STATE, "members do not share one set of evaluation points");
    if (!g->cfg_lut) {
      rc = ensure_event_classes(g, false);
      if (rc) return rc;
    }
    g->last_stream = st;
  }
  if (replan) {
    if (t_capturing) return fail(SXMC_ERR_STATE, "the chains' plans are out of date: step once before recording a graph");
    mg->seen.resize(C);
    for (size_t c = 0; c < C; c++) mg->seen[c] = mg->groups[c]->plan_generation;
    if (!multigroup_prepare(mg)) {
      mg->seen.clear();
      return fail(SXMC_ERR_STATE, "these chains cannot share a fill pass: " + mg->why_not);
    }
  }
  for (size_t c = 0; c < C; c++) {
    sxmc_group* g = mg->groups[c];
    const bool zero_launched = g->prezeroed != 1;
    int rc = group_prepare_fill(g, st, false);
    if (rc) return rc;
    g->last_step_launches = zero_launched ? 1 : 0;
  }
  // ONE pass over the tables for all chains
  sxmc_group* g0 = mg->groups[0];
  for (size_t i = 0; i < g0->classes.size(); i++) {
    const LaunchClass& c0 = g0->classes[i];
    if (c0.shape.grid <= 0) continue;
    SxChainDescsHost ch{};
    for (size_t c = 0; c < C; c++) ch.d[c] = mg->groups[c]->classes[i].d_descs;
    const bool rec = g0->prof && !t_capturing && g0->prof_n < (int)g0->ev0.size();
    SX_HIP(sx_rtc_launch_multi(mg->fill_fn[i], c0.shape.grid, c0.shape.threads, mg->lds_bytes[i], ch, c0.d_segs,
                               c0.d_blk_off, mg->fill_w[i], (unsigned)mg->groups[0]->debug_mode, st,
                               rec ? (void*)g0->ev0[g0->prof_n] : nullptr, rec ? (void*)g0->ev1[g0->prof_n] : nullptr));
    if (rec) g0->prof_n++;
  }
  // every chain's own step end -- in two launches for the whole set where every chain's end has the two-launch form
  // (the chains then pay the kernels' latency once, not C times), else chain by chain
  bool joint = mg->joint_ends;
  for (size_t c = 0; c < C && joint; c++) {
    const sxmc_group* g = mg->groups[c];
    joint = g->max_bins == g0->max_bins &&
            !step_end_takes_tail(g, false, g->cfg_lut ? g->members[0]->npoints : g->ec[0].K);
  }
  SxChainEnds ends{};
  for (size_t c = 0; c < C; c++) {
    sxmc_group* g = mg->groups[c];
    const sxmc_step_args& p = args[c];
    g->last_step_launches += (int)g0->classes.size();
    unsigned long long ne = g->members[0]->npoints;
    const SxSignalDesc* descs = g->d_descs;
    const unsigned* weight = nullptr;
    if (!g->cfg_lut) {
      const sxmc_group::EventClasses& ec = g->ec[0];
      ne = ec.K;
      descs = ec.d_descs;
      weight = ec.d_weight;
    }
    SxStepArgs a;
    a.nsignals = g->members.size();
    a.nsources = p.nsources;
    a.means = p.d_means;
    a.sigmas = p.d_sigmas;
    a.rng = p.d_rng;
    a.nll_current = p.d_nll_current;
    a.nll_proposed = p.d_nll_proposed;
    a.v_current = p.d_v_current;
    a.v_proposed = p.d_v_proposed;
    a.accepted = p.d_accepted;
    a.counter = p.d_counter;
    a.jump_buffer = p.d_jump_buffer;
    a.nparameters = p.nparameters;
    a.debug_mode = p.debug_mode;
    a.jump_width = p.d_jump_width;
    a.nexpected = p.d_nexpected;
    a.n_mc = p.d_n_mc;
    a.source_id = p.d_source_id;
    a.norms = p.d_norms;
    if (joint) {
      SxChainEnd& e = ends.c[c];
      e.lookup_descs = descs;
      e.hist_descs = g->d_descs;
      e.nrows = ne;
      e.weight = weight;
      e.sums = g->d_step_sums;
      e.ticket = g->d_ticket;
      e.nblocks = (unsigned)step_sum_blocks(ne);
      e.a = a;
      g->last_step_launches += 2;
      g->prezeroed = 1;
      for (sxmc_hist* h : g->members) {
        h->bins_valid = false;
        h->cleared_by = g;
      }
      continue;
    }
    int rc = group_step_tail(g, st, false, descs, ne, weight, a);
    if (rc) return rc;
  }
  if (joint) SX_HIP(sx_launch_chain_ends(ends, (int)C, (int)g0->members.size(), g0->max_bins, 128, st));
  return SXMC_OK;
}

int sxmc_multigroup_set_joint_step_end(sxmc_multigroup_t mg, int enable) {
  SX_REQUIRE(mg, "null multigroup");
  mg->joint_ends = enable != 0;
  return SXMC_OK;
}

// The look-ahead walk's pass (see finish2_zero_kernel): groups[0] evaluates the step's proposal (its evaluators are
// bound to a->d_v_proposed / a->d_norms), groups[1] the look-ahead vector (bound to d_v_lookahead / d_norms_lookahead).
int sxmc_multigroup_lookahead_step_async(sxmc_multigroup_t mg, sxmc_stream_t s, const sxmc_step_args* a,
                                         double* d_v_lookahead, const unsigned* d_norms_lookahead, const int* d_cap) {
  SX_FLUSH();
  SX_ORDER(s);
  TraceRange trace("sxmc: look-ahead pass (two evaluations, one or two steps)");
  SX_REQUIRE(mg && a && d_v_lookahead && d_norms_lookahead, "null argument");
  SX_REQUIRE(mg->groups.size() == 2, "the look-ahead walk steps exactly two groups: the proposal's and the look-ahead's");
  SX_REQUIRE(a->d_means && a->d_sigmas && a->d_rng && a->d_nll_current && a->d_nll_proposed && a->d_v_current &&
                 a->d_v_proposed && a->d_accepted && a->d_counter && a->d_jump_buffer && a->d_jump_width &&
                 a->d_nexpected && a->d_n_mc && a->d_source_id && a->d_norms && a->nparameters > 0,
             "null argument");
  SX_REQUIRE(a->nparameters <= 256, "the look-ahead walk stages its vectors in LDS: at most 256 parameters");
  hipStream_t st = (hipStream_t)s;
  bool replan = mg->seen.size() != 2;
  for (size_t c = 0; c < 2; c++) {
    sxmc_group* g = mg->groups[c];
    int rc = group_refresh(g);
    if (rc) return rc;
    if (!replan && mg->seen[c] != g->plan_generation) replan = true;
    rc = group_check_bound(g, true);
    if (rc) return rc;
    if (!g->same_points) return fail(SXMC_ERR_STATE, "members do not share one set of evaluation points");
    if (g->cfg_lut) return fail(SXMC_ERR_STATE, "the look-ahead walk sums over event classes: switch the lookup table off");
    if (g->sparse_ready && g->cfg_sparse) return fail(SXMC_ERR_STATE, "the look-ahead walk needs histograms that fit LDS");
    rc = ensure_event_classes(g, false);
    if (rc) return rc;
    g->last_stream = st;
  }
  sxmc_group *ga = mg->groups[0], *gb = mg->groups[1];
  SX_REQUIRE(ga->members.size() == gb->members.size() && ga->members.size() <= 1024 && ga->ec[0].K == gb->ec[0].K,
             "the two groups must hold the same members over the same data");
  if (step_end_takes_tail(ga, false, ga->ec[0].K)) {
    return fail(SXMC_ERR_STATE,
                "the look-ahead walk is not offered for this shape: the sequential step ends in the one-workgroup form "
                "(at most 256 look-ups), whose event sum is partitioned differently -- walk sequentially "
                "(sxmc_group_lookahead_supported says so beforehand)");
  }
  if (replan) {
    if (t_capturing) return fail(SXMC_ERR_STATE, "the chains' plans are out of date: step once before recording a graph");
    mg->seen.resize(2);
    for (size_t c = 0; c < 2; c++) mg->seen[c] = mg->groups[c]->plan_generation;
    if (!multigroup_prepare(mg)) {
      mg->seen.clear();
      return fail(SXMC_ERR_STATE, "these groups cannot share a fill pass: " + mg->why_not);
    }
  }
  for (size_t c = 0; c < 2; c++) {
    sxmc_group* g = mg->groups[c];
    const bool zero_launched = g->prezeroed != 1;
    int rc = group_prepare_fill(g, st, false);
    if (rc) return rc;
    g->last_step_launches = zero_launched ? 1 : 0;
  }
  for (size_t i = 0; i < ga->classes.size(); i++) {
    const LaunchClass& c0 = ga->classes[i];
    if (c0.shape.grid <= 0) continue;
    SxChainDescsHost ch{};
    ch.d[0] = ga->classes[i].d_descs;
    ch.d[1] = gb->classes[i].d_descs;
    const bool rec = ga->prof && !t_capturing && ga->prof_n < (int)ga->ev0.size();
    SX_HIP(sx_rtc_launch_multi(mg->fill_fn[i], c0.shape.grid, c0.shape.threads, mg->lds_bytes[i], ch, c0.d_segs,
                               c0.d_blk_off, mg->fill_w[i], (unsigned)mg->groups[0]->debug_mode, st,
                               rec ? (void*)ga->ev0[ga->prof_n] : nullptr, rec ? (void*)ga->ev1[ga->prof_n] : nullptr));
    if (rec) ga->prof_n++;
  }
  const sxmc_group::EventClasses &ea = ga->ec[0], &eb = gb->ec[0];
  const unsigned long long ne = ea.K;
  const int block = 128;
  // each candidate's event sum is cut exactly like the sequential step's (group_step_tail): the same blocks of 128
  // rows, the same cap, so the partial sums and their reduction round identically
  const int half = step_sum_blocks(ne);
  SxStepArgs k;
  k.nsignals = ga->members.size();
  k.nsources = a->nsources;
  k.means = a->d_means;
  k.sigmas = a->d_sigmas;
  k.rng = a->d_rng;
  k.nll_current = a->d_nll_current;
  k.nll_proposed = a->d_nll_proposed;
  k.v_current = a->d_v_current;
  k.v_proposed = a->d_v_proposed;
  k.accepted = a->d_accepted;
  k.counter = a->d_counter;
  k.jump_buffer = a->d_jump_buffer;
  k.nparameters = a->nparameters;
  k.debug_mode = a->debug_mode;
  k.jump_width = a->d_jump_width;
  k.nexpected = a->d_nexpected;
  k.n_mc = a->d_n_mc;
  k.source_id = a->d_source_id;
  k.norms = a->d_norms;
  // the pass's step end: ONE cooperative launch where both candidates' event sums fit 128 lanes of a finisher
  // (step_end2_kernel; the same switch as the sequential step's), else lookup + event sums, then step end + clearing
  const bool coop = 2 * half <= kCoopMaxWorkers && step_end_is_cooperative(ga, ne);
  if (coop) {
    SX_HIP(sx_launch_step_end2(ea.d_descs, eb.d_descs, ga->d_descs, gb->d_descs, (int)ga->members.size(),
                               std::max(ga->max_bins, gb->max_bins), ne, ea.d_weight, eb.d_weight, ga->d_coop_slots,
                               ga->d_coop_last, ga->d_ticket, half, d_norms_lookahead, d_v_lookahead, d_cap, k, st));
  } else {
    SX_HIP(sx_launch_eval_nll2(ea.d_descs, eb.d_descs, (int)ga->members.size(), ne, ea.d_weight, eb.d_weight,
                               a->d_v_proposed, d_v_lookahead, a->d_nexpected, a->d_n_mc, a->d_source_id, a->d_norms,
                               d_norms_lookahead, ga->d_step_sums, gb->d_step_sums, half, block, st));
    SX_HIP(sx_launch_finish2_zero(ga->d_descs, gb->d_descs, (int)ga->members.size(), std::max(ga->max_bins, gb->max_bins),
                                  (size_t)half, ga->d_step_sums, gb->d_step_sums, d_norms_lookahead, d_v_lookahead, d_cap, k,
                                  128, st));
  }
  for (size_t c = 0; c < 2; c++) {
    sxmc_group* g = mg->groups[c];
    g->last_step_launches += (int)ga->classes.size() + (coop ? 1 : 2);
    g->prezeroed = 1;
    for (sxmc_hist* h : g->members) {
      h->bins_valid = false;
      h->cleared_by = g;
    }
  }
  return SXMC_OK;
}

int sxmc_group_lookahead_supported(sxmc_group_t g, int* ok) {
  SX_REQUIRE(g && ok, "null argument");
  *ok = 0;
  int rc = group_refresh(g);
  if (rc) return rc;
  if (!g->same_points || g->cfg_lut || (g->sparse_ready && g->cfg_sparse) || g->members.empty()) return SXMC_OK;
  if (!g->members[0]->has_points) return SXMC_OK;
  rc = ensure_event_classes(g, false);
  if (rc) return rc;
  if (step_end_takes_tail(g, false, g->ec[0].K)) return SXMC_OK;
  // the pass keeps TWO histograms per member in LDS (the proposal's and the look-ahead's): what multigroup_prepare
  // will require of every launch of the plan
  DeviceProps props;
  if (get_props(props)) return SXMC_OK;
  for (const LaunchClass& c : g->classes) {
    if (!c.shape.lds_hist || !c.prog_simple ||
        !(c.shape.pre_width == 0 || c.shape.pre_width == 3 || c.shape.pre_width == 5) ||
        (c.shape.nobs == 0 && c.shape.pre_width != 5)) {
      return SXMC_OK;
    }
    const size_t words = c.shape.pre_width == 5 ? (size_t)(c.shape.lds_layout & 0xFFFFFFu) : c.shape.lds_bytes / 4 - 4 - 64;
    if ((4 + 2 * words + 64) * 4 > (size_t)props.lds_per_cu) return SXMC_OK;
  }
  *ok = 1;
  return SXMC_OK;
}

// The first look-ahead vector of a walk: what the step after the pending one would propose if the pending one
// were rejected (current vector + jump width x the deviates that step end will draw; generators untouched).
int sxmc_lookahead_begin(sxmc_stream_t s, int nparameters, const sxmc_rng_state* d_rng, const float* d_jump_width,
                         const double* d_v_current, double* d_v_lookahead) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(d_rng && d_jump_width && d_v_current && d_v_lookahead && nparameters > 0, "null argument");
  SX_HIP(sx_launch_peek_next_proposal(nparameters, d_rng, d_jump_width, d_v_current, d_v_lookahead, (hipStream_t)s));
  return SXMC_OK;
}

int sxmc_group_last_step_launches(sxmc_group_t g, int* launches) {
  SX_REQUIRE(g && launches, "null argument");
  *launches = g->last_step_launches;
  return SXMC_OK;
}

int sxmc_group_set_tail_kernel(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_tail = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_cooperative_step_end(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_coop = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_set_fused_step(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_fused = enable ? 1 : 0;
  return SXMC_OK;
}

int sxmc_group_step_end_timeouts(sxmc_group_t g, sxmc_stream_t s, unsigned* timeouts) {
  SX_REQUIRE(g && timeouts, "null argument");
  SX_FLUSH();
  if (s) {
    // on the chain's own stream: a copy through the legacy stream is what the runtime refuses while ANOTHER host
    // thread records a graph on a blocking stream (see sxmc_graph_begin_capture)
    SX_HIP(hipMemcpyAsync(timeouts, g->d_ticket + 6, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)s));
    SX_HIP(hipStreamSynchronize((hipStream_t)s));
  } else {
    SX_HIP(hipMemcpy(timeouts, g->d_ticket + 6, sizeof(unsigned), hipMemcpyDeviceToHost));
  }
  return SXMC_OK;
}

int sxmc_rtc_compile_check(int nobs, int nslot, int lds_hist, int pre_width, int sparse_runs, const unsigned* ops,
                           int nops, size_t* code_bytes) {
  SX_REQUIRE(nops >= 0 && nops <= SXMC_MAX_SYST && (ops || nops == 0), "bad program");
  SxRtcSpec k{};
  k.nobs = nobs;
  k.nslot = nslot;
  k.lds_hist = lds_hist;
  k.pre_width = pre_width;
  k.sparse_runs = sparse_runs;
  k.nops = nops;
  for (int i = 0; i < nops; i++) k.ops[i] = ops[i];
  std::string err;
  if (!sx_rtc_compile_only(k, code_bytes, &err)) return fail(SXMC_ERR_HIP, err);
  return SXMC_OK;
}

int sxmc_rtc_compile_check_lockstep(int nobs, int nslot, int pre_width, int nchains, const unsigned* ops, int nops,
                                    size_t* code_bytes) {
  SX_REQUIRE(nops >= 0 && nops <= SXMC_MAX_SYST && (ops || nops == 0) && nchains >= 2 && nchains <= 4, "bad program");
  SxRtcSpec k{};
  k.nobs = nobs;
  k.nslot = nslot;
  k.lds_hist = 1;
  k.pre_width = pre_width;
  k.nchain = nchains;
  k.nops = nops;
  for (int i = 0; i < nops; i++) k.ops[i] = ops[i];
  std::string err;
  if (!sx_rtc_compile_only(k, code_bytes, &err)) return fail(SXMC_ERR_HIP, err);
  return SXMC_OK;
}

int sxmc_group_synchronize(sxmc_group_t g) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_REQUIRE(g, "null group");
  SX_HIP(hipStreamSynchronize(g->last_stream));
  return SXMC_OK;
}

int sxmc_group_profile(sxmc_group_t g, int enable, int capacity) {
  SX_REQUIRE(g, "null group");
  g->prof = enable != 0;
  g->prof_n = 0;
  if (g->prof) {
    if (capacity < 1) capacity = 1;
    while ((int)g->ev0.size() < capacity) {
      hipEvent_t a, b;
      SX_HIP(hipEventCreate(&a));
      SX_HIP(hipEventCreate(&b));
      g->ev0.push_back(a);
      g->ev1.push_back(b);
    }
  }
  return SXMC_OK;
}

int sxmc_group_profile_read(sxmc_group_t g, double* fill_ms_total, int* nlaunches) {
  SX_REQUIRE(g && fill_ms_total && nlaunches, "null argument");
  double tot = 0;
  for (int i = 0; i < g->prof_n; i++) {
    SX_HIP(hipEventSynchronize(g->ev1[i]));
    float ms = 0;
    SX_HIP(hipEventElapsedTime(&ms, g->ev0[i], g->ev1[i]));
    tot += ms;
  }
  *fill_ms_total = tot;
  *nlaunches = g->prof_n;
  return SXMC_OK;
}

int sxmc_group_algorithmic_bytes(sxmc_group_t g, double* fill_read, double* hist, double* event) {
  SX_REQUIRE(g && fill_read && hist && event, "null argument");
  int rc = group_refresh(g);
  if (rc) return rc;
  double fr = 0, hb = 0, ev = 0;
  for (size_t i = 0; i < g->members.size(); i++) {
    const sxmc_hist* h = g->members[i];
    const SxSignalDesc& d = g->h_descs[i];
    // columns the fill streams: float slots minus the observables covered by the pre-binned column
    int pre_w = 0, pre_dims = 0;
    bool codes = false;
    for (const LaunchClass& c : g->classes) {
      for (int idx : c.member_idx) {
        if (idx == (int)i && c.shape.pre_width) {
          pre_w = c.shape.pre_width;
          for (int k = 0; k < d.nobs; k++) pre_dims += (c.pre_mask >> k) & 1u;
        }
        if (idx == (int)i) codes = c.codes;
      }
    }
    if (const SampleStore::Bucketed* bk = i < g->member_bucket.size() ? g->member_bucket[i] : nullptr) {
      // bucketed table: the columns that change, for the samples inside the domain of the untouched observables,
      // + one word per granule
      // (ordered: that observable's column is needed only in the granules that straddle a bin edge -- which ones
      // depends on the parameters; not counted -- and each granule has two end values besides its word)
      // (codes: the streamed fields at 16 bits each, two to a word; the float values of the ambiguous rows -- a few
      // in 10^4, which ones depends on the parameters -- are not counted either)
      const bool ord = bk->sort && bk->sort->ordered >= 0;
      const double row_bytes = (codes && bk->d_qcol) ? 4.0 * (double)((bk->nq + 1) / 2)
                                                     : 4.0 * (double)(bk->fields.size() - (ord ? 1 : 0));
      fr += (double)bk->nkept * row_bytes + ((ord ? 8.0 : 0.0) + (bk->runs > 1 ? 8.0 : 4.0)) * (double)bk->ngranules;
    } else {
      fr += (double)h->nsamples * (4.0 * (d.nslot - pre_dims) + pre_w);
    }
    if (h->total_nbins <= kLdsMaxBins) {
      hb += 4.0 * (double)h->total_nbins;                       // LDS-private, flushed once
    } else if (g->sparse_ready && g->cfg_sparse) {
      hb += 2.0 * 4.0 * (double)std::max(h->ntargets, 1);      // event-bin counters: zero + update
    } else {
      hb += 2.0 * 4.0 * (double)h->total_nbins;                 // HBM-resident histogram: zero + update
    }
    ev += 16.0 * (double)d.npoints;
  }
  *fill_read = fr;
  *hist = hb;
  *event = ev;
  return SXMC_OK;
}

// ------------------------------------------------------------------------------ NLL launch points
static int check_launch(int grid, int block) {
  if (grid < 1 || block < 1 || block > 1024) return fail(SXMC_ERR_INVALID, "bad launch shape");
  return SXMC_OK;
}

int sxmc_launch_init_device_rngs(int grid, int block, sxmc_stream_t s, int nthreads, unsigned long long seed,
                                 sxmc_rng_state* d_state) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(d_state && (long long)grid * block >= nthreads, "init_device_rngs: grid*block < nthreads");
  SX_HIP(sx_nll_init_rngs(grid, block, (hipStream_t)s, nthreads, seed, d_state));
  return SXMC_OK;
}

int sxmc_launch_pick_new_vector(int grid, int block, sxmc_stream_t s, int nthreads, sxmc_rng_state* d_rng,
                                const float* d_jump_width, const double* d_current_vector,
                                double* d_proposed_vector) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_pick_new_vector(grid, block, (hipStream_t)s, nthreads, d_rng, d_jump_width, d_current_vector,
                                d_proposed_vector));
  return SXMC_OK;
}

int sxmc_launch_jump_decider(int grid, int block, sxmc_stream_t s, sxmc_rng_state* d_rng, double* d_nll_current,
                             const double* d_nll_proposed, double* d_v_current, const double* d_v_proposed,
                             unsigned nparameters, int* d_accepted, int* d_counter, float* d_jump_buffer) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_jump_decider(grid, block, (hipStream_t)s, d_rng, d_nll_current, d_nll_proposed, d_v_current,
                             d_v_proposed, nparameters, d_accepted, d_counter, d_jump_buffer));
  return SXMC_OK;
}

int sxmc_launch_nll_event_chunks(int grid, int block, sxmc_stream_t s, const float* d_lut, const double* d_pars,
                                 size_t ne, size_t ns, const double* d_nexpected, const unsigned* d_n_mc,
                                 const short* d_source_id, const unsigned* d_norms, double* d_sums) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(ns <= 4096, "too many signals");
  SX_HIP(sx_nll_event_chunks(grid, block, (hipStream_t)s, d_lut, d_pars, ne, ns, d_nexpected, d_n_mc, d_source_id,
                             d_norms, d_sums));
  return SXMC_OK;
}

int sxmc_launch_nll_event_reduce(int grid, int block, sxmc_stream_t s, size_t nthreads, const double* d_sums,
                                 double* d_total_sum) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(grid == 1, "nll_event_reduce runs in one workgroup");
  SX_HIP(sx_nll_event_reduce(block, (hipStream_t)s, nthreads, d_sums, d_total_sum));
  return SXMC_OK;
}

int sxmc_launch_nll_total(int grid, int block, sxmc_stream_t s, size_t nparameters, const double* d_pars,
                          size_t nsignals, size_t nsources, const double* d_means, const double* d_sigmas,
                          const double* d_events_total, const double* d_nexpected, const unsigned* d_n_mc,
                          const short* d_source_id, const unsigned* d_norms, double* d_nll) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_total((hipStream_t)s, nparameters, d_pars, nsignals, nsources, d_means, d_sigmas, d_events_total,
                      d_nexpected, d_n_mc, d_source_id, d_norms, d_nll));
  return SXMC_OK;
}

int sxmc_launch_finish_nll_jump_pick_combo(int grid, int block, sxmc_stream_t s, size_t npartial_sums,
                                           const double* d_sums, size_t nsignals, size_t nsources,
                                           const double* d_means, const double* d_sigmas, sxmc_rng_state* d_rng,
                                           double* d_nll_current, double* d_nll_proposed, double* d_v_current,
                                           double* d_v_proposed, int* d_accepted, int* d_counter,
                                           float* d_jump_buffer, int nparameters, const float* d_jump_width,
                                           const double* d_nexpected, const unsigned* d_n_mc,
                                           const short* d_source_id, const unsigned* d_norms, int debug_mode) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(grid == 1, "finish_nll_jump_pick_combo runs in one workgroup");
  SX_HIP(sx_nll_finish_combo(block, (hipStream_t)s, npartial_sums, d_sums, nsignals, nsources, d_means, d_sigmas,
                             d_rng, d_nll_current, d_nll_proposed, d_v_current, d_v_proposed, d_accepted,
                             d_counter, d_jump_buffer, nparameters, d_jump_width, d_nexpected, d_n_mc, d_source_id,
                             d_norms, debug_mode != 0));
  return SXMC_OK;
}

#if SXMC_MEASURE
// (measurement build only) test hook: d_out[k] = d_x[k]^i, formed as the polynomial systematics form it
int sxmc_debug_pow_int(const double* d_x, int n, int i, double* d_out) {
  SX_FLUSH();
  SX_REQUIRE(d_x && d_out && n >= 0 && i >= 0 && i < 64, "bad arguments");
  SX_HIP(sx_launch_pow_int(d_x, n, i, d_out, nullptr));
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}

// (measurement build only) test hook: raw Philox output of state[0] (advances it by ndraws)
int sxmc_debug_philox_dump(sxmc_rng_state* d_state, unsigned* d_out, int ndraws) {
  SX_FLUSH();
  SX_HIP(sx_nll_philox_dump(nullptr, d_state, d_out, ndraws));
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}
#endif

}  // extern "C"
