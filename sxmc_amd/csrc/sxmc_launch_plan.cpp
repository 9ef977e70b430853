// sxmc_launch_plan.cpp -- a group's launch plan: device-resident tables (bucketed copies, codes, sparse structures), launch classes,
// partitions, launch shapes; refresh when bindings or settings change; event classes; the fill launches.
#include "sxmc_host.h"

using namespace sxhost;

namespace sxhost {

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

// strata of x - t inside a bucket of a boxed table (SXMC_BOX_STRATA, measurement build: 1 .. 16).  More strata make the
// boxes narrower in x - t and wider in x: at BASELINE config 3 a CPU model of the layout puts the granules whose box
// straddles an edge at 0.8 / 1.5 / 2.9 % for 1 / 2 / 4 strata at a resolution parameter of 0, 4 / 3.2 / 3.6 % at 0.01 and
// 17 / 10 / 7.8 % at 0.05 (its prior width).  Measured on the walk of the bench (one box, alternating,
// profiles/r05_boxed_strata.log): 65.2-65.9 us for 1 or 2 strata, 67.3 for 4, 71.4 for 8.  Two.
int box_strata() {
  static const int n = [] {
    const char* e = measure_env("SXMC_BOX_STRATA");
    return e ? std::min(std::max(std::atoi(e), 1), 16) : 2;
  }();
  return n;
}

// The rows of member `h`'s table sorted by their bin indices in the observables of `mask` (those no systematic
// writes) and cut into 256-row granules, bucket by bucket.  Fetched from the table's cache or built; *out =
// nullptr when bucketing does not pay for this table.  d_full_desc: the member's descriptor on the device.
// `ordered` >= 0: inside every bucket the rows are in ascending order of that observable's raw value (NaN last).
int get_bucket_sort(sxmc_hist* h, const SxSignalDesc* d_full_desc, unsigned mask, int ordered,
                    const SampleStore::BucketSort** out, int box_truth = -1) {
  *out = nullptr;
  SampleStore& st = *h->store;
  std::lock_guard<std::mutex> lock(st.pre_mutex);
  if (SampleStore::BucketSort* have = st.find_sort(mask, ordered, box_truth)) {
    *out = have->rejected ? nullptr : have;
    return SXMC_OK;
  }
  st.sorts.push_back(std::make_unique<SampleStore::BucketSort>());
  SampleStore::BucketSort* b = st.sorts.back().get();
  b->mask = mask;
  b->ordered = ordered;
  b->box_truth = box_truth;
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
    if (box_truth >= 0) {
      // BOXED observable: strata of x - t (their boundaries: quantiles, read off a first sort), x ascending inside
      const float* colx = st.d_cols + (size_t)ordered * h->pitch;
      const float* colt = st.d_cols + (size_t)box_truth * h->pitch;
      const int nstrata = box_strata();
      unsigned bounds[16] = {0};
      if (nstrata > 1) {
        SX_HIP(sx_box_keys(colx, colt, n, 1, 1, nullptr, keys0.as<unsigned>(), rows0.as<unsigned>(), nullptr));
        SX_HIP(sx_bucket_sort(keys0.as<unsigned>(), keys1.as<unsigned>(), rows0.as<unsigned>(), rows1.as<unsigned>(), n, 32,
                              nullptr));
        for (int k = 0; k + 1 < nstrata; k++) {
          SX_HIP(hipMemcpy(&bounds[k], keys1.as<unsigned>() + (size_t)((double)n * (k + 1) / nstrata), 4,
                           hipMemcpyDeviceToHost));
        }
      }
      SX_HIP(sx_box_keys(colx, colt, n, 2, nstrata, bounds, keys0.as<unsigned>(), rows0.as<unsigned>(), nullptr));
    } else {
      SX_HIP(sx_order_keys(st.d_cols + (size_t)ordered * h->pitch, n, keys0.as<unsigned>(),
                           rows0.as<unsigned>(), nullptr));
    }
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
  if (bs->ordered >= 0 && bs->box_truth >= 0) {
    // the boxed observable's column is the last of `fields`, its truth field the one before (group_rebuild)
    SX_REQUIRE(fields.size() >= 2, "a boxed layout streams its observable and the truth field");
    SX_HIP(hipMalloc((void**)&b->d_gbox, sizeof(float) * 4 * A));
    SX_HIP(hipMemset(b->d_gbox, 0xFF, sizeof(float) * 4 * A));   // (NaN: a granule never written is left to the float columns)
    SX_HIP(sx_bucket_boxes(b->d_cols + (fields.size() - 1) * b->pitch, b->d_cols + (fields.size() - 2) * b->pitch,
                           dvalid.as<unsigned>(), P, b->d_gbox, nullptr));
    {   // mean extents of the finite boxes (what a dual launch estimates the share of straddling granules from)
      std::vector<float> hb(4 * P);
      if (P) SX_HIP(hipMemcpy(hb.data(), b->d_gbox, sizeof(float) * 4 * P, hipMemcpyDeviceToHost));
      // (the mean of the lower nine tenths: the granules of a bucket's thin tails are wide whatever the layout and are
      // few; they straddle an edge at any parameters and say nothing about the rest)
      std::vector<double> ex, et;
      for (size_t p = 0; p < P; p++) {
        const float x0 = hb[4 * p], x1 = hb[4 * p + 1], t0 = hb[4 * p + 2], t1 = hb[4 * p + 3];
        if (!(std::isfinite(x0) && std::isfinite(x1) && std::isfinite(t0) && std::isfinite(t1))) continue;
        ex.push_back((double)x1 - (double)x0);
        et.push_back((double)t1 - (double)t0);
      }
      auto trimmed = [](std::vector<double>& v) {
        if (v.empty()) return 0.0;
        std::sort(v.begin(), v.end());
        const size_t keep = std::max<size_t>(1, v.size() * 9 / 10);
        double sum = 0;
        for (size_t i = 0; i < keep; i++) sum += v[i];
        return sum / (double)keep;
      };
      b->box_dx = trimmed(ex);
      b->box_dt = trimmed(et);
    }
  } else if (bs->ordered >= 0) {
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
  const bool boxed = b->sort && b->sort->ordered >= 0 && b->sort->box_truth >= 0;
  const int nq = (int)b->fields.size() - ((b->sort && b->sort->ordered >= 0) ? 1 : 0) - (boxed ? 1 : 0);
  // (below 2^22 granules a unit's byte offset into a column of codes fits 32 bits, and a unit number 28: what the
  // ordered kernel's addressing and its queue entries assume)
  if ((boxed ? nq != 1 : nq < 2) || nq > SXMC_MAX_QSLOTS || b->ngranules == 0 || b->ngranules >= ((size_t)1 << 22) || !b->sort) {
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
  static const bool gate_lifted = [] {   // (SXMC_CODES_GATE=1, measurement build: profiles/r04b_codes_gate_probe.log)
    const char* e = measure_env("SXMC_CODES_GATE");
    return e && e[0] == '1';
  }();
  if (!(ambiguous <= 2e-3) && !gate_lifted) return SXMC_OK;
  // (the codes are an extra: a table they do not fit beside -- +4 bytes per row and pair of fields -- keeps its float stream)
  if (hipMalloc((void**)&b->d_qcol, boxed ? sizeof(unsigned short) * b->pitch
                                          : sizeof(unsigned) * b->pitch * (size_t)((nq + 1) / 2)) != hipSuccess) {
    (void)hipGetLastError();
    b->d_qcol = nullptr;
    return SXMC_OK;
  }
  unsigned long long tally[2] = {0, 0};
  b->q16 = boxed;
  hipError_t e = boxed ? sx_column_codes16(b->d_cols, b->qbase[0], b->qstep[0], n, reinterpret_cast<unsigned short*>(b->d_qcol),
                                           tally, nullptr)
                       : sx_column_codes(b->d_cols, b->pitch, nq, b->qbase, b->qstep, n, b->d_qcol, tally, nullptr);
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
unsigned ordered_queue_log(size_t room, int cap) {
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
      if (prew == 5 || prew == 6) {   // (sp: index into the ordered / boxed programs built in: histograms in LDS, no runs)
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
    int box_obs_of = -1, box_truth_of = -1;
    // box_truth >= 0: `ordered` is a BOXED observable (fill_boxed_kernel) and box_truth the slot of its truth field
    auto try_bucket = [&](int ordered, int box_truth = -1) -> int {
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
      const bool shape_ok = box_truth >= 0 ? (nobs2 == 1 && fields.size() == 3 && lds_hist &&
                                              fields[1] == d.slot_col[box_truth])
                            : ordered >= 0 ? (nobs2 <= 5 && fields.size() <= 7 && (lds_hist || nobs2 >= 1))
                                           : (mask && nobs2 >= 1 && sx_fill_has_specialization(nobs2, (int)fields.size()));
      if (!shape_ok) return SXMC_OK;
      SxSignalDesc cd;
      compact_desc(d, keep, nobs2, cd, ordered);
      const std::vector<unsigned> prog2 = prog_words(cd);
      const int prew = box_truth >= 0 ? 6 : ordered >= 0 ? 5 : 3;
      const int sp = box_truth >= 0 ? sx_fill_find_boxed_program(cd.nobs, cd.nslot, (int)prog2.size(), prog2.data())
                     : ordered >= 0 ? sx_fill_find_ordered_program(cd.nobs, cd.nslot, (int)prog2.size(), prog2.data())
                                    : sx_fill_find_static_program(cd.nobs, cd.nslot, (int)prog2.size(), prog2.data());
      if (!have_kernel(cd.nobs, cd.nslot, prew, 0, prog2, sp, &rtc_fill)) return SXMC_OK;
      const SampleStore::BucketSort* bs = nullptr;
      int rc2 = get_bucket_sort(h, g->d_descs + i, mask, ordered, &bs,
                                box_truth >= 0 ? d.slot_col[box_truth] : -1);
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
      runs_mode = !lds_hist && narrow && h->has_points && h->d_table && box_truth < 0 &&
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
      box_obs_of = box_truth >= 0 ? ordered : -1;
      box_truth_of = box_truth;
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
      // The BOXED observable (fill_boxed_kernel): written by one-coefficient shift / scale / cos-theta scale and at least
      // one resolution scale, all of those against ONE field that nothing writes, and read by nothing; beside it exactly one
      // other written observable, with one-coefficient shift / scale / cos-theta scale only (one streamed field, as 16-bit
      // codes).  Histogram in LDS, codes on.  Where the ordered form also applies this one streams half the bytes.
      int boxed = -1, box_truth = -1;
      if (g->cfg_box != 0 && !g->box_blocked && g->cfg_order && lds_hist && h->total_nbins < (1 << 22) && codes_enabled(g)) {
        int nwritten = 0;
        bool others_ok = true;
        for (int k = 0; k < d.nobs; k++) {
          bool written = false, ok = true, has_res = false;
          int truth = -1;
          for (int q = 0; q < d.nsyst; q++) {
            const SxSystOp& op = d.syst[q];
            if (op.obs_slot == k) {
              written = true;
              ok = ok && op.npars == 1;
              if (op.type == SXMC_SYST_RESOLUTION_SCALE) {
                has_res = true;
                ok = ok && (truth < 0 || truth == op.extra_slot) && op.extra_slot != k;
                truth = op.extra_slot;
              }
            }
            if (op.type == SXMC_SYST_RESOLUTION_SCALE && op.extra_slot == k) ok = false;   // (read by a systematic)
          }
          if (!written) continue;
          nwritten++;
          if (has_res && ok && boxed < 0) {
            boxed = k;
            box_truth = truth;
          } else {
            others_ok = others_ok && ok && !has_res;
          }
        }
        bool truth_written = false;
        for (int q = 0; q < d.nsyst && box_truth >= 0; q++) truth_written = truth_written || d.syst[q].obs_slot == box_truth;
        if (boxed < 0 || nwritten != 2 || !others_ok || truth_written || d.nsyst > 8) boxed = -1;
        // Does it pay?  A box straddles an edge of the observable when it is not small against a bin: with fewer than a
        // few granules per bin, stratum and bucket most do.  cfg_box == 1 (tests): wherever it applies.
        if (boxed >= 0 && g->cfg_box < 0) {
          double buckets = 1.0;
          for (int k = 0; k < d.nobs; k++) {
            bool written = false;
            for (int q = 0; q < d.nsyst; q++) written = written || d.syst[q].obs_slot == k;
            if (!written) buckets *= (double)h->nbins[(size_t)k];
          }
          const double per = (double)h->nsamples / 256.0 / (buckets * box_strata());
          if (per < 4.0 * ((double)h->nbins[(size_t)boxed] + 1.0)) boxed = -1;
        }
      }
      if (boxed >= 0) {
        rc = try_bucket(boxed, box_truth);
        if (rc) return rc;
      }
      if (!bucketed && ordered >= 0) {
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
      cls->box_obs = pre_width == 6 ? box_obs_of : -1;
      cls->box_truth = pre_width == 6 ? box_truth_of : -1;
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
    const bool boxed = c.shape.pre_width == 6;
    const bool bucketed = c.shape.pre_width == 3 || c.shape.pre_width == 5 || boxed;
    const bool ordered = c.shape.pre_width == 5 || boxed;   // (the LDS layout, shapes and partition of the ordered form)
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
        d.boxes = bk->d_gbox;
        if (boxed) {   // (launch-wide: the largest of the members' mean extents)
          c.box_dx = std::max(c.box_dx, (float)bk->box_dx);
          c.box_dt = std::max(c.box_dt, (float)bk->box_dt);
        }
        g->member_bucket[(size_t)idx] = bk;
        // CODES: ordered table, histogram in LDS, 2 to 4 streamed fields, every systematic on them affine (one
        // coefficient) -- the conditions fill_ordered_body's kCodes states at compile time
        // (Histograms beyond LDS, counted at the event bins over run-walked tables -- BASELINE config 5 -- were given
        // codes too and measured: bit-identical, and SLOWER, 4.0 ms against 2.1-2.6.  With 200 bins per written
        // observable a code step is 1/220 of a bin, 0.8 % of the samples are ambiguous and 87 % of the 256-sample
        // units hold one; the fix-up then runs almost everywhere.  Codes pay where bins are coarse against 2^-16 of
        // the window: not offered there.)
        bool affine = ordered && c.shape.lds_hist && c.shape.nobs >= 1 && (boxed || c.shape.nslot - 1 >= 2) &&
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
        if (boxed && !d.qcol) {   // (no table of codes -- too many ambiguous rows, rows outside the window, no memory: the
                                  //  boxed form has no float stream of its own.  Planned again without it.)
          g->box_blocked = true;
          return group_rebuild(g);
        }
      }
      d.vec_start = prefix;
      prefix += d.nvec;
      d.step_gate = g->d_ticket + 8;     // (read by the measurement build's gated fill only)
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
      // (boxed form: one workgroup of 1024 lanes.  Config 3, one box, alternating: 64.6-66.4 us against 68.3 for 768 x 1
      // and 73.0-73.3 for 512 x 2 -- profiles/r05_boxed_ab.log)
      if (boxed) {
        static const int forced = [] {   // (SXMC_BOX_LANES, measurement build: 512 = two workgroups of 512 per CU, 768, 1024)
          const char* e = measure_env("SXMC_BOX_LANES");
          return e ? std::atoi(e) : 0;
        }();
        codes_two = forced == 512 && codes_two;
        threads = c.shape.threads = forced == 512 ? 512 : forced == 768 ? 768 : 1024;
      }
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
      // (SXMC_LDS_RESERVE, measurement build: bytes of the CU's LDS the fill leaves to other kernels' workgroups)
      static const size_t lds_reserve = [] {
        const char* e = measure_env("SXMC_LDS_RESERVE");
        return e ? (size_t)std::max(0, std::atoi(e)) : (size_t)0;
      }();
      const size_t share = ((size_t)props.lds_per_cu - lds_reserve) / (size_t)std::max(1, bpc) -
                           (fused_step_requested(g) ? 16 * 1024 : 0);
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
            const long long nb = descs[q].nbins[0];   // (slot 0 of the compacted problem)
            // (ordered form: the observable binned per sample must be the histogram's outermost dimension; the boxed
            // form makes it the outermost dimension of the LDS copy wherever it sits)
            const long long S = boxed ? (nb >= 1 ? (long long)h->total_nbins / nb : 0) : descs[q].bin_stride[0];
            all = all && S >= 1 && nb >= 1 && S * nb == (long long)h->total_nbins && S * (nb + 2) < (1ll << 22);
            if (boxed) all = all && descs[q].bin_stride[0] >= 1 && (long long)h->total_nbins % ((long long)descs[q].bin_stride[0] * nb) == 0;
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
        if (boxed && (!qlog || !c.padded_rstride)) {   // (the boxed form exists only in the padded form with queues)
          for (LaunchClass& cc : g->classes) free_class(cc);   // (what this pass has allocated so far)
          g->box_blocked = true;
          return group_rebuild(g);
        }
      } else if (boxed) {
        for (LaunchClass& cc : g->classes) free_class(cc);
        g->box_blocked = true;
        return group_rebuild(g);
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
        // (profiles/r03_c3_teams_sweep.log: 3 teams 129.4 us against 133.4-134.4 on one,
        // 128.4 against 124.9 on another, each consistently over alternating runs), so the default is one team and
        // sxmc_group_optimize tries three on the box it runs on (SXMC_PART_GROUPS forces a count for A/B runs).
        static const int forced_groups = [] {
          const char* e = measure_env("SXMC_PART_GROUPS");
          return e ? std::atoi(e) : 0;
        }();
        // (the boxed form, where nothing was asked for: 20 teams -- a workgroup then sees a twentieth of the sorted order,
        // its bucket or two, and flushes as few bins.  One box, alternating, two rounds (profiles/r05_boxed_teams_ab.log):
        // fill 63.4-63.5 us and step 77.9 against 65.0-65.1 and 79.9-80.0 for one team; 3-10 teams in between.)
        const int groups = (bucketed && c.shape.lds_hist)
                               ? (forced_groups > 0 ? forced_groups : g->cfg_teams > 0 ? g->cfg_teams : boxed ? 20 : 1) : 1;
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

  // BOXED plans, cfg_box < 0 (the default): the boxed form is fast only while few boxes straddle an edge, which depends on
  // the parameters of the evaluation.  The same members are therefore planned a second time in the ORDERED form (a twin
  // group: its own tables, partition, launch shape), group_fill launches the plan `fill_form` names, and the host moves
  // that between flushes of a walk (sxmc_group_adapt_fill_form).  Until it is asked to, the ordered form runs: a caller
  // that never asks gets round 4's kernel.  No twin (the ordered form does not apply): the plan is built again without boxes.
  if (!g->is_twin) {
    bool any_boxed = false;
    for (const LaunchClass& c : g->classes) any_boxed = any_boxed || c.shape.pre_width == 6;
    if (any_boxed && g->cfg_box < 0) {
      if (!g->twin) {
        g->twin = new sxmc_group;
        g->twin->is_twin = true;
      }
      sxmc_group* t = g->twin;
      t->members = g->members;
      t->cfg_box = 0;
      t->cfg_threads = g->cfg_threads;
      t->cfg_bpc = g->cfg_bpc;
      t->cfg_partition = g->cfg_partition;
      t->cfg_teams = g->cfg_teams;
      t->cfg_queue_log = g->cfg_queue_log;
      t->cfg_rtc = g->cfg_rtc;
      t->cfg_codes = g->cfg_codes;
      t->cfg_order = g->cfg_order;
      t->cfg_prebin = g->cfg_prebin;
      t->cfg_bucket = g->cfg_bucket;
      t->cfg_fused = g->cfg_fused;
      rc = group_rebuild(t);
      if (rc) return rc;
      bool ok = !t->classes.empty();
      for (const LaunchClass& c : t->classes) ok = ok && c.shape.pre_width != 6;
      if (!ok) {
        g->box_blocked = true;
        return group_rebuild(g);
      }
      for (LaunchClass& c : g->classes) c.dual = c.shape.pre_width == 6;
    } else if (g->twin && !any_boxed) {
      for (LaunchClass& c : g->twin->classes) free_class(c);
      g->twin->classes.clear();
    }
    if (!(any_boxed && g->cfg_box < 0)) g->fill_form = 1;   // (no twin: the plan's own launches)
    else if (g->fill_form != 1) g->fill_form = 2;
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
  g->cfg_seen_box = g->cfg_box;
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
               g->cfg_seen_order != g->cfg_order || g->cfg_seen_codes != g->cfg_codes || g->cfg_seen_box != g->cfg_box ||
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

int group_fill(sxmc_group* g, hipStream_t s, bool sparse) {
  TraceRange trace("sxmc: fill (EvalHist of all signals)");
  sparse = sparse && g->sparse_ready && g->cfg_sparse;
  int rc = group_prepare_fill(g, s, sparse);
  if (rc) return rc;
  // (a boxed plan with an ordered twin: the launches of the plan fill_form names -- sxmc_group_adapt_fill_form)
  sxmc_group* plan = (g->twin && !g->twin->classes.empty() && g->fill_form == 2 && g->cfg_box < 0) ? g->twin : g;
  for (LaunchClass& c : plan->classes) {
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

}  // namespace sxhost
