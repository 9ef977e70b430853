// sxmc_group.cpp -- the group of evaluators (one launch sequence for all signals of a step): configuration, autotune
// (EvalHist::Optimize's role, pdfz.cpp:622-814), evaluation, the MCMC step and the forms of its end.
#include "sxmc_host.h"

using namespace sxhost;

namespace sxhost {
std::atomic<int> g_stepping_groups{0};
}  // namespace sxhost

extern "C" {

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
  if (g->coop_fits >= 0) g_stepping_groups.fetch_sub(1, std::memory_order_acq_rel);
  for (LaunchClass& c : g->classes) free_class(c);
  if (g->h_pin) (void)hipHostFree(g->h_pin);
  if (g->twin) {   // (the ordered twin of a boxed plan: planned, never launched by itself)
    for (LaunchClass& c : g->twin->classes) free_class(c);
    if (g->twin->d_descs) (void)hipFree(g->twin->d_descs);
    if (g->twin->d_descs_sparse) (void)hipFree(g->twin->d_descs_sparse);
    delete g->twin;
    g->twin = nullptr;
  }
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
  for (const LaunchClass& c : g->classes) has_bucketed = has_bucketed || ((c.shape.pre_width == 3 || c.shape.pre_width == 5 || c.shape.pre_width == 6) && c.shape.lds_hist);
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

int sxmc_group_set_boxes(sxmc_group_t g, int enable) {
  SX_REQUIRE(g, "null group");
  g->cfg_box = enable < 0 ? -1 : enable ? 1 : 0;
  g->box_blocked = false;
  return SXMC_OK;
}

int sxmc_group_set_box_limit(sxmc_group_t g, double bins) {
  SX_REQUIRE(g && bins >= 0.0, "bad arguments");
  g->box_limit = (float)bins;
  return SXMC_OK;
}

int sxmc_group_set_fill_form(sxmc_group_t g, int form) {
  SX_REQUIRE(g && (form == 1 || form == 2), "form: 1 boxed, 2 ordered");
  SX_FLUSH();
  int rc = group_refresh(g);
  if (rc) return rc;
  SX_REQUIRE(g->twin && !g->twin->classes.empty() && g->cfg_box < 0, "the group's plan has one form only");
  g->fill_form = form;
  return SXMC_OK;
}

// The image of a box of the boxed tables' mean extents under the members' program at the parameters the evaluators read
// NOW (copied back from the device): the reference's operations on the box's corners (fill_boxed_body's interval form), in
// bins of the boxed observable.  What share of the granules straddles an edge is about that width.
static int box_image_width(sxmc_group* g, double* width) {
  *width = HUGE_VAL;
  const LaunchClass* cls = nullptr;
  for (const LaunchClass& c : g->classes) cls = (c.dual && !cls) ? &c : cls;
  if (!cls || cls->member_idx.empty() || cls->box_obs < 0) return SXMC_OK;
  const SxSignalDesc& d = g->h_descs[(size_t)cls->member_idx[0]];
  if (!d.params) return SXMC_OK;
  // the coefficients, in ONE asynchronous copy on the group's own stream into pinned memory: a copy through the legacy
  // stream would wait for every blocking stream of the process -- and fail outright while another chain records a graph
  int lo_idx = 1 << 30, hi_idx = -1;
  for (int q = 0; q < d.nsyst; q++) {
    if (d.syst[q].obs_slot != cls->box_obs) continue;
    lo_idx = std::min(lo_idx, (int)d.syst[q].pars[0]);
    hi_idx = std::max(hi_idx, (int)d.syst[q].pars[0]);
  }
  if (hi_idx < 0 || d.param_stride < 1 || (long)(hi_idx - lo_idx) * d.param_stride >= 256) return SXMC_OK;
  if (!g->h_pin) SX_HIP(hipHostMalloc((void**)&g->h_pin, 256 * sizeof(double)));
  const size_t span = (size_t)(hi_idx - lo_idx) * (size_t)d.param_stride + 1;
  SX_HIP(hipMemcpyAsync(g->h_pin, d.params + (long)lo_idx * d.param_stride, span * sizeof(double), hipMemcpyDeviceToHost,
                        g->last_stream));
  SX_HIP(hipStreamSynchronize(g->last_stream));
  double xl = 0.5 * (d.lower[cls->box_obs] + d.upper[cls->box_obs]), xh = xl + (double)cls->box_dx;
  const double tl = xl, th = xl + (double)cls->box_dt;
  bool fin = true;
  for (int q = 0; q < d.nsyst && fin; q++) {
    const SxSystOp& op = d.syst[q];
    if (op.obs_slot != cls->box_obs) continue;
    const double c0 = g->h_pin[(size_t)((int)op.pars[0] - lo_idx) * (size_t)d.param_stride];
    const double pc = 0.0 + c0 * 1.0;
    if (op.type == SXMC_SYST_SHIFT) {
      xl = xl + pc;
      xh = xh + pc;
    } else if (op.type == SXMC_SYST_SCALE || op.type == SXMC_SYST_CTSCALE) {
      const double s = 1 + pc;
      const double a = op.type == SXMC_SYST_SCALE ? xl * s : 1 + (xl - 1) * s;
      const double b = op.type == SXMC_SYST_SCALE ? xh * s : 1 + (xh - 1) * s;
      xl = s >= 0.0 ? a : b;
      xh = s >= 0.0 ? b : a;
    } else if (op.type == SXMC_SYST_RESOLUTION_SCALE) {
      const double dl = xl - th, dh = xh - tl, a = pc * dl, b = pc * dh;
      xl = xl + (pc >= 0.0 ? a : b);
      xh = xh + (pc >= 0.0 ? b : a);
    }
    fin = std::isfinite(xl) && std::isfinite(xh);
  }
  if (fin) *width = (xh - xl) * d.scale[cls->box_obs];
  return SXMC_OK;
}

int sxmc_group_adapt_fill_form(sxmc_group_t g, int* form, int* changed) {
  SX_REQUIRE(g, "null group");
  SX_FLUSH();
  if (form) *form = 0;
  if (changed) *changed = 0;
  if (t_capturing) return fail(SXMC_ERR_STATE, "the form of the fill is not chosen while a graph is being recorded");
  int rc = group_refresh(g);
  if (rc) return rc;
  if (!(g->twin && !g->twin->classes.empty() && g->cfg_box < 0)) return SXMC_OK;   // (one form only)
  SX_HIP(hipStreamSynchronize(g->last_stream));
  double width = HUGE_VAL;
  rc = box_image_width(g, &width);
  if (rc) return rc;
  // (a band between the two decisions: a walk that sits at the limit does not re-record its steps at every flush)
  const int want = width < 0.8 * (double)g->box_limit ? 1 : (width < (double)g->box_limit ? g->fill_form : 2);
  if (changed) *changed = want != g->fill_form ? 1 : 0;
  g->fill_form = want;
  if (form) *form = want;
  return SXMC_OK;
}

int sxmc_group_fill_form(sxmc_group_t g, int* form) {
  SX_REQUIRE(g && form, "null argument");
  SX_FLUSH();
  int rc = group_refresh(g);
  if (rc) return rc;
  *form = (g->twin && !g->twin->classes.empty() && g->cfg_box < 0) ? g->fill_form : 0;
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
  // (a plan with two forms: the launches of the form that runs now, its table named with the other form beside it)
  const bool two = g->twin && !g->twin->classes.empty() && g->cfg_box < 0;
  const sxmc_group* plan = (two && g->fill_form == 2) ? g->twin : g;
  for (size_t i = 0; i < plan->classes.size(); i++) {
    const LaunchClass& c = plan->classes[i];
    char line[512];
    const char* kind = c.shape.rtc_fill ? "runtime" : c.shape.static_prog >= 0 ? "builtin" : c.shape.nobs ? "decoded" : "generic";
    std::snprintf(line, sizeof line,
                  "launch %zu: members=%zu nobs=%d nslot=%d hist=%s program=%s table=%s%s threads=%d grid=%d partition=%d teams=%d\n",
                  i, c.member_idx.size(), c.shape.nobs, c.shape.nslot, c.shape.lds_hist ? "lds" : "global", kind,
                  c.shape.pre_width == 6 ? (c.dual ? "boxed+codes(now)|ordered+codes" : "boxed+codes")
                  : (two && plan == g->twin && c.shape.pre_width == 5) ? (c.codes ? "boxed+codes|ordered+codes(now)" : "boxed+codes|ordered(now)")
                  : c.shape.pre_width == 5 ? (c.codes ? "ordered+codes" : "ordered") : c.shape.pre_width == 3 ? "bucketed" : c.shape.pre_width ? "prebinned" : "rows",
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
// measurement build only, THE GATED STEP: the group's next sxmc_group_step_async puts its fill on `fill_stream` (null:
// back to one stream) as step `node` (0 .. 15) of a recording; sxmc_measure_stream_fork makes `to` wait for what `from`
// has queued so far (inside a recording: brings `to` into it).
int sxmc_measure_set_gated_step(sxmc_group_t g, sxmc_stream_t fill_stream, int node) {
  SX_REQUIRE(g && node >= 0 && node < 16, "bad arguments");
  g->split_fill_stream = (hipStream_t)fill_stream;
  g->split_node = node;
  return SXMC_OK;
}
int sxmc_measure_stream_fork(sxmc_stream_t from, sxmc_stream_t to) {
  SX_REQUIRE(from && to, "null stream");
  hipEvent_t ev = nullptr;
  SX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  SX_HIP(hipEventRecord(ev, (hipStream_t)from));
  SX_HIP(hipStreamWaitEvent((hipStream_t)to, ev, 0));
  return SXMC_OK;     // (the event lives as long as the process: a measurement build's leak)
}

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

namespace sxhost {
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
// RESIDENCY.  The waits inside step_end_kernel are between workgroups of one ordinary launch: they end only if the
// waited-for workgroups are resident or become resident.  Fill kernels never wait, so whatever they occupy frees up by
// itself; what could starve a step end is OTHER step ends waiting in the slots its workgroups need.  So the form is
// taken only while the step ends of every chain stepping in this process (each at most 129 workgroups) fit HALF of what
// the device holds of that kernel at once (the runtime's occupancy figure x CUs: thousands), decided once, at a group's
// first step (a recorded graph keeps the form it was recorded with); beyond that, the two-launch form.  The waits stay
// bounded besides, and a wait that gave up ends the walk at its next flush (sxmc_group_step_end_timeouts).
void note_stepping(sxmc_group* g) {
  if (g->coop_fits >= 0) return;
  const int chains = g_stepping_groups.fetch_add(1, std::memory_order_acq_rel) + 1;
  DeviceProps props;
  int capacity = 0;
  if (get_props(props) == SXMC_OK) capacity = sx_step_end_resident_capacity((int)g->members.size(), props.cus);
  g->coop_fits = (long long)chains * (kCoopMaxWorkers + 1) * 2 <= (long long)capacity ? 1 : 0;
}
bool step_end_is_cooperative(const sxmc_group* g, unsigned long long ne) {
  static const int env_default = [] {
    const char* e = std::getenv("SXMC_COOP_STEP_END");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  const int on = g->cfg_coop < 0 ? env_default : g->cfg_coop;
  return on != 0 && g->coop_fits != 0 && g->cfg_tail != 0 && step_sum_blocks(ne) <= kCoopMaxWorkers;
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
}  // namespace sxhost

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
  note_stepping(g);
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
#if SXMC_MEASURE
  // THE GATED STEP (measurement build only; sxmc_measure_set_gated_step): the fill goes to a stream of its own, ordered
  // after the previous FILL but not after the previous step end -- it runs its prologue beside that kernel and waits
  // for the proposal inside (fill_ordered_body, dbg bit 6); the step end follows on `s` once the fill has ended.
  if (g->split_fill_stream && !fused && !sparse && g->classes.size() == 1 && step_end_is_cooperative(g, ne) &&
      !step_end_takes_tail(g, sparse, ne)) {
    const int node = g->split_node & 15;
    const int saved = g->debug_mode;
    g->debug_mode = saved | 64 | (node << 24);
    rc = group_fill(g, g->split_fill_stream, sparse);
    g->debug_mode = saved;
    if (rc) return rc;
    hipEvent_t ev = nullptr;
    SX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g->split_events.push_back(ev);
    SX_HIP(hipEventRecord(ev, g->split_fill_stream));
    SX_HIP(hipStreamWaitEvent(st, ev, 0));
    a.gate = g->d_ticket + 8;
    a.gate_value = (unsigned)node + 1u;
    g->last_step_launches = (int)g->classes.size() + (zero_launched ? 1 : 0);
    return group_step_tail(g, st, sparse, descs, ne, weight, a);
  }
#endif
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

extern "C" {

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
  if (int rc_ = settle()) return rc_;
  if (s) {
    // on the chain's own stream: a copy through the legacy stream is what the runtime refuses while ANOTHER host
    // thread records a graph on a blocking stream (see sxmc_graph_begin_capture)
    SX_HIP(hipMemcpyAsync(timeouts, g->d_ticket + 6, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)s));
    SX_HIP(hipStreamSynchronize((hipStream_t)s));
  } else {
    SX_HIP(hipMemcpy(timeouts, g->d_ticket + 6, sizeof(unsigned), hipMemcpyDeviceToHost));
  }
  if (*timeouts != 0u) {
    // A wait inside a cooperative step end gave up: the steps since the last check are not valid (a finisher that gave
    // up used 0 for the missing partial, a late worker may have left a stale partial in its slot, a worker that gave up
    // left histograms uncleared).  The caller ends its walk (sxmc::MCMC and sxmc_amd/mcmc.py check at every flush of the
    // jump buffer); the group itself is put back in order here: slots emptied, count cleared, nothing taken as
    // pre-zeroed, and the two-launch step end from now on.
    hipStream_t st = (hipStream_t)s;
    SX_HIP(hipStreamSynchronize(g->last_stream));
    SX_HIP(sx_step_end_slots_init(g->d_coop_slots, g->d_coop_last, 128));
    SX_HIP(hipMemsetAsync(g->d_ticket, 0, 256, st));
    SX_HIP(hipStreamSynchronize(st));
    g->prezeroed = 0;
    for (sxmc_hist* h : g->members) h->cleared_by = nullptr;
    g->cfg_coop = 0;
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
  if (g->twin && !g->twin->classes.empty() && g->cfg_box < 0 && g->fill_form == 2) {
    // (the ordered twin's tables are what the fill streams now; histogram and event bytes do not depend on the form)
    double fr2 = 0, hb2 = 0, ev2 = 0, fr1 = 0;
    rc = sxmc_group_algorithmic_bytes(g->twin, &fr2, &hb2, &ev2);
    if (rc) return rc;
    g->fill_form = 1;
    rc = sxmc_group_algorithmic_bytes(g, &fr1, hist, event);
    g->fill_form = 2;
    *fill_read = fr2;
    return rc;
  }
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
      // (boxed: one 16-bit code per row, per granule the box (16 bytes) and the word; the three float columns of the
      // granules whose box straddles an edge -- a few per cent, which ones depends on the parameters -- are not counted)
      const bool ord = bk->sort && bk->sort->ordered >= 0;
      const bool box = ord && bk->sort->box_truth >= 0;
      const double row_bytes = (codes && bk->d_qcol) ? (bk->q16 ? 2.0 : 4.0 * (double)((bk->nq + 1) / 2))
                                                     : 4.0 * (double)(bk->fields.size() - (ord ? 1 : 0));
      fr += (double)bk->nkept * row_bytes + ((box ? 16.0 : ord ? 8.0 : 0.0) + (bk->runs > 1 ? 8.0 : 4.0)) * (double)bk->ngranules;
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

}  // extern "C"
