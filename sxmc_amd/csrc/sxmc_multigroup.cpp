// sxmc_multigroup.cpp -- several chains over the same tables in one pass (lockstep sets) and the look-ahead pass of one chain.
#include "sxmc_host.h"

using namespace sxhost;

namespace sxhost {
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
}  // namespace sxhost

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
    g->cfg_box = 0;   // (several chains per pass: the ordered form; the boxed one has no such kernel)
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
    note_stepping(g);
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
    g->cfg_box = 0;   // (several chains per pass: the ordered form; the boxed one has no such kernel)
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
    note_stepping(g);
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
  g->cfg_box = 0;   // (the look-ahead pass runs over the ordered form: the question is asked of that plan)
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

}  // extern "C"
