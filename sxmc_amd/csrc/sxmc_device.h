// sxmc_device.h -- structures shared between the host side of libsxmc_hip.so and its gfx950
// kernels.  Internal: the public boundary is include/sxmc_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/sxmc_hip.h"

#include "sxmc_device_types.h"

// Arguments of finish_nll_jump_pick_combo (nll_kernels.h:190-207) for the fused step end.
struct SxStepArgs {
  size_t nsignals, nsources;
  const double* means;
  const double* sigmas;
  sxmc_rng_state* rng;
  double* nll_current;
  double* nll_proposed;
  double* v_current;
  double* v_proposed;
  int* accepted;
  int* counter;
  float* jump_buffer;
  int nparameters;
  int debug_mode;
  const float* jump_width;
  const double* nexpected;
  const unsigned* n_mc;
  const short* source_id;
  const unsigned* norms;
  // measurement build only (the gated step): the finisher stores gate_value to *gate once the next proposal is written
  unsigned* gate = nullptr;
  unsigned gate_value = 0;
};

// The step ends of the chains of a lockstep set, launched together (sxmc_multigroup_step_async): per chain what
// eval_nll_kernel and finish_zero_kernel take.  Passed by value (kernel argument); the chain is blockIdx.y.
#define SXMC_MAX_LOCKSTEP 4
struct SxChainEnd {
  const SxSignalDesc* lookup_descs;  // the rows the event sum runs over (event classes, or the events themselves)
  const SxSignalDesc* hist_descs;    // the chain's histograms and normalisations (cleared for the next evaluation)
  unsigned long long nrows;
  const unsigned* weight;            // events per row, or null
  double* sums;                      // nblocks partial event sums
  unsigned* ticket;
  unsigned nblocks;                  // workgroups of this chain's event sum (step_sum_blocks: as in its sequential step)
  SxStepArgs a;
};
struct SxChainEnds {
  SxChainEnd c[SXMC_MAX_LOCKSTEP];
};

// Host-callable launchers implemented in the .hip files -------------------------------------
struct SxLaunchShape {
  int nobs;
  int nslot;
  int lds_hist;     // 1: LDS-privatized sub-histograms, 0: global atomics into the HBM histogram
  int threads;      // 256 / 512 / 1024
  int grid;
  size_t lds_bytes;
  int debug_mode;   // measurement hook, see fill_kernel
  int static_prog;  // index into the static program table, or -1: decode the program at run time
  void* rtc_fill;   // kernel specialised at run time for this launch (hipFunction_t), or null: built-in kernels
  void* rtc_sparse; // ... and for its sparse flavour over runs
  int sparse_runs;  // 1: the sparse flavour of this launch runs fill_sparse_kernel (bucketed table laid out in runs)
  size_t sparse_lds_bytes;
  unsigned lds_layout;  // pre_width 5: words between the LDS histogram's replicas | log2(replicas) << 24
  int pre_width;    // bytes per sample of the pre-binned column (1, 2, 4), 0 = none, 3 = bucketed table (one
                    // bin offset per 256-sample granule), 5 = bucketed table with an ordered observable
                    // (fill_ordered_kernel; static_prog then indexes the ordered programs built in)
                    // 6 = bucketed table with a BOXED observable (fill_boxed_kernel; static_prog indexes the boxed
                    // programs built in; lds_layout as for 5, always the padded form with queues)
  // profiling (sxmc_group_profile): when set, the fill is launched through hipExtLaunchKernelGGL /
  // hipExtModuleLaunchKernel with these two HIP events, which then carry the DISPATCH's own begin and end timestamps
  // -- the kernel's duration as rocprofv3 --kernel-trace reports it.  (Two hipEventRecord calls around the launch
  // also time the packets between them: +2.5-3 us per launch, a fifth of BASELINE config 2's fill.)
  void* ev_start = nullptr;
  void* ev_stop = nullptr;
  // the whole step in this launch (fill_step_kernel): HOST image of the step end's arguments (sx_tail_args_fill; passed
  // to the kernel by value) and the extra workgroups (finisher + workers) appended to the grid; null: the fill alone
  const void* tail = nullptr;
  int tail_blocks = 0;
};
bool sx_fill_has_step_form(const SxLaunchShape& sh);
size_t sx_tail_args_bytes();
void sx_tail_args_fill(void* image, const SxSignalDesc* lookup_descs, const SxSignalDesc* hist_descs, int nsig,
                       int max_bins, unsigned long long npoints, const unsigned* weight, unsigned long long* slots,
                       double* last_good, unsigned* sync, int nvb, const struct SxStepArgs& a);

// A fill kernel specialised at run time (sxmc_rtc.cpp): the template arguments of fill_body / fill_sparse_body.
struct SxRtcSpec {
  int nobs, nslot, lds_hist, pre_width, sparse_runs;
  int nchain;                    // > 1: the lockstep-chains kernel (fill_multi_body), histograms in LDS
  int max_threads;               // launch bound of the kernel (0 = 1024): several chains over an ordered table need more
                                 // registers than 1024 lanes leave (spills inside the stream loop: every reload drains
                                 // the loads in flight), so their kernels are compiled for the workgroup they get
  int nops;
  unsigned ops[SXMC_MAX_SYST];   // type | obs_slot << 4 | extra_slot << 8 | npars << 12 (0 = one coefficient)
};
bool sx_rtc_compile_only(const SxRtcSpec& k, size_t* code_bytes, std::string* err);
void* sx_rtc_get(const SxRtcSpec& k, std::string* err);
struct SxChainDescsHost {
  const SxSignalDesc* d[4];
};
hipError_t sx_rtc_launch_multi(void* fn, int grid, int threads, size_t lds_bytes, const SxChainDescsHost& chains,
                               const SxSegment* segs, const unsigned* blk_off, unsigned hist_words, unsigned dbg,
                               hipStream_t s, void* ev_start = nullptr, void* ev_stop = nullptr);
hipError_t sx_rtc_launch(void* fn, int grid, int threads, size_t lds_bytes, const SxSignalDesc* descs,
                         const SxSegment* segs, const unsigned* blk_off, unsigned w, unsigned dbg, hipStream_t s,
                         void* ev_start = nullptr, void* ev_stop = nullptr);

hipError_t sx_launch_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, unsigned* ticket, hipStream_t s);
hipError_t sx_launch_step_end(const SxSignalDesc* lookup_descs, const SxSignalDesc* hist_descs, int nsig, int max_bins,
                              unsigned long long npoints, const unsigned* weight, unsigned long long* slots,
                              double* last_good, unsigned* sync, int nvb, const SxStepArgs& a, hipStream_t s);
hipError_t sx_launch_step_end2(const SxSignalDesc* lookup_a, const SxSignalDesc* lookup_b, const SxSignalDesc* hist_a,
                               const SxSignalDesc* hist_b, int nsig, int max_bins, unsigned long long npoints,
                               const unsigned* weight_a, const unsigned* weight_b, unsigned long long* slots,
                               double* last_good, unsigned* sync, int nvb, const unsigned* norms_b, double* v_b,
                               const int* cap, const SxStepArgs& a, hipStream_t s);
hipError_t sx_step_end_slots_init(unsigned long long* slots, double* last_good, int n);
int sx_step_end_resident_capacity(int nsig, int cus);
hipError_t sx_launch_chain_ends(const SxChainEnds& e, int nchains, int nsig, int max_bins, int block, hipStream_t s);
hipError_t sx_launch_finish_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, size_t npartial,
                                 const double* sums, unsigned* ticket, const SxStepArgs& a, int block, hipStream_t s);
hipError_t sx_launch_eval_nll_finish(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                                      const unsigned* weight, double* sums, unsigned* ticket, const SxStepArgs& a,
                                      int grid, int block, hipStream_t s);
hipError_t sx_launch_eval_nll2(const SxSignalDesc* descs_a, const SxSignalDesc* descs_b, int nsig,
                               unsigned long long npoints, const unsigned* weight_a, const unsigned* weight_b,
                               const double* pars_a, const double* pars_b, const double* nexpected,
                               const unsigned* n_mc, const short* source_id, const unsigned* norms_a,
                               const unsigned* norms_b, double* sums_a, double* sums_b, int half, int block,
                               hipStream_t s);
hipError_t sx_launch_finish2_zero(const SxSignalDesc* descs_a, const SxSignalDesc* descs_b, int nsig, int max_bins,
                                  size_t npartial, const double* sums_a, const double* sums_b, const unsigned* norms_b,
                                  double* v_b, const int* cap, const SxStepArgs& a, int block, hipStream_t s);
hipError_t sx_launch_peek_next_proposal(int nparameters, const sxmc_rng_state* rng, const float* jump_width,
                                        const double* v_current, double* out, hipStream_t s);
hipError_t sx_launch_tail_step(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                               const unsigned* weight, const SxStepArgs& a, hipStream_t s);
hipError_t sx_launch_fill(const SxLaunchShape& shape, const SxSignalDesc* d_descs, const SxSegment* d_segs,
                          const unsigned* d_blk_off, hipStream_t s);
hipError_t sx_launch_fill_sparse_runs(const SxLaunchShape& shape, const SxSignalDesc* d_descs, const SxSegment* d_segs,
                                      const unsigned* d_blk_off, hipStream_t s);
bool sx_fill_static_supports_sparse_runs(int prog);

bool sx_fill_has_specialization(int nobs, int nslot);
int sx_fill_find_static_program(int nobs, int nslot, int nops, const unsigned* ops);
int sx_fill_find_ordered_program(int nobs, int nslot, int nops, const unsigned* ops);
int sx_fill_find_boxed_program(int nobs, int nslot, int nops, const unsigned* ops);
bool sx_fill_static_supports(int prog, int lds_hist, int prebin);
hipError_t sx_launch_prebin(const SxSignalDesc* d_desc, unsigned long long npad, unsigned mask, int width, void* out,
                            hipStream_t s);
// bucketed copy of a sample table (layout_kernels.hip)
hipError_t sx_bucket_keys(const SxSignalDesc* d_desc, unsigned long long nsamples, unsigned mask, const unsigned* radix,
                          unsigned outside, const unsigned* d_rows_in, unsigned* d_keys, unsigned* d_rows,
                          hipStream_t s);
hipError_t sx_order_keys(const float* d_col, unsigned long long nsamples, unsigned* d_keys, unsigned* d_rows,
                         hipStream_t s);
hipError_t sx_bucket_edges(const float* d_col, const unsigned* d_valid, unsigned long long ngranules, float* d_edges,
                           hipStream_t s);
// boxed observable (fill_boxed_kernel): sort keys by (stratum of x - t, x); per-granule boxes; one column as u16 codes
hipError_t sx_box_keys(const float* d_colx, const float* d_colt, unsigned long long nsamples, int pass, int nstrata,
                       const unsigned* bounds, unsigned* d_keys, unsigned* d_rows, hipStream_t s);
hipError_t sx_bucket_boxes(const float* d_colx, const float* d_colt, const unsigned* d_valid, unsigned long long ngranules,
                           float* d_boxes, hipStream_t s);
hipError_t sx_column_codes16(const float* col, double base, double step, unsigned long long n, unsigned short* qcol,
                             unsigned long long* tally, hipStream_t s);
hipError_t sx_bucket_sort(const unsigned* keys_in, unsigned* keys_out, const unsigned* rows_in, unsigned* rows_out,
                          unsigned long long n, int bits, hipStream_t s);
hipError_t sx_bucket_first(const unsigned* sorted_keys, unsigned long long n, unsigned* d_first, hipStream_t s);
hipError_t sx_bucket_gather(const float* cols, unsigned long long pitch, int ncols, const int* col_list,
                            const unsigned* sorted_rows, const unsigned* d_src, const unsigned* d_valid,
                            unsigned long long ngranules, float* out, unsigned long long out_pitch, hipStream_t s);
// 16-bit codes of the streamed columns of a bucketed copy (fill_ordered_body's CODES)
hipError_t sx_column_minmax(const float* cols, unsigned long long pitch, int ncols, unsigned long long n, float* out,
                            hipStream_t s);
hipError_t sx_column_codes(const float* cols, unsigned long long pitch, int nslots, const double* base, const double* step,
                           unsigned long long n, unsigned* qcol, unsigned long long* tally, hipStream_t s);
hipError_t sx_hist_cdf(const unsigned* d_bins, unsigned* d_cdf, int nbins_total, hipStream_t s);
hipError_t sx_random_sample(const unsigned* d_cdf, int nbins_total, int nobs, const int* nbins, const double* lower,
                            const double* upper, const float* cut_lo, const float* cut_hi, unsigned long long seed,
                            unsigned long long n, float dataset, float* d_out, unsigned* d_exhausted, hipStream_t s);
hipError_t sx_launch_eval_pdf(const SxSignalDesc* d_descs, int nsig, unsigned long long max_points,
                              hipStream_t s);
hipError_t sx_launch_eval_nll(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                              const unsigned* weight, const double* pars, const double* nexpected, const unsigned* n_mc,
                              const short* source_id, const unsigned* norms, double* sums,
                              int grid, int block, hipStream_t s);
hipError_t sx_launch_pow_int(const double* x, int n, int i, double* out, hipStream_t s);
hipError_t sx_launch_transpose(const float* aos, float* cols, unsigned long long nsamples,
                               int nfields, unsigned long long col_pitch, hipStream_t s);
hipError_t sx_launch_untranspose_obs(const float* cols, float* out, unsigned long long nsamples,
                                     int nobs, unsigned long long col_pitch, float dataset,
                                     hipStream_t s);
