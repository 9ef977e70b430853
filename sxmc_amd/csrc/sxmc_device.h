// sxmc_device.h -- structures shared between the host side of libsxmc_hip.so and its gfx950
// kernels.  Internal: the public boundary is include/sxmc_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sxmc_hip.h"

#define SXMC_VEC 4         // samples per lane per iteration (one 16-byte load per column)
#define SXMC_OP_NOP (-1)

// One systematic, addressed by SLOT (position among the columns a launch loads), not by field.
// Restates SystematicDescriptor (pdfz.cpp:48-54) with the parameter indices inlined.
struct SxSystOp {
  short type;
  short obs_slot;
  short extra_slot;
  short npars;
  short pars[SXMC_MAX_SYST_PARS];
  short coef_start;  // first lane of this systematic's coefficients in the coefficient table
  short pad[3];
};

// Everything the kernels need to know about one evaluator (one signal's PDF).
// Lives in device memory as an array, one entry per group member; every field is read with
// wave-uniform addresses (scalar loads).
struct SxSignalDesc {
  // --- samples: column-major, column k at cols + k*col_pitch, padded with NaN to a multiple
  //     of SXMC_VEC samples (NaN is outside every domain, so padding needs no masking)
  const float* cols;
  unsigned long long col_pitch;  // floats, multiple of 64
  unsigned long long nsamples;
  unsigned long long nvec;       // ceil(nsamples / SXMC_VEC)
  unsigned long long vec_start;  // prefix sum of nvec over the launch's members
  // --- histogram
  unsigned* bins;
  unsigned* norm;                // norm_buffer + norm_offset
  int total_nbins;
  int nobs;
  int nslot;                     // columns loaded: nobs observables + referenced extra fields
  int nsyst;
  int ncoef;                     // total polynomial coefficients of all systematics (<= 64 for the
                                 // specialized kernels: one lane each)
  int param_stride;
  const double* params;          // param_buffer + param_offset
  int slot_col[SXMC_MAX_NFIELDS];
  int bin_stride[SXMC_MAX_NFIELDS];
  double lower[SXMC_MAX_NFIELDS];
  double upper[SXMC_MAX_NFIELDS];
  double scale[SXMC_MAX_NFIELDS];  // nbins / (upper - lower), computed on the host in double
  SxSystOp syst[SXMC_MAX_SYST];
  short coef_par[64];            // coefficient lane -> parameter index
  const void* pre;               // pre-binned column of the observables no systematic writes (or null); for a
                                 // bucketed table: one bin offset per 256-sample granule
  // --- sparse counting (histograms too large for LDS, evaluation for lookup only): `bins` then points
  //     at one counter per DISTINCT EVENT BIN, `read_bins` at the events' counter slots, and the fill maps
  //     a sample's flat bin index to its slot through a one-hash bit filter and an open-addressing table
  const unsigned* sparse_filter; // filter_bits / 32 words
  const unsigned* sparse_table;  // pairs {flat bin index, slot}; empty key = 0xFFFFFFFF
  int sparse_filter_shift;       // hash >> shift selects a filter bit
  int sparse_table_shift;        // hash >> shift selects a table entry
  int sparse_real_nbins;         // the histogram's true bin count (total_nbins is the counter count here)
  int sparse_coarse_shift;       // hash >> shift selects a bit of the coarse filter (staged in LDS; two hashes per bin)
  const unsigned* sparse_coarse; // coarse two-hash bit filter, at most 128 KiB
  // --- sparse counting over a bucketed table walked in runs (fill_sparse_kernel): the event bins are grouped by
  //     bucket (the bin indices of the untouched observables), each bucket with its own small hash table
  const unsigned* sparse_dir;    // pairs {first table entry, log2(table size) | flags}, indexed by bucket key
  const unsigned* sparse_tkeys;  // table entries: the event bin's index contribution of the written observables
  const unsigned* sparse_tslot;  // ... and its counter slot
  int nbins[SXMC_MAX_NFIELDS];   // bins per observable (an index that comes out as nbins is the aliasing case)
  // --- evaluation at the data events
  const int* read_bins;
  unsigned long long npoints;
  float* pdf_out;                // pdf_buffer + pdf_offset
  int pdf_stride;
  int pad0;
  double bin_volume;
};

// One piece of fill work: units v0 + tid, + step, ... < v1 of member `sig` (a unit = SXMC_VEC samples).
struct SxSegment {
  int sig;
  int pad;
  unsigned long long v0, v1, step;
};

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
  int sparse_runs;  // 1: the sparse flavour of this launch runs fill_sparse_kernel (bucketed table laid out in runs)
  size_t sparse_lds_bytes;
  int pre_width;    // bytes per sample of the pre-binned column (1, 2, 4), 0 = none, 3 = bucketed table (one
                    // bin offset per 256-sample granule)
};

hipError_t sx_launch_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, unsigned* ticket, hipStream_t s);
hipError_t sx_launch_finish_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, size_t npartial,
                                 const double* sums, unsigned* ticket, const SxStepArgs& a, int block, hipStream_t s);
hipError_t sx_launch_eval_nll_finish(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                                      const unsigned* weight, double* sums, unsigned* ticket, const SxStepArgs& a,
                                      int grid, int block, hipStream_t s);
hipError_t sx_launch_tail_step(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                               const unsigned* weight, const SxStepArgs& a, hipStream_t s);
hipError_t sx_launch_fill(const SxLaunchShape& shape, const SxSignalDesc* d_descs, const SxSegment* d_segs,
                          const unsigned* d_blk_off, hipStream_t s);
hipError_t sx_launch_fill_sparse_runs(const SxLaunchShape& shape, const SxSignalDesc* d_descs, const SxSegment* d_segs,
                                      const unsigned* d_blk_off, hipStream_t s);
bool sx_fill_static_supports_sparse_runs(int prog);
#define SXMC_SPARSE_EMPTY 0xFFu   /* directory flag: the bucket holds no event bin */
#define SXMC_SPARSE_SLOW 0xFEu    /* directory flag: look every sample up in the global table (see the kernel) */
#define SXMC_SPARSE_SMAX_LOG2 9  /* largest per-wave table: 512 entries */
bool sx_fill_has_specialization(int nobs, int nslot);
int sx_fill_find_static_program(int nobs, int nslot, int nops, const unsigned* ops);
bool sx_fill_static_supports(int prog, int lds_hist, int prebin);
hipError_t sx_launch_prebin(const SxSignalDesc* d_desc, unsigned long long npad, unsigned mask, int width, void* out,
                            hipStream_t s);
// bucketed copy of a sample table (layout_kernels.hip)
hipError_t sx_bucket_keys(const SxSignalDesc* d_desc, unsigned long long nsamples, unsigned mask, const unsigned* radix,
                          unsigned outside, unsigned* d_keys, unsigned* d_rows, hipStream_t s);
hipError_t sx_bucket_sort(const unsigned* keys_in, unsigned* keys_out, const unsigned* rows_in, unsigned* rows_out,
                          unsigned long long n, int bits, hipStream_t s);
hipError_t sx_bucket_first(const unsigned* sorted_keys, unsigned long long n, unsigned* d_first, hipStream_t s);
hipError_t sx_bucket_gather(const float* cols, unsigned long long pitch, int ncols, const int* col_list,
                            const unsigned* sorted_rows, const unsigned* d_src, const unsigned* d_valid,
                            unsigned long long ngranules, float* out, unsigned long long out_pitch, hipStream_t s);
hipError_t sx_launch_eval_pdf(const SxSignalDesc* d_descs, int nsig, unsigned long long max_points,
                              hipStream_t s);
hipError_t sx_launch_eval_nll(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                              const unsigned* weight, const double* pars, const double* nexpected, const unsigned* n_mc,
                              const short* source_id, const unsigned* norms, double* sums,
                              int grid, int block, hipStream_t s);
hipError_t sx_launch_transpose(const float* aos, float* cols, unsigned long long nsamples,
                               int nfields, unsigned long long col_pitch, hipStream_t s);
hipError_t sx_launch_untranspose_obs(const float* cols, float* out, unsigned long long nsamples,
                                     int nobs, unsigned long long col_pitch, float dataset,
                                     hipStream_t s);
