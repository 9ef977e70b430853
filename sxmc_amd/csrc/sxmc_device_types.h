// sxmc_device_types.h -- the structures the gfx950 kernels read, shared by the host side of libsxmc_hip.so, the
// kernels built into it and the kernels it compiles at run time (hiprtc: so no standard headers here, built-in types
// only).  Internal: the public boundary is include/sxmc_hip.h.
#pragma once

// (the same values as include/sxmc_hip.h, which a run-time compilation does not see)
#ifndef SXMC_MAX_NFIELDS
#define SXMC_MAX_NFIELDS 10
#define SXMC_MAX_SYST 16
#define SXMC_MAX_SYST_PARS 8
#define SXMC_SYST_SHIFT 0
#define SXMC_SYST_SCALE 1
#define SXMC_SYST_RESOLUTION_SCALE 2
#define SXMC_SYST_CTSCALE 3
#endif

#define SXMC_VEC 4         // samples per lane per iteration (one 16-byte load per column)
#define SXMC_OP_NOP (-1)
#define SXMC_SPARSE_EMPTY 0xFFu   /* directory flag: the bucket holds no event bin */
#define SXMC_SPARSE_SLOW 0xFEu    /* directory flag: look every sample up in the global table (see the kernel) */
#define SXMC_SPARSE_SMAX_LOG2 9  /* largest per-wave table: 512 entries */
#define SXMC_MAX_QSLOTS 4        /* streamed slots a table of 16-bit codes can stand in for (fill_ordered_body) */
#define SXMC_QCODE_MAX 65533u    /* largest code of a value inside its window */
#define SXMC_QCODE_EXACT 0xFFFEu /* "outside the window: ask the exact columns" */
#define SXMC_QCODE_NEVER 0xFFFFu /* "not finite: never counted" */

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
  const float* edges;            // bucketed table with an ORDERED observable (fill_ordered_kernel): per granule the
                                 // observable's value in the granule's first and last row; its geometry sits at
                                 // index `nobs` of lower / upper / scale / bin_stride / nbins, its column in slot
                                 // nslot - 1
  const float* boxes;            // bucketed table with a BOXED observable (fill_boxed_kernel): per granule {xmin, xmax,
                                 // tmin, tmax} of the observable's raw value and of the truth field its resolution
                                 // scale reads, over the granule's rows (NaN: a row is not finite); geometry at index
                                 // `nobs`, the observable in slot nslot - 1, the truth field in slot nslot - 2; `qcol`
                                 // then holds ONE 16-bit code per row (slot 0)
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
  // --- CODES (fill_ordered_body): the streamed slots of the bucketed copy once more, each value as a 16-bit code
  //     code = floor((x - qbase) / qstep), two slots per 32-bit word (slot 2w in the high half of word w, slot 2w + 1
  //     in the low half), word column w at qcol + w * col_pitch.  Code 0xFFFE in the high half of word 0: some slot
  //     of the row lies outside its window (the exact columns decide); 0xFFFF there: some slot is not finite (the row
  //     can never be counted).  Null: the launch streams the float columns.
  const unsigned* qcol;
  double qbase[SXMC_MAX_QSLOTS];
  double qstep[SXMC_MAX_QSLOTS];
  // --- measurement build only (the gated step, DESIGN.md section 4 "one more look"): a word the step end of step k sets
  //     to k + 1 when the next proposal is written; a fill launched beside it waits for that before it reads its
  //     parameters.  Null otherwise; the product's kernels never look at it.
  unsigned* step_gate;
};

// One piece of fill work: units v0 + tid, + step, ... < v1 of member `sig` (a unit = SXMC_VEC samples).
struct SxSegment {
  int sig;
  int pad;
  unsigned long long v0, v1, step;
};

