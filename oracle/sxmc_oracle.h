/*
 * sxmc_oracle.h -- CPU restatement of the sxmc NLL hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle: a scalar, single-threaded, plain-C restatement of the
 * reference's CPU-mode loop (the `-DHEMI_CUDA_DISABLE` build of /root/reference/src/pdfz.cpp
 * and nll_kernels.cpp, where every "kernel" is a serial loop with offset 0 / stride 1).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (sxmc_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * Pinning status:
 *   - pdfz part (SetEvalPoints, bin_samples+apply_systematic, eval_pdf): PINNED by the
 *     reference's own gtest known answers (test/test_pdfz.cpp, test_pdfz_2d.cpp,
 *     test_pdfz_syst.cpp), transcribed as data into tests/golden/pdfz_known_answers.json.
 *   - nll part (nll_event_chunks, nll_event_reduce, nll_total, jump_decider,
 *     pick_new_vector): PARITY UNPINNED.  The reference has no test, fixture or golden
 *     vector for these functions, and the reference cannot be built in this image without
 *     writing stand-ins for hemi / ROOT / CUDA headers, which is not allowed.  They are
 *     restated line by line from nll_kernels.cpp:30-188 and checked only by hand-derived
 *     closed-form cases.
 *
 * Compile with: gcc -O2 -ffp-contract=off (x86-64 SSE2 double arithmetic, no FMA
 * contraction, no fast-math), which is the arithmetic of the reference's g++ CPU build.
 *
 * Behaviour where the reference has undefined behaviour (documented deviations):
 *   - a NaN field passes the reference's `x < lower || x >= upper` test and then hits
 *     `(int)NaN` (pdfz.cpp:391-397): here a NaN field is OUT of the domain (rejected).
 *   - `(x-lower)*scale` can round up to exactly nbins for x one ulp below upper, which
 *     makes the reference write one-past-the-end (pdfz.cpp:396-402): here the index
 *     arithmetic is kept unclamped exactly as written, the sample still counts in `norm`,
 *     and the bin increment is dropped iff the flat index falls outside [0, total_nbins).
 *     An evaluation point whose flat index falls outside that range (the reference would read
 *     one-past-the-end in eval_pdf) is treated as outside the domain (-1 -> NaN).
 */
#ifndef SXMC_ORACLE_H
#define SXMC_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_NFIELDS 10      /* pdfz.cpp:17 */
#define ORACLE_MAX_SYST_PARS 8

/* pdfz.h:111-116 */
enum { ORACLE_SHIFT = 0, ORACLE_SCALE = 1, ORACLE_RESOLUTION_SCALE = 2, ORACLE_CTSCALE = 3 };

/* pdfz.cpp:48-54 (SystematicDescriptor), pars inlined instead of pointed to */
typedef struct {
  short type;
  short obs;
  short extra_field;
  short npars;
  short pars[ORACLE_MAX_SYST_PARS];
} oracle_syst_t;

/* pdfz.cpp:200-215: bin strides (row-major, last dim stride 1), total bins, bin volume.
 * Returns total_nbins. */
int oracle_hist_geometry(int nobs, const double* lower, const double* upper,
                         const int* nbins, int* bin_stride, double* bin_volume);

/* pdfz.cpp:245-302: points has npoints rows of (nobs+1) floats, last = dataset id. */
void oracle_set_eval_points(size_t npoints, const float* points, int nobs,
                            const double* lower, const double* upper,
                            const int* nbins, const int* bin_stride,
                            unsigned dataset, int* read_bins);

/* pdfz.cpp:334-346 + 349-408 (+ apply_systematic 306-331): zero, then fill.
 * data is row-major [nsamples][nfields] float; parameters already offset by param_offset. */
void oracle_bin_samples(size_t nsamples, const float* data, int nobs, int nfields,
                        const int* bin_stride, const int* nbins,
                        const double* lower, const double* upper,
                        int nsyst, const oracle_syst_t* syst,
                        const double* parameters, int param_stride,
                        int total_nbins, unsigned* bins, unsigned* norm);

/* Same result as oracle_bin_samples using nthreads host threads with private
 * histograms summed at the end (integer counters: order independent). */
void oracle_bin_samples_mt(int nthreads, size_t nsamples, const float* data, int nobs,
                           int nfields, const int* bin_stride, const int* nbins,
                           const double* lower, const double* upper,
                           int nsyst, const oracle_syst_t* syst,
                           const double* parameters, int param_stride,
                           int total_nbins, unsigned* bins, unsigned* norm);

/* pdfz.cpp:411-436 */
void oracle_eval_pdf(size_t npoints, const int* read_bins, const unsigned* bins,
                     const unsigned* norm, double bin_volume,
                     float* output, int output_stride);

/* nll_kernels.cpp:89-116, CPU mode: one partial sum over all events -> sums[0] */
void oracle_nll_event_chunks(const float* lut, const double* pars, size_t ne, size_t ns,
                             const double* nexpected, const unsigned* n_mc,
                             const short* source_id, const unsigned* norms,
                             double* sums);

/* nll_kernels.cpp:119-146, CPU mode */
void oracle_nll_event_reduce(size_t nthreads, const double* sums, double* total_sum);

/* nll_kernels.cpp:149-188 */
void oracle_nll_total(size_t nparameters, const double* pars, size_t nsignals,
                      size_t nsources, const double* means, const double* sigmas,
                      const double* events_total, const double* nexpected,
                      const unsigned* n_mc, const short* source_id,
                      const unsigned* norms, double* nll);

/* nll_kernels.cpp:56-86 with the uniform deviate u supplied by the caller */
void oracle_jump_decider(double u, double* nll_current, const double* nll_proposed,
                         double* v_current, const double* v_proposed,
                         unsigned nparameters, int* accepted, int* counter,
                         float* jump_buffer, int debug_mode);

/* nll_kernels.cpp:30-53 with the unit normal deviates z[i] supplied by the caller
 * (GPU-mode form: proposed = current + width * z) */
void oracle_pick_new_vector(int n, const double* z, const float* jump_width,
                            const double* current_vector, double* proposed_vector);

#ifdef __cplusplus
}
#endif
#endif
