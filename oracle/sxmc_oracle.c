/*
 * sxmc_oracle.c -- CPU restatement of the sxmc NLL hot path.  TEST INFRASTRUCTURE ONLY.
 * See sxmc_oracle.h for scope, pinning status and the two documented deviations.
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#include "sxmc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* src/pdfz.cpp:200-215 */
int oracle_hist_geometry(int nobs, const double* lower, const double* upper,
                         const int* nbins, int* bin_stride, double* bin_volume) {
  double vol = 1.0f;
  for (int i = 0; i < nobs; i++) {
    vol *= (upper[i] - lower[i]) / nbins[i];
  }
  bin_stride[nobs - 1] = 1;
  for (int i = nobs - 2; i >= 0; i--) {
    bin_stride[i] = nbins[i + 1] * bin_stride[i + 1];
  }
  *bin_volume = vol;
  return bin_stride[0] * nbins[0];
}

/* src/pdfz.cpp:245-302 */
void oracle_set_eval_points(size_t npoints, const float* points, int nobs,
                            const double* lower, const double* upper,
                            const int* nbins, const int* bin_stride,
                            unsigned dataset, int* read_bins) {
  double bin_scale[ORACLE_MAX_NFIELDS];
  for (int iobs = 0; iobs < nobs; iobs++) {
    double span = upper[iobs] - lower[iobs];
    bin_scale[iobs] = nbins[iobs] / span;
  }

  for (size_t ipoint = 0; ipoint < npoints; ipoint++) {
    int in_pdf_domain = 1;
    int bin_id = 0;
    for (int iobs = 0; iobs < nobs; iobs++) {
      double element = points[(size_t)(nobs + 1) * ipoint + iobs];
      /* deviation: NaN is out of domain (header) */
      if (!(element >= lower[iobs] && element < upper[iobs])) {
        in_pdf_domain = 0;
        break;
      }
      bin_id += (int)((element - lower[iobs]) * bin_scale[iobs]) * bin_stride[iobs];
    }
    /* deviation: an index rounded up past the end (header) would make eval_pdf read
     * one-past-the-end; such a point is treated as outside the domain */
    if (in_pdf_domain && (unsigned)bin_id >= (unsigned)(bin_stride[0] * nbins[0])) {
      in_pdf_domain = 0;
    }
    /* dataset test overwrites bin_id, then out-of-domain wins (pdfz.cpp:289-300) */
    if (points[(size_t)(nobs + 1) * ipoint + nobs] != dataset) {
      bin_id = -2;
    }
    read_bins[ipoint] = in_pdf_domain ? bin_id : -1;
  }
}

/* src/pdfz.cpp:306-331 */
static inline void apply_systematic(const oracle_syst_t* syst, double* fields,
                                    const double* parameters, int param_stride) {
  double p = 0;
  for (short i = 0; i < syst->npars; i++) {
    p += (parameters[syst->pars[i] * param_stride] * pow(fields[syst->obs], (double)i));
  }
  switch (syst->type) {
    case ORACLE_SHIFT:
      fields[syst->obs] += p;
      break;
    case ORACLE_SCALE:
      fields[syst->obs] *= (1 + p);
      break;
    case ORACLE_CTSCALE:
      fields[syst->obs] = 1 + (fields[syst->obs] - 1) * (1 + p);
      break;
    case ORACLE_RESOLUTION_SCALE:
      fields[syst->obs] += (p * (fields[syst->obs] - fields[syst->extra_field]));
      break;
  }
}

/* The sample loop of src/pdfz.cpp:349-408 over [first, last), accumulating into bins/norm
 * with the CPU atomicAdd semantics of src/cuda_compat.h:20-24 (plain +=). */
static void bin_range(size_t first, size_t last, const float* data, int nobs, int nfields,
                      const int* bin_stride, const int* nbins, const double* lower,
                      const double* upper, int nsyst, const oracle_syst_t* syst,
                      const double* parameters, int param_stride, int total_nbins,
                      unsigned* bins, unsigned* norm) {
  double field_buffer[ORACLE_MAX_NFIELDS];
  double bin_scale[ORACLE_MAX_NFIELDS];
  for (int iobs = 0; iobs < nobs; iobs++) {
    bin_scale[iobs] = nbins[iobs] / (upper[iobs] - lower[iobs]);
  }

  unsigned thread_norm = 0;
  for (size_t isample = first; isample < last; isample++) {
    int in_pdf_domain = 1;
    int bin_id = 0;

    for (int ifield = 0; ifield < nfields; ifield++) {
      field_buffer[ifield] = data[isample * nfields + ifield];
    }
    for (int isyst = 0; isyst < nsyst; isyst++) {
      apply_systematic(syst + isyst, field_buffer, parameters, param_stride);
    }
    for (int iobs = 0; iobs < nobs; iobs++) {
      double element = field_buffer[iobs];
      /* deviation: NaN is out of domain (header) */
      if (!(element >= lower[iobs] && element < upper[iobs])) {
        in_pdf_domain = 0;
        break;
      }
      bin_id += (int)((element - lower[iobs]) * bin_scale[iobs]) * bin_stride[iobs];
    }
    if (in_pdf_domain) {
      /* deviation: drop the one-past-the-end write (header) */
      if ((unsigned)bin_id < (unsigned)total_nbins) {
        bins[bin_id] += 1;
      }
      thread_norm += 1;
    }
  }
  *norm += thread_norm;
}

/* src/pdfz.cpp:334-346 (zero_hist) then 349-408 (bin_samples) */
void oracle_bin_samples(size_t nsamples, const float* data, int nobs, int nfields,
                        const int* bin_stride, const int* nbins,
                        const double* lower, const double* upper,
                        int nsyst, const oracle_syst_t* syst,
                        const double* parameters, int param_stride,
                        int total_nbins, unsigned* bins, unsigned* norm) {
  *norm = 0;
  for (int i = 0; i < total_nbins; i++) {
    bins[i] = 0;
  }
  bin_range(0, nsamples, data, nobs, nfields, bin_stride, nbins, lower, upper, nsyst, syst,
            parameters, param_stride, total_nbins, bins, norm);
}

typedef struct {
  size_t first, last;
  const float* data;
  int nobs, nfields;
  const int* bin_stride;
  const int* nbins;
  const double* lower;
  const double* upper;
  int nsyst;
  const oracle_syst_t* syst;
  const double* parameters;
  int param_stride;
  int total_nbins;
  unsigned* bins;
  unsigned norm;
} mt_job_t;

static void* mt_worker(void* arg) {
  mt_job_t* j = (mt_job_t*)arg;
  bin_range(j->first, j->last, j->data, j->nobs, j->nfields, j->bin_stride, j->nbins,
            j->lower, j->upper, j->nsyst, j->syst, j->parameters, j->param_stride,
            j->total_nbins, j->bins, &j->norm);
  return NULL;
}

void oracle_bin_samples_mt(int nthreads, size_t nsamples, const float* data, int nobs,
                           int nfields, const int* bin_stride, const int* nbins,
                           const double* lower, const double* upper,
                           int nsyst, const oracle_syst_t* syst,
                           const double* parameters, int param_stride,
                           int total_nbins, unsigned* bins, unsigned* norm) {
  if (nthreads < 1) nthreads = 1;
  mt_job_t* jobs = (mt_job_t*)calloc((size_t)nthreads, sizeof(mt_job_t));
  pthread_t* tids = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    mt_job_t* j = &jobs[t];
    j->first = nsamples * (size_t)t / (size_t)nthreads;
    j->last = nsamples * (size_t)(t + 1) / (size_t)nthreads;
    j->data = data; j->nobs = nobs; j->nfields = nfields;
    j->bin_stride = bin_stride; j->nbins = nbins; j->lower = lower; j->upper = upper;
    j->nsyst = nsyst; j->syst = syst; j->parameters = parameters;
    j->param_stride = param_stride; j->total_nbins = total_nbins;
    j->bins = (unsigned*)calloc((size_t)total_nbins, sizeof(unsigned));
    j->norm = 0;
    pthread_create(&tids[t], NULL, mt_worker, j);
  }
  *norm = 0;
  memset(bins, 0, (size_t)total_nbins * sizeof(unsigned));
  for (int t = 0; t < nthreads; t++) {
    pthread_join(tids[t], NULL);
    for (int i = 0; i < total_nbins; i++) bins[i] += jobs[t].bins[i];
    *norm += jobs[t].norm;
    free(jobs[t].bins);
  }
  free(jobs);
  free(tids);
}

/* src/pdfz.cpp:411-436 */
void oracle_eval_pdf(size_t npoints, const int* read_bins, const unsigned* bins,
                     const unsigned* norm, double bin_volume,
                     float* output, int output_stride) {
  const double bin_norm = *norm * bin_volume;
  for (size_t ipoint = 0; ipoint < npoints; ipoint++) {
    int bin_id = read_bins[ipoint];
    double pdf_value = 0.0f;
    if (bin_id == -2) {
      pdf_value = 0.0;
    } else if (bin_id < 0) {
      pdf_value = nanf("");
    } else {
      pdf_value = bins[bin_id] / bin_norm;
    }
    output[(size_t)output_stride * ipoint] = pdf_value;
  }
}

/* src/nll_kernels.cpp:89-116 with hemiGetElementOffset()==0, stride==1 */
void oracle_nll_event_chunks(const float* lut, const double* pars, size_t ne, size_t ns,
                             const double* nexpected, const unsigned* n_mc,
                             const short* source_id, const unsigned* norms,
                             double* sums) {
  double sum = 0;
  for (size_t i = 0; i < ne; i++) {
    double s = 0;
    for (size_t j = 0; j < ns; j++) {
      float v = lut[j * ne + i];
      float eff = 1.0 * norms[j] / n_mc[j];
      short sid = source_id[j];
      s += pars[sid] * nexpected[j] * eff * (!isnan(v) ? v : 0);
    }
    if (s > 0) {
      sum += log(s);
    }
  }
  if (!isnan(sum)) {
    sums[0] = sum;
  }
}

/* src/nll_kernels.cpp:119-146, CPU branch */
void oracle_nll_event_reduce(size_t nthreads, const double* sums, double* total_sum) {
  double thread_sum = 0.0;
  for (size_t i = 0; i < nthreads; i++) {
    thread_sum += sums[i];
  }
  total_sum[0] = thread_sum;
}

/* src/nll_kernels.cpp:149-188 */
void oracle_nll_total(size_t nparameters, const double* pars, size_t nsignals,
                      size_t nsources, const double* means, const double* sigmas,
                      const double* events_total, const double* nexpected,
                      const unsigned* n_mc, const short* source_id,
                      const unsigned* norms, double* nll) {
  double sum = -events_total[0];
  if (isnan(sum)) {
    nll[0] = 1e18;
    return;
  }
  for (unsigned i = 0; i < nsignals; i++) {
    short sid = source_id[i];
    sum += pars[sid] * nexpected[i] * norms[i] / n_mc[i];
  }
  for (unsigned i = 0; i < nparameters; i++) {
    if (i < nsources && pars[i] < 0) {
      nll[0] = 1e18;
      return;
    }
    if (sigmas[i] > 0) {
      double x = (pars[i] - means[i]) / sigmas[i];
      sum += 0.5 * x * x;
    }
  }
  nll[0] = sum;
}

/* src/nll_kernels.cpp:56-86, u supplied */
void oracle_jump_decider(double u, double* nll_current, const double* nll_proposed,
                         double* v_current, const double* v_proposed,
                         unsigned nparameters, int* accepted, int* counter,
                         float* jump_buffer, int debug_mode) {
  double np = nll_proposed[0];
  double nc = nll_current[0];
  if (debug_mode || (np < nc || u <= exp(nc - np))) {
    nll_current[0] = np;
    for (unsigned i = 0; i < nparameters; i++) {
      v_current[i] = v_proposed[i];
    }
    accepted[0] += 1;
  }
  int count = counter[0];
  for (unsigned i = 0; i < nparameters; i++) {
    jump_buffer[count * (nparameters + 1) + i] = v_current[i];
  }
  jump_buffer[count * (nparameters + 1) + nparameters] = nll_current[0];
  counter[0] = count + 1;
}

/* src/nll_kernels.cpp:30-53, device form (current + width * z), z supplied */
void oracle_pick_new_vector(int n, const double* z, const float* jump_width,
                            const double* current_vector, double* proposed_vector) {
  for (int i = 0; i < n; i++) {
    if (jump_width[i] > 0) {
      proposed_vector[i] = current_vector[i] + jump_width[i] * z[i];
    } else {
      proposed_vector[i] = current_vector[i];
    }
  }
}
