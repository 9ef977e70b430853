// nll_device.h -- device functions of the NLL half of the MCMC step, shared by nll_kernels.hip (the
// reference's separate entry points) and pdfz_kernels.hip (the fused lookup + event sum + step end).
// See nll_kernels.hip for the reference lines each one restates.
#pragma once

#include "sxmc_device.h"

#pragma clang fp contract(off)

namespace sxdev {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------ RNG
struct Philox4 {
  unsigned x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long counter_lo,
                                                 unsigned long long counter_hi,
                                                 unsigned long long key) {
  unsigned c0 = (unsigned)counter_lo, c1 = (unsigned)(counter_lo >> 32);
  unsigned c2 = (unsigned)counter_hi, c3 = (unsigned)(counter_hi >> 32);
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ Philox4 rng_next(sxmc_rng_state* st) {
  const Philox4 r = philox4x32_10(st->offset, st->subsequence, st->seed);
  st->offset += 1;
  return r;
}

// (0, 1], like curand_uniform
__device__ __forceinline__ double rng_uniform(sxmc_rng_state* st) {
  const Philox4 r = rng_next(st);
  return ((double)r.x + 1.0) * 2.3283064365386963e-10;
}

// unit normal (Box-Muller in double precision)
__device__ __forceinline__ double rng_normal(sxmc_rng_state* st) {
  const Philox4 r = rng_next(st);
  const double u1 = ((double)r.x + 1.0) * 2.3283064365386963e-10;  // (0,1]
  const double u2 = (double)r.y * 2.3283064365386963e-10;          // [0,1)
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// ------------------------------------------------------------------------------------ device parts
__device__ __forceinline__ void pick_new_vector_device(int n, sxmc_rng_state* rng,
                                                       const float* jump_width,
                                                       const double* current_vector,
                                                       double* proposed_vector) {
  const int offset = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = gridDim.x * blockDim.x;
  for (int i = offset; i < n; i += stride) {
    if (jump_width[i] > 0) {  // fixed parameters carry width -1 (mcmc.cpp:204-207)
      const double u = rng_normal(&rng[i]);
      proposed_vector[i] = current_vector[i] + jump_width[i] * u;
    } else {
      proposed_vector[i] = current_vector[i];
    }
  }
}

__device__ __forceinline__ void jump_decider_device(sxmc_rng_state* rng, double* nll_current,
                                                    const double* nll_proposed, double* v_current,
                                                    const double* v_proposed, unsigned nparameters,
                                                    int* accepted, int* counter, float* jump_buffer,
                                                    bool debug_mode) {
  const double u = rng_uniform(&rng[0]);
  const double np = nll_proposed[0];
  const double nc = nll_current[0];
  if (debug_mode || (np < nc || u <= exp(nc - np))) {  // Metropolis, nll_kernels.cpp:69-77
    nll_current[0] = np;
    for (unsigned i = 0; i < nparameters; i++) v_current[i] = v_proposed[i];
    accepted[0] += 1;
  }
  const int count = counter[0];
  for (unsigned i = 0; i < nparameters; i++) {
    jump_buffer[count * (nparameters + 1) + i] = (float)v_current[i];
  }
  jump_buffer[count * (nparameters + 1) + nparameters] = (float)nll_current[0];
  counter[0] = count + 1;
}

__device__ __forceinline__ void nll_total_device(size_t nparameters, size_t nsignals, size_t nsources,
                                                 const double* pars, const double* means,
                                                 const double* sigmas, const double* events_total,
                                                 const double* nexpected, const unsigned* n_mc,
                                                 const short* source_id, const unsigned* norms,
                                                 double* nll) {
  double sum = -events_total[0];
  if (isnan(sum)) {
    nll[0] = 1e18;
    return;
  }
  for (unsigned i = 0; i < nsignals; i++) {
    const short sid = source_id[i];
    sum += pars[sid] * nexpected[i] * norms[i] / n_mc[i];
  }
  for (unsigned i = 0; i < nparameters; i++) {
    if (i < nsources && pars[i] < 0) {  // steep penalty for negative rates
      nll[0] = 1e18;
      return;
    }
    if (sigmas[i] > 0) {
      const double x = (pars[i] - means[i]) / sigmas[i];
      sum += 0.5 * x * x;
    }
  }
  nll[0] = sum;
}

// Sum of sums[0..n) over the workgroup; the total is returned to every thread.
__device__ __forceinline__ double block_sum(size_t n, const double* sums, double* s_wave /*[17]*/) {
  double t = 0.0;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) t += sums[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_down(t, off, kWave);
  const int wave = threadIdx.x / kWave;
  const int nwaves = (blockDim.x + kWave - 1) / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) s_wave[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < nwaves; w++) tot += s_wave[w];
    s_wave[16] = tot;
  }
  __syncthreads();
  return s_wave[16];
}

// The fused end of an MCMC step (nll_kernels.cpp:230-271): reduce the event partial sums, total the
// NLL at the proposed vector, Metropolis accept/reject + append to the jump buffer, draw the next
// proposal.  One workgroup.  Every small array is staged into LDS by all lanes at once and the
// vector copy / buffer append / proposal run one parameter per lane: the kernel is a handful of
// memory latencies long instead of one per element.
constexpr int kStage = 256;

__device__ __forceinline__ void finish_step_device(size_t npartial_sums, const double* sums, size_t nsignals,
                                                   size_t nsources, const double* means, const double* sigmas,
                                                   sxmc_rng_state* rng, double* nll_current,
                                                   double* nll_proposed, double* v_current, double* v_proposed,
                                                   int* accepted, int* counter, float* jump_buffer,
                                                   int nparameters, const float* jump_width,
                                                   const double* nexpected, const unsigned* n_mc,
                                                   const short* source_id, const unsigned* norms,
                                                   bool debug_mode) {
  __shared__ double s_wave[17];
  __shared__ double s_vprop[kStage], s_means[kStage], s_sigmas[kStage], s_nexp[kStage];
  __shared__ unsigned s_nmc[kStage], s_norms[kStage];
  __shared__ short s_sid[kStage];
  __shared__ int s_accept, s_count;
  __shared__ double s_nllcur;

  const bool staged = nparameters <= kStage && nsignals <= (size_t)kStage;
  if (staged) {
    for (int i = threadIdx.x; i < nparameters; i += blockDim.x) {
      s_vprop[i] = v_proposed[i];
      s_means[i] = means[i];
      s_sigmas[i] = sigmas[i];
    }
    for (int i = threadIdx.x; i < (int)nsignals; i += blockDim.x) {
      s_nexp[i] = nexpected[i];
      s_nmc[i] = n_mc[i];
      s_norms[i] = norms[i];
      s_sid[i] = source_id[i];
    }
  }
  double total_sum = block_sum(npartial_sums, sums, s_wave);  // barriers inside: staging is visible after

  if (threadIdx.x == 0) {
    if (staged) {
      nll_total_device(nparameters, nsignals, nsources, s_vprop, s_means, s_sigmas, &total_sum, s_nexp, s_nmc,
                       s_sid, s_norms, nll_proposed);
    } else {
      nll_total_device(nparameters, nsignals, nsources, v_proposed, means, sigmas, &total_sum, nexpected, n_mc,
                       source_id, norms, nll_proposed);
    }
    // jump_decider_device (nll_kernels.cpp:56-86), scalar part
    const double u = rng_uniform(&rng[0]);
    const double np = nll_proposed[0];
    const double nc = nll_current[0];
    const bool accept = debug_mode || (np < nc || u <= exp(nc - np));
    if (accept) {
      nll_current[0] = np;
      accepted[0] += 1;
    }
    const int count = counter[0];
    counter[0] = count + 1;
    s_accept = accept ? 1 : 0;
    s_count = count;
    s_nllcur = accept ? np : nc;
  }
  __threadfence_block();
  __syncthreads();

  const bool accept = s_accept != 0;
  const size_t row = (size_t)s_count * (size_t)(nparameters + 1);
  for (int i = threadIdx.x; i < nparameters; i += blockDim.x) {
    // accepted: v_current <- v_proposed; every step: append v_current (as float) to the jump buffer
    const double cur = accept ? (staged ? s_vprop[i] : v_proposed[i]) : v_current[i];
    if (accept) v_current[i] = cur;
    jump_buffer[row + i] = (float)cur;
    // pick_new_vector_device (nll_kernels.cpp:30-53): next proposal around the (new) current vector
    if (jump_width[i] > 0) {
      const double z = rng_normal(&rng[i]);
      v_proposed[i] = cur + jump_width[i] * z;
    } else {
      v_proposed[i] = cur;
    }
  }
  if (threadIdx.x == 0) jump_buffer[row + nparameters] = (float)s_nllcur;
}


}  // namespace sxdev
