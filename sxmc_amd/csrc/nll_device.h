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
// BD > 0: the workgroup's LOGICAL size -- the first BD lanes of a larger launch do the work of a workgroup of BD (the
// others have left): the fused step kernel runs its step-end roles in workgroups of the fill's size (see
// fill_step_kernel in pdfz_kernels.hip), and the partition of every sum must be the one a launch of BD lanes has.
template <int BD = 0>
__device__ __forceinline__ double block_sum(size_t n, const double* sums, double* s_wave /*[17]*/) {
  const unsigned bdim = BD > 0 ? (unsigned)BD : blockDim.x;
  // (a thread's terms are added in index order, as before; sixteen loads are in flight at a time -- the reference's
  //  launch shape hands 16 384 partial sums to 128 lanes, mcmc.cpp:37-45, and one load per addition made that 41 us)
  double t = 0.0;
  size_t i = threadIdx.x;
  const size_t bd = bdim;
  for (; i + 15 * bd < n; i += 16 * bd) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = sums[i + (size_t)k * bd];
#pragma unroll
    for (int k = 0; k < 16; k++) t += v[k];
  }
  for (; i < n; i += bd) t += sums[i];
  // a block size that is not a multiple of 64 leaves the last wave partly empty: what a shuffle reads from
  // a lane that does not exist is undefined, so those contributions are replaced by zero
  const int wave = threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int live = min(kWave, (int)bdim - wave * kWave);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const double o = __shfl_down(t, off, kWave);
    t += (lane + off < live) ? o : 0.0;
  }
  const int nwaves = ((int)bdim + kWave - 1) / kWave;
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
// proposal.  One workgroup, one launch per step, so it is built to be a few memory latencies long:
//   phase A (all lanes): everything that does not depend on the accept decision -- every input is
//     loaded once, lane j forms signal j's expected-rate term and lane i parameter i's constraint term
//     (the divisions), lane i draws its proposal deviate (lane 0 its uniform first: same consumption
//     order of generator 0 as jump_decider followed by pick_new_vector), the partial sums are reduced;
//   phase B (lane 0): the terms are added in the reference's order (so the value is the sequential
//     loop's, bit for bit), accept or reject;
//   phase C (all lanes): vector copy, jump-buffer append and next proposal, one parameter per lane.
// Vectors longer than kStage fall back to the plain sequential form.
constexpr int kStage = 256;

// Returns (to every thread) whether the proposal was accepted.
// `wait`: called by every thread of the workgroup, once, after everything that does not need the partial sums has
// been loaded and computed (phase A) and before the sums are read -- the cooperative step end (step_end_kernel) waits
// there for the workgroups that are still summing, so that phase A runs under their look-ups.
template <int BD = 0, typename Wait>
__device__ __forceinline__ bool finish_step_device_w(size_t npartial_sums, const double* sums, size_t nsignals,
                                                     size_t nsources, const double* means, const double* sigmas,
                                                     sxmc_rng_state* rng, double* nll_current,
                                                     double* nll_proposed, double* v_current, double* v_proposed,
                                                     int* accepted, int* counter, float* jump_buffer,
                                                     int nparameters, const float* jump_width,
                                                     const double* nexpected, const unsigned* n_mc,
                                                     const short* source_id, const unsigned* norms,
                                                     bool debug_mode, Wait&& wait) {
  __shared__ double s_wave[17];
  __shared__ double s_vprop[kStage], s_vcur[kStage], s_pen[kStage], s_z[kStage], s_term[kStage];
  __shared__ float s_jw[kStage];
  __shared__ unsigned char s_flag[kStage];  // bit 0: constraint term present, bit 1: negative source rate
  __shared__ int s_accept, s_count;
  __shared__ double s_nllcur, s_u;

  const unsigned bdim = BD > 0 ? (unsigned)BD : blockDim.x;
  const bool staged = nparameters <= kStage && nsignals <= (size_t)kStage;
  if (!staged) {   // (never with BD > 0: the host offers the fused kernel only for vectors that are staged)
    wait();
    double total_sum = block_sum<BD>(npartial_sums, sums, s_wave);
    if (threadIdx.x == 0) {
      nll_total_device(nparameters, nsignals, nsources, v_proposed, means, sigmas, &total_sum, nexpected, n_mc,
                       source_id, norms, nll_proposed);
      jump_decider_device(rng, nll_current, nll_proposed, v_current, v_proposed, nparameters, accepted, counter,
                          jump_buffer, debug_mode);
    }
    __threadfence_block();
    __syncthreads();
    pick_new_vector_device(nparameters, rng, jump_width, v_current, v_proposed);
    __syncthreads();
    return nll_current[0] == nll_proposed[0];   // (long vectors: accepted, or a tie, which changes nothing)
  }

  // ---- phase A
  double nc = 0.0;
  int count = 0;
  if (threadIdx.x == 0) {
    nc = nll_current[0];
    count = counter[0];
  }
  for (int i = threadIdx.x; i < nparameters; i += (int)bdim) {
    const double p = v_proposed[i], mean = means[i], sigma = sigmas[i];
    const float jw = jump_width[i];
    s_vprop[i] = p;
    s_vcur[i] = v_current[i];
    s_jw[i] = jw;
    unsigned char flag = 0;
    double pen = 0.0;
    if (sigma > 0) {  // nll_kernels.cpp:181-184
      const double x = (p - mean) / sigma;
      pen = 0.5 * x * x;
      flag |= 1;
    }
    if ((size_t)i < nsources && p < 0) flag |= 2;  // :176-179
    s_pen[i] = pen;
    s_flag[i] = flag;
    if (i == 0 || jw > 0) {
      sxmc_rng_state st = rng[i];
      if (i == 0) s_u = rng_uniform(&st);            // jump_decider's draw (nll_kernels.cpp:63)
      s_z[i] = (jw > 0) ? rng_normal(&st) : 0.0;     // pick_new_vector's draw (:42), free parameters only
      rng[i].offset = st.offset;
    }
  }
  for (int j = threadIdx.x; j < (int)nsignals; j += (int)bdim) {
    s_term[j] = v_proposed[source_id[j]] * nexpected[j] * norms[j] / n_mc[j];  // :169-172
  }
  wait();
  const double total_sum = block_sum<BD>(npartial_sums, sums, s_wave);  // barriers inside: LDS is visible after

  // ---- phase B
  if (threadIdx.x == 0) {
    double sum = -total_sum;
    bool bad = isnan(sum);
    if (!bad) {
      for (unsigned j = 0; j < nsignals; j++) sum += s_term[j];
      for (int i = 0; i < nparameters; i++) {
        if (s_flag[i] & 2) {
          bad = true;
          break;
        }
        if (s_flag[i] & 1) sum += s_pen[i];
      }
    }
    const double np = bad ? 1e18 : sum;
    nll_proposed[0] = np;
    const bool accept = debug_mode || (np < nc || s_u <= exp(nc - np));  // Metropolis, :69-77
    if (accept) {
      nll_current[0] = np;
      accepted[0] += 1;
    }
    counter[0] = count + 1;
    s_accept = accept ? 1 : 0;
    s_count = count;
    s_nllcur = accept ? np : nc;
  }
  __syncthreads();

  // ---- phase C
  const bool accept = s_accept != 0;
  const size_t row = (size_t)s_count * (size_t)(nparameters + 1);
  for (int i = threadIdx.x; i < nparameters; i += (int)bdim) {
    const double cur = accept ? s_vprop[i] : s_vcur[i];
    if (accept) v_current[i] = cur;
    jump_buffer[row + i] = (float)cur;
    v_proposed[i] = (s_jw[i] > 0) ? cur + s_jw[i] * s_z[i] : cur;  // :40-47
  }
  if (threadIdx.x == 0) jump_buffer[row + nparameters] = (float)s_nllcur;
  return accept;
}

__device__ __forceinline__ bool finish_step_device(size_t npartial_sums, const double* sums, size_t nsignals,
                                                   size_t nsources, const double* means, const double* sigmas,
                                                   sxmc_rng_state* rng, double* nll_current,
                                                   double* nll_proposed, double* v_current, double* v_proposed,
                                                   int* accepted, int* counter, float* jump_buffer,
                                                   int nparameters, const float* jump_width,
                                                   const double* nexpected, const unsigned* n_mc,
                                                   const short* source_id, const unsigned* norms,
                                                   bool debug_mode) {
  return finish_step_device_w<0>(npartial_sums, sums, nsignals, nsources, means, sigmas, rng, nll_current, nll_proposed,
                              v_current, v_proposed, accepted, counter, jump_buffer, nparameters, jump_width, nexpected,
                              n_mc, source_id, norms, debug_mode, [] {});
}

// What the NEXT call of finish_step_device would propose if the step it decides were rejected -- i.e. from the
// vector v_current as it stands: v_current + jump_width * z with the deviates that call will draw, WITHOUT
// advancing the generators (the call itself draws them again).  Generator 0 hands the decider its uniform first.
// The look-ahead walk evaluates this vector beside the proposal (sxmc_multigroup_lookahead_step_async).
__device__ __forceinline__ void peek_next_proposal_device(int nparameters, const sxmc_rng_state* rng,
                                                          const float* jump_width, const double* v_current,
                                                          double* out) {
  for (int i = threadIdx.x; i < nparameters; i += blockDim.x) {
    const float jw = jump_width[i];
    double v = v_current[i];
    if (jw > 0) {
      sxmc_rng_state st = rng[i];
      if (i == 0) (void)rng_uniform(&st);
      v = v + jw * rng_normal(&st);
    }
    out[i] = v;
  }
}

}  // namespace sxdev
