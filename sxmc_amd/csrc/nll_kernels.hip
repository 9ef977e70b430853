// nll_kernels.hip -- gfx950 kernels for the NLL half of the MCMC step (wave64 only).
//
// What is computed is fixed by the reference (file:line relative to /root/reference):
//   init_device_rngs            src/nll_kernels.cpp:18-27
//   pick_new_vector(_device)    src/nll_kernels.cpp:30-53, 191-197
//   jump_decider(_device)       src/nll_kernels.cpp:56-86, 200-206
//   nll_event_chunks            src/nll_kernels.cpp:89-116
//   nll_event_reduce(_device)   src/nll_kernels.cpp:119-146, 209-212
//   nll_total(_device)          src/nll_kernels.cpp:149-188, 215-227
//   finish_nll_jump_pick_combo  src/nll_kernels.cpp:230-271
// The reduction is a wave64 shuffle tree followed by one LDS hop across waves (the reference's
// shared-memory tree has its barrier outside the loop, nll_kernels.cpp:139-143).  Random numbers
// come from a counter-based Philox4x32-10 stream per parameter instead of cuRAND XORWOW: chains
// are statistically, not bitwise, comparable with the reference (whose CPU and GPU targets do
// not agree with each other either).
#include "nll_device.h"

#pragma clang fp contract(off)

namespace {

using namespace sxdev;

__global__ void init_rngs_kernel(int n, unsigned long long seed, sxmc_rng_state* state) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  state[idx].seed = seed;
  state[idx].subsequence = (unsigned long long)idx;
  state[idx].offset = 0;
  state[idx].reserved = 0;
}

#ifndef SXMC_MEASURE
#define SXMC_MEASURE 0
#endif
#if SXMC_MEASURE
// (measurement build only) raw generator output for tests: out[4*i..4*i+3] = draw i of state[0]
__global__ void philox_dump_kernel(sxmc_rng_state* state, unsigned* out, int ndraws) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int i = 0; i < ndraws; i++) {
    const Philox4 r = rng_next(&state[0]);
    out[4 * i + 0] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
  }
}
#endif

// ------------------------------------------------------------------------------------ kernels
__global__ void pick_new_vector_kernel(int nthreads, sxmc_rng_state* rng, const float* jump_width,
                                       const double* current_vector, double* proposed_vector) {
  pick_new_vector_device(nthreads, rng, jump_width, current_vector, proposed_vector);
}

__global__ void jump_decider_kernel(sxmc_rng_state* rng, double* nll_current, const double* nll_proposed,
                                    double* v_current, const double* v_proposed, unsigned nparameters,
                                    int* accepted, int* counter, float* jump_buffer) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    jump_decider_device(rng, nll_current, nll_proposed, v_current, v_proposed, nparameters, accepted,
                        counter, jump_buffer, false);
  }
}

__global__ void nll_event_chunks_kernel(const float* __restrict__ lut, const double* __restrict__ pars,
                                        size_t ne, size_t ns, const double* __restrict__ nexpected,
                                        const unsigned* __restrict__ n_mc,
                                        const short* __restrict__ source_id,
                                        const unsigned* __restrict__ norms, double* sums) {
  extern __shared__ double s_coef[];  // [ns]: pars[sid] * nexpected * eff, the event-independent factor
  for (size_t j = threadIdx.x; j < ns; j += blockDim.x) {
    const float eff = (float)(1.0 * norms[j] / n_mc[j]);
    s_coef[j] = pars[source_id[j]] * nexpected[j] * eff;
  }
  __syncthreads();
  const size_t offset = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // The reference's launch shape (64 x 256 lanes, mcmc.cpp:37-45) gives a lane a handful of events and the kernel is
  // nothing but their loads' latency: the table values of an event are asked for sixteen signals at a time (indices
  // past the last signal repeat it: no branches between the loads), and the NEXT event's first sixteen while this
  // event's logarithm is taken.  The terms are added in signal order, the events in index order, as before.
  constexpr int U = 16;
  double sum = 0;
  float nxt[U];
  auto request = [&](size_t i, size_t j0, float* v) {
    const size_t ic = i < ne ? i : (ne ? ne - 1 : 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t j = j0 + (size_t)u < ns ? j0 + (size_t)u : ns - 1;
      v[u] = lut[j * ne + ic];
    }
  };
  if (ne == 0 || ns == 0) {
    if (offset < stride) sums[offset] = 0.0;
    return;
  }
  request(offset, 0, nxt);
  for (size_t i = offset; i < ne; i += stride) {
    double s = 0;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = nxt[u];
    for (size_t j0 = 0; j0 < ns; j0 += U) {
      if (j0 > 0) request(i, j0, v);
      if (j0 + U >= ns) request(i + stride, 0, nxt);   // (past the last event: a clamped, unused read)
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (j0 + (size_t)u < ns) s = s + s_coef[j0 + u] * (double)(!isnan(v[u]) ? v[u] : 0.0f);  // NaN: empty histogram
      }
    }
    if (s > 0) sum += log(s);
  }
  if (!isnan(sum)) sums[offset] = sum;
}

__global__ void nll_event_reduce_kernel(size_t nthreads, const double* sums, double* total_sum) {
  __shared__ double s_wave[17];
  const double t = block_sum(nthreads, sums, s_wave);
  if (threadIdx.x == 0) total_sum[0] = t;
}

__global__ void nll_total_kernel(size_t npars, const double* pars, size_t nsignals, size_t nsources,
                                 const double* means, const double* sigmas, const double* events_total,
                                 const double* nexpected, const unsigned* n_mc, const short* source_id,
                                 const unsigned* norms, double* nll) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    nll_total_device(npars, nsignals, nsources, pars, means, sigmas, events_total, nexpected, n_mc,
                     source_id, norms, nll);
  }
}

__global__ void finish_nll_jump_pick_combo_kernel(size_t npartial_sums, const double* sums, size_t nsignals,
                                                  size_t nsources, const double* means, const double* sigmas,
                                                  sxmc_rng_state* rng, double* nll_current,
                                                  double* nll_proposed, double* v_current, double* v_proposed,
                                                  int* accepted, int* counter, float* jump_buffer,
                                                  int nparameters, const float* jump_width,
                                                  const double* nexpected, const unsigned* n_mc,
                                                  const short* source_id, const unsigned* norms,
                                                  bool debug_mode) {
  finish_step_device(npartial_sums, sums, nsignals, nsources, means, sigmas, rng, nll_current, nll_proposed,
                     v_current, v_proposed, accepted, counter, jump_buffer, nparameters, jump_width, nexpected,
                     n_mc, source_id, norms, debug_mode);
}

}  // namespace

// ------------------------------------------------------------------------------------ launchers
extern "C" {

hipError_t sx_nll_init_rngs(int grid, int block, hipStream_t s, int n, unsigned long long seed,
                            sxmc_rng_state* st) {
  hipLaunchKernelGGL(init_rngs_kernel, dim3(grid), dim3(block), 0, s, n, seed, st);
  return hipGetLastError();
}

#if SXMC_MEASURE
hipError_t sx_nll_philox_dump(hipStream_t s, sxmc_rng_state* st, unsigned* out, int ndraws) {
  hipLaunchKernelGGL(philox_dump_kernel, dim3(1), dim3(64), 0, s, st, out, ndraws);
  return hipGetLastError();
}
#endif

hipError_t sx_nll_pick_new_vector(int grid, int block, hipStream_t s, int n, sxmc_rng_state* rng,
                                  const float* jw, const double* cur, double* prop) {
  hipLaunchKernelGGL(pick_new_vector_kernel, dim3(grid), dim3(block), 0, s, n, rng, jw, cur, prop);
  return hipGetLastError();
}

hipError_t sx_nll_jump_decider(int grid, int block, hipStream_t s, sxmc_rng_state* rng, double* nc,
                               const double* np, double* vc, const double* vp, unsigned n, int* acc,
                               int* cnt, float* jb) {
  hipLaunchKernelGGL(jump_decider_kernel, dim3(grid), dim3(block), 0, s, rng, nc, np, vc, vp, n, acc, cnt, jb);
  return hipGetLastError();
}

hipError_t sx_nll_event_chunks(int grid, int block, hipStream_t s, const float* lut, const double* pars,
                               size_t ne, size_t ns, const double* nexpected, const unsigned* n_mc,
                               const short* source_id, const unsigned* norms, double* sums) {
  hipLaunchKernelGGL(nll_event_chunks_kernel, dim3(grid), dim3(block), ns * sizeof(double), s, lut, pars, ne,
                     ns, nexpected, n_mc, source_id, norms, sums);
  return hipGetLastError();
}

hipError_t sx_nll_event_reduce(int block, hipStream_t s, size_t n, const double* sums, double* total) {
  hipLaunchKernelGGL(nll_event_reduce_kernel, dim3(1), dim3(block), 0, s, n, sums, total);
  return hipGetLastError();
}

hipError_t sx_nll_total(hipStream_t s, size_t npars, const double* pars, size_t nsignals, size_t nsources,
                        const double* means, const double* sigmas, const double* events_total,
                        const double* nexpected, const unsigned* n_mc, const short* source_id,
                        const unsigned* norms, double* nll) {
  hipLaunchKernelGGL(nll_total_kernel, dim3(1), dim3(64), 0, s, npars, pars, nsignals, nsources, means, sigmas,
                     events_total, nexpected, n_mc, source_id, norms, nll);
  return hipGetLastError();
}

hipError_t sx_nll_finish_combo(int block, hipStream_t s, size_t npartial, const double* sums, size_t nsignals,
                               size_t nsources, const double* means, const double* sigmas, sxmc_rng_state* rng,
                               double* nll_current, double* nll_proposed, double* v_current, double* v_proposed,
                               int* accepted, int* counter, float* jump_buffer, int nparameters,
                               const float* jump_width, const double* nexpected, const unsigned* n_mc,
                               const short* source_id, const unsigned* norms, bool debug_mode) {
  hipLaunchKernelGGL(finish_nll_jump_pick_combo_kernel, dim3(1), dim3(block), 0, s, npartial, sums, nsignals,
                     nsources, means, sigmas, rng, nll_current, nll_proposed, v_current, v_proposed, accepted,
                     counter, jump_buffer, nparameters, jump_width, nexpected, n_mc, source_id, norms,
                     debug_mode);
  return hipGetLastError();
}

}  // extern "C"
