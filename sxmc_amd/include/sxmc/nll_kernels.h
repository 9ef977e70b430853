// nll_kernels.h -- the reference's NLL / MCMC-step kernel entry points (src/nll_kernels.h:44-207) on
// MI355X.  Each function has the reference kernel's name and argument list, preceded by the launch
// shape the caller hands to its launch macro -- (grid, block, dynamic shared memory bytes, stream) --
// exactly as in mcmc.cpp:252-256, 314-348, 396-414:
//
//     SXMC_KERNEL_LAUNCH(nll_event_chunks, nnllblocks, nllblocksize, 0, 0, lut, pars, ...);
//
// The shared-memory argument is accepted and ignored (the kernels size their own LDS).  All array
// arguments are device pointers, e.g. from sxmc::DeviceArray accessors.
#pragma once

#include <cstddef>

#include "device_array.h"

/** One counter-based generator state per parameter; replaces curandStateXORWOW (nll_kernels.h:25-29). */
typedef sxmc_rng_state RNGState;

namespace sxmc {
namespace launch {

/** init_device_rngs (nll_kernels.h:44-45, raw <<<>>> launch at mcmc.cpp:121-126) */
inline void init_device_rngs(int grid, int block, size_t, sxmc_stream_t stream, int nthreads,
                             unsigned long long seed, RNGState* state) {
  check(sxmc_launch_init_device_rngs(grid, block, stream, nthreads, seed, state));
}

/** pick_new_vector (nll_kernels.h:60-63) */
inline void pick_new_vector(int grid, int block, size_t, sxmc_stream_t stream, int nthreads, RNGState* rng,
                            const float* jump_width, const double* current_vector, double* proposed_vector) {
  check(sxmc_launch_pick_new_vector(grid, block, stream, nthreads, rng, jump_width, current_vector,
                                    proposed_vector));
}

/** jump_decider (nll_kernels.h:86-89) */
inline void jump_decider(int grid, int block, size_t, sxmc_stream_t stream, RNGState* rng, double* nll_current,
                         const double* nll_proposed, double* v_current, const double* v_proposed,
                         unsigned nparameters, int* accepted, int* counter, float* jump_buffer) {
  check(sxmc_launch_jump_decider(grid, block, stream, rng, nll_current, nll_proposed, v_current, v_proposed,
                                 nparameters, accepted, counter, jump_buffer));
}

/** nll_event_chunks (nll_kernels.h:107-114) */
inline void nll_event_chunks(int grid, int block, size_t, sxmc_stream_t stream, const float* lut,
                             const double* pars, const size_t ne, const size_t ns, const double* nexpected,
                             const unsigned* n_mc, const short* source_id, const unsigned* norms, double* sums) {
  check(sxmc_launch_nll_event_chunks(grid, block, stream, lut, pars, ne, ns, nexpected, n_mc, source_id, norms,
                                     sums));
}

/** nll_event_reduce (nll_kernels.h:126-127) */
inline void nll_event_reduce(int grid, int block, size_t, sxmc_stream_t stream, const size_t nthreads,
                             const double* sums, double* total_sum) {
  check(sxmc_launch_nll_event_reduce(grid, block, stream, nthreads, sums, total_sum));
}

/** nll_total (nll_kernels.h:149-159) */
inline void nll_total(int grid, int block, size_t, sxmc_stream_t stream, const size_t nparameters,
                      const double* pars, const size_t nsignals, const size_t nsources, const double* means,
                      const double* sigmas, const double* events_total, const double* nexpected,
                      const unsigned* n_mc, const short* source_id, const unsigned* norms, double* nll) {
  check(sxmc_launch_nll_total(grid, block, stream, nparameters, pars, nsignals, nsources, means, sigmas,
                              events_total, nexpected, n_mc, source_id, norms, nll));
}

/** finish_nll_jump_pick_combo (nll_kernels.h:190-207) */
inline void finish_nll_jump_pick_combo(int grid, int block, size_t, sxmc_stream_t stream,
                                       const size_t npartial_sums, const double* sums, const size_t nsignals,
                                       const size_t nsources, const double* means, const double* sigmas,
                                       RNGState* rng, double* nll_current, double* nll_proposed,
                                       double* v_current, double* v_proposed, int* accepted, int* counter,
                                       float* jump_buffer, int nparameters, const float* jump_width,
                                       const double* nexpected, const unsigned* n_mc, const short* source_id,
                                       const unsigned* norms, const bool debug_mode = false) {
  check(sxmc_launch_finish_nll_jump_pick_combo(grid, block, stream, npartial_sums, sums, nsignals, nsources,
                                               means, sigmas, rng, nll_current, nll_proposed, v_current,
                                               v_proposed, accepted, counter, jump_buffer, nparameters,
                                               jump_width, nexpected, n_mc, source_id, norms,
                                               debug_mode ? 1 : 0));
}

}  // namespace launch
}  // namespace sxmc

#define SXMC_KERNEL_LAUNCH(name, grid, block, shmem, stream, ...) \
  ::sxmc::launch::name((grid), (block), (shmem), (sxmc_stream_t)(size_t)(stream), __VA_ARGS__)
