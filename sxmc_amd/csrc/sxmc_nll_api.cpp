// sxmc_nll_api.cpp -- the NLL launch points with the reference's argument lists (nll_kernels.h:60-207) and, in the measurement build,
// the test hooks.
#include "sxmc_host.h"

using namespace sxhost;

extern "C" {

// ------------------------------------------------------------------------------ NLL launch points
static int check_launch(int grid, int block) {
  if (grid < 1 || block < 1 || block > 1024) return fail(SXMC_ERR_INVALID, "bad launch shape");
  return SXMC_OK;
}

int sxmc_launch_init_device_rngs(int grid, int block, sxmc_stream_t s, int nthreads, unsigned long long seed,
                                 sxmc_rng_state* d_state) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(d_state && (long long)grid * block >= nthreads, "init_device_rngs: grid*block < nthreads");
  SX_HIP(sx_nll_init_rngs(grid, block, (hipStream_t)s, nthreads, seed, d_state));
  return SXMC_OK;
}

int sxmc_launch_pick_new_vector(int grid, int block, sxmc_stream_t s, int nthreads, sxmc_rng_state* d_rng,
                                const float* d_jump_width, const double* d_current_vector,
                                double* d_proposed_vector) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_pick_new_vector(grid, block, (hipStream_t)s, nthreads, d_rng, d_jump_width, d_current_vector,
                                d_proposed_vector));
  return SXMC_OK;
}

int sxmc_launch_jump_decider(int grid, int block, sxmc_stream_t s, sxmc_rng_state* d_rng, double* d_nll_current,
                             const double* d_nll_proposed, double* d_v_current, const double* d_v_proposed,
                             unsigned nparameters, int* d_accepted, int* d_counter, float* d_jump_buffer) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_jump_decider(grid, block, (hipStream_t)s, d_rng, d_nll_current, d_nll_proposed, d_v_current,
                             d_v_proposed, nparameters, d_accepted, d_counter, d_jump_buffer));
  return SXMC_OK;
}

int sxmc_launch_nll_event_chunks(int grid, int block, sxmc_stream_t s, const float* d_lut, const double* d_pars,
                                 size_t ne, size_t ns, const double* d_nexpected, const unsigned* d_n_mc,
                                 const short* d_source_id, const unsigned* d_norms, double* d_sums) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(ns <= 4096, "too many signals");
  SX_HIP(sx_nll_event_chunks(grid, block, (hipStream_t)s, d_lut, d_pars, ne, ns, d_nexpected, d_n_mc, d_source_id,
                             d_norms, d_sums));
  return SXMC_OK;
}

int sxmc_launch_nll_event_reduce(int grid, int block, sxmc_stream_t s, size_t nthreads, const double* d_sums,
                                 double* d_total_sum) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(grid == 1, "nll_event_reduce runs in one workgroup");
  SX_HIP(sx_nll_event_reduce(block, (hipStream_t)s, nthreads, d_sums, d_total_sum));
  return SXMC_OK;
}

int sxmc_launch_nll_total(int grid, int block, sxmc_stream_t s, size_t nparameters, const double* d_pars,
                          size_t nsignals, size_t nsources, const double* d_means, const double* d_sigmas,
                          const double* d_events_total, const double* d_nexpected, const unsigned* d_n_mc,
                          const short* d_source_id, const unsigned* d_norms, double* d_nll) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_HIP(sx_nll_total((hipStream_t)s, nparameters, d_pars, nsignals, nsources, d_means, d_sigmas, d_events_total,
                      d_nexpected, d_n_mc, d_source_id, d_norms, d_nll));
  return SXMC_OK;
}

int sxmc_launch_finish_nll_jump_pick_combo(int grid, int block, sxmc_stream_t s, size_t npartial_sums,
                                           const double* d_sums, size_t nsignals, size_t nsources,
                                           const double* d_means, const double* d_sigmas, sxmc_rng_state* d_rng,
                                           double* d_nll_current, double* d_nll_proposed, double* d_v_current,
                                           double* d_v_proposed, int* d_accepted, int* d_counter,
                                           float* d_jump_buffer, int nparameters, const float* d_jump_width,
                                           const double* d_nexpected, const unsigned* d_n_mc,
                                           const short* d_source_id, const unsigned* d_norms, int debug_mode) {
  SX_FLUSH();
  SX_ORDER(s);
  if (int rc = check_launch(grid, block)) return rc;
  SX_REQUIRE(grid == 1, "finish_nll_jump_pick_combo runs in one workgroup");
  SX_HIP(sx_nll_finish_combo(block, (hipStream_t)s, npartial_sums, d_sums, nsignals, nsources, d_means, d_sigmas,
                             d_rng, d_nll_current, d_nll_proposed, d_v_current, d_v_proposed, d_accepted,
                             d_counter, d_jump_buffer, nparameters, d_jump_width, d_nexpected, d_n_mc, d_source_id,
                             d_norms, debug_mode != 0));
  return SXMC_OK;
}

#if SXMC_MEASURE
// (measurement build only) test hook: d_out[k] = d_x[k]^i, formed as the polynomial systematics form it
int sxmc_debug_pow_int(const double* d_x, int n, int i, double* d_out) {
  SX_FLUSH();
  SX_REQUIRE(d_x && d_out && n >= 0 && i >= 0 && i < 64, "bad arguments");
  SX_HIP(sx_launch_pow_int(d_x, n, i, d_out, nullptr));
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}

// (measurement build only) test hook: raw Philox output of state[0] (advances it by ndraws)
int sxmc_debug_philox_dump(sxmc_rng_state* d_state, unsigned* d_out, int ndraws) {
  SX_FLUSH();
  SX_HIP(sx_nll_philox_dump(nullptr, d_state, d_out, ndraws));
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}
#endif

}  // extern "C"

