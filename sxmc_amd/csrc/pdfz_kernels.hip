// pdfz_kernels.hip -- gfx950 kernels for the pdfz::EvalHist half of the NLL evaluation:
// zero, histogram fill with per-sample systematics, PDF lookup.  Written for CDNA4 only
// (wave64, 160 KiB LDS per CU, 256 CUs); no other target is supported.
//
// What is computed is fixed by the reference (file:line relative to /root/reference):
//   zero_hist    src/pdfz.cpp:334-346
//   bin_samples  src/pdfz.cpp:349-408  (+ apply_systematic 306-331)
//   eval_pdf     src/pdfz.cpp:411-436
// How it is computed is not: the samples are stored column-major and read with one 16-byte
// load per lane per column, each workgroup owns a contiguous slice of the concatenated sample
// index space of ALL signals, accumulates into an LDS-private uint32 histogram with ds_add_u32
// and flushes only its non-zero bins with one global atomic each; in-domain counts are kept in
// a register per lane and reduced once per workgroup.
//
// Exactness: double arithmetic is IEEE add/sub/mul with NO contraction (compile flag
// -ffp-contract=off plus the pragma below), in the reference's operation order, so bin
// indices are bit-identical to the reference CPU loop.  x^i of the polynomial systematics is
// rounded once (pow_step in fill_kernels.inc.h), like libm's pow for these exponents.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstring>
#include <vector>

#include "nll_device.h"

#include "fill_kernels.inc.h"

#pragma clang fp contract(off)

namespace {

using namespace sxfill;


// ------------------------------------------------------------------------------------ zero
// bins <- 0, norm <- 0 for every member (blockIdx.y = member).
__global__ __launch_bounds__(256) void zero_kernel(const SxSignalDesc* __restrict__ descs, unsigned* ticket) {
  const SxSignalDesc& d = descs[blockIdx.y];
  unsigned* bins = d.bins;
  const unsigned n = (unsigned)d.total_nbins;
  const unsigned n4 = n >> 2;
  uint4* b4 = reinterpret_cast<uint4*>(bins);  // hipMalloc'd: 256-byte aligned
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    b4[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
    if (threadIdx.x == 0) *d.norm = 0u;
    if (threadIdx.x == 0 && blockIdx.y == 0 && ticket) *ticket = 0u;  // arrival counter of the fused step end
  }
}


// Shape-agnostic fallback (any nobs/nslot up to SXMC_MAX_NFIELDS): one sample per lane per
// iteration, fields in a dynamically indexed array.  Correctness path for shapes without a
// specialization; same arithmetic.
template <bool LDS_HIST>
__global__ __launch_bounds__(1024) void fill_kernel_generic(const SxSignalDesc* __restrict__ descs,
                                                            const SxSegment* __restrict__ segs,
                                                            const unsigned* __restrict__ blk_off,
                                                            unsigned hist_words) {
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned nthreads = blockDim.x;
  unsigned* s_norm = lds;
  unsigned* hist = lds + 4;

  bool lds_clean = false;
  const unsigned seg_end = blk_off[blockIdx.x + 1];

  for (unsigned si = blk_off[blockIdx.x]; si < seg_end; ++si) {
    const SxSegment& sg = segs[si];
    const SxSignalDesc& d = descs[sg.sig];
    const unsigned long long v0 = sg.v0;
    const unsigned long long v1 = sg.v1;
    const unsigned long long step = sg.step;
    const bool sparse = !LDS_HIST && d.sparse_table != nullptr;
    const unsigned B = sparse ? (unsigned)d.sparse_real_nbins : (unsigned)d.total_nbins;
    gptr<unsigned> gbins = to_global(d.bins);
    if (!lds_clean) {
      // whole LDS histogram (sized for the largest member), once per workgroup
      if (LDS_HIST) {
        for (unsigned b = tid; b < hist_words; b += nthreads) hist[b] = 0u;
      }
      if (tid == 0) *s_norm = 0u;
      __syncthreads();
    }
    const int nobs = d.nobs, nslot = d.nslot, nsyst = d.nsyst;
    gptr<const double> params = to_global(d.params);
    const int pstride = d.param_stride;
    unsigned cnt = 0;
    // v counts 4-sample units; a workgroup covers units [c, c + nthreads) for c = v0, v0 + step, ...
    for (unsigned long long c = v0; c < v1; c += step)
    for (unsigned long long i = c * SXMC_VEC + tid;
         i < (c + nthreads < v1 ? c + nthreads : v1) * SXMC_VEC; i += nthreads) {
      double f[SXMC_MAX_NFIELDS];
      for (int k = 0; k < nslot; k++) {
        f[k] = (double)to_global(d.cols)[(unsigned long long)d.slot_col[k] * d.col_pitch + i];
      }
      for (int s = 0; s < nsyst; s++) {
        const SxSystOp& op = d.syst[s];
        double x = f[op.obs_slot];
        double p = 0.0, pw = 1.0, pl = 0.0;
        for (int t = 0; t < op.npars; t++) {
          p = p + params[(long)op.pars[t] * pstride] * pw;
          pow_step(pw, pl, x);
        }
        switch (op.type) {
          case SXMC_SYST_SHIFT: x = x + p; break;
          case SXMC_SYST_SCALE: x = x * (1 + p); break;
          case SXMC_SYST_CTSCALE: x = 1 + (x - 1) * (1 + p); break;
          case SXMC_SYST_RESOLUTION_SCALE: x = x + (p * (x - f[op.extra_slot])); break;
          default: break;
        }
        f[op.obs_slot] = x;
      }
      bool ok = true;
      int bin = 0;
      for (int k = 0; k < nobs && ok; k++) {
        const double x = f[k];
        ok = (x >= d.lower[k]) && (x < d.upper[k]);
        if (ok) bin += (int)((x - d.lower[k]) * d.scale[k]) * d.bin_stride[k];
      }
      if (ok) {
        cnt += 1u;
        if ((unsigned)bin < B) {
          if (LDS_HIST) {
            __hip_atomic_fetch_add(&hist[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else if (sparse) {
            sparse_count(d, gbins, (unsigned)bin);
          } else {
            __hip_atomic_fetch_add(&gbins[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
    if ((tid & (kWave - 1)) == 0 && cnt != 0u) {
      __hip_atomic_fetch_add(s_norm, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (LDS_HIST) {
      for (unsigned b = tid; b < B; b += nthreads) {
        const unsigned c = hist[b];
        if (c != 0u) {
          __hip_atomic_fetch_add(&gbins[b], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hist[b] = 0u;
        }
      }
    }
    if (tid == 0) {
      const unsigned c = *s_norm;
      if (c != 0u) __hip_atomic_fetch_add(to_global(d.norm), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_norm = 0u;
    }
    __syncthreads();
    lds_clean = true;
  }
}

// ------------------------------------------------------------------------------------ eval
__device__ __forceinline__ float pdf_value(int rb, gptr<const unsigned> bins, double bin_norm) {
  // pdfz.cpp:423-434: -2 -> 0, other negatives -> NaN, else (float)(bins / (norm * volume))
  if (rb == -2) return 0.0f;
  if (rb < 0) return __int_as_float(0x7fc00000);
  return (float)((double)bins[rb] / bin_norm);
}

__global__ __launch_bounds__(256) void eval_pdf_kernel(const SxSignalDesc* __restrict__ descs) {
  const SxSignalDesc& d = descs[blockIdx.y];
  if (d.read_bins == nullptr) return;
  gptr<const int> rbs = to_global(d.read_bins);
  gptr<const unsigned> bins = to_global((const unsigned*)d.bins);
  const unsigned long long n = d.npoints;
  const double bin_norm = (double)(*to_global((const unsigned*)d.norm)) * d.bin_volume;
  gptr<float> out = to_global(d.pdf_out);
  const long stride = d.pdf_stride;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    out[stride * (long)i] = pdf_value(rbs[i], bins, bin_norm);
  }
}

// eval_pdf for all members fused with nll_event_chunks (nll_kernels.cpp:89-116): per event the
// lookup-table values are produced, stored (the lut is the API contract) and consumed in
// registers.  One partial sum per workgroup.  The kernel is a chain of dependent gathers
// (read_bins -> bins), so the per-member pointers are staged in LDS and the members are walked
// eight at a time with their loads issued together.
struct EvalMember {
  const int* read_bins;
  const unsigned* bins;
  float* out;
  long stride;
  double bin_norm;
  double coef;  // pars[sid] * nexpected * eff, the event-independent factor of nll_kernels.cpp:107
};

// Body shared by eval_nll_kernel and eval_nll_finish_kernel: returns the workgroup's partial event sum
// (valid in thread 0).  A row is an event -- or, with `weight`, a class of events that fall into the same
// bin of every member (the descriptors then point at the class tables and carry no lookup-table output):
// its log term counts weight[row] times.
// `ready`: called by every thread, once, after the loads that do not depend on the evaluation (the rows' event-bin
// tables, the first row's weight: set by SetEvalPoints) have been ISSUED and before anything the fill writes is read
// (histograms, normalisations): the fused step kernel waits there for the fill's workgroups, with those loads in
// flight under the wait.
template <int BD = 0, typename Ready>   // (BD > 0: the workgroup's logical size, see block_sum in nll_device.h)
__device__ __forceinline__ double eval_nll_block_part(const SxSignalDesc* __restrict__ descs, int nsig,
                                                      unsigned long long npoints, const unsigned* __restrict__ weight,
                                                      const double* __restrict__ pars,
                                                      const double* __restrict__ nexpected,
                                                      const unsigned* __restrict__ n_mc,
                                                      const short* __restrict__ source_id,
                                                      const unsigned* __restrict__ norms, double* sh,
                                                      unsigned block_index, unsigned nblocks, Ready&& ready) {
  double* s_wave = sh;  // [16] wave sums, then nsig EvalMember records
  EvalMember* s_mem = reinterpret_cast<EvalMember*>(sh + 16);
  const unsigned bdim = BD > 0 ? (unsigned)BD : blockDim.x;

  // The events' bin indices of the first members are requested before anything else: their addresses
  // need only the descriptors (uniform, scalar loads), so the loads fly while the per-member factors
  // below are fetched and staged.
  constexpr int U = 16;
  int rb[U];
  unsigned long long i = (unsigned long long)block_index * bdim + threadIdx.x;
  // (clamped indices, no branches: all U table addresses are fetched together, then all U loads issued)
  auto load_read_bins = [&](unsigned long long ev, int j0) {
    const unsigned long long evc = ev < npoints ? ev : npoints - 1;
    gptr<const int> table[U];
#pragma unroll
    for (int u = 0; u < U; u++) table[u] = to_global(descs[min(j0 + u, nsig - 1)].read_bins);
#pragma unroll
    for (int u = 0; u < U; u++) rb[u] = table[u][evc];
#pragma unroll
    for (int u = 0; u < U; u++) rb[u] = (j0 + u < nsig && ev < npoints) ? rb[u] : -2;
  };
  if (npoints == 0 || nsig <= 0) {   // uniform: nothing to look up
    ready();
    return 0.0;
  }
  load_read_bins(i, 0);
  unsigned wfirst = weight ? to_global(weight)[i < npoints ? i : npoints - 1] : 1u;
  ready();

  for (int j = threadIdx.x; j < nsig; j += (int)bdim) {
    const SxSignalDesc& d = descs[j];
    s_mem[j].read_bins = d.read_bins;
    s_mem[j].bins = d.bins;
    s_mem[j].out = d.pdf_out;
    s_mem[j].stride = d.pdf_stride;
    s_mem[j].bin_norm = (double)(*to_global((const unsigned*)d.norm)) * d.bin_volume;
    const float eff = (float)(1.0 * norms[j] / n_mc[j]);             // nll_kernels.cpp:105
    s_mem[j].coef = pars[source_id[j]] * nexpected[j] * eff;         // left-to-right, :107
  }
  __syncthreads();

  double sum = 0.0;
  bool requested = true;
  const unsigned long long step = (unsigned long long)nblocks * bdim;
  for (; i < npoints; i += step) {
    double s = 0.0;
    // (the row's weight is asked for before the gathers, not after the logarithm: one memory round trip less on a
    //  path that is nothing but round trips)
    const unsigned wrow = requested ? wfirst : (weight ? to_global(weight)[i] : 1u);
    for (int j0 = 0; j0 < nsig; j0 += U) {
      if (!requested) load_read_bins(i, j0);
      requested = false;
      unsigned count[U];
#pragma unroll
      for (int u = 0; u < U; u++) count[u] = (rb[u] >= 0) ? to_global(s_mem[j0 + u].bins)[rb[u]] : 0u;
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (j0 + u < nsig) {
          // pdfz.cpp:423-434: -2 -> 0, other negatives -> NaN, else (float)(bins / (norm * volume))
          float v;
          if (rb[u] == -2) v = 0.0f;
          else if (rb[u] < 0) v = __int_as_float(0x7fc00000);
          else v = (float)((double)count[u] / s_mem[j0 + u].bin_norm);
          if (s_mem[j0 + u].out) to_global(s_mem[j0 + u].out)[s_mem[j0 + u].stride * (long)i] = v;
          s = s + s_mem[j0 + u].coef * (double)(!isnan(v) ? v : 0.0f);
        }
      }
    }
    if (s > 0) sum += weight ? (double)wrow * log(s) : log(s);
  }

#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off, kWave);
  const int wave = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) s_wave[wave] = sum;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < (int)(bdim / kWave); w++) t += s_wave[w];
  }
  return t;
}

template <int BD = 0>
__device__ __forceinline__ double eval_nll_block_part(const SxSignalDesc* __restrict__ descs, int nsig,
                                                      unsigned long long npoints, const unsigned* __restrict__ weight,
                                                      const double* __restrict__ pars,
                                                      const double* __restrict__ nexpected,
                                                      const unsigned* __restrict__ n_mc,
                                                      const short* __restrict__ source_id,
                                                      const unsigned* __restrict__ norms, double* sh,
                                                      unsigned block_index, unsigned nblocks) {
  return eval_nll_block_part<BD>(descs, nsig, npoints, weight, pars, nexpected, n_mc, source_id, norms, sh, block_index,
                                 nblocks, [] {});
}

__device__ __forceinline__ double eval_nll_block(const SxSignalDesc* __restrict__ descs, int nsig,
                                                 unsigned long long npoints, const unsigned* __restrict__ weight,
                                                 const double* __restrict__ pars,
                                                 const double* __restrict__ nexpected,
                                                 const unsigned* __restrict__ n_mc,
                                                 const short* __restrict__ source_id,
                                                 const unsigned* __restrict__ norms, double* sh) {
  return eval_nll_block_part(descs, nsig, npoints, weight, pars, nexpected, n_mc, source_id, norms, sh, blockIdx.x,
                             gridDim.x);
}

__global__ __launch_bounds__(256) void eval_nll_kernel(const SxSignalDesc* __restrict__ descs, int nsig,
                                                       unsigned long long npoints,
                                                       const unsigned* __restrict__ weight,
                                                       const double* __restrict__ pars,
                                                       const double* __restrict__ nexpected,
                                                       const unsigned* __restrict__ n_mc,
                                                       const short* __restrict__ source_id,
                                                       const unsigned* __restrict__ norms,
                                                       double* __restrict__ sums) {
  extern __shared__ double sh[];
  const double t = eval_nll_block(descs, nsig, npoints, weight, pars, nexpected, n_mc, source_id, norms, sh);
  if (threadIdx.x == 0 && !isnan(t)) sums[blockIdx.x] = t;
}

// The whole end of an MCMC step in one launch: lookup + event partial sums as above, and the
// workgroup that arrives LAST (a ticket counter, no waiting, so no co-residency assumption) also runs
// finish_nll_jump_pick_combo.  Hand-off of the partial sums follows the agent-scope release/acquire
// recipe for gfx950: partial stored, drained and released by lane 0 before its ticket; the last arriver
// acquires (invalidating its CU's L1) before any of its lanes read the partials.
__global__ __launch_bounds__(256) void eval_nll_finish_kernel(const SxSignalDesc* __restrict__ descs, int nsig,
                                                              unsigned long long npoints,
                                                              const unsigned* __restrict__ weight, double* sums,
                                                              unsigned* ticket, SxStepArgs a) {
  extern __shared__ double sh[];
  const double t =
      eval_nll_block(descs, nsig, npoints, weight, a.v_proposed, a.nexpected, a.n_mc, a.source_id, a.norms, sh);
  int* s_last = reinterpret_cast<int*>(sh + 15);  // last wave-sum slot: at most 4 waves are in use
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = isnan(t) ? 0.0 : t;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = arrived == gridDim.x - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *s_last = last;
  }
  __syncthreads();
  if (!*s_last) return;
  __syncthreads();
  sxdev::finish_step_device(gridDim.x, sums, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                            a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                            a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                            a.debug_mode != 0);
}

// LOOK-AHEAD WALK (sxmc_multigroup_lookahead_step_async).  A Metropolis step that rejects leaves the chain where it
// was, and the NEXT proposal -- current vector + jump width x the next deviates -- is then known before the
// step is decided.  One pass over the tables (the lockstep fill of two chains) evaluates the likelihood at BOTH
// the step's proposal A and that look-ahead vector B; the step end below decides the step from A and, if it
// rejected, the following step from B at once: 1 + P(reject) steps per pass, the chain -- every row of the
// jump buffer, every generator state -- exactly the sequential one (pre-fetching, Brockwell 2006).
// Lookup + event sums of both candidates in one launch: the first half of the grid works on A, the second on B.
__global__ __launch_bounds__(256) void eval_nll2_kernel(const SxSignalDesc* __restrict__ descs_a,
                                                        const SxSignalDesc* __restrict__ descs_b, int nsig,
                                                        unsigned long long npoints,
                                                        const unsigned* __restrict__ weight_a,
                                                        const unsigned* __restrict__ weight_b,
                                                        const double* __restrict__ pars_a,
                                                        const double* __restrict__ pars_b,
                                                        const double* __restrict__ nexpected,
                                                        const unsigned* __restrict__ n_mc,
                                                        const short* __restrict__ source_id,
                                                        const unsigned* __restrict__ norms_a,
                                                        const unsigned* __restrict__ norms_b,
                                                        double* __restrict__ sums_a, double* __restrict__ sums_b,
                                                        unsigned half) {
  extern __shared__ double sh[];
  const bool second = blockIdx.x >= half;
  // (eval_nll_block strides by gridDim.x: give each half its own picture of the grid)
  const double t = eval_nll_block_part(second ? descs_b : descs_a, nsig, npoints, second ? weight_b : weight_a,
                                       second ? pars_b : pars_a, nexpected, n_mc, source_id,
                                       second ? norms_b : norms_a, sh, second ? blockIdx.x - half : blockIdx.x, half);
  if (threadIdx.x == 0 && !isnan(t)) (second ? sums_b : sums_a)[second ? blockIdx.x - half : blockIdx.x] = t;
}

// Step end of the look-ahead walk + the clearing for the next pass.  Workgroup 0: finish_nll_jump_pick_combo for
// the step (candidate A = a.v_proposed) and, if that rejected and the walk may go on (the step count stays below
// *cap), for the next step: its proposal -- written into a.v_proposed by the first call -- IS the look-ahead
// vector (same current vector, same deviates, same arithmetic), so the second call runs on the same buffers with
// B's event sums and normalisations.  Then the next look-ahead vector is written to v_b.  The other workgroups
// clear both candidates' histograms.
__global__ __launch_bounds__(256) void finish2_zero_kernel(const SxSignalDesc* __restrict__ descs_a,
                                                           const SxSignalDesc* __restrict__ descs_b, int nsig,
                                                           unsigned zblocks, size_t npartial, const double* sums_a,
                                                           const double* sums_b, const unsigned* norms_b, double* v_b,
                                                           const int* cap, SxStepArgs a) {
  if (blockIdx.x == 0) {
    __shared__ int s_before;
    if (threadIdx.x == 0) s_before = a.counter[0];
    __syncthreads();
    const int limit = cap ? cap[0] : 0x7FFFFFFF;
    if (s_before < limit) {      // (a pass launched after the walk reached its stop: nothing to do)
      const bool accepted =
          sxdev::finish_step_device(npartial, sums_a, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                                    a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                                    a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                                    a.debug_mode != 0);
      __syncthreads();
      if (!accepted && s_before + 1 < limit) {
        sxdev::finish_step_device(npartial, sums_b, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                                  a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                                  a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, norms_b,
                                  a.debug_mode != 0);
        __syncthreads();
      }
      sxdev::peek_next_proposal_device(a.nparameters, a.rng, a.jump_width, a.v_current, v_b);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nsig; j += blockDim.x) {
      *descs_a[j].norm = 0u;
      *descs_b[j].norm = 0u;
    }
    return;
  }
  const unsigned b = blockIdx.x - 1u;
  const unsigned which = b / (zblocks * (unsigned)nsig);
  const unsigned bb = b - which * zblocks * (unsigned)nsig;
  const SxSignalDesc& d = (which ? descs_b : descs_a)[bb / zblocks];
  const unsigned chunk = bb % zblocks;
  unsigned* bins = d.bins;
  const unsigned n = (unsigned)d.total_nbins;
  const unsigned n4 = n >> 2;
  uint4* b4 = reinterpret_cast<uint4*>(bins);
  const unsigned stride = zblocks * blockDim.x;
  for (unsigned i = chunk * blockDim.x + threadIdx.x; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (chunk == 0 && threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
}

// first look-ahead vector of a walk (the state as sxmc_launch_pick_new_vector left it)
__global__ void peek_next_proposal_kernel(int nparameters, const sxmc_rng_state* rng, const float* jump_width,
                                          const double* v_current, double* out) {
  sxdev::peek_next_proposal_device(nparameters, rng, jump_width, v_current, out);
}

// The whole end of an MCMC step in ONE workgroup: lookup + event sum over the rows (events, or classes of events
// with a weight each), finish_nll_jump_pick_combo, and the clearing of histograms and normalisations the next
// evaluation would start with.  With a few thousand rows (event classes at BASELINE configs 1-3) the end of a step
// is a chain of memory latencies, not work: as its own kernels (lookup + event sum over ~40 workgroups, then the
// step end beside the zeroing) it costs two launch ramps, two kernel boundaries and a round trip of the partial
// sums through memory; here it is one ramp, and the step's other inputs are in flight under the gathers.
__global__ __launch_bounds__(1024) void tail_step_kernel(const SxSignalDesc* __restrict__ descs, int nsig,
                                                         unsigned long long npoints,
                                                         const unsigned* __restrict__ weight, SxStepArgs a) {
  extern __shared__ double sh[];
  __shared__ double s_total;
  const double t =
      eval_nll_block(descs, nsig, npoints, weight, a.v_proposed, a.nexpected, a.n_mc, a.source_id, a.norms, sh);
  if (threadIdx.x == 0) s_total = isnan(t) ? 0.0 : t;   // (nll_kernels.cpp:113-115: a NaN partial is not stored)
  __syncthreads();
  sxdev::finish_step_device(1, &s_total, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                            a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                            a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                            a.debug_mode != 0);
  // (the normalisations were last read before the barriers inside finish_step_device)
  for (int j = 0; j < nsig; j++) {
    const SxSignalDesc& d = descs[j];
    const unsigned n = (unsigned)d.total_nbins, n4 = n >> 2;
    uint4* b4 = reinterpret_cast<uint4*>(d.bins);
    for (unsigned i = threadIdx.x; i < n4; i += blockDim.x) b4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < (n & 3u)) d.bins[(n4 << 2) + threadIdx.x] = 0u;
    if (threadIdx.x == 0) *d.norm = 0u;
  }
}

// finish_nll_jump_pick_combo (workgroup 0) and, beside it in the same launch, what zero_kernel does for
// the NEXT evaluation (all other workgroups): a walk that does not look at histograms or normalisations
// between steps saves a launch per step.  Workgroup 0 clears the normalisations itself, after it has read
// them; the other workgroups touch only the histograms (counters), which nothing reads at this point.
__global__ __launch_bounds__(256) void finish_zero_kernel(const SxSignalDesc* __restrict__ descs, int nsig,
                                                          unsigned zblocks, size_t npartial, const double* sums,
                                                          unsigned* ticket, SxStepArgs a) {
  if (blockIdx.x == 0) {
    sxdev::finish_step_device(npartial, sums, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                              a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                              a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                              a.debug_mode != 0);
    __syncthreads();
    for (int j = threadIdx.x; j < nsig; j += blockDim.x) *descs[j].norm = 0u;
    if (threadIdx.x == 0 && ticket) *ticket = 0u;
    return;
  }
  const unsigned b = blockIdx.x - 1u;
  const SxSignalDesc& d = descs[b / zblocks];
  const unsigned chunk = b % zblocks;
  unsigned* bins = d.bins;
  const unsigned n = (unsigned)d.total_nbins;
  const unsigned n4 = n >> 2;
  uint4* b4 = reinterpret_cast<uint4*>(bins);
  const unsigned stride = zblocks * blockDim.x;
  for (unsigned i = chunk * blockDim.x + threadIdx.x; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (chunk == 0 && threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
}

// THE STEP END IN ONE LAUNCH (cooperative).  eval_nll_kernel + finish_zero_kernel are two launches whose work is a chain
// of memory round trips: at BASELINE config 3 they take 9.5 + 6.6 us, 18-20 us of a 150 us step with their boundaries
// (profiles/r03_c3_kernel_stats.csv), and more than the fill itself at config 2.  Earlier one-launch forms lost what
// the boundary saved: the LAST workgroup to arrive ran the whole step end cold (round 2: 15.7 against 14.7 us); one
// workgroup doing everything is far too slow beyond a few hundred look-ups; and a first cooperative form of this
// round -- partials published with a release fence and an arrival COUNTER, the finisher acquiring -- cost 17.8 us
// against 16.2 (profiles/r04_step_end_ab_*): a fence is an L2 write-back, the counter another round trip, the acquire
// an L2 invalidate, and a kernel boundary inside a replayed graph is only 0-1.3 us.  This form has no fences and no
// counters.  The roles are fixed at launch:
//   * workgroups 0 .. W-1 (the workers: W = the virtual blocks of the event sum, at most 128) do the look-ups and the
//     event sum exactly as eval_nll_kernel's workgroups do -- the same blocks of 128 rows, the same partial sums, so
//     the NLL and with it the chain stay bit-identical -- and hand their partial over with ONE device-scope atomic
//     store into their own 64-bit slot (the partial IS the message: nothing else of the worker's needs to be visible);
//   * workgroup W (the finisher) meanwhile does everything of finish_nll_jump_pick_combo that does not need the sums
//     (phase A of finish_step_device: every input loaded, the expected-rate and constraint terms, the next proposal's
//     deviates); then lane i polls slot i until it holds a value (device-scope atomic loads), the values go to LDS,
//     and what is left after the last worker's partial lands is a barrier, the reduction and phases B and C;
//   * once the finisher holds ALL partials -- every look-up is done, nothing reads the histograms any more -- it
//     empties the slots again, which is also the workers' signal: each polls its own slot until it is empty, then
//     clears its share of the histograms for the next evaluation (what zero_kernel would do) while the finisher
//     finishes; the finisher clears the normalisations, which it and the workers read.
// A slot holds the partial's bit pattern, or kSlotEmpty, or kSlotNaN for "my partial came out NaN" -- then the value
// of the last step whose partial did not is used, which is what eval_nll_kernel's unwritten array element amounts to
// (nll_kernels.cpp:113-115 does not store a NaN partial); both markers are NaN patterns no stored partial can have.
// Waiting inside a kernel needs the waited-for workgroups to be resident or to become resident: the grid is at most
// 129 workgroups of 128 lanes with a few KB of LDS (the host takes this form only then), which the device holds many
// times over beside any fill kernel; the workers wait only for the finisher and the finisher only for the workers,
// all of one launch; and every wait is BOUNDED -- a lane that does not see its slot change within ~0.3 s gives up,
// counts a timeout (sync[6], read by sxmc_group_step_end_timeouts) and goes on, so the grid always drains.
constexpr unsigned kEndSpinLimit = 200000u;
constexpr unsigned long long kSlotEmpty = 0x7FF8DEADBEEF0001ull, kSlotNaN = 0x7FF8DEADBEEF0002ull;

// The roles of the cooperative step end.  `role` < W: worker `role`; `role` == W: the finisher.  BD > 0: the logical
// workgroup size when the roles run inside a launch of larger workgroups (fill_step_kernel); lanes >= BD have left.
// `ready`: called by every thread of the workgroup, once, before anything the fill writes is read (see
// eval_nll_block_part): nothing for step_end_kernel, whose launch follows the fill's; the wait for the fill's workgroups
// in fill_step_kernel.
template <int BD, typename Ready>
__device__ __forceinline__ void step_end_role(unsigned role, unsigned W, const SxSignalDesc* __restrict__ lookup_descs,
                                              const SxSignalDesc* __restrict__ hist_descs, int nsig,
                                              unsigned long long npoints, const unsigned* __restrict__ weight,
                                              unsigned long long* slots, double* last_good, unsigned* sync,
                                              unsigned nvb, unsigned zblocks, const SxStepArgs& a, double* sh,
                                              Ready&& ready) {
  const unsigned bdim = BD > 0 ? (unsigned)BD : blockDim.x;
  unsigned* const timeouts = sync + 6;
  if (role == W) {
    // ---- the finisher (its phase A reads the normalisations: after the fill)
    ready();
    __shared__ double s_part[128];
    sxdev::finish_step_device_w<BD>(nvb, s_part, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                                    a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                                    a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                                    a.debug_mode != 0, [&] {
                                      const unsigned i = threadIdx.x;
                                      if (i < nvb) {
                                        unsigned long long v = kSlotEmpty;
                                        unsigned it = 0;
                                        for (; it < kEndSpinLimit; it++) {
                                          v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                          if (v != kSlotEmpty) break;
                                          __builtin_amdgcn_s_sleep(1);
                                        }
                                        double t;
                                        if (v == kSlotEmpty) {   // gave up: flagged, the step is not valid
                                          __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                          t = 0.0;
                                        } else if (v == kSlotNaN) {
                                          t = last_good[i];
                                        } else {
                                          t = __longlong_as_double((long long)v);
                                          last_good[i] = t;
                                        }
                                        s_part[i] = t;
                                      }
                                      __syncthreads();
                                      // every partial is in: the look-ups are over.  Empty the slots -- for the next
                                      // launch, and as the workers' signal that the histograms may be cleared.
                                      if (i < nvb) {
                                        __hip_atomic_store(slots + i, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                      }
                                    });
    __syncthreads();
#if SXMC_MEASURE
    // (measurement build, the gated step: the next proposal is written -- published with a release at agent scope -- and
    // the fill that waits for it beside this kernel may read it)
    if (a.gate && threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_store(a.gate, a.gate_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    for (int j = threadIdx.x; j < nsig; j += (int)bdim) *hist_descs[j].norm = 0u;
    if (threadIdx.x == 0) sync[0] = 0u;   // (the ticket of the other step-end forms: as finish_zero_kernel leaves it)
    return;
  }
  // ---- a worker: its virtual block of the event sum, as eval_nll_kernel's workgroup does it
  const unsigned vb = role;
  const double t = eval_nll_block_part<BD>(lookup_descs, nsig, npoints, weight, a.v_proposed, a.nexpected, a.n_mc,
                                           a.source_id, a.norms, sh, vb, nvb, ready);
  __shared__ int s_clear;
  if (threadIdx.x == 0) {
    __hip_atomic_store(slots + vb, isnan(t) ? kSlotNaN : (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the store has landed before this lane starts reading the slot)
    // the histograms may be cleared once the finisher has every worker's partial: it then empties the slots
    unsigned it = 0;
    for (; it < kEndSpinLimit; it++) {
      if (__hip_atomic_load(slots + vb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kSlotEmpty) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (it == kEndSpinLimit) __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_clear = it < kEndSpinLimit ? 1 : 0;
  }
  __syncthreads();
  if (!s_clear) return;
  const unsigned npieces = zblocks * (unsigned)nsig;
  for (unsigned p = role; p < npieces; p += W) {
    const SxSignalDesc& d = hist_descs[p / zblocks];
    const unsigned chunk = p % zblocks;
    unsigned* bins = d.bins;
    const unsigned n = (unsigned)d.total_nbins;
    const unsigned n4 = n >> 2;
    uint4* b4 = reinterpret_cast<uint4*>(bins);
    const unsigned stride = zblocks * bdim;
    for (unsigned i = chunk * bdim + threadIdx.x; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (chunk == 0 && threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
  }
}

__global__ __launch_bounds__(128) void step_end_kernel(const SxSignalDesc* __restrict__ lookup_descs,
                                                       const SxSignalDesc* __restrict__ hist_descs, int nsig,
                                                       unsigned long long npoints, const unsigned* __restrict__ weight,
                                                       unsigned long long* slots, double* last_good, unsigned* sync,
                                                       unsigned nvb, unsigned zblocks, SxStepArgs a) {
  extern __shared__ double sh[];
  step_end_role<0>(blockIdx.x, gridDim.x - 1u, lookup_descs, hist_descs, nsig, npoints, weight, slots, last_good, sync,
                   nvb, zblocks, a, sh, [] {});
}

// THE WHOLE STEP IN ONE LAUNCH: fill_step_kernel = a fill kernel + the roles above as extra workgroups of the SAME launch.
// What is left between the fill and the cooperative step end as two launches is the second kernel's ramp (~2.5 us), the
// gap before it (~1.3 us) and its first loads (descriptors, event-bin tables, weights, staged factors: two memory round
// trips) -- all of which can sit under the fill's tail, whose workgroups finish over ~8 us.  Blocks
// 0 .. nfill-1 are the fill's workgroups, unchanged, each counting itself done (after its flush has landed: every lane
// waits for its own atomics, then one device-scope increment); block nfill is the finisher, blocks nfill+1 .. the
// workers.  A role workgroup starts on a CU as soon as a fill workgroup has left it (they all carry the fill's LDS
// allotment, one workgroup per CU), does what does not depend on the histograms, waits for the count to reach nfill,
// acquires, and goes on as step_end_role.  Who waits for whom: the roles for fill workgroups and the workers for the
// finisher -- all EARLIER blocks of the same launch, which the dispatcher has started before them; the fill
// workgroups wait for nobody; only the finisher (one workgroup per launch) waits for later blocks.  So the launch
// makes progress whatever else shares the device, and every wait is bounded as in step_end_kernel.
// The roles run with the fill's workgroup size; lanes >= 128 leave at once and the rest behave as a workgroup of 128
// (BD = 128 throughout: the same virtual blocks, partial sums and reduction tree as step_end_kernel -- bit-identical).
struct SxTailArgs {   // a kernel argument, by value (pointers that arrive as arguments are uniform and known to be global)
  const SxSignalDesc* lookup_descs;
  const SxSignalDesc* hist_descs;
  int nsig;
  unsigned nvb, zblocks, pad;
  unsigned long long npoints;
  const unsigned* weight;
  unsigned long long* slots;
  double* last_good;
  unsigned* sync;
  SxStepArgs a;
};

template <typename Fill>
__device__ __forceinline__ void fill_step_body(unsigned nfill, const SxTailArgs& t, unsigned dbg, Fill&& fill) {
  if (blockIdx.x < nfill) {
    fill();
    // this workgroup's flush has landed before it counts itself done: every lane waits for the acknowledgement of its
    // own device-scope atomics (they are performed at the memory side: acknowledged = performed), then one lane adds
    // to the count at the same scope.  (No release fence: on gfx950 that is a write-back of the XCD's whole L2, which
    // every workgroup would pay at the tail of the launch, and nothing but those atomics has to be published.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(t.sync + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x >= 128u) return;   // (whole waves: the roles are workgroups of 128)
  if (sxfill::sx_dbg(dbg) & 8u) return;   // measurement build only (RESULTS ARE WRONG): the fill's part of the launch alone
  SX_WG_STAMP(0);                    // (measurement build: a role's entry, its sight of the fill's end, its exit)
  extern __shared__ double sh_tail[];
  const unsigned role_index = blockIdx.x - nfill;            // 0: the finisher, 1 ..: workers
  const unsigned W = t.nvb;
  const unsigned role = role_index == 0 ? W : role_index - 1u;
  // The wait for the fill.  No acquire fence after it: what the roles read of the fill's -- counters and normalisations,
  // all written by device-scope atomics at the memory side -- is read here for the first time since the launch began
  // (the launch itself started with the caches invalidated, and the fill's workgroups read none of it), so no stale
  // copy can sit in this CU's L1 or this XCD's L2; and an agent-scope acquire on gfx950 invalidates the XCD's WHOLE L2,
  // once per role workgroup, under the other roles' look-ups (measured: the roles took 28 us instead of 14).
  auto fill_done = [&] {
    if (threadIdx.x == 0) {
      unsigned* done = t.sync + 7;
      unsigned it = 0;
      for (; it < kEndSpinLimit; it++) {
        if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nfill) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (it == kEndSpinLimit) __hip_atomic_fetch_add(t.sync + 6, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SX_WG_STAMP(1);
    __syncthreads();
  };
  step_end_role<128>(role, W, t.lookup_descs, t.hist_descs, t.nsig, t.npoints, t.weight, t.slots, t.last_good, t.sync,
                     t.nvb, t.zblocks, t.a, sh_tail, fill_done);
  // the finisher is the last to need the count of this launch: it zeroes it for the next one
  if (role == W && threadIdx.x == 0) {
    __hip_atomic_store(t.sync + 7, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SX_WG_STAMP(2);
}

template <int NOBS, int NSLOT, typename PROG>
__global__ __launch_bounds__(1024) void fill_ordered_step_kernel(const SxSignalDesc* __restrict__ descs,
                                                                 const SxSegment* __restrict__ segs,
                                                                 const unsigned* __restrict__ blk_off, unsigned layout,
                                                                 unsigned dbg, unsigned nfill, SxTailArgs tail) {
  fill_step_body(nfill, tail, dbg, [&] {
    SxChainDescs one;
    one.d[0] = one.d[1] = one.d[2] = one.d[3] = descs;
    fill_ordered_body<NOBS, NSLOT, PROG, 1, true>(one, segs, blk_off, layout, dbg);
  });
}

template <int NOBS, int NSLOT, typename PROG, int PREW>
__global__ __launch_bounds__(1024) void fill_step_kernel(const SxSignalDesc* __restrict__ descs,
                                                         const SxSegment* __restrict__ segs,
                                                         const unsigned* __restrict__ blk_off, unsigned hist_words,
                                                         unsigned dbg, unsigned nfill, SxTailArgs tail) {
  fill_step_body(nfill, tail, dbg, [&] { fill_body<NOBS, NSLOT, true, PROG, PREW>(descs, segs, blk_off, hist_words, dbg); });
}

// The look-ahead pass's step end as ONE cooperative launch: eval_nll2_kernel + finish2_zero_kernel with the hand-over
// of step_end_kernel.  Workers 0 .. nvb-1 sum candidate A's virtual blocks, nvb .. 2 nvb-1 candidate B's (each exactly
// as eval_nll2_kernel's workgroups do: same blocks, same partial sums -- the walk stays the sequential chain bit for
// bit); slot w belongs to worker w.  The finisher (workgroup 2 nvb) does phase A of the step's
// finish_nll_jump_pick_combo while they work, then lane w polls slot w (2 nvb <= 128 lanes); once it holds every
// partial of BOTH candidates it empties the slots -- the workers' signal to clear both candidates' histograms -- and
// goes on as finish2_zero_kernel's workgroup 0: the step from A, if that rejected the following step from B, the
// next look-ahead vector, the normalisations cleared.
__global__ __launch_bounds__(128) void step_end2_kernel(const SxSignalDesc* __restrict__ lookup_a,
                                                        const SxSignalDesc* __restrict__ lookup_b,
                                                        const SxSignalDesc* __restrict__ hist_a,
                                                        const SxSignalDesc* __restrict__ hist_b, int nsig,
                                                        unsigned long long npoints, const unsigned* __restrict__ weight_a,
                                                        const unsigned* __restrict__ weight_b, unsigned long long* slots,
                                                        double* last_good, unsigned* sync, unsigned nvb, unsigned zblocks,
                                                        const unsigned* norms_b, double* v_b, const int* cap, SxStepArgs a) {
  extern __shared__ double sh[];
  const unsigned W = gridDim.x - 1u;   // == 2 * nvb
  unsigned* const timeouts = sync + 6;
  if (blockIdx.x == W) {
    // ---- the finisher
    __shared__ double s_part[128];     // [0, nvb): candidate A's partial sums, [nvb, 2 nvb): candidate B's
    __shared__ int s_before;
    if (threadIdx.x == 0) s_before = a.counter[0];
    __syncthreads();
    const int limit = cap ? cap[0] : 0x7FFFFFFF;
    auto collect = [&] {
      const unsigned i = threadIdx.x;
      if (i < W) {
        unsigned long long v = kSlotEmpty;
        for (unsigned it = 0; it < kEndSpinLimit; it++) {
          v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v != kSlotEmpty) break;
          __builtin_amdgcn_s_sleep(1);
        }
        double t;
        if (v == kSlotEmpty) {
          __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t = 0.0;
        } else if (v == kSlotNaN) {
          t = last_good[i];
        } else {
          t = __longlong_as_double((long long)v);
          last_good[i] = t;
        }
        s_part[i] = t;
      }
      __syncthreads();
      if (i < W) __hip_atomic_store(slots + i, kSlotEmpty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (s_before < limit) {      // (a pass launched after the walk reached its stop decides nothing)
      const bool accepted = sxdev::finish_step_device_w(
          nvb, s_part, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current, a.nll_proposed, a.v_current,
          a.v_proposed, a.accepted, a.counter, a.jump_buffer, a.nparameters, a.jump_width, a.nexpected, a.n_mc,
          a.source_id, a.norms, a.debug_mode != 0, collect);
      __syncthreads();
      if (!accepted && s_before + 1 < limit) {
        sxdev::finish_step_device(nvb, s_part + nvb, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                                  a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                                  a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, norms_b,
                                  a.debug_mode != 0);
        __syncthreads();
      }
      sxdev::peek_next_proposal_device(a.nparameters, a.rng, a.jump_width, a.v_current, v_b);
    } else {
      collect();                 // (the workers still hand over and wait for their signal)
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nsig; j += blockDim.x) {
      *hist_a[j].norm = 0u;
      *hist_b[j].norm = 0u;
    }
    return;
  }
  // ---- a worker
  const bool second = blockIdx.x >= nvb;
  const unsigned vb = second ? blockIdx.x - nvb : blockIdx.x;
  const double t = eval_nll_block_part(second ? lookup_b : lookup_a, nsig, npoints, second ? weight_b : weight_a,
                                       second ? v_b : a.v_proposed, a.nexpected, a.n_mc, a.source_id,
                                       second ? norms_b : a.norms, sh, vb, nvb);
  __shared__ int s_clear;
  if (threadIdx.x == 0) {
    __hip_atomic_store(slots + blockIdx.x, isnan(t) ? kSlotNaN : (unsigned long long)__double_as_longlong(t),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned it = 0;
    for (; it < kEndSpinLimit; it++) {
      if (__hip_atomic_load(slots + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kSlotEmpty) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (it == kEndSpinLimit) __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_clear = it < kEndSpinLimit ? 1 : 0;
  }
  __syncthreads();
  if (!s_clear) return;
  const unsigned npieces = 2u * zblocks * (unsigned)nsig;
  for (unsigned p = blockIdx.x; p < npieces; p += W) {
    const unsigned which = p / (zblocks * (unsigned)nsig);
    const unsigned q = p - which * zblocks * (unsigned)nsig;
    const SxSignalDesc& d = (which ? hist_b : hist_a)[q / zblocks];
    const unsigned chunk = q % zblocks;
    unsigned* bins = d.bins;
    const unsigned n = (unsigned)d.total_nbins;
    const unsigned n4 = n >> 2;
    uint4* b4 = reinterpret_cast<uint4*>(bins);
    const unsigned stride = zblocks * blockDim.x;
    for (unsigned i = chunk * blockDim.x + threadIdx.x; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (chunk == 0 && threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
  }
}

// LOCKSTEP SETS: the step ends of all chains of a set in two launches instead of two per chain.  The chains have
// their own events (event classes, weights), parameter vectors, generators and jump buffers, so nothing is shared
// but the launch: chain = blockIdx.y, and inside a chain every workgroup does exactly what it does in
// eval_nll_kernel / finish_zero_kernel launched for that chain alone (same blocks of the event sum, same order of
// the partial sums) -- the chains stay bit-identical to chains stepped one at a time.  These kernels are a few
// microseconds of memory latency each; C chains cost the latency once instead of C times.
__global__ __launch_bounds__(256) void eval_nll_chains_kernel(SxChainEnds e, int nsig) {
  extern __shared__ double sh[];
  const SxChainEnd& c = e.c[blockIdx.y];
  if (blockIdx.x >= c.nblocks) return;   // (uniform: a chain with fewer rows than the set's largest)
  const double t = eval_nll_block_part(c.lookup_descs, nsig, c.nrows, c.weight, c.a.v_proposed, c.a.nexpected, c.a.n_mc,
                                       c.a.source_id, c.a.norms, sh, blockIdx.x, c.nblocks);
  if (threadIdx.x == 0 && !isnan(t)) c.sums[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void finish_zero_chains_kernel(SxChainEnds e, int nsig, unsigned zblocks) {
  const SxChainEnd& c = e.c[blockIdx.y];
  const SxStepArgs& a = c.a;
  if (blockIdx.x == 0) {
    sxdev::finish_step_device(c.nblocks, c.sums, a.nsignals, a.nsources, a.means, a.sigmas, a.rng, a.nll_current,
                              a.nll_proposed, a.v_current, a.v_proposed, a.accepted, a.counter, a.jump_buffer,
                              a.nparameters, a.jump_width, a.nexpected, a.n_mc, a.source_id, a.norms,
                              a.debug_mode != 0);
    __syncthreads();
    for (int j = threadIdx.x; j < nsig; j += blockDim.x) *c.hist_descs[j].norm = 0u;
    if (threadIdx.x == 0 && c.ticket) *c.ticket = 0u;
    return;
  }
  const unsigned b = blockIdx.x - 1u;
  const SxSignalDesc& d = c.hist_descs[b / zblocks];
  const unsigned chunk = b % zblocks;
  unsigned* bins = d.bins;
  const unsigned n = (unsigned)d.total_nbins;
  const unsigned n4 = n >> 2;
  uint4* b4 = reinterpret_cast<uint4*>(bins);
  const unsigned stride = zblocks * blockDim.x;
  for (unsigned i = chunk * blockDim.x + threadIdx.x; i < n4; i += stride) b4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (chunk == 0 && threadIdx.x < (n & 3u)) bins[(n4 << 2) + threadIdx.x] = 0u;
}

#if SXMC_WG_STAMPS
extern "C" {
__device__ unsigned long long sx_wg_stamps[3 * 4096];   // (the definition; fill_kernels.inc.h declares it)
}
extern "C" int sxmc_debug_read_wg_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sx_wg_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif

// ------------------------------------------------------------------------------------ pre-binning
// Builds the pre-binned column of one evaluator: for every sample, sum_k idx_k * stride_k over the
// observables in `mask` with the fill kernel's arithmetic (pdfz.cpp:388-398), or all ones when one of
// them is outside its domain (NaN included) or the row is padding.  width = bytes per sample.
__global__ __launch_bounds__(256) void prebin_kernel(const SxSignalDesc* __restrict__ dp, unsigned mask, int width,
                                                     void* out) {
  const SxSignalDesc& d = *dp;
  const unsigned long long npad = d.nvec * SXMC_VEC;
  const unsigned sentinel = width == 1 ? 0xFFu : width == 2 ? 0xFFFFu : 0xFFFFFFFFu;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += step) {
    unsigned val = sentinel;
    if (i < d.nsamples) {
      bool ok = true;
      int part = 0;
      for (int k = 0; k < d.nobs; k++) {
        if (!((mask >> k) & 1u)) continue;
        const double x = (double)to_global(d.cols)[(unsigned long long)k * d.col_pitch + i];
        ok = ok && (x >= d.lower[k]) && (x < d.upper[k]);
        part += (int)((x - d.lower[k]) * d.scale[k]) * d.bin_stride[k];
      }
      if (ok) val = (unsigned)part;
    }
    if (width == 1) static_cast<unsigned char*>(out)[i] = (unsigned char)val;
    else if (width == 2) static_cast<unsigned short*>(out)[i] = (unsigned short)val;
    else static_cast<unsigned*>(out)[i] = val;
  }
}

#if SXMC_MEASURE
// (measurement build only) test hook: out[k] = x[k]^i as the polynomial systematics form it
__global__ void pow_int_kernel(const double* __restrict__ x, int n, int i, double* __restrict__ out) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    double h = 1.0, l = 0.0;
    for (int t = 0; t < i; t++) pow_step(h, l, x[k]);
    out[k] = h;
  }
}
#endif

// ------------------------------------------------------------------------------------ layout
// Row-major [n][F] -> column-major with pitch; pads [n, nvec*4) with NaN in every column.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ aos, float* __restrict__ cols,
                                                        unsigned long long n, int F,
                                                        unsigned long long pitch) {
  const unsigned long long npad = (n + SXMC_VEC - 1) / SXMC_VEC * SXMC_VEC;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += step) {
    for (int k = 0; k < F; k++) {
      cols[(unsigned long long)k * pitch + i] = (i < n) ? aos[i * F + k] : __int_as_float(0x7fc00000);
    }
  }
}

// GetSamples (pdfz.h:542-556): rows of nobs observables + the dataset id.
__global__ __launch_bounds__(256) void untranspose_obs_kernel(const float* __restrict__ cols, float* __restrict__ out,
                                                              unsigned long long n, int nobs,
                                                              unsigned long long pitch, float dataset) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    for (int k = 0; k < nobs; k++) out[i * (nobs + 1) + k] = cols[(unsigned long long)k * pitch + i];
    out[i * (nobs + 1) + nobs] = dataset;
  }
}

// A fill launch; with the launch shape's two profiling events set, through hipExtLaunchKernelGGL, which stamps them
// with the dispatch's own begin and end (what rocprofv3 --kernel-trace reports as the kernel's duration).
template <typename K, typename... A>
void launch_fill(const SxLaunchShape& sh, K k, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
  if (sh.ev_start && sh.ev_stop) {
    hipExtLaunchKernelGGL(k, grid, block, (std::uint32_t)lds, s, (hipEvent_t)sh.ev_start, (hipEvent_t)sh.ev_stop, 0u,
                          args...);
  } else {
    hipLaunchKernelGGL(k, grid, block, lds, s, args...);
  }
}

// Which built-in fills also exist fused with the step end (fill_step_kernel / fill_ordered_step_kernel): the ordered
// programs, and the LDS-histogram kernels of the EMPTY program over a pre-binned column (BASELINE config 2).
template <bool LDS_HIST, typename PROG, int PREW>
constexpr bool has_step_form() {
  return LDS_HIST && PROG::n == 0 && (PREW == 1 || PREW == 2);
}

template <int NOBS, int NSLOT, bool LDS_HIST, typename PROG, int PREW = 0>
hipError_t launch_fill_k(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                         const unsigned* blk_off, hipStream_t s) {
  // LDS: 4 header words + hist_words + 64 trash words
  // LDS-histogram launches: hist_words bins + 64 trash words; others: room for the sparse coarse filter
  const unsigned hist_words = (unsigned)(sh.lds_bytes / 4 - 4 - (LDS_HIST ? 64 : 0));
  if constexpr (has_step_form<LDS_HIST, PROG, PREW>()) {
    if (sh.tail) {   // the whole step in this launch: the fill's workgroups + finisher + workers
      auto ks = fill_step_kernel<NOBS, NSLOT, PROG, PREW>;
      if (sh.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ks),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
        if (e != hipSuccess) return e;
      }
      hipLaunchKernelGGL(ks, dim3(sh.grid + sh.tail_blocks), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off,
                         hist_words, (unsigned)sh.debug_mode, (unsigned)sh.grid, *static_cast<const SxTailArgs*>(sh.tail));
      return hipGetLastError();
    }
  }
  if (sh.tail) return hipErrorInvalidValue;   // (the host asks sx_fill_has_step_form first)
  auto k = fill_kernel<NOBS, NSLOT, LDS_HIST, PROG, PREW>;
  if (sh.lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
    if (e != hipSuccess) return e;
  }
  launch_fill(sh, k, dim3(sh.grid), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off, hist_words,
              (unsigned)sh.debug_mode);
  return hipGetLastError();
}

typedef hipError_t (*FillLauncher)(const SxLaunchShape&, const SxSignalDesc*, const SxSegment*, const unsigned*,
                                   hipStream_t);

template <int NOBS, int NSLOT, typename PROG>
hipError_t launch_fill_sparse_k(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                                const unsigned* blk_off, hipStream_t s) {
  auto k = fill_sparse_kernel<NOBS, NSLOT, PROG>;
  if (sh.sparse_lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sh.sparse_lds_bytes);
    if (e != hipSuccess) return e;
  }
  const unsigned nwaves = (unsigned)sh.threads / 64u;
  const unsigned smax = (unsigned)(sh.sparse_lds_bytes / 4 / nwaves / 2);   // entries per wave (keys + counts)
  launch_fill(sh, k, dim3(sh.grid), dim3(sh.threads), sh.sparse_lds_bytes, s, descs, segs, blk_off, smax,
              (unsigned)sh.debug_mode);
  return hipGetLastError();
}

// Static programs (LDS-histogram launches only).  Slots: observables 0..nobs-1, then the
// referenced extra fields in ascending order.
struct StaticEntry {
  int nobs, nslot, nops;
  unsigned ops[4];
  FillLauncher fn;           // LDS histogram, all observables binned in the kernel
  FillLauncher fn_pre[2];    // LDS histogram, observables no systematic writes pre-binned, 1 / 2 bytes per sample
                             // (with an LDS-sized histogram the partial index of the table's programs is < 65535)
  FillLauncher fn_g;         // histogram beyond LDS capacity (dense global atomics or sparse event-bin counters)
  FillLauncher fn_g_pre[2];
  FillLauncher fn_gran[2];   // bucketed table (every observable given is binned, + one offset per granule): LDS / beyond LDS
  FillLauncher fn_sruns;     // bucketed table walked in runs, event bins counted in per-wave LDS tables
};
#define SX_SHIFT(o) sx_op(SXMC_SYST_SHIFT, o)
#define SX_SCALE(o) sx_op(SXMC_SYST_SCALE, o)
#define SX_CTSC(o) sx_op(SXMC_SYST_CTSCALE, o)
#define SX_RES(o, e) sx_op(SXMC_SYST_RESOLUTION_SCALE, o, e)
#define SX_NOPRE {nullptr, nullptr}
#define SX_NOG nullptr, {nullptr, nullptr}
#define SX_PRE(NO, NS, ...) \
  {launch_fill_k<NO, NS, true, StaticProg<__VA_ARGS__>, 1>, launch_fill_k<NO, NS, true, StaticProg<__VA_ARGS__>, 2>}
#define SX_G(NO, NS, ...) \
  launch_fill_k<NO, NS, false, StaticProg<__VA_ARGS__>, 0>, \
  {launch_fill_k<NO, NS, false, StaticProg<__VA_ARGS__>, 1>, launch_fill_k<NO, NS, false, StaticProg<__VA_ARGS__>, 2>}
#define SX_NOGRAN {nullptr, nullptr}, nullptr
#define SX_GRAN(NO, NS, ...) \
  {launch_fill_k<NO, NS, true, StaticProg<__VA_ARGS__>, kPreGranule>, launch_fill_k<NO, NS, false, StaticProg<__VA_ARGS__>, kPreGranule>}, \
  launch_fill_sparse_k<NO, NS, StaticProg<__VA_ARGS__>>
#define SX_P1(NO, NS, PRE, G, GR, A) {NO, NS, 1, {A, 0, 0, 0}, launch_fill_k<NO, NS, true, StaticProg<A>>, PRE, G, GR}
#define SX_P2(NO, NS, PRE, G, GR, A, B) {NO, NS, 2, {A, B, 0, 0}, launch_fill_k<NO, NS, true, StaticProg<A, B>>, PRE, G, GR}
#define SX_P3(NO, NS, PRE, G, GR, A, B, C) {NO, NS, 3, {A, B, C, 0}, launch_fill_k<NO, NS, true, StaticProg<A, B, C>>, PRE, G, GR}
// (no systematics: EVERY observable is untouched, so the pre-binned column carries the whole flat index and no float
// column is streamed at all -- 1 or 2 bytes per sample instead of 4 per observable)
#define SX_P0(NO) \
  {NO, NO, 0, {0, 0, 0, 0}, launch_fill_k<NO, NO, true, StaticProg<>>, \
   {launch_fill_k<NO, NO, true, StaticProg<>, 1>, launch_fill_k<NO, NO, true, StaticProg<>, 2>}, SX_NOG, SX_NOGRAN}
const StaticEntry kStaticPrograms[] = {
    // no systematics at all (BASELINE config 2)
    SX_P0(1), SX_P0(2), SX_P0(3),
    // 1-D (bench_sxmc pdfz: one shift; config/example.json: scale + resolution_scale).  These are also what a
    // bucketed higher-dimensional table with ONE observable written by systematics reduces to.
    SX_P1(1, 1, SX_NOPRE, SX_NOG, SX_GRAN(1, 1, SX_SHIFT(0)), SX_SHIFT(0)),
    SX_P1(1, 1, SX_NOPRE, SX_NOG, SX_GRAN(1, 1, SX_SCALE(0)), SX_SCALE(0)),
    SX_P1(1, 1, SX_NOPRE, SX_NOG, SX_GRAN(1, 1, SX_CTSC(0)), SX_CTSC(0)),
    SX_P2(1, 1, SX_NOPRE, SX_NOG, SX_GRAN(1, 1, SX_SHIFT(0), SX_SCALE(0)), SX_SHIFT(0), SX_SCALE(0)),
    SX_P1(1, 2, SX_NOPRE, SX_NOG, SX_GRAN(1, 2, SX_RES(0, 1)), SX_RES(0, 1)),
    SX_P2(1, 2, SX_NOPRE, SX_NOG, SX_GRAN(1, 2, SX_SCALE(0), SX_RES(0, 1)), SX_SCALE(0), SX_RES(0, 1)),
    SX_P3(1, 2, SX_NOPRE, SX_NOG, SX_GRAN(1, 2, SX_SHIFT(0), SX_SCALE(0), SX_RES(0, 1)), SX_SHIFT(0), SX_SCALE(0),
          SX_RES(0, 1)),
    // 2-D
    SX_P1(2, 2, SX_PRE(2, 2, SX_SHIFT(0)), SX_NOG, SX_NOGRAN, SX_SHIFT(0)),
    SX_P1(2, 2, SX_PRE(2, 2, SX_SCALE(0)), SX_NOG, SX_NOGRAN, SX_SCALE(0)),
    SX_P1(2, 2, SX_PRE(2, 2, SX_SHIFT(1)), SX_NOG, SX_NOGRAN, SX_SHIFT(1)),
    SX_P2(2, 3, SX_PRE(2, 3, SX_SCALE(0), SX_RES(0, 2)), SX_NOG, SX_NOGRAN, SX_SCALE(0), SX_RES(0, 2)),
    // (also what BASELINE configs 3 and 5 reduce to once bucketed: e and r are written, the rest is not)
    SX_P3(2, 3, SX_NOPRE, SX_NOG, SX_GRAN(2, 3, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 2)), SX_SHIFT(1), SX_SCALE(0),
          SX_RES(0, 2)),
    // 3-D (BASELINE config 3: shift(r) + scale(e) + resolution_scale(e | e_true))
    SX_P1(3, 3, SX_PRE(3, 3, SX_SHIFT(0)), SX_NOG, SX_NOGRAN, SX_SHIFT(0)),
    SX_P1(3, 3, SX_PRE(3, 3, SX_SCALE(0)), SX_NOG, SX_NOGRAN, SX_SCALE(0)),
    SX_P1(3, 4, SX_PRE(3, 4, SX_RES(0, 3)), SX_NOG, SX_NOGRAN, SX_RES(0, 3)),
    SX_P2(3, 4, SX_PRE(3, 4, SX_SCALE(0), SX_RES(0, 3)), SX_NOG, SX_NOGRAN, SX_SCALE(0), SX_RES(0, 3)),
    SX_P3(3, 4, SX_PRE(3, 4, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 3)),
          SX_G(3, 4, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 3)), SX_NOGRAN, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 3)),
    // 5-D (BASELINE config 5: the same three systematics, histograms beyond LDS capacity)
    SX_P3(5, 6, SX_PRE(5, 6, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 5)),
          SX_G(5, 6, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 5)), SX_NOGRAN, SX_SHIFT(1), SX_SCALE(0), SX_RES(0, 5)),
};
constexpr int kNumStatic = (int)(sizeof(kStaticPrograms) / sizeof(kStaticPrograms[0]));

// Ordered programs built in (bucketed table with an ordered observable, fill_ordered_kernel): slots are the
// observables binned per sample, the fields only read, then the ordered observable.  Anything else: hiprtc.
template <int NOBS, int NSLOT, typename PROG>
hipError_t launch_fill_ordered_k(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                                 const unsigned* blk_off, hipStream_t s) {
  if (sh.tail) {   // the whole step in this launch: the fill's workgroups + finisher + workers
    auto ks = fill_ordered_step_kernel<NOBS, NSLOT, PROG>;
    if (sh.lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ks),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ks, dim3(sh.grid + sh.tail_blocks), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off,
                       sh.lds_layout, (unsigned)sh.debug_mode, (unsigned)sh.grid, *static_cast<const SxTailArgs*>(sh.tail));
    return hipGetLastError();
  }
  auto k = fill_ordered_kernel<NOBS, NSLOT, PROG>;
  if (sh.lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
    if (e != hipSuccess) return e;
  }
  launch_fill(sh, k, dim3(sh.grid), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off, sh.lds_layout,
              (unsigned)sh.debug_mode);
  return hipGetLastError();
}
struct OrderedEntry {
  int nobs, nslot, nops;
  unsigned ops[4];
  FillLauncher fn;
};
#define SX_O1(NO, NS, A) {NO, NS, 1, {A, 0, 0, 0}, launch_fill_ordered_k<NO, NS, StaticProg<A>>}
#define SX_O3(NO, NS, A, B, C) {NO, NS, 3, {A, B, C, 0}, launch_fill_ordered_k<NO, NS, StaticProg<A, B, C>>}
const OrderedEntry kOrderedPrograms[] = {
    // one observable, one shift / scale (bench_sxmc pdfz): nothing is streamed but the granule words
    SX_O1(0, 1, SX_SHIFT(0)),
    SX_O1(0, 1, SX_SCALE(0)),
    // BASELINE configs 3 and 5 bucketed: e (scale + resolution_scale against e_true) binned per sample, r (shift) ordered
    SX_O3(1, 3, SX_SHIFT(2), SX_SCALE(0), SX_RES(0, 1)),
};
constexpr int kNumOrdered = (int)(sizeof(kOrderedPrograms) / sizeof(kOrderedPrograms[0]));

// Boxed programs built in (bucketed table with a boxed observable, fill_boxed_kernel): slot 0 the observable binned per
// sample, slot 1 the truth field, slot 2 the boxed observable.  Anything else: hiprtc.
template <int NOBS, int NSLOT, typename PROG>
hipError_t launch_fill_boxed_k(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                               const unsigned* blk_off, hipStream_t s) {
  auto k = fill_boxed_kernel<NOBS, NSLOT, PROG>;
  if (sh.lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
    if (e != hipSuccess) return e;
  }
  launch_fill(sh, k, dim3(sh.grid), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off, sh.lds_layout,
              (unsigned)sh.debug_mode);
  return hipGetLastError();
}
#define SX_B3(NO, NS, A, B, C) {NO, NS, 3, {A, B, C, 0}, launch_fill_boxed_k<NO, NS, StaticProg<A, B, C>>}
const OrderedEntry kBoxedPrograms[] = {
    // BASELINE config 3 bucketed: r (shift) binned per sample, e (scale + resolution_scale against e_true) boxed
    SX_B3(1, 3, SX_SHIFT(0), SX_SCALE(2), SX_RES(2, 1)),
};
constexpr int kNumBoxed = (int)(sizeof(kBoxedPrograms) / sizeof(kBoxedPrograms[0]));
template <int NOBS, int NSLOT>
hipError_t launch_fill_dyn(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                           const unsigned* blk_off, hipStream_t s) {
  return sh.lds_hist ? launch_fill_k<NOBS, NSLOT, true, DynamicProg>(sh, descs, segs, blk_off, s)
                     : launch_fill_k<NOBS, NSLOT, false, DynamicProg>(sh, descs, segs, blk_off, s);
}

}  // namespace

bool sx_fill_has_specialization(int nobs, int nslot) {
  return nobs >= 1 && nobs <= 5 && nslot >= nobs && nslot <= nobs + 2;
}

// does static program `prog` have a kernel for this histogram mode, without / with a pre-binned column
// (prebin = 1) / for a bucketed table (prebin = 3)?
bool sx_fill_static_supports(int prog, int lds_hist, int prebin) {
  if (prog < 0 || prog >= kNumStatic) return false;
  const StaticEntry& e = kStaticPrograms[prog];
  if (prebin == kPreGranule) return e.fn_gran[lds_hist ? 0 : 1] != nullptr;
  if (prebin) return (lds_hist ? e.fn_pre[0] : e.fn_g_pre[0]) != nullptr;
  return (lds_hist ? e.fn : e.fn_g) != nullptr;
}

// Does the launch described by `sh` also exist fused with the step end (see fill_step_kernel)?
bool sx_fill_has_step_form(const SxLaunchShape& sh) {
  if (sh.rtc_fill || !sh.lds_hist || sh.grid <= 0) return false;
  if (sh.pre_width == kPreOrdered) return sh.static_prog >= 0 && sh.static_prog < kNumOrdered;
  if (sh.pre_width == 1 || sh.pre_width == 2) {
    return sh.static_prog >= 0 && sh.static_prog < kNumStatic && kStaticPrograms[sh.static_prog].nops == 0 &&
           kStaticPrograms[sh.static_prog].fn_pre[sh.pre_width - 1] != nullptr;
  }
  return false;
}
size_t sx_tail_args_bytes() { return sizeof(SxTailArgs); }
// Fills the host image of a group's SxTailArgs (the device struct is private to this file).
void sx_tail_args_fill(void* image, const SxSignalDesc* lookup_descs, const SxSignalDesc* hist_descs, int nsig,
                       int max_bins, unsigned long long npoints, const unsigned* weight, unsigned long long* slots,
                       double* last_good, unsigned* sync, int nvb, const SxStepArgs& a) {
  SxTailArgs t;
  std::memset(&t, 0, sizeof t);   // (padding too: the host compares images bytewise to see whether to upload again)
  int zb = (max_bins / 4 + 127) / 128;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  t.lookup_descs = lookup_descs;
  t.hist_descs = hist_descs;
  t.nsig = nsig;
  t.nvb = (unsigned)nvb;
  t.zblocks = (unsigned)zb;
  t.npoints = npoints;
  t.weight = weight;
  t.slots = slots;
  t.last_good = last_good;
  t.sync = sync;
  std::memcpy(&t.a, &a, sizeof a);
  std::memcpy(image, &t, sizeof t);
}

bool sx_fill_static_supports_sparse_runs(int prog) {
  return prog >= 0 && prog < kNumStatic && kStaticPrograms[prog].fn_sruns != nullptr;
}

hipError_t sx_launch_fill_sparse_runs(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                                      const unsigned* blk_off, hipStream_t s) {
  if (sh.grid <= 0) return hipSuccess;
  if (sh.rtc_sparse) {
    const unsigned smax = (unsigned)(sh.sparse_lds_bytes / 4 / ((unsigned)sh.threads / 64u) / 2);
    return sx_rtc_launch(sh.rtc_sparse, sh.grid, sh.threads, sh.sparse_lds_bytes, descs, segs, blk_off, smax,
                         (unsigned)sh.debug_mode, s, sh.ev_start, sh.ev_stop);
  }
  if (!sx_fill_static_supports_sparse_runs(sh.static_prog)) return hipErrorInvalidValue;
  return kStaticPrograms[sh.static_prog].fn_sruns(sh, descs, segs, blk_off, s);
}

hipError_t sx_launch_prebin(const SxSignalDesc* d_desc, unsigned long long npad, unsigned mask, int width, void* out,
                            hipStream_t s) {
  if (npad == 0) return hipSuccess;
  unsigned long long bx = (npad + 255) / 256;
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(prebin_kernel, dim3((unsigned)bx), dim3(256), 0, s, d_desc, mask, width, out);
  return hipGetLastError();
}

int sx_fill_find_static_program(int nobs, int nslot, int nops, const unsigned* ops) {
  for (int i = 0; i < kNumStatic; i++) {
    const StaticEntry& e = kStaticPrograms[i];
    if (e.nobs != nobs || e.nslot != nslot || e.nops != nops) continue;
    bool same = true;
    for (int k = 0; k < nops; k++) same = same && e.ops[k] == ops[k];
    if (same) return i;
  }
  return -1;
}

int sx_fill_find_ordered_program(int nobs, int nslot, int nops, const unsigned* ops) {
  for (int i = 0; i < kNumOrdered; i++) {
    const OrderedEntry& e = kOrderedPrograms[i];
    if (e.nobs != nobs || e.nslot != nslot || e.nops != nops) continue;
    bool same = true;
    for (int k = 0; k < nops; k++) same = same && e.ops[k] == ops[k];
    if (same) return i;
  }
  return -1;
}

int sx_fill_find_boxed_program(int nobs, int nslot, int nops, const unsigned* ops) {
  for (int i = 0; i < kNumBoxed; i++) {
    const OrderedEntry& e = kBoxedPrograms[i];
    if (e.nobs != nobs || e.nslot != nslot || e.nops != nops) continue;
    bool same = true;
    for (int k = 0; k < nops; k++) same = same && e.ops[k] == ops[k];
    if (same) return i;
  }
  return -1;
}

hipError_t sx_launch_fill(const SxLaunchShape& sh, const SxSignalDesc* descs, const SxSegment* segs,
                          const unsigned* blk_off, hipStream_t s) {
  if (sh.grid <= 0) return hipSuccess;
  if (sh.pre_width == kPreBoxed && !sh.rtc_fill) {
    if (sh.static_prog < 0 || sh.static_prog >= kNumBoxed) return hipErrorInvalidValue;
    return kBoxedPrograms[sh.static_prog].fn(sh, descs, segs, blk_off, s);
  }
  if (sh.pre_width == kPreOrdered && !sh.rtc_fill) {
    if (sh.static_prog < 0 || sh.static_prog >= kNumOrdered) return hipErrorInvalidValue;
    return kOrderedPrograms[sh.static_prog].fn(sh, descs, segs, blk_off, s);
  }
  if (sh.rtc_fill) {
    const unsigned hist_words = (sh.pre_width == kPreOrdered || sh.pre_width == kPreBoxed) ? sh.lds_layout
                                                            : (unsigned)(sh.lds_bytes / 4 - 4 - (sh.lds_hist ? 64 : 0));
    return sx_rtc_launch(sh.rtc_fill, sh.grid, sh.threads, sh.lds_bytes, descs, segs, blk_off, hist_words,
                         (unsigned)sh.debug_mode, s, sh.ev_start, sh.ev_stop);
  }
  if (sh.static_prog >= 0 && sh.static_prog < kNumStatic) {
    const StaticEntry& e = kStaticPrograms[sh.static_prog];
    FillLauncher fn = nullptr;
    if (sh.pre_width == 0) {
      fn = sh.lds_hist ? e.fn : e.fn_g;
    } else if (sh.pre_width == kPreGranule) {
      fn = e.fn_gran[sh.lds_hist ? 0 : 1];
    } else if (sh.pre_width == 1 || sh.pre_width == 2) {
      fn = sh.lds_hist ? e.fn_pre[sh.pre_width - 1] : e.fn_g_pre[sh.pre_width - 1];
    }
    if (!fn) return hipErrorInvalidValue;
    return fn(sh, descs, segs, blk_off, s);
  }
#define SX_CASE(NO, NS) \
  if (sh.nobs == NO && sh.nslot == NS) return launch_fill_dyn<NO, NS>(sh, descs, segs, blk_off, s);
  SX_CASE(1, 1) SX_CASE(1, 2) SX_CASE(1, 3)
  SX_CASE(2, 2) SX_CASE(2, 3) SX_CASE(2, 4)
  SX_CASE(3, 3) SX_CASE(3, 4) SX_CASE(3, 5)
  SX_CASE(4, 4) SX_CASE(4, 5) SX_CASE(4, 6)
  SX_CASE(5, 5) SX_CASE(5, 6) SX_CASE(5, 7)
#undef SX_CASE
  if (sh.lds_hist) {
    auto k = fill_kernel_generic<true>;
    if (sh.lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh.lds_bytes);
      if (e != hipSuccess) return e;
    }
    launch_fill(sh, k, dim3(sh.grid), dim3(sh.threads), sh.lds_bytes, s, descs, segs, blk_off,
                (unsigned)(sh.lds_bytes / 4 - 4 - 64));
  } else {
    launch_fill(sh, fill_kernel_generic<false>, dim3(sh.grid), dim3(sh.threads), (size_t)64, s, descs, segs, blk_off, 0u);
  }
  return hipGetLastError();
}

hipError_t sx_launch_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, unsigned* ticket, hipStream_t s) {
  if (nsig == 0) return hipSuccess;
  int bx = (max_bins / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(zero_kernel, dim3(bx, nsig), dim3(256), 0, s, d_descs, ticket);
  return hipGetLastError();
}

hipError_t sx_launch_eval_pdf(const SxSignalDesc* d_descs, int nsig, unsigned long long max_points,
                              hipStream_t s) {
  if (nsig == 0 || max_points == 0) return hipSuccess;
  unsigned long long bx = (max_points + 255) / 256;
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(eval_pdf_kernel, dim3((unsigned)bx, nsig), dim3(256), 0, s, d_descs);
  return hipGetLastError();
}

hipError_t sx_launch_eval_nll(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                              const unsigned* weight, const double* pars, const double* nexpected, const unsigned* n_mc,
                              const short* source_id, const unsigned* norms, double* sums,
                              int grid, int block, hipStream_t s) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  hipLaunchKernelGGL(eval_nll_kernel, dim3(grid), dim3(block), shmem, s, d_descs, nsig, npoints, weight, pars,
                     nexpected, n_mc, source_id, norms, sums);
  return hipGetLastError();
}

hipError_t sx_launch_eval_nll_finish(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                                      const unsigned* weight, double* sums, unsigned* ticket, const SxStepArgs& a,
                                      int grid, int block, hipStream_t s) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  hipLaunchKernelGGL(eval_nll_finish_kernel, dim3(grid), dim3(block), shmem, s, d_descs, nsig, npoints, weight, sums,
                     ticket, a);
  return hipGetLastError();
}

hipError_t sx_launch_eval_nll2(const SxSignalDesc* descs_a, const SxSignalDesc* descs_b, int nsig,
                               unsigned long long npoints, const unsigned* weight_a, const unsigned* weight_b,
                               const double* pars_a, const double* pars_b, const double* nexpected,
                               const unsigned* n_mc, const short* source_id, const unsigned* norms_a,
                               const unsigned* norms_b, double* sums_a, double* sums_b, int half, int block,
                               hipStream_t s) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  hipLaunchKernelGGL(eval_nll2_kernel, dim3(2 * half), dim3(block), shmem, s, descs_a, descs_b, nsig, npoints, weight_a,
                     weight_b, pars_a, pars_b, nexpected, n_mc, source_id, norms_a, norms_b, sums_a, sums_b,
                     (unsigned)half);
  return hipGetLastError();
}

hipError_t sx_launch_finish2_zero(const SxSignalDesc* descs_a, const SxSignalDesc* descs_b, int nsig, int max_bins,
                                  size_t npartial, const double* sums_a, const double* sums_b, const unsigned* norms_b,
                                  double* v_b, const int* cap, const SxStepArgs& a, int block, hipStream_t s) {
  int zb = (max_bins / 4 + block - 1) / block;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(finish2_zero_kernel, dim3(1 + 2u * (unsigned)zb * (unsigned)nsig), dim3(block), 0, s, descs_a,
                     descs_b, nsig, (unsigned)zb, npartial, sums_a, sums_b, norms_b, v_b, cap, a);
  return hipGetLastError();
}

hipError_t sx_launch_peek_next_proposal(int nparameters, const sxmc_rng_state* rng, const float* jump_width,
                                        const double* v_current, double* out, hipStream_t s) {
  hipLaunchKernelGGL(peek_next_proposal_kernel, dim3(1), dim3(256), 0, s, nparameters, rng, jump_width, v_current, out);
  return hipGetLastError();
}

hipError_t sx_launch_tail_step(const SxSignalDesc* d_descs, int nsig, unsigned long long npoints,
                               const unsigned* weight, const SxStepArgs& a, hipStream_t s) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  hipLaunchKernelGGL(tail_step_kernel, dim3(1), dim3(1024), shmem, s, d_descs, nsig, npoints, weight, a);
  return hipGetLastError();
}

hipError_t sx_launch_finish_zero(const SxSignalDesc* d_descs, int nsig, int max_bins, size_t npartial,
                                 const double* sums, unsigned* ticket, const SxStepArgs& a, int block, hipStream_t s) {
  int zb = (max_bins / 4 + block - 1) / block;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(finish_zero_kernel, dim3(1 + (unsigned)zb * (unsigned)nsig), dim3(block), 0, s, d_descs, nsig,
                     (unsigned)zb, npartial, sums, ticket, a);
  return hipGetLastError();
}

hipError_t sx_launch_step_end(const SxSignalDesc* lookup_descs, const SxSignalDesc* hist_descs, int nsig, int max_bins,
                              unsigned long long npoints, const unsigned* weight, unsigned long long* slots,
                              double* last_good, unsigned* sync, int nvb, const SxStepArgs& a, hipStream_t s) {
  const int block = 128;   // (the rows of a virtual block of the event sum: eval_nll_kernel's workgroup)
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  int zb = (max_bins / 4 + block - 1) / block;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(step_end_kernel, dim3((unsigned)nvb + 1u), dim3(block), shmem, s, lookup_descs, hist_descs, nsig,
                     npoints, weight, slots, last_good, sync, (unsigned)nvb, (unsigned)zb, a);
  return hipGetLastError();
}

hipError_t sx_launch_step_end2(const SxSignalDesc* lookup_a, const SxSignalDesc* lookup_b, const SxSignalDesc* hist_a,
                               const SxSignalDesc* hist_b, int nsig, int max_bins, unsigned long long npoints,
                               const unsigned* weight_a, const unsigned* weight_b, unsigned long long* slots,
                               double* last_good, unsigned* sync, int nvb, const unsigned* norms_b, double* v_b,
                               const int* cap, const SxStepArgs& a, hipStream_t s) {
  const int block = 128;
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  int zb = (max_bins / 4 + block - 1) / block;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(step_end2_kernel, dim3(2u * (unsigned)nvb + 1u), dim3(block), shmem, s, lookup_a, lookup_b, hist_a,
                     hist_b, nsig, npoints, weight_a, weight_b, slots, last_good, sync, (unsigned)nvb, (unsigned)zb,
                     norms_b, v_b, cap, a);
  return hipGetLastError();
}

// Workgroups of the cooperative step end the device holds at once (the runtime's occupancy figure for step_end_kernel
// with this launch's LDS x the CU count; 0: could not be asked) -- what the host's residency check counts against
int sx_step_end_resident_capacity(int nsig, int cus) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, step_end_kernel, 128, shmem) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return per_cu * cus;
}

// the cooperative step end's slots start out empty
hipError_t sx_step_end_slots_init(unsigned long long* slots, double* last_good, int n) {
  std::vector<unsigned long long> h((size_t)n, kSlotEmpty);
  hipError_t e = hipMemcpy(slots, h.data(), sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(last_good, 0, sizeof(double) * (size_t)n);
  return e;
}

hipError_t sx_launch_chain_ends(const SxChainEnds& e, int nchains, int nsig, int max_bins, int block, hipStream_t s) {
  const size_t shmem = 16 * sizeof(double) + (size_t)((nsig + 15) / 16 * 16) * sizeof(EvalMember);  // whole chunks
  unsigned widest = 1;
  for (int c = 0; c < nchains; c++) widest = e.c[c].nblocks > widest ? e.c[c].nblocks : widest;
  hipLaunchKernelGGL(eval_nll_chains_kernel, dim3(widest, (unsigned)nchains), dim3(block), shmem, s, e, nsig);
  hipError_t rc = hipGetLastError();
  if (rc != hipSuccess) return rc;
  int zb = (max_bins / 4 + block - 1) / block;
  if (zb < 1) zb = 1;
  if (zb > 1024) zb = 1024;
  hipLaunchKernelGGL(finish_zero_chains_kernel, dim3(1 + (unsigned)zb * (unsigned)nsig, (unsigned)nchains), dim3(block), 0,
                     s, e, nsig, (unsigned)zb);
  return hipGetLastError();
}

#if SXMC_MEASURE
hipError_t sx_launch_pow_int(const double* x, int n, int i, double* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pow_int_kernel, dim3((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), dim3(256), 0, s, x, n, i, out);
  return hipGetLastError();
}
#endif

hipError_t sx_launch_transpose(const float* aos, float* cols, unsigned long long nsamples, int nfields,
                               unsigned long long col_pitch, hipStream_t s) {
  unsigned long long npad = (nsamples + SXMC_VEC - 1) / SXMC_VEC * SXMC_VEC;
  if (npad == 0) return hipSuccess;
  unsigned long long bx = (npad + 255) / 256;
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)bx), dim3(256), 0, s, aos, cols, nsamples, nfields,
                     col_pitch);
  return hipGetLastError();
}

hipError_t sx_launch_untranspose_obs(const float* cols, float* out, unsigned long long nsamples, int nobs,
                                     unsigned long long col_pitch, float dataset, hipStream_t s) {
  if (nsamples == 0) return hipSuccess;
  unsigned long long bx = (nsamples + 255) / 256;
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(untranspose_obs_kernel, dim3((unsigned)bx), dim3(256), 0, s, cols, out, nsamples, nobs,
                     col_pitch, dataset);
  return hipGetLastError();
}
