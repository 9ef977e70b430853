// layout_kernels.hip -- one-off device work that builds the BUCKETED copy of an evaluator's sample table.
//
// An observable that no systematic writes has the same bin index at every evaluation (the reference
// recomputes it per sample per step, pdfz.cpp:388-398).  Samples are therefore grouped, once, by the
// tuple of bin indices of those observables ("bucket"): inside a bucket their contribution to the flat
// bin index is one constant, so the fill streams only the columns that do change (the observables some
// systematic writes + the truth fields they reference) and reads one word per 256 samples ("granule")
// for the rest.  Samples outside the domain in an untouched observable can never be counted and are
// dropped from the copy.  Counters are integers, so the order in which samples are visited does not
// change a single count: results stay bit-identical (the parity tests compare with the
// unbucketed evaluation and with the CPU restatement of the reference).
//
// Steps (host side: sxmc_launch_plan.cpp, get_bucket_sort / get_bucketed): key per sample -> stable radix sort of (key, row) ->
// first row of every key -> host lays the granules out -> gather of the streamed columns.
#include "nll_device.h"

#include <hipcub/hipcub.hpp>

#include <cstring>

#pragma clang fp contract(off)

namespace {

// key = sum over the observables in `mask` of idx_k * radix_k, idx_k = (int)((x - lo) * scale) exactly as
// the fill computes it (pdfz.cpp:388-398; idx_k == nbins_k can happen one ulp below the upper edge and is
// a bucket of its own: radix bases are nbins_k + 1); `outside` for rows outside the domain (NaN included).
struct KeyPlan {
  unsigned mask;
  unsigned outside;
  unsigned radix[SXMC_MAX_NFIELDS];
};

// rows_in: the rows in an order to keep among equal keys (the sort that follows is stable), or null: 0, 1, 2, ...
__global__ __launch_bounds__(256) void bucket_key_kernel(const SxSignalDesc* __restrict__ dp, KeyPlan plan,
                                                         const unsigned* __restrict__ rows_in,
                                                         unsigned* __restrict__ keys, unsigned* __restrict__ rows) {
  const SxSignalDesc& d = *dp;
  const unsigned long long n = d.nsamples;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const unsigned long long row = rows_in ? rows_in[i] : i;
    bool ok = true;
    unsigned key = 0;
    for (int k = 0; k < d.nobs; k++) {
      if (!((plan.mask >> k) & 1u)) continue;
      const double x = (double)d.cols[(unsigned long long)k * d.col_pitch + row];
      ok = ok && (x >= d.lower[k]) && (x < d.upper[k]);
      key += (unsigned)(int)((x - d.lower[k]) * d.scale[k]) * plan.radix[k];
    }
    keys[i] = ok ? key : plan.outside;
    rows[i] = (unsigned)row;
  }
}

// ORDERED observable (fill_ordered_kernel): key = the float's bits mapped so that unsigned order is numeric order
// (-inf ... -0 +0 ... +inf, every NaN last).
__global__ __launch_bounds__(256) void order_key_kernel(const float* __restrict__ col, unsigned long long n,
                                                        unsigned* __restrict__ keys, unsigned* __restrict__ rows) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const float x = col[i];
    const unsigned u = __float_as_uint(x);
    keys[i] = (x != x) ? 0xFFFFFFFFu : ((u >> 31) ? ~u : (u | 0x80000000u));
    rows[i] = (unsigned)i;
  }
}

// per granule: the ordered observable's value in its first and in its last row (granules are never empty here)
__global__ __launch_bounds__(256) void bucket_edges_kernel(const float* __restrict__ col,
                                                           const unsigned* __restrict__ valid,
                                                           unsigned long long ngranules, float* __restrict__ edges) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; p < ngranules; p += step) {
    const unsigned nv = valid[p] ? valid[p] : 1u;
    edges[2 * p] = col[p * 256ull];
    edges[2 * p + 1] = col[p * 256ull + nv - 1u];
  }
}

// BOXED observable (fill_boxed_kernel): rows of a bucket are to lie in small boxes of (x, t) -- the observable's raw
// value and the truth field its resolution scale reads.  Pass 1: key = x - t as an order-preserving pattern (its sorted
// order gives the strata's boundaries: quantiles of x - t).  Pass 2: key = stratum of x - t in the top bits, x below:
// inside a stratum rows ascend in x.  Any grouping is CORRECT (the boxes are taken from the rows themselves); this one
// keeps them small in both directions.  Rows with a NaN sort last.
struct BoxStrata {
  int n;                 // strata (1 .. 16)
  unsigned bound[16];    // ascending patterns: stratum = how many of bound[0 .. n-2] are <= the row's
};
__global__ __launch_bounds__(256) void box_key_kernel(const float* __restrict__ colx, const float* __restrict__ colt,
                                                      unsigned long long n, int pass, BoxStrata strata,
                                                      unsigned* __restrict__ keys, unsigned* __restrict__ rows) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const float x = colx[i], d = x - colt[i];
    const unsigned ud = __float_as_uint(d), kd = (ud >> 31) ? ~ud : (ud | 0x80000000u);
    unsigned key;
    if (pass == 1) {
      key = (d != d) ? 0xFFFFFFFFu : kd;
    } else {
      const unsigned ux = __float_as_uint(x), kx = (ux >> 31) ? ~ux : (ux | 0x80000000u);
      unsigned s = 0;
      for (int k = 0; k + 1 < strata.n; k++) s += strata.bound[k] <= kd ? 1u : 0u;
      key = (d != d || x != x) ? 0xFFFFFFFFu : ((s << 28) | (kx >> 4));
    }
    keys[i] = key;
    rows[i] = (unsigned)i;
  }
}

// per granule: the box of its valid rows, {xmin, xmax, tmin, tmax}; NaN in all four if a valid row holds a value that
// is not finite (such a granule is left to the float columns at every evaluation).  One wave per granule.
__global__ __launch_bounds__(256) void bucket_boxes_kernel(const float* __restrict__ colx, const float* __restrict__ colt,
                                                           const unsigned* __restrict__ valid,
                                                           unsigned long long ngranules, float* __restrict__ boxes) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
  for (unsigned long long p = wave; p < ngranules; p += nwaves) {
    const unsigned nv = valid[p];
    float xl = __builtin_inff(), xh = -__builtin_inff(), tl = __builtin_inff(), th = -__builtin_inff();
    bool bad = false;
    for (unsigned i = lane; i < nv; i += 64u) {
      const float x = colx[p * 256ull + i], t = colt[p * 256ull + i];
      bad = bad || !(fabsf(x) < __builtin_inff()) || !(fabsf(t) < __builtin_inff());
      xl = fminf(xl, x);
      xh = fmaxf(xh, x);
      tl = fminf(tl, t);
      th = fmaxf(th, t);
    }
    for (int off = 32; off > 0; off >>= 1) {
      xl = fminf(xl, __shfl_down(xl, off, 64));
      xh = fmaxf(xh, __shfl_down(xh, off, 64));
      tl = fminf(tl, __shfl_down(tl, off, 64));
      th = fmaxf(th, __shfl_down(th, off, 64));
    }
    const bool anybad = __ballot(bad) != 0ull || nv == 0u;
    if (lane == 0) {
      const float nanv = __int_as_float(0x7fc00000);
      boxes[4 * p] = anybad ? nanv : xl;
      boxes[4 * p + 1] = anybad ? nanv : xh;
      boxes[4 * p + 2] = anybad ? nanv : tl;
      boxes[4 * p + 3] = anybad ? nanv : th;
    }
  }
}

// first[k] = position of the first row with key k in the sorted order (first[] pre-filled with all ones)
__global__ __launch_bounds__(256) void bucket_first_kernel(const unsigned* __restrict__ sorted_keys,
                                                           unsigned long long n, unsigned* __restrict__ first) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const unsigned k = sorted_keys[i];
    if (i == 0 || sorted_keys[i - 1] != k) first[k] = (unsigned)i;
  }
}

// One workgroup per granule per pass: row i of granule p comes from sorted position src[p] + i when
// i < valid[p]; the rest of the granule is padding (NaN: outside every domain).
struct GatherCols {
  int ncols;
  int col[SXMC_MAX_NFIELDS];
};

__global__ __launch_bounds__(256) void bucket_gather_kernel(const float* __restrict__ cols, unsigned long long pitch,
                                                            GatherCols gc, const unsigned* __restrict__ sorted_rows,
                                                            const unsigned* __restrict__ src,
                                                            const unsigned* __restrict__ valid,
                                                            unsigned long long ngranules, float* __restrict__ out,
                                                            unsigned long long out_pitch) {
  for (unsigned long long p = blockIdx.x; p < ngranules; p += gridDim.x) {
    const unsigned i = threadIdx.x;
    const bool live = i < valid[p];
    const unsigned row = live ? sorted_rows[(unsigned long long)src[p] + i] : 0u;
    for (int c = 0; c < gc.ncols; c++) {
      out[(unsigned long long)c * out_pitch + p * 256ull + i] =
          live ? cols[(unsigned long long)gc.col[c] * pitch + row] : __int_as_float(0x7fc00000);
    }
  }
}

// CODES (fill_ordered_body in fill_kernels.inc.h).  The streamed columns of a bucketed copy once more, each value as a
// 16-bit code inside a window [base, base + 65534 step]: code = floor((x - base) / step), so that
// |x - (base + (code + 1/2) step)| <= step / 2 -- CHECKED here in double, with the slack the fill's error bound
// grants (2^-20 relative); a value that fails the check, or lies outside the window, marks its row "ask the exact
// columns" (high half of word 0 = 0xFFFE), a value that is not finite marks it "never counted" (0xFFFF: NaN and
// +-inf stay outside every domain under systematics with finite coefficients; the granules' padding rows are NaN).
struct CodePlan {
  int nslots;
  double base[SXMC_MAX_QSLOTS], step[SXMC_MAX_QSLOTS];
};

__device__ __forceinline__ unsigned ordered_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}

// finite minimum and maximum of every column, as order-preserving bit patterns (mm[2c], mm[2c + 1]; pre-set to
// 0xFFFFFFFF / 0 by the host)
__global__ __launch_bounds__(256) void column_minmax_kernel(const float* __restrict__ cols, unsigned long long pitch,
                                                            int ncols, unsigned long long n, unsigned* __restrict__ mm) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (int c = 0; c < ncols; c++) {
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
      const float x = cols[(unsigned long long)c * pitch + i];
      if (fabsf(x) < __builtin_inff()) {
        const unsigned b = ordered_bits(x);
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63u) == 0u) {
      if (lo != 0xFFFFFFFFu) atomicMin(&mm[2 * c], lo);
      if (hi != 0u) atomicMax(&mm[2 * c + 1], hi);
    }
  }
}

__global__ __launch_bounds__(256) void column_codes_kernel(const float* __restrict__ cols, unsigned long long pitch,
                                                           CodePlan plan, unsigned long long n,
                                                           unsigned* __restrict__ qcol, unsigned* __restrict__ tally) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  unsigned nexact = 0u, nnever = 0u;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    unsigned code[SXMC_MAX_QSLOTS] = {0u, 0u, 0u, 0u};
    bool never = false, exact = false;
    for (int m = 0; m < plan.nslots; m++) {
      const double x = (double)cols[(unsigned long long)m * pitch + i];
      if (!(fabs(x) < __builtin_inf())) {
        never = true;
        continue;
      }
      const double t = (x - plan.base[m]) / plan.step[m];
      if (!(t >= 0.0 && t < (double)(SXMC_QCODE_MAX + 1u))) {
        exact = true;
        continue;
      }
      const unsigned q = (unsigned)t;
      const double centre = plan.base[m] + ((double)q + 0.5) * plan.step[m];
      if (!(fabs(x - centre) <= 0.5 * plan.step[m] * (1.0 + 0x1p-20))) exact = true;
      code[m] = q;
    }
    if (never) code[0] = SXMC_QCODE_NEVER; else if (exact) code[0] = SXMC_QCODE_EXACT;
    nexact += (!never && exact) ? 1u : 0u;
    nnever += never ? 1u : 0u;
    for (int w = 0; w < (plan.nslots + 1) / 2; w++) {
      qcol[(unsigned long long)w * pitch + i] = (code[2 * w] << 16) | (2 * w + 1 < plan.nslots ? code[2 * w + 1] : 0u);
    }
  }
  if (nexact) atomicAdd(&tally[0], nexact);
  if (nnever) atomicAdd(&tally[1], nnever);
}

// ONE column as 16-bit codes, one per row (fill_boxed_kernel streams 8 bytes per lane and unit): the same window, check
// and special values as above (0xFFFE: outside the window or failing the check, 0xFFFF: not finite)
__global__ __launch_bounds__(256) void column_codes16_kernel(const float* __restrict__ col, double base, double qstep,
                                                             unsigned long long n, unsigned short* __restrict__ qcol,
                                                             unsigned* __restrict__ tally) {
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  unsigned nexact = 0u, nnever = 0u;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const double x = (double)col[i];
    unsigned code;
    if (!(fabs(x) < __builtin_inf())) {
      code = SXMC_QCODE_NEVER;
      nnever++;
    } else {
      const double t = (x - base) / qstep;
      bool exact = !(t >= 0.0 && t < (double)(SXMC_QCODE_MAX + 1u));
      const unsigned q = exact ? 0u : (unsigned)t;
      const double centre = base + ((double)q + 0.5) * qstep;
      if (!(fabs(x - centre) <= 0.5 * qstep * (1.0 + 0x1p-20))) exact = true;
      code = exact ? SXMC_QCODE_EXACT : q;
      nexact += exact ? 1u : 0u;
    }
    qcol[i] = (unsigned short)code;
  }
  if (nexact) atomicAdd(&tally[0], nexact);
  if (nnever) atomicAdd(&tally[1], nnever);
}

// EvalHist::RandomSample (pdfz.cpp:817-922) without leaving the device: a bin is drawn with probability
// proportional to its content (inverse CDF: `cdf` is the inclusive prefix sum of the histogram), then a point
// uniform inside the bin (what TH1::GetRandom does); redrawn while it falls outside the cuts, as the reference
// does (:853-857).  Counter-based generator: event e, attempt t -> Philox4x32-10(counter = (e, t), key = seed).
struct SampleGeom {
  int nobs;
  int has_cuts;
  int nbins[3];
  double lower[3], width[3];
  float cut_lo[3], cut_hi[3];
};

__global__ __launch_bounds__(256) void random_sample_kernel(const unsigned* __restrict__ cdf, unsigned nbins_total,
                                                            SampleGeom g, unsigned long long seed,
                                                            unsigned long long n, float dataset,
                                                            float* __restrict__ out, unsigned* __restrict__ exhausted) {
  const unsigned total = cdf[nbins_total - 1];
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += step) {
    float x[3] = {0.0f, 0.0f, 0.0f};
    for (unsigned attempt = 0; attempt < 1024; attempt++) {
      const sxdev::Philox4 r = sxdev::philox4x32_10(e, attempt, seed);
      const unsigned target = (unsigned)(((unsigned long long)r.x * total) >> 32);   // uniform in [0, total)
      unsigned lo = 0, hi = nbins_total - 1;                                          // first bin with cdf > target
      while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (cdf[mid] > target) hi = mid; else lo = mid + 1;
      }
      unsigned flat = lo;
      const unsigned u[3] = {r.y, r.z, r.w};
      bool ok = true;
      for (int k = g.nobs - 1; k >= 0; k--) {
        const unsigned idx = flat % (unsigned)g.nbins[k];
        flat /= (unsigned)g.nbins[k];
        x[k] = (float)(g.lower[k] + ((double)idx + ((double)u[k] + 0.5) * 2.3283064365386963e-10) * g.width[k]);
        if (g.has_cuts) ok = ok && !(x[k] > g.cut_hi[k] || x[k] < g.cut_lo[k]);
      }
      if (ok) break;
      // the reference redraws until the point passes the cuts (pdfz.cpp:838-905); a point that has not passed after
      // 1024 draws is reported to the host, which fails the call -- never handed on as an event
      if (attempt == 1023) atomicAdd(exhausted, 1u);
    }
    for (int k = 0; k < g.nobs; k++) out[e * (unsigned long long)(g.nobs + 1) + k] = x[k];
    out[e * (unsigned long long)(g.nobs + 1) + g.nobs] = dataset;
  }
}

unsigned grid_for(unsigned long long n, unsigned cap) {
  unsigned long long b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

}  // namespace

hipError_t sx_bucket_keys(const SxSignalDesc* d_desc, unsigned long long nsamples, unsigned mask, const unsigned* radix,
                          unsigned outside, const unsigned* d_rows_in, unsigned* d_keys, unsigned* d_rows,
                          hipStream_t s) {
  if (nsamples == 0) return hipSuccess;
  KeyPlan plan;
  plan.mask = mask;
  plan.outside = outside;
  for (int k = 0; k < SXMC_MAX_NFIELDS; k++) plan.radix[k] = radix[k];
  hipLaunchKernelGGL(bucket_key_kernel, dim3(grid_for(nsamples, 16384)), dim3(256), 0, s, d_desc, plan, d_rows_in, d_keys,
                     d_rows);
  return hipGetLastError();
}

hipError_t sx_order_keys(const float* d_col, unsigned long long nsamples, unsigned* d_keys, unsigned* d_rows,
                         hipStream_t s) {
  if (nsamples == 0) return hipSuccess;
  hipLaunchKernelGGL(order_key_kernel, dim3(grid_for(nsamples, 16384)), dim3(256), 0, s, d_col, nsamples, d_keys, d_rows);
  return hipGetLastError();
}

hipError_t sx_box_keys(const float* d_colx, const float* d_colt, unsigned long long nsamples, int pass, int nstrata,
                       const unsigned* bounds, unsigned* d_keys, unsigned* d_rows, hipStream_t s) {
  if (nsamples == 0) return hipSuccess;
  if (nstrata < 1 || nstrata > 16) return hipErrorInvalidValue;
  BoxStrata st{};
  st.n = nstrata;
  for (int k = 0; k + 1 < nstrata; k++) st.bound[k] = bounds ? bounds[k] : 0u;
  hipLaunchKernelGGL(box_key_kernel, dim3(grid_for(nsamples, 16384)), dim3(256), 0, s, d_colx, d_colt, nsamples, pass, st,
                     d_keys, d_rows);
  return hipGetLastError();
}

hipError_t sx_bucket_boxes(const float* d_colx, const float* d_colt, const unsigned* d_valid, unsigned long long ngranules,
                           float* d_boxes, hipStream_t s) {
  if (ngranules == 0) return hipSuccess;
  hipLaunchKernelGGL(bucket_boxes_kernel, dim3(grid_for(ngranules * 64ull, 16384)), dim3(256), 0, s, d_colx, d_colt, d_valid,
                     ngranules, d_boxes);
  return hipGetLastError();
}

hipError_t sx_bucket_edges(const float* d_col, const unsigned* d_valid, unsigned long long ngranules, float* d_edges,
                           hipStream_t s) {
  if (ngranules == 0) return hipSuccess;
  hipLaunchKernelGGL(bucket_edges_kernel, dim3(grid_for(ngranules, 4096)), dim3(256), 0, s, d_col, d_valid, ngranules,
                     d_edges);
  return hipGetLastError();
}

// stable sort of (key, row) pairs on the low `bits` bits of the key
hipError_t sx_bucket_sort(const unsigned* keys_in, unsigned* keys_out, const unsigned* rows_in, unsigned* rows_out,
                          unsigned long long n, int bits, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (n > 0x7FFFFFFFull) return hipErrorInvalidValue;
  size_t temp_bytes = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys_out, rows_in, rows_out, (int)n, 0,
                                                    bits, s);
  if (e != hipSuccess) return e;
  void* temp = nullptr;
  e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
  if (e != hipSuccess) return e;
  e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, rows_in, rows_out, (int)n, 0, bits, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(temp);
  return e != hipSuccess ? e : e2;
}

hipError_t sx_bucket_first(const unsigned* sorted_keys, unsigned long long n, unsigned* d_first, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(bucket_first_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, sorted_keys, n, d_first);
  return hipGetLastError();
}

hipError_t sx_bucket_gather(const float* cols, unsigned long long pitch, int ncols, const int* col_list,
                            const unsigned* sorted_rows, const unsigned* d_src, const unsigned* d_valid,
                            unsigned long long ngranules, float* out, unsigned long long out_pitch, hipStream_t s) {
  if (ngranules == 0 || ncols == 0) return hipSuccess;
  GatherCols gc;
  gc.ncols = ncols;
  for (int c = 0; c < SXMC_MAX_NFIELDS; c++) gc.col[c] = c < ncols ? col_list[c] : 0;
  const unsigned grid = (unsigned)(ngranules > 65536ull * 4 ? 65536ull * 4 : ngranules);
  hipLaunchKernelGGL(bucket_gather_kernel, dim3(grid), dim3(256), 0, s, cols, pitch, gc, sorted_rows, d_src, d_valid,
                     ngranules, out, out_pitch);
  return hipGetLastError();
}

// finite minimum / maximum of `ncols` columns of `n` rows: out[2c], out[2c + 1] (min > max: no finite value)
hipError_t sx_column_minmax(const float* cols, unsigned long long pitch, int ncols, unsigned long long n, float* out,
                            hipStream_t s) {
  if (ncols <= 0 || ncols > SXMC_MAX_NFIELDS) return hipErrorInvalidValue;
  unsigned h[2 * SXMC_MAX_NFIELDS];
  for (int c = 0; c < ncols; c++) {
    h[2 * c] = 0xFFFFFFFFu;
    h[2 * c + 1] = 0u;
  }
  unsigned* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, sizeof(unsigned) * 2 * ncols);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(d, h, sizeof(unsigned) * 2 * ncols, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && n) {
    hipLaunchKernelGGL(column_minmax_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, cols, pitch, ncols, n, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(unsigned) * 2 * ncols, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return e;
  if (e2 != hipSuccess) return e2;
  for (int c = 0; c < 2 * ncols; c++) {
    const bool none = h[2 * (c / 2)] == 0xFFFFFFFFu && h[2 * (c / 2) + 1] == 0u;
    const unsigned b = h[c], u = (b >> 31) ? (b & 0x7FFFFFFFu) : ~b;
    float x;
    std::memcpy(&x, &u, 4);
    out[c] = none ? ((c & 1) ? -1.0f : 1.0f) : x;
  }
  return hipSuccess;
}

// the table of codes of `nslots` columns (CodePlan above); tally[0]: rows marked "ask the exact columns", tally[1]:
// rows marked "never counted"
hipError_t sx_column_codes(const float* cols, unsigned long long pitch, int nslots, const double* base, const double* step,
                           unsigned long long n, unsigned* qcol, unsigned long long* tally, hipStream_t s) {
  if (nslots < 1 || nslots > SXMC_MAX_QSLOTS) return hipErrorInvalidValue;
  CodePlan plan{};
  plan.nslots = nslots;
  for (int m = 0; m < nslots; m++) {
    plan.base[m] = base[m];
    plan.step[m] = step[m];
  }
  unsigned* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, sizeof(unsigned) * 2);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d, 0, sizeof(unsigned) * 2, s);
  if (e == hipSuccess && n) {
    hipLaunchKernelGGL(column_codes_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, cols, pitch, plan, n, qcol, d);
    e = hipGetLastError();
  }
  unsigned h[2] = {0u, 0u};
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (tally) {
    tally[0] = h[0];
    tally[1] = h[1];
  }
  return e != hipSuccess ? e : e2;
}

// one column as 16-bit codes, one per row (column_codes16_kernel); tally as above
hipError_t sx_column_codes16(const float* col, double base, double step, unsigned long long n, unsigned short* qcol,
                             unsigned long long* tally, hipStream_t s) {
  unsigned* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, sizeof(unsigned) * 2);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d, 0, sizeof(unsigned) * 2, s);
  if (e == hipSuccess && n) {
    hipLaunchKernelGGL(column_codes16_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, col, base, step, n, qcol, d);
    e = hipGetLastError();
  }
  unsigned h[2] = {0u, 0u};
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (tally) {
    tally[0] = h[0];
    tally[1] = h[1];
  }
  return e != hipSuccess ? e : e2;
}

// inclusive prefix sum of the histogram (counts fit 32 bits: the reference's norm is a uint32 too)
hipError_t sx_hist_cdf(const unsigned* d_bins, unsigned* d_cdf, int nbins_total, hipStream_t s) {
  size_t temp_bytes = 0;
  hipError_t e = hipcub::DeviceScan::InclusiveSum(nullptr, temp_bytes, d_bins, d_cdf, nbins_total, s);
  if (e != hipSuccess) return e;
  void* temp = nullptr;
  e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
  if (e != hipSuccess) return e;
  e = hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, d_bins, d_cdf, nbins_total, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(temp);
  return e != hipSuccess ? e : e2;
}

hipError_t sx_random_sample(const unsigned* d_cdf, int nbins_total, int nobs, const int* nbins, const double* lower,
                            const double* upper, const float* cut_lo, const float* cut_hi, unsigned long long seed,
                            unsigned long long n, float dataset, float* d_out, unsigned* d_exhausted, hipStream_t s) {
  if (n == 0) return hipSuccess;
  SampleGeom g{};
  g.nobs = nobs;
  g.has_cuts = (cut_lo && cut_hi) ? 1 : 0;
  for (int k = 0; k < nobs && k < 3; k++) {
    g.nbins[k] = nbins[k];
    g.lower[k] = lower[k];
    g.width[k] = (upper[k] - lower[k]) / nbins[k];
    g.cut_lo[k] = cut_lo ? cut_lo[k] : 0.0f;
    g.cut_hi[k] = cut_hi ? cut_hi[k] : 0.0f;
  }
  hipLaunchKernelGGL(random_sample_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, s, d_cdf, (unsigned)nbins_total, g, seed,
                     n, dataset, d_out, d_exhausted);
  return hipGetLastError();
}
