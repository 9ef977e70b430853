// fill_kernels.inc.h -- the histogram-fill kernels of libsxmc_hip.so as templates, written for gfx950 only.
// Included by pdfz_kernels.hip (the specialisations built into the library) AND compiled at run time through
// hiprtc for any other program of systematics (sxmc_rtc.cpp), so it depends on nothing but sxmc_device_types.h
// and the HIP device built-ins: no standard-library headers.
// What is computed is fixed by the reference: bin_samples src/pdfz.cpp:349-408, apply_systematic :306-331.
#pragma once

#include "sxmc_device_types.h"

#pragma clang fp contract(off)

// MEASUREMENT BUILD (make VARIANT=_measure EXTRA=-DSXMC_MEASURE=1 -> libsxmc_hip_measure.so; the run-time kernels of
// such a library are compiled the same way).  The kernels take a word `dbg` of measurement hooks -- the stream alone,
// the arithmetic alone, no histogram update, no drain, a scale on the codes' error bound: RESULTS ARE WRONG with any of
// them set.  The product library is built with SXMC_MEASURE 0: sx_dbg() is then the constant 0, every branch on it is
// folded away at compile time, and nothing in the product ABI can set it (tests/test_abi.py).
#ifndef SXMC_MEASURE
#define SXMC_MEASURE 0
#endif

namespace sxfill {

__device__ __forceinline__ constexpr unsigned sx_dbg(unsigned dbg_arg) { return SXMC_MEASURE ? dbg_arg : 0u; }

// (stand-ins for std::integral_constant / std::index_sequence)
template <int N>
struct IntC {
  static constexpr int value = N;
  constexpr operator int() const { return N; }
};
template <unsigned long... I>
struct ISeq {};
template <int N, unsigned long... I>
struct MakeISeq : MakeISeq<N - 1, (unsigned long)(N - 1), I...> {};
template <unsigned long... I>
struct MakeISeq<0, I...> {
  typedef ISeq<I...> type;
};

constexpr int kWave = 64;

// Pointers read out of a descriptor are generic (flat) to the compiler; every buffer they name is
// HBM, so say so: global_* instructions keep vmcnt in order (flat_* would force vmcnt(0)
// lgkmcnt(0) before every use and defeat the load pipelining below).
template <typename T>
using gptr = __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ gptr<T> to_global(T* p) {
  return (gptr<T>)p;
}

__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

// a * b + c on 24-bit operands in ONE instruction (the compiler splits __mul24(a, b) + c into a multiply and an add)
__device__ __forceinline__ int mad24(int a, int b, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// LDS word of histogram bin b in the swizzled layout of fill_ordered_body (an involution inside each 64-word block)
__device__ __forceinline__ unsigned lds_slot(unsigned b) { return b ^ ((b >> 6) & 63u); }

__device__ __forceinline__ double uniform_d(double x) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// x^i for the polynomial systematics (p = sum_i c_i * pow(x, i), pdfz.cpp:310-314).  The reference calls libm's
// pow, whose result for these small integer exponents is the correctly rounded power (up to the rare cases libm
// itself misrounds).  Repeated multiplication rounds i - 1 times and differs from that in the last bit for
// i >= 3 (~20 % of the values), so the running power is kept as an unevaluated sum h + l (error-free product
// by FMA, renormalised): h is x^i rounded ONCE -- exact for i <= 1, the plain product for i = 2, the correctly
// rounded power beyond unless x^i lies within 2^-100 of a rounding boundary.
__device__ __forceinline__ void pow_step(double& h, double& l, double x) {
  const double ph = h * x;
  const double pl = __builtin_fma(h, x, -ph) + l * x;
  const double s = ph + pl;
  l = pl - (s - ph);
  h = s;
}

// ------------------------------------------------------------------------------------ fill
// Compile-time slot dispatch: the slot index is wave-uniform (it comes from the descriptor),
// so this is a scalar branch to code that addresses the slot's registers directly.
template <int NSLOT, typename Fn>
__device__ __forceinline__ void with_slot(int slot, Fn&& fn) {
  switch (slot) {
    case 0: fn(IntC<0>{}); break;
    case 1: if constexpr (NSLOT > 1) fn(IntC<1>{}); break;
    case 2: if constexpr (NSLOT > 2) fn(IntC<2>{}); break;
    case 3: if constexpr (NSLOT > 3) fn(IntC<3>{}); break;
    case 4: if constexpr (NSLOT > 4) fn(IntC<4>{}); break;
    case 5: if constexpr (NSLOT > 5) fn(IntC<5>{}); break;
    case 6: if constexpr (NSLOT > 6) fn(IntC<6>{}); break;
    default: break;
  }
}

// apply_systematic (pdfz.cpp:316-330) on SXMC_VEC samples of one slot, p already formed.
template <int NSLOT>
__device__ __forceinline__ void apply_transform(double (&f)[NSLOT][SXMC_VEC], int type,
                                                double (&x)[SXMC_VEC], int extra_slot,
                                                const double (&p)[SXMC_VEC]) {
  switch (type) {
    case SXMC_SYST_SHIFT:
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) x[q] = x[q] + p[q];
      break;
    case SXMC_SYST_SCALE:
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) x[q] = x[q] * (1 + p[q]);
      break;
    case SXMC_SYST_CTSCALE:
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) x[q] = 1 + (x[q] - 1) * (1 + p[q]);
      break;
    case SXMC_SYST_RESOLUTION_SCALE:
      with_slot<NSLOT>(extra_slot, [&](auto E) {
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) x[q] = x[q] + (p[q] * (x[q] - f[E][q]));
      });
      break;
    default:
      break;
  }
}

__device__ __forceinline__ double readlane_d(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// The systematics of one member as a "program" held in two lane-indexed registers:
//   lane s of `opword` = type | obs_slot << 4 | extra_slot << 8 | npars << 12 | coef_start << 16
//   lane c of `coef`   = the c-th polynomial coefficient, already read from the parameter buffer
__device__ __forceinline__ unsigned pack_opword(const SxSystOp& op) {
  return (unsigned)op.type | ((unsigned)op.obs_slot << 4) | ((unsigned)op.extra_slot << 8) |
         ((unsigned)op.npars << 12) | ((unsigned)op.coef_start << 16);
}

// DYNAMIC program: one systematic (apply_systematic, pdfz.cpp:306-331) on SXMC_VEC samples per
// lane, decoded at run time.  A wave-uniform loop over the systematics reads `opword` / `coef`
// back with v_readlane, so any number of systematics of any kind runs through one copy of this
// code.  It costs scalar-unit time (decode + branches), which the CU's 16 waves share: the
// STATIC programs below exist because that, not HBM, bounded the kernel.
template <int NSLOT>
__device__ __forceinline__ void apply_op(double (&f)[NSLOT][SXMC_VEC], unsigned w, double coef) {
  const int type = (int)(w & 15u);
  const int obs_slot = (int)((w >> 4) & 15u);
  const int extra_slot = (int)((w >> 8) & 15u);
  const int npars = (int)((w >> 12) & 15u);
  const int cstart = (int)(w >> 16);
  with_slot<NSLOT>(obs_slot, [&](auto K) {
    double p[SXMC_VEC];
    if (npars == 1) {
      // p = 0 + c0 * pow(x, 0) = 0 + c0 * 1 for every x (pdfz.cpp:310-314)
      const double pc = 0.0 + readlane_d(coef, cstart) * 1.0;
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) p[q] = pc;
    } else {
      // p = sum_i c_i * x^i at the current x; x^i by repeated multiplication
      double pw[SXMC_VEC], pl[SXMC_VEC];
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        p[q] = 0.0;
        pw[q] = 1.0;
        pl[q] = 0.0;
      }
      for (int i = 0; i < npars; i++) {
        const double c = readlane_d(coef, cstart + i);
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          p[q] = p[q] + c * pw[q];
          pow_step(pw[q], pl[q], f[K][q]);
        }
      }
    }
    apply_transform<NSLOT>(f, type, f[K], extra_slot, p);
  });
}

// STATIC program: the list of one-coefficient systematics is a template argument, so the sample
// loop is straight-line vector code with the p's in scalar registers -- no decode, no branches.
// The host picks a static kernel when a launch's program matches one in the table at the end of
// this file, and the dynamic kernel otherwise.
constexpr unsigned sx_op(int type, int obs_slot, int extra_slot = 0, int npars = 0) {
  return (unsigned)type | ((unsigned)obs_slot << 4) | ((unsigned)extra_slot << 8) | ((unsigned)npars << 12);
}
// coefficients of the polynomial p = sum_i c_i x^i (pdfz.cpp:310-314); the field is 0 in the table's one-coefficient words
constexpr int sx_op_npars(unsigned w) { return ((w >> 12) & 15u) ? (int)((w >> 12) & 15u) : 1; }
template <unsigned... OPS>
constexpr int sx_prog_ncoef() {
  constexpr unsigned ops[] = {OPS..., 0u};
  int c = 0;
  for (int k = 0; k < (int)sizeof...(OPS); k++) c += sx_op_npars(ops[k]);
  return c;
}
template <unsigned... OPS>
constexpr int sx_prog_cstart(int i) {
  constexpr unsigned ops[] = {OPS..., 0u};
  int c = 0;
  for (int k = 0; k < i; k++) c += sx_op_npars(ops[k]);
  return c;
}
template <unsigned... OPS>
struct StaticProg {
  static constexpr bool dynamic = false;
  static constexpr int n = (int)sizeof...(OPS);
  static constexpr int ncoef = sx_prog_ncoef<OPS...>();
  // bit k set: some systematic writes slot k (its value changes from evaluation to evaluation)
  static constexpr unsigned touched = (0u | ... | (1u << ((OPS >> 4) & 15u)));
};
struct DynamicProg {
  static constexpr bool dynamic = true;
  static constexpr int n = 0;
  static constexpr int ncoef = 0;
  static constexpr unsigned touched = ~0u;
};

// one systematic of a static program; c = its coefficients as read from the parameter buffer (wave-uniform)
template <int NSLOT, unsigned OPC>
__device__ __forceinline__ void apply_static(double (&f)[NSLOT][SXMC_VEC], const double* c) {
  constexpr int type = (int)(OPC & 15u), K = (int)((OPC >> 4) & 15u), E = (int)((OPC >> 8) & 15u);
  constexpr int NP = sx_op_npars(OPC);
  static_assert(K < NSLOT && E < NSLOT, "slot out of range");
#pragma unroll
  for (int q = 0; q < SXMC_VEC; q++) {
    // p = sum_i c_i * pow(x, i) at the current x (pdfz.cpp:310-314): 0 + c_0 * 1 for one coefficient; x^i by
    // repeated multiplication otherwise, as in the run-time decoded program
    double pc = 0.0 + c[0] * 1.0;
    if constexpr (NP > 1) {
      double pw = 1.0 * f[K][q], pl = 0.0;
#pragma unroll
      for (int i = 1; i < NP; i++) {
        pc = pc + c[i] * pw;
        pow_step(pw, pl, f[K][q]);
      }
    }
    if constexpr (type == SXMC_SYST_SHIFT) f[K][q] = f[K][q] + pc;
    if constexpr (type == SXMC_SYST_SCALE) f[K][q] = f[K][q] * (1 + pc);
    if constexpr (type == SXMC_SYST_CTSCALE) f[K][q] = 1 + (f[K][q] - 1) * (1 + pc);
    if constexpr (type == SXMC_SYST_RESOLUTION_SCALE) f[K][q] = f[K][q] + (pc * (f[K][q] - f[E][q]));
  }
}

template <int NSLOT, unsigned... OPS, unsigned long... I>
__device__ __forceinline__ void run_static(double (&f)[NSLOT][SXMC_VEC], const double* c, StaticProg<OPS...>,
                                           ISeq<I...>) {
  (apply_static<NSLOT, OPS>(f, c + sx_prog_cstart<OPS...>((int)I)), ...);
}

typedef float vfloat4 __attribute__((ext_vector_type(4)));  // one 16-byte load per lane

// How the sample columns are read: NONTEMPORAL loads.  Right for a table far larger than the 256 MiB Infinity Cache,
// which is streamed once per evaluation and would only evict itself (+3 % at BASELINE config 3) -- and, MEASURED, also
// for one that fits (config 2: 80 MB, read again by the next evaluation): default-policy loads, which could let that
// replay hit on die, were 5-7 % SLOWER at every launch shape (profiles/r03_c2_sweep_policy_x_shape.log: 33 300
// against 35 700 evals/s at 1024 x 1).  A library built with -DSXMC_CACHED_LOADS=1 (make VARIANT=_cached
// EXTRA=-DSXMC_CACHED_LOADS=1; SXMC_HIP_LIB selects it) repeats that measurement.
// Measurement build only (make VARIANT=_stamps EXTRA=-DSXMC_WG_STAMPS=1; profiles/r04b_wg_tail_codes.log): every workgroup of
// the ordered fill leaves the 100 MHz real-time counter at its entry, at the end of its stream and at its exit.
#ifndef SXMC_WG_STAMPS
#define SXMC_WG_STAMPS 0
#endif
#if SXMC_WG_STAMPS && !defined(__HIPCC_RTC__)
extern "C" __device__ unsigned long long sx_wg_stamps[3 * 4096];
#define SX_WG_STAMP(which)                                                                      \
  do {                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 4096u) {                                               \
      sx_wg_stamps[(which)*4096 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();               \
    }                                                                                           \
  } while (0)
#else
#define SX_WG_STAMP(which) \
  do {                     \
  } while (0)
#endif
#ifndef SXMC_CACHED_LOADS
#define SXMC_CACHED_LOADS 0
#endif
template <typename T>
__device__ __forceinline__ T stream_load(gptr<const T> p) {
  if constexpr (SXMC_CACHED_LOADS) return *p;
  else return __builtin_nontemporal_load(p);
}

// SPARSE COUNTING.  A histogram too large for LDS costs one scattered HBM atomic per sample plus zeroing
// the whole array, but the likelihood only ever looks it up at the data events' bins.  When an evaluation
// is for lookup, such a member counts into one counter per distinct event bin instead: flat bin index ->
// one-hash bit filter (rejects ~98 % of the samples with one cached word) -> open-addressing table ->
// counter slot.  Same counts at the bins that are read, same norm.
typedef unsigned vuint2 __attribute__((ext_vector_type(2)));

// table part: the bit filter said "maybe"
__device__ __forceinline__ void sparse_lookup(const SxSignalDesc& d, gptr<unsigned> counters, unsigned bin) {
  const unsigned mask = (1u << (32 - d.sparse_table_shift)) - 1u;
  unsigned hp = (bin * 0x85EBCA6Bu) >> d.sparse_table_shift;
  for (unsigned probe = 0; probe <= mask; probe++) {
    const vuint2 e = to_global(reinterpret_cast<const vuint2*>(d.sparse_table))[hp];
    if (e[0] == bin) {
      __hip_atomic_fetch_add(&counters[e[1]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (e[0] == 0xFFFFFFFFu) return;
    hp = (hp + 1u) & mask;
  }
}

__device__ __forceinline__ void sparse_count(const SxSignalDesc& d, gptr<unsigned> counters, unsigned bin) {
  const unsigned hb = (bin * 0x9E3779B1u) >> d.sparse_filter_shift;
  if (!((to_global(d.sparse_filter)[hb >> 5] >> (hb & 31u)) & 1u)) return;
  sparse_lookup(d, counters, bin);
}

// PRE-BINNING.  An observable that no systematic writes has the same value, hence the same bin index
// and the same in/out-of-domain status, at every evaluation.  For static programs the host builds, once,
// a narrow column holding sum_k idx_k * stride_k over those observables (all ones = outside the domain)
// with exactly the arithmetic below, and the fill streams that column (PREW = 1, 2 or 4 bytes per
// sample) instead of the float columns it replaces.  PREW = 0: every observable is read and binned here.
//
// BUCKETING (PREW = kPreGranule).  The same fact taken further (layout_kernels.hip): the evaluator keeps a
// copy of the table sorted by the bin indices of the untouched observables, holding only the columns that
// change (samples outside the domain in an untouched observable are not in it at all).  The kernel then
// sees a lower-dimensional problem -- every observable it is given is binned here -- plus ONE word per
// 256-sample granule: the granule's constant contribution to the flat bin index.  A wave reads exactly one
// granule per step (units are 4 samples, a wave covers 64 consecutive units), so that word is wave-uniform.
constexpr int kPreGranule = 3;
template <int PREW> struct PreVec { typedef unsigned type; };
template <> struct PreVec<2> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct PreVec<4> { typedef unsigned type __attribute__((ext_vector_type(4))); };

template <int NSLOT, int PREW>
struct Columns {
  vfloat4 v[NSLOT];
  typename PreVec<PREW>::type pre;
};

// slot k is streamed unless it is an observable that the pre-binned column covers
template <int NOBS, int PREW, typename PROG>
constexpr bool slot_loaded(int k) {
  return PREW == 0 || PREW == kPreGranule || k >= NOBS || ((PROG::touched >> k) & 1u);
}

template <int PREW>
__device__ __forceinline__ unsigned pre_value(const typename PreVec<PREW>::type& p, int q) {
  if constexpr (PREW == 1) return (p >> (8 * q)) & 0xFFu;
  if constexpr (PREW == 2) return (p[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
  if constexpr (PREW == 4) return p[q];
  if constexpr (PREW == kPreGranule) return p;
  return 0u;
}
template <int PREW>
constexpr unsigned pre_sentinel() {
  return PREW == 1 ? 0xFFu : PREW == 2 ? 0xFFFFu : 0xFFFFFFFFu;
}

template <int NOBS, int NSLOT, int PREW, typename PROG>
__device__ __forceinline__ void load_columns(Columns<NSLOT, PREW>& c, const gptr<const vfloat4> (&col)[NSLOT],
                                             gptr<const typename PreVec<PREW>::type> pre, unsigned long long v) {
  // Issue order is pinned (sched_barrier): the wait-counter bookkeeping at the loop header merges
  // the prologue's and the steady state's load order, and only identical orders give counted
  // waits (vmcnt(#loads)) instead of a full drain.
#pragma unroll
  for (int k = 0; k < NSLOT; k++) {
    if constexpr (true) {
      if (slot_loaded<NOBS, PREW, PROG>(k)) {
        c.v[k] = stream_load<vfloat4>(&col[k][v]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if constexpr (PREW == kPreGranule) {
    c.pre = pre[v >> 6];   // one word per granule: every lane of the wave asks for the same address
    __builtin_amdgcn_sched_barrier(0);
  } else if constexpr (PREW != 0) {
    c.pre = stream_load<typename PreVec<PREW>::type>(&pre[v]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The histogram fill.  grid = a few workgroups per CU.  The host cuts the 4-sample units of all
// members into SEGMENTS (member, first unit, end unit, step) and gives each workgroup a short list of
// them (sxmc_launch_plan.cpp: build_partition): either one strided segment of one member ("interleaved":
// workgroups that share a member read neighbouring 8 KiB chunks at the same time, like a grid-stride
// copy) or a contiguous slice that may span members ("sliced", for many tiny members).
//
// LDS layout (LDS_HIST): word 0 workgroup in-domain counter, words 4.. the histogram (hist_words
// = largest member), then 64 "trash" words, one per lane: a sample that is outside the domain
// adds to its lane's trash word instead of being branched around, so the whole per-sample path
// is unpredicated vector code (no exec-mask juggling on the scalar unit).
template <int NOBS, int NSLOT, bool LDS_HIST, typename PROG, int PREW>
__device__ __forceinline__ void fill_body(const SxSignalDesc* __restrict__ descs, const SxSegment* __restrict__ segs,
                                          const unsigned* __restrict__ blk_off, unsigned hist_words, unsigned dbg_arg) {
  const unsigned dbg = sx_dbg(dbg_arg);   // (0 in the product build: see SXMC_MEASURE)
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned nthreads = blockDim.x;
  const unsigned lane = tid & (kWave - 1);

  unsigned* s_norm = lds;
  unsigned* hist = lds + 4;
  const unsigned trash = hist_words + lane;

  bool lds_clean = false;
  const unsigned* coarse_in_lds = nullptr;
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

    // ---- the member's systematics: coefficients (and, for the dynamic program, the op words into
    // lane-indexed registers).  Requested FIRST: loads return in order, so waiting for a coefficient that
    // was asked for after the columns below would drain those too.
    const int nsyst = d.nsyst;
    unsigned opword = 0u;
    double coef = 0.0;
    double craw[PROG::ncoef > 0 ? PROG::ncoef : 1];
    if constexpr (PROG::dynamic) {
      if ((int)lane < nsyst) opword = pack_opword(d.syst[lane]);
      if ((int)lane < d.ncoef) coef = to_global(d.params)[(long)d.coef_par[lane] * d.param_stride];
    } else {
      // the coefficients' parameter indices come with the descriptor (scalar), the values are one uniform load each
#pragma unroll
      for (int q = 0; q < PROG::ncoef; q++) craw[q] = to_global(d.params)[(long)d.coef_par[q] * d.param_stride];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the first units' columns are requested before anything else is set up: their addresses need
    // only the descriptor, and the loads fly while LDS is cleared and the geometry arrives
    gptr<const vfloat4> col[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      col[k] = to_global(reinterpret_cast<const vfloat4*>(d.cols + (unsigned long long)d.slot_col[k] * d.col_pitch));
    }
    gptr<const typename PreVec<PREW>::type> precol =
        to_global(reinterpret_cast<const typename PreVec<PREW>::type*>(d.pre));
    const unsigned long long vlast = v1 - 1;
    const unsigned long long vfirst = v0 + tid;
    Columns<NSLOT, PREW> bufA;
    load_columns<NOBS, NSLOT, PREW, PROG>(bufA, col, precol, vfirst < v1 ? vfirst : vlast);

    if (!lds_clean) {
      // whole LDS histogram (sized for the largest member), once per workgroup
      if (LDS_HIST) {
        for (unsigned b = tid; b < hist_words; b += nthreads) hist[b] = 0u;
      }
      if (tid == 0) *s_norm = 0u;
      __syncthreads();
    }
    // sparse counting: stage the member's coarse bit filter in LDS (it rejects most samples without
    // touching L2); members that share their tables share it, so it is reloaded only when it changes
    const int cshift = sparse ? d.sparse_coarse_shift : 0;
    if (sparse && d.sparse_coarse != coarse_in_lds) {
      const unsigned cwords = 1u << (32 - cshift - 5);
      for (unsigned b = tid; b < cwords && b < hist_words; b += nthreads) hist[b] = to_global(d.sparse_coarse)[b];
      coarse_in_lds = d.sparse_coarse;
      __syncthreads();
    }

    // ---- wave-uniform geometry into scalar registers
    double lo[NOBS], hi[NOBS], sc[NOBS];
    int st[NOBS];
#pragma unroll
    for (int k = 0; k < NOBS; k++) {
      lo[k] = d.lower[k];
      hi[k] = d.upper[k];
      sc[k] = d.scale[k];
      st[k] = d.bin_stride[k];
    }

    unsigned cnt = 0;

    // ---- the sample loop.  One stage = wait for the unit's raw columns, widen them to double,
    // immediately re-issue the loads for the next unit into the same registers, then do the arithmetic
    // under them.  ONE unit in flight per lane, on purpose: HBM delivers most when a CU keeps about 32 KiB
    // of loads in flight (tools/hbm_probe.hip: 7.1 TB/s there, 5.5-6.5 TB/s at twice that), and 512 lanes
    // per CU x 3-4 columns x 16 bytes is that much; a second unit in flight (tried: a two-deep register
    // ring) or more waves per CU only queue up behind the memory system (-8 %); for SHORT launches too (config 2,
    // ~10 units per lane: two units in flight changed nothing, profiles/r03_c2_sweep_two_units_in_flight.log --
    // such a launch is bound by its fixed cost, not by a lane's chain of round trips).  Loads are unconditional
    // (index clamped into the slice) so the wait counters stay exact; lanes past the end of the slice
    // are treated like out-of-domain samples.
    auto stage = [&](Columns<NSLOT, PREW>& buf, const unsigned long long vc) {
      double f[NSLOT][SXMC_VEC];
#pragma unroll
      for (int k = 0; k < NSLOT; k++) {
        if (slot_loaded<NOBS, PREW, PROG>(k)) {
          f[k][0] = (double)buf.v[k].x;
          f[k][1] = (double)buf.v[k].y;
          f[k][2] = (double)buf.v[k].z;
          f[k][3] = (double)buf.v[k].w;
        } else {
          f[k][0] = f[k][1] = f[k][2] = f[k][3] = 0.0;
        }
      }
      typename PreVec<PREW>::type prebits = buf.pre;
      // Pin every widening BEFORE the buffer is re-loaded: if the compiler sinks one of them
      // below, that column's registers stay live across the reload, the reload lands in fresh
      // registers and the loop latch copies them back behind a vmcnt(0) that drains the ring.
#pragma unroll
      for (int k = 0; k < NSLOT; k++) {
        if (slot_loaded<NOBS, PREW, PROG>(k)) {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) asm volatile("" : "+v"(f[k][q]));
        }
      }
      if constexpr (PREW != 0) asm volatile("" : "+v"(prebits));
      // dbg: measurement hooks (libsxmc_hip_measure.so's sxmc_group_set_debug_mode), the constant 0 in the product:
      //   bit 1: every reload hits one cached address -> the kernel without its HBM stream
      //   bit 0: skip the arithmetic and the histogram -> the HBM stream alone
      //   bit 2: skip only the histogram update
      const unsigned long long vl = vc + step;
      load_columns<NOBS, NSLOT, PREW, PROG>(buf, col, precol, (vl < v1 && !(dbg & 2u)) ? vl : vlast);
      if (dbg & 1u) {
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) cnt += (f[k][q] == 12345.678) ? 1u : 0u;
        }
        if constexpr (PREW != 0) cnt += (pre_value<PREW>(prebits, 0) == 12345u) ? 1u : 0u;
        return;
      }

      if constexpr (PROG::dynamic) {
        for (int s = 0; s < nsyst; s++) {
          apply_op<NSLOT>(f, (unsigned)__builtin_amdgcn_readlane((int)opword, s), coef);
        }
      } else {
        run_static<NSLOT>(f, craw, PROG{}, typename MakeISeq<PROG::n>::type{});
      }

      // lanes past the end of the slice hold clamped duplicates: count them as failures
      const unsigned dead = (vc < v1) ? 0u : 1u;
      unsigned okbin[SXMC_VEC];   // flat bin index of a sample to be counted, or all ones (non-LDS modes)
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        // pdfz.cpp:388-398.  `bad` counts failed domain tests (each a vector compare feeding an
        // add-with-carry, nothing on the scalar unit); the tests are written so that NaN fails.
        unsigned bad = dead;
        int bin = 0;
        if constexpr (PREW != 0) {
          const unsigned pv = pre_value<PREW>(prebits, q);
          if constexpr (PREW != kPreGranule) bad += (pv == pre_sentinel<PREW>()) ? 1u : 0u;
          bin = (int)pv;
        }
#pragma unroll
        for (int k = 0; k < NOBS; k++) {
          if (!slot_loaded<NOBS, PREW, PROG>(k)) continue;  // covered by the pre-binned column
          const double x = f[k][q];
          bad += !(x >= lo[k]) ? 1u : 0u;
          bad += !(x < hi[k]) ? 1u : 0u;
          const int idx = (int)((x - lo[k]) * sc[k]);
          if (LDS_HIST) {
            // histogram fits LDS => every index and stride is far below 2^23
            // (the last observable of a full-dimensional problem has stride 1; a bucketed one need not)
            bin = (k == NOBS - 1 && PREW != kPreGranule) ? bin + idx : __mul24(idx, st[k]) + bin;
          } else {
            bin += idx * st[k];
          }
        }
        const unsigned in_domain = (bad == 0u) ? 1u : 0u;
        cnt += in_domain;
        // in domain but index out of range (the reference's one-past-the-end case) still counts in the
        // norm; it and every failure are not histogrammed
        const bool store = (bad == 0u) && ((unsigned)bin < B) && !(dbg & 4u);
        if (LDS_HIST) {
          const unsigned slot = store ? (unsigned)bin : trash;   // failures go to the lane's trash word
          __hip_atomic_fetch_add(&hist[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          okbin[q] = store ? (unsigned)bin : 0xFFFFFFFFu;
        }
      }
      if (!LDS_HIST) {
        if (sparse) {
          // all four filter words are requested before any is tested: one L2 round trip, not four
          // level 1: coarse filter in LDS, two bits per bin (passes ~3 % of the non-members at 1e5 event bins)
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            const unsigned hc = (okbin[q] != 0xFFFFFFFFu) ? (okbin[q] * 0xC2B2AE35u) >> cshift : 0u;
            const unsigned hd = (okbin[q] != 0xFFFFFFFFu) ? (okbin[q] * 0x27D4EB2Fu) >> cshift : 0u;
            if (!((hist[hc >> 5] >> (hc & 31u)) & (hist[hd >> 5] >> (hd & 31u)) & 1u)) okbin[q] = 0xFFFFFFFFu;
          }
          // level 2: fine filter through L2; all four words are requested before any is tested
          unsigned hb[SXMC_VEC], word[SXMC_VEC];
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            hb[q] = (okbin[q] != 0xFFFFFFFFu) ? (okbin[q] * 0x9E3779B1u) >> d.sparse_filter_shift : 0u;
            word[q] = to_global(d.sparse_filter)[hb[q] >> 5];
          }
          // level 3: the table, for the ~1 % that remain
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            if (okbin[q] != 0xFFFFFFFFu && ((word[q] >> (hb[q] & 31u)) & 1u)) sparse_lookup(d, gbins, okbin[q]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            if (okbin[q] != 0xFFFFFFFFu) {
              __hip_atomic_fetch_add(&gbins[okbin[q]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
    };

    // wave-uniform trip count: every lane runs the same number of stages
    unsigned long long v = vfirst;
    const unsigned long long niter = (v1 - v0 + step - 1) / step;
    for (unsigned long long it = 0; it < niter; ++it, v += step) stage(bufA, v);

    // ---- in-domain count: lane registers -> wave -> workgroup -> one global atomic
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
    if (lane == 0 && cnt != 0u) {
      __hip_atomic_fetch_add(s_norm, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();

    // ---- flush: only non-zero bins reach HBM.  No step depends on what an earlier one read (the LDS reads of
    // several steps are in flight together); the words are cleared afterwards, if another member follows.
    if (LDS_HIST) {
#pragma unroll 4
      for (unsigned b = tid; b < B; b += nthreads) {
        const unsigned c = hist[b];
        if (c != 0u) __hip_atomic_fetch_add(&gbins[b], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (si + 1u < seg_end) {
        __syncthreads();
        for (unsigned b = tid; b < B; b += nthreads) hist[b] = 0u;
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

// ORDERED OBSERVABLE (PREW = kPreOrdered).  Bucketing taken one step further, to an observable that IS written --
// but only by one-coefficient shift / scale / cos-theta-scale systematics and read by nothing else.  Each of those
// is a monotone map of the sample's value for the evaluation's parameters (IEEE addition and multiplication by a
// constant round monotonically), and so is the binning that follows: along rows sorted by the raw value the
// observable's "extended bin" (-1 below the domain, the index inside, a large value at or above the upper edge)
// is a step function.  The bucketed copy keeps the rows of each bucket in that order; a 256-row granule whose
// first and last row land in the same extended bin has every row there, so the observable's contribution to the
// flat index is one constant for the granule -- computed per evaluation from the two end values, with exactly the
// per-sample arithmetic -- and its column is not read at all.  Only granules that straddle a bin edge ("mixed":
// at most nbins + 1 per bucket) take the per-sample path over the column; granules entirely outside the domain
// are skipped.  NaN anywhere (a NaN end value or coefficient) makes a granule mixed: the per-sample path decides,
// as it always has.  Counts are integers and every sample is binned by the same arithmetic or by an argument
// about that arithmetic, so the histograms stay bit-identical (tests: every ordered case against the unordered
// evaluation and the CPU restatement, samples placed within ulps of the edges).
//
// Slots: 0 .. NOBS-1 the observables binned per sample, NOBS .. NSLOT-2 fields that are only read, NSLOT-1 the
// ordered observable's column (geometry at index NOBS of the descriptor's arrays).  Granule word `pre`:
// bits 0-23 the bucket's bin offset, bits 24-31 rows in the granule - 1.  Lane l of a wave works out the codes of
// the wave's next 64 granules in one go (one 8-byte read per granule), each step then takes its own with
// v_readlane: scalar from there on.  NCHAIN > 1: lockstep chains (see fill_multi_body), each with its own codes.
// The kernel itself (fill_ordered_body) follows fill_multi_body below; fill_sparse_body has an ORDERED variant.
constexpr int kPreOrdered = 5;
constexpr int kPreBoxed = 6;     // bucketed table with a boxed observable (fill_boxed_body, further down)
constexpr unsigned kOrdMixed = 0xFFFFFFFEu, kOrdSkip = 0xFFFFFFFFu;
typedef float vfloat2 __attribute__((ext_vector_type(2)));

template <unsigned OPC>
__device__ __forceinline__ void apply_ordered_scalar(double& x, const double* c) {
  constexpr int type = (int)(OPC & 15u);
  static_assert(sx_op_npars(OPC) == 1 && (type == SXMC_SYST_SHIFT || type == SXMC_SYST_SCALE || type == SXMC_SYST_CTSCALE),
                "not a monotone systematic");
  const double pc = 0.0 + c[0] * 1.0;   // (as apply_static)
  if constexpr (type == SXMC_SYST_SHIFT) x = x + pc;
  if constexpr (type == SXMC_SYST_SCALE) x = x * (1 + pc);
  if constexpr (type == SXMC_SYST_CTSCALE) x = 1 + (x - 1) * (1 + pc);
}
// the systematics of the program that write slot ORD, on one value
template <int ORD, unsigned... OPS, unsigned long... I>
__device__ __forceinline__ void run_ordered_scalar(double& x, const double* c, StaticProg<OPS...>, ISeq<I...>) {
  ([&] {
    if constexpr ((int)((OPS >> 4) & 15u) == ORD) apply_ordered_scalar<OPS>(x, c + sx_prog_cstart<OPS...>((int)I));
  }(), ...);
}
// the systematics that write slot ORD (WANT) or the others (!WANT), on the lane's samples
template <int NSLOT, int ORD, bool WANT, unsigned... OPS, unsigned long... I>
__device__ __forceinline__ void run_static_part(double (&f)[NSLOT][SXMC_VEC], const double* c, StaticProg<OPS...>,
                                                ISeq<I...>) {
  ([&] {
    if constexpr (((int)((OPS >> 4) & 15u) == ORD) == WANT) apply_static<NSLOT, OPS>(f, c + sx_prog_cstart<OPS...>((int)I));
  }(), ...);
}
// CODES (fill_ordered_body).  shift, scale, cos-theta scale and resolution scale with ONE coefficient are affine maps
// of the sample's fields (pdfz.cpp:316-330), and so is their composition: after the program, observable k is
//     x_k = sum_m A[k][m] * field_m + C[k]                       (in real arithmetic)
// with A and C functions of the evaluation's parameters alone.  The reference evaluates that in double, operation
// by operation; its result differs from the real-arithmetic value by a few units in the last place of the largest
// intermediate.  So a sample whose real-arithmetic bin coordinate (x_k - lo_k) * scale_k lies further from every
// integer than (what is not known about the field values) + (those roundings) lands in the same bin either way --
// and the fill may work that bin out from ANY approximation of the fields that comes with an error bound.  The
// bucketed copy therefore holds each streamed field a second time as a 16-bit code (layout_kernels.hip:
// code = floor((x - qbase) / qstep), so x = qbase + (code + 1/2) * qstep +- qstep / 2): 2 bytes per field and sample
// instead of 4.  Per evaluation every wave composes the program into A and C (AffineForm below, in double), folds
// window, binning and scale into single-precision coefficients
//     u_k = sum_m alpha[k][m] * code_m + gamma[k]
// and an error bound eps[k] that covers the half code step, the single-precision evaluation of u and the double
// roundings of the reference's own arithmetic (bounded through `mag`, a bound on every intermediate's magnitude).
// A sample with eps <= frac(u_k) <= 1 - eps in every observable is binned from its codes: floor(u_k) IS the
// reference's index (or outside the domain when it is not in [0, nbins)).  (The kernel asks it one-sidedly: it
// evaluates u' = u + e, e = 1.01 eps in the constant term, and tests fract(u') >= 2e; the reference's value lies in
// [u' - 2e, u'].)  Any other sample -- about 2 eps of them,
// a few in 10^4 -- is "ambiguous": its row number goes into a queue in LDS, and at the end of the stream the
// workgroup reads those rows' float values and bins them with the reference's arithmetic (exact_one).  Coefficients
// that are not finite, or so large that eps reaches 1/8, switch the codes off for that evaluation: the float
// columns are streamed as before.  Counts are integers; every sample is binned either by the reference's arithmetic
// or by an argument about it: bit-identical histograms (tests/test_gpu_codes.py: against the float stream and the
// CPU restatement, samples placed within ulps of the transformed edges, windows that exclude samples, queues that
// overflow).
template <int NQ>
struct AffineForm {
  double a[NQ];   // coefficients of the NQ streamed fields
  double c;       // constant
  double mag;     // >= |value| of the slot at any point of the program, for any field values inside the windows
};

template <int NQ, unsigned OPC>
__device__ __forceinline__ void apply_affine(AffineForm<NQ> (&f)[NQ], const double* coef) {
  constexpr int type = (int)(OPC & 15u), K = (int)((OPC >> 4) & 15u), E = (int)((OPC >> 8) & 15u);
  static_assert(sx_op_npars(OPC) == 1 && K < NQ, "not an affine systematic");
  const double p = 0.0 + coef[0] * 1.0;
  const double ap = __builtin_fabs(p);
  if constexpr (type == SXMC_SYST_SHIFT) {
    f[K].c = f[K].c + p;
    f[K].mag = f[K].mag + ap;
  }
  if constexpr (type == SXMC_SYST_SCALE) {
    const double s = 1 + p;
#pragma unroll
    for (int m = 0; m < NQ; m++) f[K].a[m] = f[K].a[m] * s;
    f[K].c = f[K].c * s;
    f[K].mag = f[K].mag * (1 + ap);
  }
  if constexpr (type == SXMC_SYST_CTSCALE) {
    const double s = 1 + p;
#pragma unroll
    for (int m = 0; m < NQ; m++) f[K].a[m] = f[K].a[m] * s;
    f[K].c = 1 + (f[K].c - 1) * s;
    f[K].mag = 1 + (f[K].mag + 1) * (1 + ap);
  }
  if constexpr (type == SXMC_SYST_RESOLUTION_SCALE) {
    static_assert(E < NQ, "slot out of range");
#pragma unroll
    for (int m = 0; m < NQ; m++) f[K].a[m] = f[K].a[m] + p * (f[K].a[m] - f[E].a[m]);
    f[K].c = f[K].c + p * (f[K].c - f[E].c);
    f[K].mag = f[K].mag + ap * (f[K].mag + f[E].mag);
  }
}
// the systematics of the program that do NOT write slot ORD, composed
template <int NQ, int ORD, unsigned... OPS, unsigned long... I>
__device__ __forceinline__ void run_affine(AffineForm<NQ> (&f)[NQ], const double* c, StaticProg<OPS...>, ISeq<I...>) {
  ([&] {
    if constexpr ((int)((OPS >> 4) & 15u) != ORD) apply_affine<NQ, OPS>(f, c + sx_prog_cstart<OPS...>((int)I));
  }(), ...);
}
// can the part of the program that does not write slot ORD be composed that way?
template <int ORD, unsigned... OPS>
constexpr bool prog_is_affine(StaticProg<OPS...>) {
  return (true && ... && ((int)((OPS >> 4) & 15u) == ORD || sx_op_npars(OPS) == 1));
}
__device__ __forceinline__ float uniform_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// SPARSE COUNTING OVER A BUCKETED TABLE, WALKED IN RUNS.  Histograms beyond LDS capacity, evaluated for lookup
// (BASELINE config 5).  The table is bucketed (layout_kernels.hip) and laid out so that every WAVE walks its own
// run of consecutive granules of the sorted order: a wave stays inside one bucket -- one tuple of bin indices of
// the untouched observables -- for many steps.  The event bins are grouped by the same buckets on the host, each
// bucket with a small open-addressing table keyed by the index contribution of the WRITTEN observables.  On
// entering a bucket a wave copies that table into its private slice of LDS; a sample then costs one LDS probe
// (no L2 filter, no global table, no global atomic), hits are counted in LDS beside the keys, and the counts go to
// the global event-bin counters once, when the wave leaves the bucket.  Counts are integers: same counters as
// every other evaluation form, bit for bit.
// Left to the global table (sparse_lookup): buckets with more event bins than a wave's slice holds, buckets whose
// key contains an index equal to nbins, and single samples whose written observable's index comes out as nbins
// (pdfz.cpp:388-398 lets such an index alias into the next row of the flat index; the bucket tables assume the
// canonical decomposition, the global table is keyed by the flat index itself).
typedef unsigned vuint2g __attribute__((ext_vector_type(2)));
typedef unsigned vuint4g __attribute__((ext_vector_type(4)));

template <int NOBS, int NSLOT, typename PROG, bool ORDERED = false>
__device__ __forceinline__ void fill_sparse_body(const SxSignalDesc* __restrict__ descs,
                                                 const SxSegment* __restrict__ segs,
                                                 const unsigned* __restrict__ blk_off, unsigned smax, unsigned dbg_arg) {
  const unsigned dbg = sx_dbg(dbg_arg);   // (0 in the product build: see SXMC_MEASURE)
  static_assert(!PROG::dynamic, "static programs only");
  // ORDERED: the table's last slot is an ordered observable (see fill_ordered_body further down): its index
  // contribution is one constant per granule, worked out from the granule's end values, except in the granules
  // that straddle one of its bin edges; geometry at index NOBS of the descriptor's arrays
  static_assert(!ORDERED || (NOBS >= 1 && NOBS < NSLOT), "unsupported");
  constexpr int ORD = NSLOT - 1, NSTREAM = ORDERED ? NSLOT - 1 : NSLOT;
  typedef typename MakeISeq<PROG::n>::type Seq;
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;
  unsigned* wkeys = lds + (size_t)wave * 2u * smax;   // this wave's table keys ...
  unsigned* wcnt = wkeys + smax;                      // ... and the hit counts beside them
  for (unsigned b = lane; b < smax; b += kWave) wcnt[b] = 0u;

  const unsigned seg_end = blk_off[blockIdx.x + 1];
  for (unsigned si = blk_off[blockIdx.x]; si < seg_end; ++si) {
    const SxSegment& sg = segs[si];
    const SxSignalDesc& d = descs[sg.sig];
    const unsigned long long v0 = sg.v0;
    const unsigned long long v1 = sg.v1;
    const unsigned long long step = sg.step;
    const unsigned B = (unsigned)d.sparse_real_nbins;
    gptr<unsigned> gcnt = to_global(d.bins);
    gptr<const unsigned> tkeys = to_global(d.sparse_tkeys);
    gptr<const unsigned> tslot = to_global(d.sparse_tslot);
    gptr<const vuint2g> dir = to_global(reinterpret_cast<const vuint2g*>(d.sparse_dir));

    double craw[PROG::ncoef > 0 ? PROG::ncoef : 1];
#pragma unroll
    for (int q = 0; q < PROG::ncoef; q++) craw[q] = to_global(d.params)[(long)d.coef_par[q] * d.param_stride];
    __builtin_amdgcn_sched_barrier(0);

    gptr<const vfloat4> col[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      col[k] = to_global(reinterpret_cast<const vfloat4*>(d.cols + (unsigned long long)d.slot_col[k] * d.col_pitch));
    }
    gptr<const vuint2g> gkp = to_global(reinterpret_cast<const vuint2g*>(d.pre));  // {bucket key, bin offset} per granule
    const unsigned long long vlast = v1 - 1;
    const unsigned long long vfirst = v0 + tid;

    vfloat4 raw[NSLOT];
    vuint2g kp;
    auto load = [&](unsigned long long v) {
#pragma unroll
      for (int k = 0; k < NSTREAM; k++) {
        raw[k] = __builtin_nontemporal_load(&col[k][v]);
        __builtin_amdgcn_sched_barrier(0);
      }
      kp = gkp[v >> 6];
      __builtin_amdgcn_sched_barrier(0);
    };
    load(vfirst < v1 ? vfirst : vlast);

    // The host sends this kernel only geometries whose bin counts AND strides are below 2^23 (sxmc_launch_plan.cpp:
    // group_rebuild, `narrow`), so idx * stride + bin is ONE v_mad_i32_i24 (24-bit signed operands, 32-bit result);
    // round 2 formed it from two 24-bit products of a split stride: 5 vector instructions per observable and sample.
    double lo[NOBS], hi[NOBS], sc[NOBS];
    int stv[NOBS];
    unsigned nb[NOBS];
#pragma unroll
    for (int k = 0; k < NOBS; k++) {
      lo[k] = d.lower[k];
      hi[k] = d.upper[k];
      sc[k] = d.scale[k];
      stv[k] = d.bin_stride[k];
      asm volatile("" : "+v"(stv[k]));   // (in a vector register: the multiply-add takes one scalar operand)
      nb[k] = (unsigned)d.nbins[k];
    }

    // ordered observable: geometry, and the codes of the wave's next 64 granules (one per lane)
    const double olo = d.lower[ORDERED ? NOBS : 0], ohi = d.upper[ORDERED ? NOBS : 0], osc = d.scale[ORDERED ? NOBS : 0];
    const int ost = d.bin_stride[ORDERED ? NOBS : 0];
    int ostv = ost;
    asm volatile("" : "+v"(ostv));
    const unsigned onb = (unsigned)d.nbins[ORDERED ? NOBS : 0];
    gptr<const vfloat2> edges = to_global(reinterpret_cast<const vfloat2*>(d.edges));
    const unsigned long long vwave = v0 + (tid - lane);
    bool wild = false;
#pragma unroll
    for (int q = 0; q < PROG::ncoef; q++) wild = wild || !(__builtin_fabs(craw[q]) < __builtin_inf());
    unsigned codes = 0u;

    unsigned cnt = 0;
    // the bucket this wave is in: its table sits in wkeys[0 .. 1 << cur_log2), counts in wcnt
    unsigned cur_key = 0xFFFFFFFFu, cur_off = 0u, cur_info = SXMC_SPARSE_EMPTY, cur_probes = 1u;

    auto flush = [&]() {
      if (cur_info <= SXMC_SPARSE_SMAX_LOG2) {
        const unsigned S = 1u << cur_info;
        for (unsigned b = lane; b < S; b += kWave) {
          const unsigned c = wcnt[b];
          if (c != 0u) {
            __hip_atomic_fetch_add(&gcnt[tslot[cur_off + b]], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wcnt[b] = 0u;
          }
        }
      }
    };

    unsigned long long v = vfirst;
    const unsigned long long niter = (v1 - v0 + step - 1) / step;
    for (unsigned long long it = 0; it < niter; ++it, v += step) {
      unsigned code = 0u;
      if constexpr (ORDERED) {
        const int j = (int)(it & 63ull);
        if (j == 0) {
          const unsigned long long vg = vwave + (it + lane) * step;
          const bool live = vg < v1;
          const vfloat2 e = edges[(live ? vg : vlast) >> 6];
          double x0 = (double)e.x, x1 = (double)e.y;
          run_ordered_scalar<ORD>(x0, craw, PROG{}, Seq{});
          run_ordered_scalar<ORD>(x1, craw, PROG{}, Seq{});
          const bool nan = wild || !(x0 == x0) || !(x1 == x1);
          const int i0 = (int)((x0 - olo) * osc), i1 = (int)((x1 - olo) * osc);
          const int e0 = !(x0 >= olo) ? -1 : (!(x0 < ohi) ? 0x7FFFFFFF : i0);
          const int e1 = !(x1 >= olo) ? -1 : (!(x1 < ohi) ? 0x7FFFFFFF : i1);
          const bool outside = e0 < 0 || e0 == 0x7FFFFFFF;
          // (an index equal to nbins -- one ulp below the upper edge -- aliases into the next row of the flat
          // index: such granules take the per-sample path, which knows how to deal with it)
          const bool mixedg = nan || e0 != e1 || (!outside && (unsigned)e0 >= onb);
          const unsigned cst = (unsigned)__mul24(e0, ost);
          codes = !live ? kOrdSkip : mixedg ? kOrdMixed : outside ? kOrdSkip : cst;
        }
        code = (unsigned)__builtin_amdgcn_readlane((int)codes, j);
      }
      double f[NSLOT][SXMC_VEC];
#pragma unroll
      for (int k = 0; k < NSTREAM; k++) {
        f[k][0] = (double)raw[k].x;
        f[k][1] = (double)raw[k].y;
        f[k][2] = (double)raw[k].z;
        f[k][3] = (double)raw[k].w;
      }
      const unsigned key = (unsigned)__builtin_amdgcn_readfirstlane((int)kp[0]);
      const unsigned pre = (unsigned)__builtin_amdgcn_readfirstlane((int)kp[1]);
      // (the bucket state is wave-uniform; say so, or the loop-carried copies live in vector registers and every
      // test on them becomes exec-mask code)
      cur_key = (unsigned)uniform_i((int)cur_key);
      cur_off = (unsigned)uniform_i((int)cur_off);
      cur_info = (unsigned)uniform_i((int)cur_info);
      cur_probes = (unsigned)uniform_i((int)cur_probes);
#pragma unroll
      for (int k = 0; k < NSTREAM; k++) {
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) asm volatile("" : "+v"(f[k][q]));
      }
      const bool mixed = ORDERED && code == kOrdMixed;
      vfloat4 rawo = {0.0f, 0.0f, 0.0f, 0.0f};
      if (mixed) {   // a granule that straddles a bin edge of the ordered observable: its column too, this once
        rawo = __builtin_nontemporal_load(&col[ORD][v < v1 ? v : vlast]);
        __builtin_amdgcn_sched_barrier(0);
      }
      const unsigned long long vl = v + step;
      load(vl < v1 ? vl : vlast);
      if (dbg & 1u) {
#pragma unroll
        for (int k = 0; k < NSTREAM; k++) {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) cnt += (f[k][q] == 12345.678) ? 1u : 0u;
        }
        continue;
      }
      if (ORDERED && code == kOrdSkip) continue;   // (wave-uniform) the whole granule is outside the domain

      // ---- a new bucket: counts of the old one go to the global counters, the new table comes into LDS
      if (key != cur_key) {
        flush();
        const vuint2g e = dir[key];
        cur_key = key;
        cur_off = (unsigned)__builtin_amdgcn_readfirstlane((int)e[0]);
        const unsigned info = (unsigned)__builtin_amdgcn_readfirstlane((int)e[1]);
        cur_info = info & 0xFFu;            // log2(table size) >= 2, or a flag
        cur_probes = (info >> 8) & 0xFFu;   // cells a key may have been pushed along
        if (cur_info <= SXMC_SPARSE_SMAX_LOG2) {
          const unsigned S = 1u << cur_info;
          for (unsigned b = lane; b < S; b += kWave) wkeys[b] = tkeys[cur_off + b];
        }
      }

      if constexpr (ORDERED) {
        f[ORD][0] = f[ORD][1] = f[ORD][2] = f[ORD][3] = 0.0;
        run_static_part<NSLOT, ORD, false>(f, craw, PROG{}, Seq{});
        if (mixed) {
          f[ORD][0] = (double)rawo.x;
          f[ORD][1] = (double)rawo.y;
          f[ORD][2] = (double)rawo.z;
          f[ORD][3] = (double)rawo.w;
          run_static_part<NSLOT, ORD, true>(f, craw, PROG{}, Seq{});
        }
      } else {
        run_static<NSLOT>(f, craw, PROG{}, Seq{});
      }

      // The per-sample part, written for the vector unit's instruction count (this kernel is co-bound by vector issue
      // and the stream): domain tests are compares into scalar mask registers combined on the scalar unit (NaN
      // fails), the in-domain count is an add-with-carry from that mask.  pdfz.cpp:388-398.
      const bool live = v < v1;   // lanes past the end of the slice hold clamped duplicates
      unsigned p2[SXMC_VEC];
      bool fast[SXMC_VEC], slow[SXMC_VEC];
      const bool table_here = cur_info <= SXMC_SPARSE_SMAX_LOG2;
      const bool all_slow = cur_info == SXMC_SPARSE_SLOW;
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        bool ind = live, alias = false;
        int bin = (ORDERED && !mixed) ? (int)code : 0;
        if (mixed) {
          const double x = f[ORD][q];
          ind = ind & (x >= olo) & (x < ohi);
          const int idx = (int)((x - olo) * osc);
          alias = alias | ((unsigned)idx >= onb);
          bin = mad24(idx, ostv, bin);
        }
#pragma unroll
        for (int k = 0; k < NOBS; k++) {
          const double x = f[k][q];
          ind = ind & (x >= lo[k]) & (x < hi[k]);
          const int idx = (int)((x - lo[k]) * sc[k]);
          alias = alias | ((unsigned)idx >= nb[k]);   // (one ulp below the upper edge the index can come out as nbins)
          bin = mad24(idx, stv[k], bin);              // (what a sample outside the domain gives is not used)
        }
        cnt += ind ? 1u : 0u;
        p2[q] = (unsigned)bin;
        const bool count_it = ind && !(dbg & 4u);
        fast[q] = count_it && table_here && !alias;
        slow[q] = count_it && (all_slow || alias);
      }
      if (table_here) {
        // The table is cut into cells of four keys (one 16-byte LDS read); a key sits in its home cell or, when
        // that was full, in one of the next cur_probes - 1 cells.  The trip count is the same for every lane and
        // almost always one: no divergent probing loop (a per-lane while loop here cost more than the stream).
        const unsigned cshift = 34u - cur_info, cmask = (1u << (cur_info - 2u)) - 1u;   // cells = size / 4
        unsigned cell[SXMC_VEC];
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) cell[q] = cur_info > 2u ? (p2[q] * 0x9E3779B1u) >> cshift : 0u;
        for (unsigned pr = 0; pr < cur_probes; pr++) {
          vuint4g kk[SXMC_VEC];
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            kk[q] = *reinterpret_cast<const vuint4g*>(&wkeys[4u * ((cell[q] + pr) & cmask)]);
          }
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            // keys are distinct, so at most one of the four matches: its position by arithmetic, not by branches
            const unsigned key = fast[q] ? p2[q] : 0xFFFFFFFEu;   // (never a table key)
            const unsigned m = (kk[q][1] == key ? 1u : 0u) + (kk[q][2] == key ? 2u : 0u) + (kk[q][3] == key ? 3u : 0u);
            if ((kk[q][0] == key) | (m != 0u)) {
              __hip_atomic_fetch_add(&wcnt[4u * ((cell[q] + pr) & cmask) + m], 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        if (slow[q]) {
          const unsigned flat = pre + p2[q];
          if (flat < B) sparse_lookup(d, gcnt, flat);
        }
      }
    }
    flush();
    cur_info = SXMC_SPARSE_EMPTY;

#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, kWave);
    if (lane == 0 && cnt != 0u) {
      __hip_atomic_fetch_add(to_global(d.norm), cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// LOCKSTEP CHAINS.  Several MCMC chains over the SAME sample tables (the fake experiments in flight on one GPU:
// BASELINE config 4's per-GPU shape) evaluate different parameter vectors on identical samples.  Stepped one by
// one, each streams the tables again and the vector units idle under the stream (config 3: 175 us of stream
// against 122 us of arithmetic per evaluation).  Here ONE pass streams the tables and bins every sample under
// each chain's parameters into that chain's LDS histogram: the bytes per evaluation divide by the number of
// chains, the arithmetic does not change -- per chain exactly the operations of fill_body, so every chain's
// counts are those of its own evaluation, bit for bit.  Static programs, histograms in LDS (NCHAIN of them),
// PREW 0 (rows) or kPreGranule (bucketed).  descs.d[c]: the launch's members as chain c's evaluators describe
// them (same tables and geometry; their own parameters, histogram, normalisation slot).
struct SxChainDescs {
  const SxSignalDesc* d[4];
};

template <int NOBS, int NSLOT, typename PROG, int PREW, int NCHAIN>
__device__ __forceinline__ void fill_multi_body(SxChainDescs chains, const SxSegment* __restrict__ segs,
                                                const unsigned* __restrict__ blk_off, unsigned hist_words) {
  static_assert(!PROG::dynamic && (PREW == 0 || PREW == kPreGranule) && NCHAIN >= 2 && NCHAIN <= 4, "unsupported");
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned nthreads = blockDim.x;
  const unsigned lane = tid & (kWave - 1);
  // LDS: words 0..3 the chains' in-domain counters, then NCHAIN histograms of hist_words, then 64 trash words
  unsigned* s_norm = lds;
  unsigned* hist = lds + 4;
  const unsigned trash = NCHAIN * hist_words + lane;

  bool lds_clean = false;
  const unsigned seg_end = blk_off[blockIdx.x + 1];
  for (unsigned si = blk_off[blockIdx.x]; si < seg_end; ++si) {
    const SxSegment& sg = segs[si];
    const SxSignalDesc& d = chains.d[0][sg.sig];     // tables and geometry: the same for every chain
    const unsigned long long v0 = sg.v0;
    const unsigned long long v1 = sg.v1;
    const unsigned long long step = sg.step;
    const unsigned B = (unsigned)d.total_nbins;

    double craw[NCHAIN][PROG::ncoef > 0 ? PROG::ncoef : 1];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      const SxSignalDesc& dc = chains.d[c][sg.sig];
#pragma unroll
      for (int q = 0; q < PROG::ncoef; q++) craw[c][q] = to_global(dc.params)[(long)dc.coef_par[q] * dc.param_stride];
    }
    __builtin_amdgcn_sched_barrier(0);

    gptr<const vfloat4> col[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      col[k] = to_global(reinterpret_cast<const vfloat4*>(d.cols + (unsigned long long)d.slot_col[k] * d.col_pitch));
    }
    gptr<const typename PreVec<PREW>::type> precol =
        to_global(reinterpret_cast<const typename PreVec<PREW>::type*>(d.pre));
    const unsigned long long vlast = v1 - 1;
    const unsigned long long vfirst = v0 + tid;
    Columns<NSLOT, PREW> buf;
    load_columns<NOBS, NSLOT, PREW, PROG>(buf, col, precol, vfirst < v1 ? vfirst : vlast);

    if (!lds_clean) {
      for (unsigned b = tid; b < NCHAIN * hist_words; b += nthreads) hist[b] = 0u;
      if (tid < 4) s_norm[tid] = 0u;
      __syncthreads();
    }

    double lo[NOBS], hi[NOBS], sc[NOBS];
    int st[NOBS];
#pragma unroll
    for (int k = 0; k < NOBS; k++) {
      lo[k] = d.lower[k];
      hi[k] = d.upper[k];
      sc[k] = d.scale[k];
      st[k] = d.bin_stride[k];
    }
    unsigned cnt[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) cnt[c] = 0u;

    unsigned long long v = vfirst;
    const unsigned long long niter = (v1 - v0 + step - 1) / step;
    for (unsigned long long it = 0; it < niter; ++it, v += step) {
      double f0[NSLOT][SXMC_VEC];
#pragma unroll
      for (int k = 0; k < NSLOT; k++) {
        f0[k][0] = (double)buf.v[k].x;
        f0[k][1] = (double)buf.v[k].y;
        f0[k][2] = (double)buf.v[k].z;
        f0[k][3] = (double)buf.v[k].w;
      }
      typename PreVec<PREW>::type prebits = buf.pre;
#pragma unroll
      for (int k = 0; k < NSLOT; k++) {
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) asm volatile("" : "+v"(f0[k][q]));
      }
      if constexpr (PREW != 0) asm volatile("" : "+v"(prebits));
      const unsigned long long vl = v + step;
      load_columns<NOBS, NSLOT, PREW, PROG>(buf, col, precol, vl < v1 ? vl : vlast);

      const unsigned dead = (v < v1) ? 0u : 1u;
#pragma unroll
      for (int c = 0; c < NCHAIN; c++) {
        double f[NSLOT][SXMC_VEC];
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) f[k][q] = f0[k][q];
        }
        run_static<NSLOT>(f, craw[c], PROG{}, typename MakeISeq<PROG::n>::type{});
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          unsigned bad = dead;
          int bin = 0;
          if constexpr (PREW != 0) bin = (int)pre_value<PREW>(prebits, q);
#pragma unroll
          for (int k = 0; k < NOBS; k++) {
            const double x = f[k][q];
            bad += !(x >= lo[k]) ? 1u : 0u;
            bad += !(x < hi[k]) ? 1u : 0u;
            const int idx = (int)((x - lo[k]) * sc[k]);
            bin = (k == NOBS - 1 && PREW != kPreGranule) ? bin + idx : __mul24(idx, st[k]) + bin;
          }
          const unsigned in_domain = (bad == 0u) ? 1u : 0u;
          cnt[c] += in_domain;
          const bool store = (bad == 0u) && ((unsigned)bin < B);
          const unsigned slot = store ? (unsigned)c * hist_words + (unsigned)bin : trash;
          __hip_atomic_fetch_add(&hist[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }

#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      unsigned t = cnt[c];
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_down(t, off, kWave);
      if (lane == 0 && t != 0u) {
        __hip_atomic_fetch_add(&s_norm[c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      const SxSignalDesc& dc = chains.d[c][sg.sig];
      gptr<unsigned> gbins = to_global(dc.bins);
#pragma unroll 4
      for (unsigned b = tid; b < B; b += nthreads) {
        const unsigned n = hist[(unsigned)c * hist_words + b];
        if (n != 0u) __hip_atomic_fetch_add(&gbins[b], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (si + 1u < seg_end) {
        __syncthreads();
        for (unsigned b = tid; b < B; b += nthreads) hist[(unsigned)c * hist_words + b] = 0u;
      }
      if (tid == 0) {
        const unsigned n = s_norm[c];
        if (n != 0u) __hip_atomic_fetch_add(to_global(dc.norm), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_norm[c] = 0u;
      }
    }
    __syncthreads();
    lds_clean = true;
  }
}

template <int NOBS, int NSLOT, typename PROG, int NCHAIN, bool LDS_HIST = true>
__device__ __forceinline__ void fill_ordered_body(SxChainDescs chains, const SxSegment* __restrict__ segs,
                                                  const unsigned* __restrict__ blk_off, unsigned layout,
                                                  unsigned dbg_arg) {
  const unsigned dbg = sx_dbg(dbg_arg);   // (0 in the product build: see SXMC_MEASURE)
  static_assert(!PROG::dynamic && NSLOT >= 1 && NOBS >= 0 && NOBS < NSLOT && NCHAIN >= 1 && NCHAIN <= 4, "unsupported");
  // LDS_HIST false: a histogram beyond LDS capacity, one global atomic per counted sample (the dense evaluation of
  // such a table: CreateHistogram; evaluations for lookup run fill_sparse_body); the granule word is then the
  // bucket's whole 32-bit bin offset, so a granule's row count is not available: NOBS >= 1
  static_assert(LDS_HIST || (NCHAIN == 1 && NOBS >= 1), "unsupported");
  constexpr int ORD = NSLOT - 1;       // the ordered observable's column
  constexpr int NSTREAM = NSLOT - 1;   // columns every granule streams
  constexpr int NG = NOBS > 0 ? NOBS : 1, NS = NSTREAM > 0 ? NSTREAM : 1, NC = PROG::ncoef > 0 ? PROG::ncoef : 1;
  // CODES (see AffineForm above): the streamed slots as 16-bit codes, two to a word
  constexpr bool kCodes = LDS_HIST && NOBS >= 1 && NSTREAM >= 2 && NSTREAM <= SXMC_MAX_QSLOTS && prog_is_affine<ORD>(PROG{});
  constexpr int NQ = kCodes ? NSTREAM : 1, QW = (NQ + 1) / 2;
#ifndef SXMC_ORD_RING
#define SXMC_ORD_RING 4   // (8 with a launch bound of 768 lanes, 145 VGPRs, now that the waits are the counted ones: 82.3-83.3 us
                          //  against 80.1-80.4 at the same shape -- profiles/r05_ring_depth_ab.log: not what is in flight)
#endif
  constexpr int kRing = SXMC_ORD_RING;   // units of codes a lane holds: one being worked on, the others in flight
#ifndef SXMC_DRAIN_BATCH
#define SXMC_DRAIN_BATCH 1
#endif
  // Granules whose float columns the drain loads together.  ONE: measured on one box, alternating
  // (profiles/r04b_drain_batch_ab.log), two cost 8 us and three 12 us of an 82 us launch instead of saving round
  // trips -- the kernel sits at 123 of its 128 registers, a second granule's 13 spill (76 bytes of scratch, none of it
  // inside the stream loop, and still).
  constexpr int kDrainStream = 1;                                  // ... by a drain in the middle of the stream
#ifndef SXMC_DRAIN_FINAL
#define SXMC_DRAIN_FINAL SXMC_DRAIN_BATCH
#endif
  constexpr int kDrainFinal = NCHAIN > 1 ? 1 : SXMC_DRAIN_FINAL;   // ... by the one after it (the ring's registers are free)
#ifndef SXMC_DRAIN_EVERY
#define SXMC_DRAIN_EVERY 0
#endif
  // units between early drains of the queues (a power of two, a multiple of the ring; 0: only when a queue is full and
  // at the end of the stream).  MEASURED, one box, alternating, three rounds (profiles/r05_early_drain_ab.log): every 64,
  // 32 or 16 units, with one or several granules loaded together by the last drain -- 81.0-83.2 us whatever the form,
  // the spread of the product build's own three runs (81.0-82.6).  The queues' round trips are not what is left of the
  // launch; the product build drains only when it must.
  constexpr int kDrainEvery = NCHAIN > 1 ? 0 : SXMC_DRAIN_EVERY;
  static_assert(kDrainEvery == 0 || (kDrainEvery % kRing == 0 && (kDrainEvery & (kDrainEvery - 1)) == 0), "bad SXMC_DRAIN_EVERY");
  static_assert(64 % kRing == 0, "a block of 64 units is a whole number of rounds of the ring");
  typedef typename MakeISeq<PROG::n>::type Seq;
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned nthreads = blockDim.x;
  const unsigned lane = tid & (kWave - 1);
  // LDS: words 0..3 the chains' in-domain counters, then per chain R replicas of the histogram, rstride words
  // apart, then 64 spare words, then the queue of ambiguous rows (codes): 4 header words + 2 words per entry.
  // `layout` = rstride | log2(R) << 24 | log2(queue entries) << 28 (0: no queue).  Inside a granule only the
  // observables binned per sample vary, so a wave's 64 updates go to a handful of bins, strided by the other
  // observables' strides: few banks, many lanes per word.  Two remedies: the word of bin b is b with its low six
  // bits XORed by the next six (bins a multiple of 64 apart land in different banks), and lane l updates replica
  // l mod R (rstride = 16 mod 64: the replicas of one bin sit in different banks too).  The flush adds the replicas up.
  unsigned* s_norm = lds;
  unsigned* hist = lds + 4;
  // (bit 27: the host laid the LDS out for the codes' padded form of the histogram -- see `outer` below)
  const unsigned rstride = layout & 0xFFFFFFu, rlog = (layout >> 24) & 7u, R = 1u << rlog;
  const bool outer_layout = ((layout >> 27) & 1u) != 0u;
  const unsigned cstride = rstride << rlog;            // one chain's replicas
  const unsigned myrep = (lane & (R - 1u)) * rstride;
  // the queues of the codes path: every WAVE owns a slice of (1 << qlog) / waves entries of two words behind the spare
  // words -- first the ambiguous rows {row, granule's bin offset | chain << 28}, then, in the last eighth, whole
  // granules left to the float columns {wave's first unit | chain << 28, the ordered observable's code}.  Filled and
  // emptied by its wave alone: no atomics, no barriers.
  const unsigned qlog = layout >> 28;
  const unsigned qwave = qlog ? (1u << qlog) / (nthreads / kWave) : 0u;      // entries per wave
  const unsigned gq_min = (unsigned)(kRing * NCHAIN) + 1u;                    // (a round of the ring must fit)
  const unsigned gq_cap = qwave / 8u > gq_min ? qwave / 8u : gq_min, rq_cap = qwave > gq_cap ? qwave - gq_cap : 0u;
  unsigned* qrows = lds + 4 + (LDS_HIST ? NCHAIN * cstride : 0u) + 64 + 4 + (tid / kWave) * (2u * qwave);
  unsigned* qgran = qrows + 2u * rq_cap;

  bool lds_clean = false;
  SX_WG_STAMP(0);
  const unsigned seg_end = blk_off[blockIdx.x + 1];
  for (unsigned si = blk_off[blockIdx.x]; si < seg_end; ++si) {
    const SxSegment& sg = segs[si];
    const SxSignalDesc& d = chains.d[0][sg.sig];     // tables and geometry: the same for every chain
    const unsigned long long v0 = sg.v0;
    const unsigned long long v1 = sg.v1;
    const unsigned long long step = sg.step;
    const unsigned B = (unsigned)d.total_nbins;
    gptr<unsigned> gbins0 = to_global(d.bins);

    double craw[NCHAIN][NC];
#if SXMC_MEASURE
    // (measurement build, dbg bit 6: THE GATED STEP -- this fill was launched beside the previous step's step end and
    // may do everything that does not depend on the proposal before that is written: its parameters are read further
    // down, after the wait for the gate.  Node index of the step inside its recorded graph: dbg bits 24-27.)
    const bool gated = (dbg & 64u) != 0u;
    auto load_parameters = [&]() {
#pragma unroll
      for (int c = 0; c < NCHAIN; c++) {
        const SxSignalDesc& dc = chains.d[c][sg.sig];
#pragma unroll
        for (int q = 0; q < PROG::ncoef; q++) craw[c][q] = to_global(dc.params)[(long)dc.coef_par[q] * dc.param_stride];
      }
    };
    if (!gated) load_parameters();
#else
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      const SxSignalDesc& dc = chains.d[c][sg.sig];
#pragma unroll
      for (int q = 0; q < PROG::ncoef; q++) craw[c][q] = to_global(dc.params)[(long)dc.coef_par[q] * dc.param_stride];
    }
#endif
    __builtin_amdgcn_sched_barrier(0);

    gptr<const vfloat4> col[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      col[k] = to_global(reinterpret_cast<const vfloat4*>(d.cols + (unsigned long long)d.slot_col[k] * d.col_pitch));
    }
    gptr<const unsigned> precol = to_global(reinterpret_cast<const unsigned*>(d.pre));
    gptr<const vfloat2> edges = to_global(reinterpret_cast<const vfloat2*>(d.edges));
    const unsigned long long vlast = v1 - 1;
    const unsigned long long vfirst = v0 + tid;
    const unsigned long long vwave = v0 + (tid - lane);   // the wave's first unit: a granule boundary

    // the table of codes, when the launch has one (dbg & 8: measurement / test hook, stream the floats regardless);
    // every wave needs room for at least 16 queue entries
    bool want_q = false;
    gptr<const vuint4g> qcol[QW];
    if constexpr (kCodes) {
      want_q = d.qcol != nullptr && !(dbg & 8u) && rq_cap >= 8u;
#pragma unroll
      for (int w = 0; w < QW; w++) {
        qcol[w] = to_global(reinterpret_cast<const vuint4g*>(d.qcol + (unsigned long long)w * d.col_pitch));
      }
    }

    vfloat4 raw[NS];
    unsigned rawpre;
    auto load = [&](unsigned long long v) {
#pragma unroll
      for (int k = 0; k < NSTREAM; k++) {
        raw[k] = __builtin_nontemporal_load(&col[k][v]);
        __builtin_amdgcn_sched_barrier(0);
      }
      rawpre = precol[v >> 6];
      __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned long long niter = (v1 - v0 + step - 1) / step;
    // a ring of kRing units of codes: one is worked on while the loads of the others are in flight.  (32-bit unit
    // numbers: the host offers codes only for tables below 2^22 granules, so a unit's byte offset fits 32 bits and
    // the loads take the form base in scalar registers + 32-bit offset.)
    vuint4g rq[kRing][QW];
    const unsigned vfirst32 = (unsigned)vfirst, step32 = (unsigned)step, vlast32 = (unsigned)vlast;
    if (!want_q) load(vfirst < v1 ? vfirst : vlast);

    if (!lds_clean) {
      if (LDS_HIST) {
        for (unsigned b = tid; b < NCHAIN * cstride; b += nthreads) hist[b] = 0u;
      }
      if (tid < 4) s_norm[tid] = 0u;
      __syncthreads();
    }

    double lo[NG], hi[NG], sc[NG];
    int st[NG];
#pragma unroll
    for (int k = 0; k < NOBS; k++) {
      lo[k] = d.lower[k];
      hi[k] = d.upper[k];
      sc[k] = d.scale[k];
      st[k] = d.bin_stride[k];
    }
    const double olo = d.lower[NOBS], ohi = d.upper[NOBS], osc = d.scale[NOBS];
    const int ost = d.bin_stride[NOBS];
#if SXMC_MEASURE
    if (gated) {
      // the gate: node 0 of a recorded graph follows the previous replay in stream order (it re-arms the gate), node s
      // waits until the step end of node s - 1 has written the proposal.  Bounded: a wait that gives up goes on.
      const unsigned node = (dbg >> 24) & 15u;
      gptr<unsigned> gate = to_global(d.step_gate);
      if (node == 0u) {
        if (blockIdx.x == 0u && tid == 0u && si == blk_off[0]) {
          __hip_atomic_store(gate, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        // (ONE lane of the workgroup polls, the others wait at the barrier: 3 072 waves polling one word at memory
        // scope beside the step end slowed that kernel down -- 109.7 us per step against 101.3 ungated)
        if (tid == 0u) {
          for (unsigned spin = 0; spin < 100000u; spin++) {
            if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= node) break;
            __builtin_amdgcn_s_sleep(4);
          }
        }
        __syncthreads();
      }
      load_parameters();
    }
#endif
    // a coefficient that is not finite: no claim about monotone maps, every granule takes the per-sample path
    bool wild[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      wild[c] = false;
#pragma unroll
      for (int q = 0; q < PROG::ncoef; q++) wild[c] = wild[c] || !(__builtin_fabs(craw[c][q]) < __builtin_inf());
    }

    unsigned ncnt[NCHAIN], vcnt[NCHAIN], codes[NCHAIN];   // (ncnt: wave-uniform part of the in-domain count)
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) ncnt[c] = vcnt[c] = codes[c] = 0u;
    unsigned sink = 0u;                     // keeps the loads alive in the stream-only debug mode
    // the strides in vector registers (an integer multiply-add takes one scalar operand, and the base is one)
    int stv[NG], ostv = ost;
#pragma unroll
    for (int k = 0; k < NOBS; k++) {
      stv[k] = st[k];
      asm volatile("" : "+v"(stv[k]));
    }
    asm volatile("" : "+v"(ostv));

    // ---- codes of the wave's next 64 granules in the ordered observable, one per lane (every 64 steps)
    auto granule_codes = [&](unsigned long long it) {
      const unsigned long long vg = vwave + (it + lane) * step;
      const bool live = vg < v1;
      const vfloat2 e = edges[(live ? vg : vlast) >> 6];
#pragma unroll
      for (int c = 0; c < NCHAIN; c++) {
        double x0 = (double)e.x, x1 = (double)e.y;
        run_ordered_scalar<ORD>(x0, craw[c], PROG{}, Seq{});
        run_ordered_scalar<ORD>(x1, craw[c], PROG{}, Seq{});
        const bool nan = wild[c] || !(x0 == x0) || !(x1 == x1);
        const int i0 = (int)((x0 - olo) * osc), i1 = (int)((x1 - olo) * osc);
        const int e0 = !(x0 >= olo) ? -1 : (!(x0 < ohi) ? 0x7FFFFFFF : i0);
        const int e1 = !(x1 >= olo) ? -1 : (!(x1 < ohi) ? 0x7FFFFFFF : i1);
        unsigned code = (nan || e0 != e1) ? kOrdMixed
                        : ((e0 < 0 || e0 == 0x7FFFFFFF) ? kOrdSkip
                                                       : (unsigned)(LDS_HIST ? __mul24(e0, ost) : e0 * ost));
        codes[c] = live ? code : kOrdSkip;
      }
    };

    // ---- one granule of one chain with the reference's arithmetic on the float columns.  f0: the streamed slots
    // widened; rawo: the ordered observable's unit (read only when `mixed`).  pdfz.cpp:388-398.
    // CODES, PADDED FORM of the LDS histogram (`outer`): one observable binned per sample and that one the histogram's
    // outermost dimension (stride S, S * nbins = B).  Bin b = idx * S + r then lives in word (idx + 1) * S' + r of a
    // replica, S' = S or S + 1, whichever is odd: lanes that differ in idx hit different banks without a swizzle, and
    // the rows idx = -1 and idx = nbins are guard rows -- a sample that is outside the domain or ambiguous is sent
    // there by clamping its index instead of being predicated away.  Per sample the address is then ONE multiply-add.
    bool outer = false;
    unsigned oS = 1u, oSp = 1u;
    float oinvS = 1.0f;
    auto codes_word = [&](unsigned bin) -> unsigned {     // (off the hot path: the exact arithmetic's bins)
      if (!outer) return lds_slot(bin);
      const unsigned q = (unsigned)(((float)bin + 0.5f) * oinvS);   // bin / S: exact while S * (nbins + 1) < 2^22
      return bin + oSp + (oSp - oS) * q;
    };
    // ALL: every in-domain sample counts in vcnt (the float stream); otherwise only those whose index is out of range
    // -- over codes the norm is the sum of the LDS histogram plus exactly those.
    auto exact_granule = [&](auto ALL, int c, bool mixed, const double (&f0)[NS][SXMC_VEC], const vfloat4& rawo,
                             unsigned off, unsigned codec) {
      constexpr bool kCountAll = decltype(ALL)::value != 0;
      double f[NSLOT][SXMC_VEC];
#pragma unroll
      for (int k = 0; k < NSTREAM; k++) {
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) f[k][q] = f0[k][q];
      }
      f[ORD][0] = f[ORD][1] = f[ORD][2] = f[ORD][3] = 0.0;
      run_static_part<NSLOT, ORD, false>(f, craw[c], PROG{}, Seq{});
      // The per-sample part, written for the vector unit's instruction count (what bounds several chains in
      // one pass): the domain tests are compares into scalar mask registers, combined and counted (popcount)
      // on the scalar unit; the histogram update runs under that mask.
      auto samples = [&](auto MIXED) {
        constexpr bool kMixed = decltype(MIXED)::value != 0;
        int base = (int)(off + (kMixed ? 0u : codec));
        asm volatile("" : "+v"(base));   // (in a vector register: idx * stride + base is then one multiply-add)
        const unsigned cbase = (unsigned)c * cstride + myrep;
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          bool ind = true;
          int bin = base;
#pragma unroll
          for (int k = 0; k < NOBS; k++) {
            const double x = f[k][q];
            ind = ind & (x >= lo[k]) & (x < hi[k]);      // (NaN fails)
            const int idx = (int)((x - lo[k]) * sc[k]);
            bin = LDS_HIST ? mad24(idx, stv[k], bin) : idx * stv[k] + bin;
          }
          if constexpr (kMixed) {
            const double x = f[ORD][q];
            ind = ind & (x >= olo) & (x < ohi);
            const int idx = (int)((x - olo) * osc);
            bin = LDS_HIST ? mad24(idx, ostv, bin) : idx * ostv + bin;
          }
          if constexpr (kCountAll) {
            vcnt[c] += ind ? 1u : 0u;                    // (an add-with-carry straight from the compare mask;
                                                         //  a ballot + s_bcnt1 costs a v_cndmask and a v_cmp instead)
          } else {
            vcnt[c] += (ind && !((unsigned)bin < B)) ? 1u : 0u;
          }
          // in domain but index out of range (the reference's one-past-the-end case) still counts in the norm
          if (ind && ((unsigned)bin < B) && !(dbg & 4u)) {
            if constexpr (LDS_HIST) {
              const unsigned wd = kCountAll ? lds_slot((unsigned)bin) : codes_word((unsigned)bin);
              __hip_atomic_fetch_add(&hist[cbase + wd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              __hip_atomic_fetch_add(&gbins0[(unsigned)bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      };
      if (mixed) {   // (the wait for the extra column sits in here, off the common path)
        f[ORD][0] = (double)rawo.x;
        f[ORD][1] = (double)rawo.y;
        f[ORD][2] = (double)rawo.z;
        f[ORD][3] = (double)rawo.w;
        run_static_part<NSLOT, ORD, true>(f, craw[c], PROG{}, Seq{});
        samples(IntC<1>{});
      } else {
        samples(IntC<0>{});
      }
    };

    // ---- codes: the program composed, per chain, into single-precision coefficients over the codes (AffineForm);
    // a chain whose parameters rule that out sends the whole evaluation back to the float columns
    float af[NCHAIN][NG][NQ], gf[NCHAIN][NG], thr[NCHAIN][NG];
    unsigned nbk[NG];
    bool use_q = want_q;
    if constexpr (kCodes) {
      if (want_q) {
#pragma unroll
        for (int k = 0; k < NOBS; k++) nbk[k] = (unsigned)d.nbins[k];
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) {
          AffineForm<NQ> form[NQ];
#pragma unroll
          for (int m = 0; m < NQ; m++) {
#pragma unroll
            for (int n = 0; n < NQ; n++) form[m].a[n] = (m == n) ? 1.0 : 0.0;
            form[m].c = 0.0;
            const double wlo = d.qbase[m], whi = d.qbase[m] + 65534.0 * d.qstep[m];
            form[m].mag = __builtin_fmax(__builtin_fabs(wlo), __builtin_fabs(whi));
          }
          run_affine<NQ, ORD>(form, craw[c], PROG{}, Seq{});
          bool ok = !wild[c];
#pragma unroll
          for (int k = 0; k < NOBS; k++) {
            double sum_abs = 0.0, g = form[k].c - lo[k];
#pragma unroll
            for (int m = 0; m < NQ; m++) {
              const double alpha = form[k].a[m] * d.qstep[m] * sc[k];
              sum_abs = sum_abs + __builtin_fabs(alpha);
              g = g + form[k].a[m] * (d.qbase[m] + 0.5 * d.qstep[m]);
              af[c][k][m] = uniform_f((float)alpha);
            }
            g = g * sc[k];
            // THE BOUND, term by term.  Claim: |u_codes - u_ref| <= eps, where u_ref is the product (x - lo) * scale as
            // the reference's double arithmetic forms it for this sample (pdfz.cpp:306-331, 388-398) and u_codes =
            // g + sum_m alpha_m * code_m as the lanes evaluate it in single precision (before e is added).
            //  Q  the codes: the table keeps code_m only for rows with |x_m - centre_m| <= qstep_m / 2 * (1 + 2^-20), checked
            //     per row in double when the table is built (column_codes_kernel; other rows carry SXMC_QCODE_EXACT), so
            //     in real arithmetic |sum_m A_km (x_m - centre_m)| * scale <= sum_abs / 2 * (1 + 2^-20).  Not a margin: a
            //     sample sits anywhere in its cell, the worst case is reached.  The 2^-19 here leaves another 2^-20 for
            //     the roundings of that check itself (centre_m: two, <= 2^-52 |centre_m|), where |centre_m| / qstep_m <=
            //     2^31; a window further from zero than that borrows from D below (8 of its 512 units).
            //  S  single precision, in units of 2^-24 mu (mu bounds every partial sum: |g| + e + sum_abs * 65534, with
            //     nbins + 0.25 to spare):  1  alpha_m rounded to 24 bits, times codes below 2^16, summed over m
            //                              1  the constant term g + e rounded to 24 bits
            //                              NQ one rounding per fused multiply-add (codes convert exactly)
            //     <= NQ + 2 <= 6 units; budget 8 units = mu * 2^-21.
            //  D  double precision, in units of 2^-53 (mag + |lo|) scale (`mag` bounds the slot's value at every point of
            //     the program for field values inside the windows; an operation's rounding is <= 2^-53 of its result,
            //     and what follows amplifies it by no more than it grows mag -- induction over apply_affine's cases):
            //                              32 the reference's program: <= 8 systematics, <= 4 roundings each (ctscale:
            //                                 x - 1, 1 + p, the product, + 1)
            //                              2  the reference's x - lo and * scale
            //                              32 this composition of A and C (the same operations on the coefficients)
            //                              11 g (2 NQ + 3 operations) and alpha (2 each, relative: below 2^-52 mu)
            //                              8  borrowed by Q, see there
            //     <= 85 units; budget 512 units = 2^-44 (mag + |lo|) scale.
            // eps = Q + S + D.  Measured on the GPU (tests/test_gpu_codes_margin.py): with S + D and
            // the slack below scaled by s, samples built to sit on the threshold are binned as the CPU restatement bins them down to
            // s = s_min < 1/2; Q scaled by anything below 1 misplaces samples, as it must.
            const double mu = sum_abs * 65536.0 + __builtin_fabs(g) + (double)nbk[k] + 0.25;
            const double epsq = 0.5 * sum_abs * (1.0 + 0x1p-19);
            const double eps = epsq + mu * 0x1p-21 + (form[k].mag + __builtin_fabs(lo[k])) * sc[k] * 0x1p-44;
            ok = ok && (eps < 0.125);                      // (NaN fails)
            // The kernel evaluates u' = u + e (e = 1.01 eps, added to the constant term: no instruction) and asks ONE
            // question, fract(u') >= 2e: u_ref lies in (u' - 2e, u'), so if u' is at least 2e above floor(u') both have
            // the same floor -- and a u_ref within eps of 0 or nbins, where the reference's domain test (on x, before the
            // subtraction) could disagree with the index, is never trusted.  The slack: 1.01 for the conversion of 2e to
            // single precision (it may round down: 2^-24 relative; the e inside the sums is mu's + 0.25); 2^-23 because
            // v_fract_f32 of a negative u' rounds (1 - 2^-30 becomes the largest value below 1).
            double e = eps * 1.01, slack = 0x1p-23;
#if SXMC_MEASURE
            // (measurement build: dbg bits 8-15 = 1 + 64 * the scale on everything but Q, bits 16-23 = 1 + 64 * a scale on
            // the whole threshold; 0 = unscaled.  RESULTS ARE WRONG below the bound: that is what is being measured.)
            {
              const unsigned fr = (dbg >> 8) & 0xFFu, ft = (dbg >> 16) & 0xFFu;
              const double rs = fr ? (double)(fr - 1u) / 64.0 : 1.0, ts = ft ? (double)(ft - 1u) / 64.0 : 1.0;
              e = (epsq + rs * (e - epsq)) * ts;
              slack = slack * rs * ts;
            }
#endif
            gf[c][k] = uniform_f((float)(g + e));
            thr[c][k] = uniform_f((float)(2.0 * e + slack));
          }
          use_q = use_q && (uniform_i(ok ? 1 : 0) != 0);
        }
        if (!use_q) load(vfirst < v1 ? vfirst : vlast);    // (the float stream's first unit, late this once)
        if (use_q && outer_layout && NOBS == 1 && (unsigned)st[0] * nbk[0] == B) {
          outer = true;
          oS = (unsigned)st[0];
          oSp = oS | 1u;
          oinvS = 1.0f / (float)oS;
        }
      }
    }

    if (!use_q) {
      // ================================================================ the float columns, one unit in flight
      unsigned long long v = vfirst;
      for (unsigned long long it = 0; it < niter; ++it, v += step) {
        const int j = (int)(it & 63ull);
        if (j == 0) granule_codes(it);
        unsigned code[NCHAIN];
        bool anymixed = false;
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) {
          code[c] = (unsigned)__builtin_amdgcn_readlane((int)codes[c], j);
          anymixed = anymixed || code[c] == kOrdMixed;
        }

        double f0[NS][SXMC_VEC];
#pragma unroll
        for (int k = 0; k < NSTREAM; k++) {
          f0[k][0] = (double)raw[k].x;
          f0[k][1] = (double)raw[k].y;
          f0[k][2] = (double)raw[k].z;
          f0[k][3] = (double)raw[k].w;
        }
        const unsigned prebits = (unsigned)uniform_i((int)rawpre);
#pragma unroll
        for (int k = 0; k < NSTREAM; k++) {
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) asm volatile("" : "+v"(f0[k][q]));
        }
        // a mixed granule: this once the ordered observable's column is needed too
        vfloat4 rawo = {0.0f, 0.0f, 0.0f, 0.0f};
        if (anymixed) {
          rawo = __builtin_nontemporal_load(&col[ORD][v < v1 ? v : vlast]);
          __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long vl = v + step;
        load((vl < v1 && !(dbg & 2u)) ? vl : vlast);
        if (dbg & 1u) {
#pragma unroll
          for (int k = 0; k < NSTREAM; k++) {
#pragma unroll
            for (int q = 0; q < SXMC_VEC; q++) sink += (f0[k][q] == 12345.678) ? 1u : 0u;
          }
          sink += (prebits == 12345u) ? 1u : 0u;
          continue;
        }
        const unsigned off = LDS_HIST ? prebits & 0xFFFFFFu : prebits, nvalid = (prebits >> 24) + 1u;

#pragma unroll
        for (int c = 0; c < NCHAIN; c++) {
          if (code[c] == kOrdSkip) continue;          // (wave-uniform) outside the ordered observable's domain
          const bool mixed = code[c] == kOrdMixed;
          if constexpr (NOBS == 0) {
            if (!mixed) {
              // nothing varies inside the granule: all its rows go to one bin
              const unsigned bin = off + code[c];
              ncnt[c] += nvalid;
              if (lane == 0) {
                if (bin < B && !(dbg & 4u)) {
                  __hip_atomic_fetch_add(&hist[(unsigned)c * cstride + lds_slot(bin)], nvalid, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
              continue;
            }
          }
          exact_granule(IntC<1>{}, c, mixed, f0, rawo, off, code[c]);
        }
      }
    } else if constexpr (kCodes) {
      // ================================================================ the codes, kRing - 1 units in flight
      unsigned nrow = 0u, ngran = 0u;                      // entries in the wave's queues (wave-uniform)

      // ---- a granule left to the float columns: one entry in the wave's queue (there is room: the loop below
      // empties the queues before a round of the ring could overfill them)
      const unsigned vwave32 = vfirst32 - lane;
      auto push_granule = [&](int c, unsigned it, unsigned code) {
        if (lane == 0) {
          qgran[2u * ngran] = (vwave32 + it * step32) | ((unsigned)c << 28);
          qgran[2u * ngran + 1u] = code;
        }
        ngran += 1u;
      };
      // ---- the ambiguous rows of one granule of one chain (bit q of `rare`: the lane's q-th sample) into the
      // wave's queue; false: they do not fit
      auto push_rows = [&](int c, unsigned rare, unsigned v, unsigned offcode) -> bool {
        unsigned long long mask[SXMC_VEC];
        unsigned total = 0u;
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          mask[q] = __builtin_amdgcn_ballot_w64(((rare >> q) & 1u) != 0u);
          total += (unsigned)__builtin_popcountll(mask[q]);
        }
        if (nrow + total > rq_cap) return false;
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          if (mask[q] == 0ull) continue;
          const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[q] >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask[q], 0u));
          if ((rare >> q) & 1u) {
            const unsigned pos = nrow + below;
            qrows[2u * pos] = v * SXMC_VEC + (unsigned)q;
            qrows[2u * pos + 1u] = NCHAIN > 1 ? (offcode | ((unsigned)c << 28)) : offcode;
          }
          nrow += (unsigned)__builtin_popcountll(mask[q]);
        }
        return true;
      };

      // ---- one granule of one chain from its codes, general form.  No predication (a sample that is not counted
      // adds to the lane's spare word, so the lane's four samples are four independent instruction chains), no
      // in-domain counter (the norm is the histogram's sum), no range check of the flat index (the caller sends
      // granules whose offset could push it past the end to the float columns), no memory access but the four LDS
      // additions: ambiguous rows go into the wave's queue.
      const unsigned spare = (unsigned)NCHAIN * cstride + lane;   // (one of the 64 spare words behind the histograms)
      auto coarse_granule = [&](int c, const vuint4g (&w)[QW], unsigned it, unsigned v, unsigned offcode) {
        int base = (int)offcode;
        asm volatile("" : "+v"(base));
        const unsigned cbase = (unsigned)c * cstride + myrep;
        unsigned word[SXMC_VEC];
        unsigned rare = 0u;                              // bit q: the lane's q-th sample is ambiguous
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          float cv[NQ];
#pragma unroll
          for (int m = 0; m < NQ; m++) {
            const unsigned cw = w[m >> 1][q];
            cv[m] = (float)((m & 1) ? (cw & 0xFFFFu) : (cw >> 16));
          }
          bool amb = false, ind = true;
          int bin = base;
#pragma unroll
          for (int k = 0; k < NOBS; k++) {
            float u = gf[c][k];
#pragma unroll
            for (int m = 0; m < NQ; m++) u = __builtin_fmaf(af[c][k][m], cv[m], u);
            const float fr = __builtin_amdgcn_fractf(u);            // u - floor(u), in [0, 1)
            amb = amb | !(fr >= thr[c][k]);                          // (NaN: ambiguous)
            const int idx = (int)__builtin_floorf(u);
            ind = ind & ((unsigned)idx < nbk[k]);
            bin = mad24(idx, stv[k], bin);
          }
          word[q] = (ind && !amb && !(dbg & 4u)) ? cbase + lds_slot((unsigned)bin) : spare;
          rare |= amb ? (1u << q) : 0u;
        }
        // rows with a special code in the high half of word 0: one test per lane
        const unsigned wmax = max(max(w[0][0], w[0][1]), max(w[0][2], w[0][3]));
        const bool special = wmax >= ((unsigned)SXMC_QCODE_EXACT << 16);
        if (__builtin_amdgcn_ballot_w64(rare != 0u || special) != 0ull) {   // (one step in seven at config 3)
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            const unsigned hi16 = w[0][q] >> 16;
            if (hi16 >= SXMC_QCODE_EXACT) {              // the arithmetic above meant nothing for this row
              word[q] = spare;
              rare = hi16 >= SXMC_QCODE_NEVER ? (rare & ~(1u << q)) : (rare | (1u << q));
            }
          }
          if (!push_rows(c, rare, v, offcode)) {         // more than the queue holds: the whole granule, later
            push_granule(c, it, (unsigned)__builtin_amdgcn_readlane((int)codes[c], (int)(it & 63u)));
            return;
          }
        }
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          __hip_atomic_fetch_add(&hist[word[q]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };

      // ---- the same in the padded form of the histogram (`outer`): per sample two conversions, the multiply-adds
      // (two samples per packed instruction), the fraction, ONE compare (fract(u') against 2e, see above) whose mask
      // stays in scalar registers, the floor as an integer, a select and a clamp that send ambiguous and
      // out-of-domain samples to the guard rows, one multiply-add for the LDS address.
      const int oS4 = (int)(4u * oSp);
      const int oclamp = (dbg & 4u) ? -1 : (int)nbk[0];   // (measurement hook: everything to the guard row)
      auto coarse_outer = [&](int c, const vuint4g (&w)[QW], unsigned it, unsigned v, unsigned offcode) {
        // byte address of (idx = -1, r): the header words, the chain's replica, the granule's offset
        int base4 = (int)(4u * (4u + (unsigned)c * cstride + myrep + offcode) + 4u * oSp);
        asm volatile("" : "+v"(base4));
        int addr[SXMC_VEC];
        bool amb[SXMC_VEC];
        // (the multiply-adds of two samples in one packed instruction: v_pk_fma_f32, IEEE per component)
        float us[SXMC_VEC];
#pragma unroll
        for (int p = 0; p < SXMC_VEC; p += 2) {
          vfloat2 u2 = {gf[c][0], gf[c][0]};
#pragma unroll
          for (int m = 0; m < NQ; m++) {
            const unsigned cw0 = w[m >> 1][p], cw1 = w[m >> 1][p + 1];
            const vfloat2 cv = {(float)((m & 1) ? (cw0 & 0xFFFFu) : (cw0 >> 16)),
                                (float)((m & 1) ? (cw1 & 0xFFFFu) : (cw1 >> 16))};
            const vfloat2 a2 = {af[c][0][m], af[c][0][m]};
            u2 = __builtin_elementwise_fma(a2, cv, u2);
          }
          us[p] = u2.x;
          us[p + 1] = u2.y;
        }
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          const float u = us[q];
          amb[q] = !(__builtin_amdgcn_fractf(u) >= thr[c][0]);      // closer to a bin edge than the bound (or NaN)
          int idx;
          asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(idx) : "v"(u));
          int ie;
          asm("v_med3_i32 %0, %1, -1, %2" : "=v"(ie) : "v"(amb[q] ? -1 : idx), "s"(oclamp));
          addr[q] = mad24(ie, oS4, base4);
        }
        const unsigned wmax = max(max(w[0][0], w[0][1]), max(w[0][2], w[0][3]));
        const bool special = wmax >= ((unsigned)SXMC_QCODE_EXACT << 16);
        const bool anyamb = (amb[0] | amb[1]) | (amb[2] | amb[3]);
        if (__builtin_amdgcn_ballot_w64(anyamb || special) != 0ull) {   // (one step in seven at config 3)
          unsigned rare = 0u;
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            rare |= amb[q] ? (1u << q) : 0u;
            const unsigned hi16 = w[0][q] >> 16;
            if (hi16 >= SXMC_QCODE_EXACT) {              // the arithmetic above meant nothing for this row
              addr[q] = base4 - oS4;                     // (the guard row)
              rare = hi16 >= SXMC_QCODE_NEVER ? (rare & ~(1u << q)) : (rare | (1u << q));
            }
          }
          if (!push_rows(c, rare, v, offcode)) {
            push_granule(c, it, (unsigned)__builtin_amdgcn_readlane((int)codes[c], (int)(it & 63u)));
            return;
          }
        }
        if (dbg & 32u) {                                   // (measurement hook: no LDS additions at all)
          sink += (unsigned)(addr[0] ^ addr[1] ^ addr[2] ^ addr[3]) == 12345u ? 1u : 0u;
          return;
        }
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          unsigned* wp = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + addr[q]);
          __hip_atomic_fetch_add(wp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };

      // largest value the per-sample part can add to a granule's offset: a granule whose offset + span stays below
      // the histogram's size needs no range check per sample (padded form: the offset must also lie inside a row)
      unsigned span = 0u;
#pragma unroll
      for (int k = 0; k < NOBS; k++) span += (nbk[k] - 1u) * (unsigned)st[k];

      // ---- per block of 64 units: the granules' codes in the ordered observable (granule_codes) and their words
      // (lane l: bin offset + ordered code of the block's l-th granule, per chain -- or the code itself when that says
      // "mixed" / "skip", which no offset reaches: one v_readlane and one compare per unit and chain decide the path)
      unsigned oc64[NCHAIN];
      auto granule_words = [&](unsigned it) {
        unsigned vg = vwave32 + (it + lane) * step32;
        vg = vg < vlast32 ? vg : vlast32;
        unsigned pre64 = precol[vg >> 6];
        asm volatile("" : "+v"(pre64));   // (the wait goes here, once per block, not into the inner loop)
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) oc64[c] = codes[c] < kOrdMixed ? (pre64 & 0xFFFFFFu) + codes[c] : codes[c];
      };

      // (OUTER: compile-time copy of `outer`, so that the stream loop exists once per form, without the other form's
      // code between its branches; `limit`: a granule whose offset reaches it cannot be binned from codes)
      const unsigned climit = outer ? oS : (B > span ? B - span : 0u);
      auto process_unit = [&](auto OUTER, unsigned it, const vuint4g (&w)[QW]) {
        constexpr bool kOuter = decltype(OUTER)::value != 0;
        const unsigned v = vfirst32 + it * step32;
        const int j = (int)(it & 63u);
        unsigned offcode[NCHAIN];
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) offcode[c] = (unsigned)__builtin_amdgcn_readlane((int)oc64[c], j);
        if (dbg & 1u) {
#pragma unroll
          for (int q = 0; q < QW; q++) sink += (w[q][0] == 12345u) ? 1u : 0u;
          sink += (offcode[0] == 12345u) ? 1u : 0u;
          return;
        }
#pragma unroll
        for (int c = 0; c < NCHAIN; c++) {
          if (offcode[c] < climit) {
            if constexpr (kOuter) coarse_outer(c, w, it, v, offcode[c]);
            else coarse_granule(c, w, it, v, offcode[c]);
          } else {
            // a granule that straddles an edge of the ordered observable, or an offset from which the flat index
            // could leave the histogram (or its row): the float columns decide, after the stream
            // (kOrdSkip, wave-uniform: outside the ordered observable's domain, nothing to count)
            const unsigned code = (unsigned)__builtin_amdgcn_readlane((int)codes[c], j);
            if (code != kOrdSkip) push_granule(c, it, code);
          }
          // (several chains: one after the other, not interleaved -- the registers of one chain's four samples are
          // all the launch bound of 1024 lanes leaves room for)
          if constexpr (NCHAIN > 1) __builtin_amdgcn_sched_barrier(0);
        }
      };

      // ---- what the queues hold, decided with the reference's arithmetic on the float columns
      auto drain = [&](auto KD) {
        constexpr int kDrain = decltype(KD)::value;        // granules whose float columns are loaded together
        if (dbg & 16u) nrow = ngran = 0u;                  // (measurement hook: what the queues hold is dropped)
        // (nothing else of this wave's is in flight here, so every wait is a whole memory round trip: the loads of
        // the first kDrain granules go out with the first rows' and share one)
        vfloat4 rawf[kDrain][NS], rawo[kDrain];
        unsigned prew[kDrain], gcode[kDrain], gc[kDrain];
        auto load_granule = [&](int s, unsigned g) {
          const unsigned w0 = (unsigned)uniform_i((int)qgran[2u * g]);
          gcode[s] = (unsigned)uniform_i((int)qgran[2u * g + 1u]);
          gc[s] = w0 >> 28;
          const unsigned vw = w0 & 0x0FFFFFFFu;
          const unsigned vc = vw + lane < vlast32 ? vw + lane : vlast32;
#pragma unroll
          for (int k = 0; k < NSTREAM; k++) rawf[s][k] = __builtin_nontemporal_load(&col[k][vc]);
          rawo[s] = vfloat4{0.0f, 0.0f, 0.0f, 0.0f};
          if (gcode[s] == kOrdMixed) rawo[s] = __builtin_nontemporal_load(&col[ORD][vc]);
          prew[s] = precol[vc >> 6];
        };
        unsigned row = 0u, w1 = 0u;
        float rowf[NS];
        const bool myrow = lane < nrow;
        if (myrow) {
          row = qrows[2u * lane];
          w1 = qrows[2u * lane + 1u];
#pragma unroll
          for (int k = 0; k < NSTREAM; k++) rowf[k] = ((gptr<const float>)col[k])[row];
        }
#pragma unroll
        for (int s = 0; s < kDrain; s++) {
          if ((unsigned)s < ngran) load_granule(s, (unsigned)s);
        }
        // ---- rows: one row with the reference's arithmetic on its float values (an ambiguous sample)
        auto exact_row = [&](int c, unsigned offcode) {
          double f[NSLOT][SXMC_VEC];
#pragma unroll
          for (int k = 0; k < NSTREAM; k++) f[k][0] = f[k][1] = f[k][2] = f[k][3] = (double)rowf[k];
          f[ORD][0] = f[ORD][1] = f[ORD][2] = f[ORD][3] = 0.0;
          run_static_part<NSLOT, ORD, false>(f, craw[c], PROG{}, Seq{});
          bool ind = true;
          int bin = (int)offcode;
#pragma unroll
          for (int k = 0; k < NOBS; k++) {
            const double x = f[k][0];
            ind = ind & (x >= lo[k]) & (x < hi[k]);
            const int idx = (int)((x - lo[k]) * sc[k]);
            bin = mad24(idx, stv[k], bin);
          }
          vcnt[c] += (ind && !((unsigned)bin < B)) ? 1u : 0u;   // (the others are counted with the histogram)
          if (ind && ((unsigned)bin < B) && !(dbg & 4u)) {
            __hip_atomic_fetch_add(&hist[(unsigned)c * cstride + myrep + codes_word((unsigned)bin)], 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        };
        for (unsigned i0 = 0; i0 < nrow; i0 += kWave) {
          if (i0 != 0u && i0 + lane < nrow) {              // (further batches: their own round trip)
            row = qrows[2u * (i0 + lane)];
            w1 = qrows[2u * (i0 + lane) + 1u];
#pragma unroll
            for (int k = 0; k < NSTREAM; k++) rowf[k] = ((gptr<const float>)col[k])[row];
          }
          if (i0 + lane < nrow) {
            const unsigned cc = NCHAIN > 1 ? w1 >> 28 : 0u, oc = NCHAIN > 1 ? (w1 & 0x0FFFFFFFu) : w1;
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) {
              if (cc == (unsigned)c) exact_row(c, oc);
            }
          }
        }
        // ---- whole granules
        for (unsigned g0 = 0; g0 < ngran; g0 += (unsigned)kDrain) {
          if (g0 != 0u) {
#pragma unroll
            for (int s = 0; s < kDrain; s++) {
              if (g0 + (unsigned)s < ngran) load_granule(s, g0 + (unsigned)s);
            }
          }
#pragma unroll
          for (int s = 0; s < kDrain; s++) {
            if (g0 + (unsigned)s >= ngran) break;
            const bool mixed = gcode[s] == kOrdMixed;
            const unsigned off = (unsigned)uniform_i((int)prew[s]) & 0xFFFFFFu;
            double f0[NS][SXMC_VEC];
#pragma unroll
            for (int k = 0; k < NSTREAM; k++) {
              f0[k][0] = (double)rawf[s][k].x;
              f0[k][1] = (double)rawf[s][k].y;
              f0[k][2] = (double)rawf[s][k].z;
              f0[k][3] = (double)rawf[s][k].w;
            }
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) {
              if (gc[s] == (unsigned)c) exact_granule(IntC<0>{}, c, mixed, f0, rawo[s], off, gcode[s]);
            }
          }
        }
        nrow = ngran = 0u;
      };

      // ---- the stream.  The inner loop holds no memory access but the ring's own loads and the LDS additions, so the
      // wait for a unit is a counted one (the loads of the kRing - 1 younger units stay in flight); it ends at every
      // 64th unit (the next granules' codes and words: loads and a wait) and when the granule queue could not take
      // another round (the queues are then emptied: loads and waits too).  A unit's registers are re-loaded right
      // after it has been worked on.
      const unsigned niter32 = (unsigned)niter;
      const unsigned vfirst16 = vfirst32 * 16u, vlast16 = vlast32 * 16u, step16 = step32 * 16u;
      const unsigned vcap16 = (dbg & 2u) ? 0u : vlast16;   // (measurement hook 2: every load hits one address)
      auto issue = [&](int slot, unsigned it) {
        unsigned o = vfirst16 + it * step16;               // (it * step16: scalar)
        o = o < vcap16 ? o : vcap16;
#pragma unroll
        for (int w = 0; w < QW; w++) {
          rq[slot][w] = __builtin_nontemporal_load((gptr<const vuint4g>)((gptr<const char>)qcol[w] + (unsigned long long)o));
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto stream = [&](auto OUTER) {
        unsigned it = 0u;
#pragma unroll
        for (int i = 0; i < kRing; i++) issue(i, (unsigned)i);
        while (it < niter32) {
          if ((it & 63u) == 0u) {
            granule_codes((unsigned long long)it);
            granule_words(it);
          }
          const unsigned block_end = niter32 < (it | 63u) + 1u ? niter32 : (it | 63u) + 1u;
          bool full = false;
          while (it < block_end && !full) {
#pragma unroll
            for (int i = 0; i < kRing; i++) {
#pragma unroll
              for (int q = 0; q < QW; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) asm volatile("" : "+v"(rq[i][q][e]));   // (the wait for this unit goes here)
              }
              if (it + (unsigned)i < block_end) process_unit(OUTER, it + (unsigned)i, rq[i]);
              issue(i, it + (unsigned)(i + kRing));     // (unconditional: the waits above count on it)
            }
            // (only the segment's last round can be a partial one: blocks end at multiples of 64)
            it = block_end - it < (unsigned)kRing ? block_end : it + (unsigned)kRing;
            full = ngran + (unsigned)(kRing * NCHAIN) > gq_cap;
            // EARLY DRAIN (measurement builds: -DSXMC_DRAIN_EVERY=32; off in the product, see kDrainEvery): what the queues
            // hold is also decided every kDrainEvery units, in the middle of the stream -- the wave stalls for the round
            // trips of its rows and granules while the CU's other waves keep the stream going
            if constexpr (kDrainEvery > 0) {
              full = full || ((it & (unsigned)(kDrainEvery - 1)) == 0u && (nrow | ngran) != 0u && it < niter32);
            }
          }
          if (full) {
            drain(IntC<kDrainStream>{});
            // (what the drain's loads leave in the compiler's model of outstanding loads reached the loop header and made
            // it wait for ALL of the ring's loads at the first unit of every round -- s_waitcnt vmcnt(0) where vmcnt(3)
            // was meant.  An explicit full wait here, where nothing of the ring's is worth keeping in flight anyway,
            // clears that model: the stream loop's waits are then the counted ones.)
            __builtin_amdgcn_s_waitcnt(0);
          }
        }
      };
      if constexpr (NOBS == 1) {
        if (outer) stream(IntC<1>{});
        else stream(IntC<0>{});
      } else {
        stream(IntC<0>{});
      }

      // (each wave empties its own queues as it leaves the stream.  Sharing what is left over the workgroup -- wave w
      // takes granules w, w + W, ... of all queues, every load issued before any wait -- was built and measured on one
      // box, alternating: 87.9 us against 81.3: the barrier makes every wave wait for the slowest one before any of the
      // work starts, whereas now the early waves empty their queues under the others' streaming.)
      drain(IntC<kDrainFinal>{});
    }

#if SXMC_WG_STAMPS
    __syncthreads();   // (measurement build: the stamp is the workgroup's last wave leaving the stream)
    SX_WG_STAMP(1);
#endif
    // in-domain counts: lane registers -> wave -> workgroup
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      unsigned t = vcnt[c] + ((dbg & 1u) ? sink : 0u);
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_down(t, off, kWave);
      t += ncnt[c];
      if (lane == 0 && t != 0u) {
        __hip_atomic_fetch_add(&s_norm[c], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) {
      const SxSignalDesc& dc = chains.d[c][sg.sig];
      gptr<unsigned> gbins = to_global(dc.bins);
      // One word per lane and step (neighbouring lanes -> neighbouring bins: the atomics of a wave fall into one or
      // two cache lines), and nothing in a step depends on what an earlier step read, so the reads of several steps
      // are in flight together.  (A first version cleared each non-zero word at once: 64 dependent LDS round trips
      // per lane, 5 us of every launch with the atomics only 1 us of it, as in-kernel timestamps showed.)  The
      // words are cleared only if the workgroup has another segment to count.
      const bool more = si + 1u < seg_end;
      const unsigned wend = (B + 63u) & ~63u;
      unsigned total = 0u;
      if (outer) {
        // the padded form: word (idx + 1) * S' + r of a replica is bin idx * S + r; the guard rows and the pad word
        // of each row (r = S, when S is even) are not bins
        const unsigned nwords = (unsigned)d.nbins[0] * oSp;
        const float invSp = 1.0f / (float)oSp;
#pragma unroll 4
        for (unsigned w = tid; w < nwords; w += nthreads) {
          const unsigned idx = (unsigned)(((float)w + 0.5f) * invSp);   // w / S' (exact: S' * (nbins + 1) < 2^22)
          const unsigned r = w - idx * oSp;
          unsigned n = 0u;
          for (unsigned rep = 0; rep < R; rep++) n += hist[(unsigned)c * cstride + rep * rstride + oSp + w];
          if (r < oS) {
            total += n;
            if (n != 0u) __hip_atomic_fetch_add(&gbins[idx * oS + r], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
#pragma unroll 4
      for (unsigned w = tid; LDS_HIST && !outer && w < wend; w += nthreads) {
        unsigned n = 0u;
        for (unsigned r = 0; r < R; r++) n += hist[(unsigned)c * cstride + r * rstride + w];
        total += n;
        // (a word of the last block whose bin would be >= B was never written)
        if (n != 0u) __hip_atomic_fetch_add(&gbins[lds_slot(w)], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (use_q) {   // (codes: the in-domain count is the histogram's sum + what vcnt holds; workgroup-uniform)
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kWave);
        if (lane == 0 && total != 0u) {
          __hip_atomic_fetch_add(&s_norm[c], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
      }
      if (LDS_HIST && more) {
        __syncthreads();
        for (unsigned b = 4u * tid; b < cstride; b += 4u * nthreads) {
          *reinterpret_cast<vuint4g*>(&hist[(unsigned)c * cstride + b]) = vuint4g{0u, 0u, 0u, 0u};
        }
      }
      if (tid == 0) {
        const unsigned n = s_norm[c];
        if (n != 0u) __hip_atomic_fetch_add(to_global(dc.norm), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_norm[c] = 0u;
      }
    }
    __syncthreads();
    lds_clean = true;
  }
  SX_WG_STAMP(2);
}

// ================================================================================================ BOXED OBSERVABLE
// fill_boxed_body: a bucketed table whose rows are grouped, inside every bucket, so that ONE observable with TWO input
// fields -- written by one-coefficient shift / scale / cos-theta scale / resolution scale against a field nothing
// writes (BASELINE config 3: e, scaled and resolution-scaled against e_true) -- is a per-granule constant, the way the
// ordered observable of fill_ordered_body is, and the fill streams only the OTHER written observable: one field, as
// 16-bit codes, 2 bytes per sample instead of 4.
//   Layout (sxmc_launch_plan.cpp, layout_kernels.hip): rows of a bucket sorted by (stratum of x - t, x), x the boxed
// observable's raw value and t its truth field, so that the 256 rows of a granule lie in a small BOX
// [xmin, xmax] x [tmin, tmax], kept per granule (16 bytes).
//   Per evaluation one lane per granule runs the reference's operations on the box in INTERVAL form: every IEEE
// operation the program applies (x + p, x * (1 + p), 1 + (x - 1) * (1 + p), x + p * (x - t), then the domain test
// and (int)((x - lo) * scale)) is monotone in each operand under round-to-nearest -- rounding never reverses an
// order -- so the same operations on the box's corners, the corner chosen by the sign of the coefficient, bound every
// row's result FROM BOTH SIDES, exactly, with no error term: interval arithmetic in which the endpoints round the way
// the values between them do.  (A resolution scale with p < 0 uses x twice with opposite monotonicity; the interval
// [xlo + p (xhi - tlo), xhi + p (xlo - thi)] is then wider than the image, never narrower.)  Every intermediate
// endpoint must be finite and no coefficient NaN; then nothing between the endpoints is NaN either.  If both ends
// land in the same bin (or outside the domain on the same side) so does every row of the granule: the observable
// costs nothing per sample.  Otherwise ("mixed": the box straddles an edge; a few per cent of the granules) the
// granule goes to a queue and is binned from its three float columns with the reference's arithmetic, as the
// granules that straddle an edge of the ordered observable are in fill_ordered_body.
//   The streamed observable: codes exactly as in fill_ordered_body (THE BOUND there, with NQ = 1), the LDS histogram
// in the padded form with THAT observable as the outermost dimension of the LDS copy whatever its place in the real
// histogram (the flush translates: a word (idx, hi * s + lo) is bin hi * s * nbins + idx * s + lo).
// Slots: 0 the streamed observable, 1 the truth field, 2 the boxed observable (geometry at index 1).
template <unsigned OPC, int XT>
__device__ __forceinline__ void apply_box_interval(double& xl, double& xh, double tl, double th, bool& fin, const double* c) {
  constexpr int type = (int)(OPC & 15u), E = (int)((OPC >> 8) & 15u);
  static_assert(sx_op_npars(OPC) == 1, "not a one-coefficient systematic");
  const double pc = 0.0 + c[0] * 1.0;   // (as apply_static)
  if constexpr (type == SXMC_SYST_SHIFT) {
    xl = xl + pc;
    xh = xh + pc;
  }
  if constexpr (type == SXMC_SYST_SCALE) {
    const double s = 1 + pc, a = xl * s, b = xh * s;
    xl = s >= 0.0 ? a : b;
    xh = s >= 0.0 ? b : a;
  }
  if constexpr (type == SXMC_SYST_CTSCALE) {
    const double s = 1 + pc, a = 1 + (xl - 1) * s, b = 1 + (xh - 1) * s;
    xl = s >= 0.0 ? a : b;
    xh = s >= 0.0 ? b : a;
  }
  if constexpr (type == SXMC_SYST_RESOLUTION_SCALE) {
    static_assert(E == XT, "the boxed observable reads one truth field");
    const double dl = xl - th, dh = xh - tl;            // x - t, smallest and largest
    const double a = pc * dl, b = pc * dh;
    const double ml = pc >= 0.0 ? a : b, mh = pc >= 0.0 ? b : a;
    fin = fin && (__builtin_fabs(dl) < __builtin_inf()) && (__builtin_fabs(dh) < __builtin_inf()) &&
          (__builtin_fabs(ml) < __builtin_inf()) && (__builtin_fabs(mh) < __builtin_inf());
    xl = xl + ml;
    xh = xh + mh;
  }
  fin = fin && (__builtin_fabs(xl) < __builtin_inf()) && (__builtin_fabs(xh) < __builtin_inf());   // (NaN fails)
}
template <int BX, int XT, unsigned... OPS, unsigned long... I>
__device__ __forceinline__ void run_box_interval(double& xl, double& xh, double tl, double th, bool& fin, const double* c,
                                                 StaticProg<OPS...>, ISeq<I...>) {
  ([&] {
    if constexpr ((int)((OPS >> 4) & 15u) == BX) apply_box_interval<OPS, XT>(xl, xh, tl, th, fin, c + sx_prog_cstart<OPS...>((int)I));
  }(), ...);
}
// can the program run that way?  The boxed observable's systematics have one coefficient and read at most slot XT,
// which nothing writes; the others (composed into an affine map of their own field) have one coefficient too and
// read nothing but their own slot
template <int BX, int XT, unsigned... OPS>
constexpr bool prog_is_boxable(StaticProg<OPS...>) {
  return (true && ... &&
          (sx_op_npars(OPS) == 1 && (int)((OPS >> 4) & 15u) != XT &&
           ((int)(OPS & 15u) != SXMC_SYST_RESOLUTION_SCALE ||
            ((int)((OPS >> 4) & 15u) == BX && (int)((OPS >> 8) & 15u) == XT))));
}

#ifndef SXMC_BOX_DRAIN
#define SXMC_BOX_DRAIN 2
#endif
#ifndef SXMC_BOX_GQ
#define SXMC_BOX_GQ 0        // (A/B builds: granules a wave queues before it empties its queues in mid-stream; 0: what fits)
#endif
#ifndef SXMC_BOX_PREFETCH
#define SXMC_BOX_PREFETCH 0  // 1: the next block's boxes and words are loaded while the current block is worked on (measured,
                             // one box, alternating: 65.2-66.8 us with, 64.6-65.3 without -- the compiler moves the
                             // prefetched word to another register at once and waits for everything to do so)
#endif
template <int NOBS, int NSLOT, typename PROG>
__device__ __forceinline__ void fill_boxed_body(const SxSignalDesc* __restrict__ descs, const SxSegment* __restrict__ segs,
                                                const unsigned* __restrict__ blk_off, unsigned layout, unsigned dbg_arg) {
  const unsigned dbg = sx_dbg(dbg_arg);   // (0 in the product build: see SXMC_MEASURE)
  static_assert(!PROG::dynamic && NOBS == 1 && NSLOT == 3, "one streamed observable, one truth field, the boxed observable");
  constexpr int BX = 2, XT = 1;        // the boxed observable's slot, its truth field's
  static_assert(prog_is_boxable<BX, XT>(PROG{}), "not a program the boxed form can run");
  constexpr int NC = PROG::ncoef > 0 ? PROG::ncoef : 1;
#ifndef SXMC_BOX_RING
#define SXMC_BOX_RING 8   // (16, 125 VGPRs: 65.9-66.2 us against 64.1-64.2, one box alternating -- profiles/r05_ring_depth_ab.log)
#endif
  constexpr int kRing = SXMC_BOX_RING;   // units of codes (8 bytes each) a lane holds: one worked on, the others in flight
  constexpr int kDrain = SXMC_BOX_DRAIN;   // queued granules whose float columns the drain loads together
  static_assert(64 % kRing == 0, "a block of 64 units is a whole number of rounds of the ring");
  typedef typename MakeISeq<PROG::n>::type Seq;
  extern __shared__ unsigned lds[];
  const unsigned tid = threadIdx.x;
  const unsigned nthreads = blockDim.x;
  const unsigned lane = tid & (kWave - 1);
  // LDS as in fill_ordered_body's padded form: 4 header words, R replicas of (nbins + 2) rows of S' words, 64 spare
  // words, the waves' queues.  `layout` = rstride | log2(R) << 24 | 1 << 27 | log2(queue entries) << 28.
  unsigned* s_norm = lds;
  unsigned* hist = lds + 4;
  const unsigned rstride = layout & 0xFFFFFFu, rlog = (layout >> 24) & 7u, R = 1u << rlog;
  const unsigned cstride = rstride << rlog;
  const unsigned myrep = (lane & (R - 1u)) * rstride;
  const unsigned qlog = layout >> 28;
  const unsigned qwave = qlog ? (1u << qlog) / (nthreads / kWave) : 0u;      // entries per wave
  // (half of a wave's slice for whole granules: at config 3 a few per cent of ~100 granules per wave)
  const unsigned gq_min = (unsigned)kRing + 1u;
  const unsigned gq_cap = qwave / 2u > gq_min ? qwave / 2u : gq_min, rq_cap = qwave > gq_cap ? qwave - gq_cap : 0u;
  const unsigned gq_drain = (SXMC_BOX_GQ > 0 && (unsigned)SXMC_BOX_GQ + (unsigned)kRing < gq_cap) ? (unsigned)SXMC_BOX_GQ + (unsigned)kRing : gq_cap;
  unsigned* qrows = lds + 4 + cstride + 64 + 4 + (tid / kWave) * (2u * qwave);
  unsigned* qgran = qrows + 2u * rq_cap;

  bool lds_clean = false;
  const unsigned seg_end = blk_off[blockIdx.x + 1];
  for (unsigned si = blk_off[blockIdx.x]; si < seg_end; ++si) {
    const SxSegment& sg = segs[si];
    const SxSignalDesc& d = descs[sg.sig];
    const unsigned long long v0 = sg.v0, v1 = sg.v1, step = sg.step;
    const unsigned B = (unsigned)d.total_nbins;
    double craw[NC];
#pragma unroll
    for (int q = 0; q < PROG::ncoef; q++) craw[q] = to_global(d.params)[(long)d.coef_par[q] * d.param_stride];
    __builtin_amdgcn_sched_barrier(0);

    gptr<const vfloat4> col[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
      col[k] = to_global(reinterpret_cast<const vfloat4*>(d.cols + (unsigned long long)d.slot_col[k] * d.col_pitch));
    }
    gptr<const unsigned> precol = to_global(reinterpret_cast<const unsigned*>(d.pre));
    gptr<const vfloat4> boxes = to_global(reinterpret_cast<const vfloat4*>(d.boxes));
    // (no table of codes: every granule goes to the float columns; the ring then loads from the first float column --
    // twice as long -- so that its loads stay unconditional and its waits counted)
    gptr<const vuint2g> qcol = to_global(d.qcol != nullptr ? reinterpret_cast<const vuint2g*>(d.qcol)
                                                           : reinterpret_cast<const vuint2g*>(d.cols));
    const unsigned vfirst32 = (unsigned)(v0 + tid), step32 = (unsigned)step, vlast32 = (unsigned)(v1 - 1);
    const unsigned vwave32 = vfirst32 - lane;             // the wave's first unit: a granule boundary
    const unsigned niter32 = (unsigned)((v1 - v0 + step - 1) / step);

    if (!lds_clean) {
      for (unsigned b = tid; b < cstride; b += nthreads) hist[b] = 0u;
      if (tid < 4) s_norm[tid] = 0u;
      __syncthreads();
    }

    // geometry: the streamed observable (index 0) and the boxed one (index NOBS)
    const double lo = d.lower[0], hi = d.upper[0], sc = d.scale[0];
    const int st = d.bin_stride[0];
    const unsigned nbk = (unsigned)d.nbins[0];
    const double olo = d.lower[NOBS], ohi = d.upper[NOBS], osc = d.scale[NOBS];
    const int ost = d.bin_stride[NOBS];
    int stv = st, ostv = ost;
    asm volatile("" : "+v"(stv));
    asm volatile("" : "+v"(ostv));
    // the LDS copy of the histogram: row idx + 1 of S' words per index of the streamed observable, word hi * s + lo of a
    // row for the bin hi * (s * nbins) + idx * s + lo (s: the observable's stride in the real histogram)
    const unsigned s_in = (unsigned)st, s_out = s_in * nbk;          // strides of the index and of what lies above it
    const unsigned oS = B / nbk, oSp = oS | 1u;
    const float inv_in = 1.0f / (float)s_in, inv_out = 1.0f / (float)s_out;
    // (x / y for x, y < 2^22, exact)
    auto fdiv = [](unsigned x, float inv) -> unsigned { return (unsigned)(((float)x + 0.5f) * inv); };
    // word of real bin `bin` (< B) inside a replica
    auto hist_word = [&](unsigned bin) -> unsigned {
      const unsigned h = fdiv(bin, inv_out), rem = bin - h * s_out, idx = fdiv(rem, inv_in);
      return (idx + 1u) * oSp + h * s_in + (rem - idx * s_in);
    };

    bool wild = false;
#pragma unroll
    for (int q = 0; q < PROG::ncoef; q++) wild = wild || !(__builtin_fabs(craw[q]) < __builtin_inf());

    // ---- the streamed observable's program composed into single precision over its code (fill_ordered_body, THE BOUND)
    float af, gf, thr;
    bool use_q = d.qcol != nullptr && rq_cap >= 8u && !wild;
    {
      AffineForm<1> form[1];
      form[0].a[0] = 1.0;
      form[0].c = 0.0;
      const double wlo = d.qbase[0], whi = d.qbase[0] + 65534.0 * d.qstep[0];
      form[0].mag = __builtin_fmax(__builtin_fabs(wlo), __builtin_fabs(whi));
      run_affine<1, BX>(form, craw, PROG{}, Seq{});
      const double alpha = form[0].a[0] * d.qstep[0] * sc;
      const double sum_abs = __builtin_fabs(alpha);
      double g = form[0].c - lo;
      g = g + form[0].a[0] * (d.qbase[0] + 0.5 * d.qstep[0]);
      g = g * sc;
      const double mu = sum_abs * 65536.0 + __builtin_fabs(g) + (double)nbk + 0.25;
      const double epsq = 0.5 * sum_abs * (1.0 + 0x1p-19);
      const double eps = epsq + mu * 0x1p-21 + (form[0].mag + __builtin_fabs(lo)) * sc * 0x1p-44;
      const bool ok = eps < 0.125;                       // (NaN fails)
      const double e = eps * 1.01, slack = 0x1p-23;
      af = uniform_f((float)alpha);
      gf = uniform_f((float)(g + e));
      thr = uniform_f((float)(2.0 * e + slack));
      use_q = use_q && (uniform_i(ok ? 1 : 0) != 0);
    }

    unsigned vcnt = 0u;                      // in domain, flat index past the end: counted in the norm only
    unsigned nrow = 0u, ngran = 0u;          // entries in the wave's queues (wave-uniform)

    // ---- per block of 64 units, one lane per granule: the box through the program, and the granule's words
    //   gword: kOrdSkip (outside the boxed observable's domain), kOrdMixed (the float columns decide), or the word
    //          of (idx = -1, bin 0 of the granule's row part) inside a replica's padded copy;  greal: the granule's
    //          offset in the real histogram (what the ambiguous rows' queue entries carry)
    unsigned gword = kOrdSkip, greal = 0u;
    vfloat4 bx_next;
    unsigned pre_next;
    auto meta_load = [&](unsigned it) {
      unsigned vg = vwave32 + (it + lane) * step32;
      vg = vg <= vlast32 ? vg : vlast32;
      bx_next = boxes[vg >> 6];
      pre_next = precol[vg >> 6];
    };
    auto granule_meta = [&](unsigned it) {
      unsigned vg = vwave32 + (it + lane) * step32;
      const bool live = vg <= vlast32;
      vg = live ? vg : vlast32;
      if constexpr (!SXMC_BOX_PREFETCH) meta_load(it);
      const vfloat4 bx = bx_next;
      const unsigned pre = pre_next;
      if constexpr (SXMC_BOX_PREFETCH) meta_load(it + 64u);   // (the next block's: in flight behind the ring's loads)
      double xl = (double)bx.x, xh = (double)bx.y;
      const double tl = (double)bx.z, th = (double)bx.w;
      bool fin = (__builtin_fabs(xl) < __builtin_inf()) && (__builtin_fabs(xh) < __builtin_inf()) &&
                 (__builtin_fabs(tl) < __builtin_inf()) && (__builtin_fabs(th) < __builtin_inf());
      run_box_interval<BX, XT>(xl, xh, tl, th, fin, craw, PROG{}, Seq{});
      const int i0 = (int)((xl - olo) * osc), i1 = (int)((xh - olo) * osc);
      const int e0 = !(xl >= olo) ? -1 : (!(xl < ohi) ? 0x7FFFFFFF : i0);
      const int e1 = !(xh >= olo) ? -1 : (!(xh < ohi) ? 0x7FFFFFFF : i1);
      const unsigned off = (pre & 0xFFFFFFu) + (unsigned)e0 * (unsigned)ost;    // (meaningless unless e0 is an index)
      const unsigned h = fdiv(off, inv_out), rem = off - h * s_out;
      // binned from codes only if every index of the streamed observable keeps the flat index canonical and in range
      const bool canon = off < (1u << 22) && rem < s_in && (h + 1u) * s_out <= B;
      const bool mixed = !use_q || !fin || e0 != e1 || !canon;
      const bool skip = fin && e0 == e1 && (e0 < 0 || e0 == 0x7FFFFFFF);
      gword = !live ? kOrdSkip : skip ? kOrdSkip : mixed ? kOrdMixed : h * s_in + rem;
      greal = off;
    };

    // ---- queues (fill_ordered_body): a granule left to the float columns; the ambiguous rows of a granule
    auto push_granule = [&](unsigned it) {
      if (lane == 0) qgran[2u * ngran] = vwave32 + it * step32;
      ngran += 1u;
    };
    auto push_rows = [&](unsigned rare, unsigned v, unsigned real_off) -> bool {
      unsigned long long mask[SXMC_VEC];
      unsigned total = 0u;
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        mask[q] = __builtin_amdgcn_ballot_w64(((rare >> q) & 1u) != 0u);
        total += (unsigned)__builtin_popcountll(mask[q]);
      }
      if (nrow + total > rq_cap) return false;
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        if (mask[q] == 0ull) continue;
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[q] >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)mask[q], 0u));
        if ((rare >> q) & 1u) {
          const unsigned pos = nrow + below;
          qrows[2u * pos] = v * SXMC_VEC + (unsigned)q;
          qrows[2u * pos + 1u] = real_off;
        }
        nrow += (unsigned)__builtin_popcountll(mask[q]);
      }
      return true;
    };

    // ---- one unit of codes: four samples of the streamed observable (coarse_outer of fill_ordered_body with one field)
    const int oS4 = (int)(4u * oSp);
    const int oclamp = (dbg & 4u) ? -1 : (int)nbk;   // (measurement hook: everything to the guard row)
    auto coarse_unit = [&](const vuint2g& w, unsigned it, unsigned v, unsigned word0) {
      int base4 = (int)(4u * (4u + myrep + word0) + 4u * oSp);      // byte address of (idx = -1 ... + 1 row = idx 0)
      asm volatile("" : "+v"(base4));
      const float c0 = (float)(w.x & 0xFFFFu), c1 = (float)(w.x >> 16), c2 = (float)(w.y & 0xFFFFu), c3 = (float)(w.y >> 16);
      const vfloat2 a2 = {af, af}, g2 = {gf, gf};
      const vfloat2 u01 = __builtin_elementwise_fma(a2, vfloat2{c0, c1}, g2);
      const vfloat2 u23 = __builtin_elementwise_fma(a2, vfloat2{c2, c3}, g2);
      const float us[SXMC_VEC] = {u01.x, u01.y, u23.x, u23.y};
      int addr[SXMC_VEC];
      bool amb[SXMC_VEC];
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        const float u = us[q];
        amb[q] = !(__builtin_amdgcn_fractf(u) >= thr);               // closer to a bin edge than the bound (or NaN)
        int idx;
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(idx) : "v"(u));
        int ie;
        asm("v_med3_i32 %0, %1, -1, %2" : "=v"(ie) : "v"(amb[q] ? -1 : idx), "s"(oclamp));
        addr[q] = mad24(ie, oS4, base4);
      }
      const float cmax = __builtin_fmaxf(__builtin_fmaxf(c0, c1), __builtin_fmaxf(c2, c3));
      const bool special = cmax >= (float)SXMC_QCODE_EXACT;
      const bool anyamb = (amb[0] | amb[1]) | (amb[2] | amb[3]);
      if (__builtin_amdgcn_ballot_w64(anyamb || special) != 0ull) {
        const float cs[SXMC_VEC] = {c0, c1, c2, c3};
        unsigned rare = 0u;
#pragma unroll
        for (int q = 0; q < SXMC_VEC; q++) {
          rare |= amb[q] ? (1u << q) : 0u;
          if (cs[q] >= (float)SXMC_QCODE_EXACT) {          // the arithmetic above meant nothing for this row
            addr[q] = base4 - oS4;                         // (the guard row)
            rare = cs[q] >= (float)SXMC_QCODE_NEVER ? (rare & ~(1u << q)) : (rare | (1u << q));
          }
        }
        const unsigned real_off = (unsigned)__builtin_amdgcn_readlane((int)greal, (int)(it & 63u));
        if (!push_rows(rare, v, real_off)) {               // more than the queue holds: the whole granule, later
          push_granule(it);
          return;
        }
      }
      if (dbg & 32u) {                                     // (measurement hook: no LDS additions at all)
        vcnt += (unsigned)(addr[0] ^ addr[1] ^ addr[2] ^ addr[3]) == 12345u ? 1u : 0u;
        return;
      }
      // (issuing the additions on their byte addresses by hand -- the pointer form costs one v_add of the LDS block's
      // relocated base, 0, per sample -- was measured, one box, alternating: 63.2-64.5 us against 62.9-65.6: nothing)
#pragma unroll
      for (int q = 0; q < SXMC_VEC; q++) {
        unsigned* wp = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + addr[q]);
        __hip_atomic_fetch_add(wp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };

    // ---- what the queues hold, with the reference's arithmetic on the float columns (pdfz.cpp:306-331, 388-398)
    auto count_exact = [&](bool ind, int bin) {
      vcnt += (ind && !((unsigned)bin < B)) ? 1u : 0u;     // (the others are counted with the histogram)
      if (ind && ((unsigned)bin < B)) {
        __hip_atomic_fetch_add(&hist[myrep + hist_word((unsigned)bin)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    };
    auto drain = [&]() {
      if (dbg & 16u) nrow = ngran = 0u;                  // (measurement hook: what the queues hold is dropped)
      vfloat4 rawf[kDrain][NSLOT];
      unsigned prew[kDrain];
      auto load_granule = [&](int s, unsigned g) {
        const unsigned vw = (unsigned)uniform_i((int)qgran[2u * g]);
        const unsigned vc = vw + lane < vlast32 ? vw + lane : vlast32;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) rawf[s][k] = __builtin_nontemporal_load(&col[k][vc]);
        prew[s] = precol[vc >> 6];
      };
      unsigned row = 0u, w1 = 0u;
      float rowf = 0.0f;
      if (lane < nrow) {
        row = qrows[2u * lane];
        w1 = qrows[2u * lane + 1u];
        rowf = ((gptr<const float>)col[0])[row];
      }
#pragma unroll
      for (int s = 0; s < kDrain; s++) {
        if ((unsigned)s < ngran) load_granule(s, (unsigned)s);
      }
      // rows: the streamed observable of one row (its granule's other indices are the entry's offset)
      for (unsigned i0 = 0; i0 < nrow; i0 += kWave) {
        if (i0 != 0u && i0 + lane < nrow) {
          row = qrows[2u * (i0 + lane)];
          w1 = qrows[2u * (i0 + lane) + 1u];
          rowf = ((gptr<const float>)col[0])[row];
        }
        if (i0 + lane < nrow) {
          double f[NSLOT][SXMC_VEC];
#pragma unroll
          for (int k = 0; k < NSLOT; k++) f[k][0] = f[k][1] = f[k][2] = f[k][3] = 0.0;
          f[0][0] = f[0][1] = f[0][2] = f[0][3] = (double)rowf;
          run_static_part<NSLOT, BX, false>(f, craw, PROG{}, Seq{});
          const double x = f[0][0];
          const bool ind = (x >= lo) & (x < hi);
          const int idx = (int)((x - lo) * sc);
          count_exact(ind, mad24(idx, stv, (int)w1));
        }
      }
      // whole granules: all three columns
      for (unsigned g0 = 0; g0 < ngran; g0 += (unsigned)kDrain) {
        if (g0 != 0u) {
#pragma unroll
          for (int s = 0; s < kDrain; s++) {
            if (g0 + (unsigned)s < ngran) load_granule(s, g0 + (unsigned)s);
          }
        }
#pragma unroll
        for (int s = 0; s < kDrain; s++) {
          if (g0 + (unsigned)s >= ngran) break;
          const unsigned off = (unsigned)uniform_i((int)prew[s]) & 0xFFFFFFu;
          double f[NSLOT][SXMC_VEC];
#pragma unroll
          for (int k = 0; k < NSLOT; k++) {
            f[k][0] = (double)rawf[s][k].x;
            f[k][1] = (double)rawf[s][k].y;
            f[k][2] = (double)rawf[s][k].z;
            f[k][3] = (double)rawf[s][k].w;
          }
          run_static<NSLOT>(f, craw, PROG{}, Seq{});
#pragma unroll
          for (int q = 0; q < SXMC_VEC; q++) {
            const double x = f[0][q], y = f[BX][q];
            const bool ind = (x >= lo) & (x < hi) & (y >= olo) & (y < ohi);      // (NaN fails: padding rows too)
            const int idx = (int)((x - lo) * sc), oidx = (int)((y - olo) * osc);
            count_exact(ind, mad24(oidx, ostv, mad24(idx, stv, (int)off)));
          }
        }
      }
      nrow = ngran = 0u;
    };

    // ---- the stream: a ring of kRing units of codes, counted waits (fill_ordered_body)
    vuint2g rq[kRing];
    const unsigned vfirst8 = vfirst32 * 8u, vlast8 = vlast32 * 8u, step8 = step32 * 8u;
    auto issue = [&](int slot, unsigned it) {
      unsigned o = vfirst8 + it * step8;
      o = o < vlast8 ? o : vlast8;
      rq[slot] = __builtin_nontemporal_load((gptr<const vuint2g>)((gptr<const char>)qcol + (unsigned long long)o));
      __builtin_amdgcn_sched_barrier(0);
    };
    const unsigned climit = (unsigned)uniform_i((int)oS);   // a granule word below this: binned from codes (a scalar compare)
    {
      unsigned it = 0u;
      if constexpr (SXMC_BOX_PREFETCH) meta_load(0u);
#pragma unroll
      for (int i = 0; i < kRing; i++) issue(i, (unsigned)i);
      while (it < niter32) {
        if ((it & 63u) == 0u) granule_meta(it);
        const unsigned block_end = niter32 < (it | 63u) + 1u ? niter32 : (it | 63u) + 1u;
        bool full = false;
        while (it < block_end && !full) {
#pragma unroll
          for (int i = 0; i < kRing; i++) {
#pragma unroll
            for (int e = 0; e < 2; e++) asm volatile("" : "+v"(rq[i][e]));   // (the wait for this unit goes here)
            const unsigned iu = it + (unsigned)i;
            if (iu < block_end) {
              const unsigned word0 = (unsigned)__builtin_amdgcn_readlane((int)gword, (int)(iu & 63u));
              if (dbg & 1u) {                                  // (measurement hook: the stream alone)
                vcnt += (rq[i].x == 12345u && word0 == 77u) ? 1u : 0u;
              } else if (word0 < climit) coarse_unit(rq[i], iu, vfirst32 + iu * step32, word0);
              else if (word0 == kOrdMixed) push_granule(iu);
            }
            issue(i, iu + (unsigned)kRing);   // (unconditional: the waits above count on it)
          }
          it = block_end - it < (unsigned)kRing ? block_end : it + (unsigned)kRing;
          full = ngran + (unsigned)kRing > gq_drain;
        }
        if (full) {
          drain();
          // (what the drain's loads leave in the compiler's model of outstanding loads would reach the loop header and add
          // waits far below the ring's depth to every unit -- s_waitcnt vmcnt(2) where vmcnt(7) is meant.  An explicit
          // full wait here clears that model: the loop's waits are then the counted ones and nothing else.)
          __builtin_amdgcn_s_waitcnt(0);
        }
      }
    }
    drain();

    // in-domain counts: lane registers -> wave -> workgroup
    {
      unsigned t = vcnt;
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_down(t, off, kWave);
      if (lane == 0 && t != 0u) __hip_atomic_fetch_add(&s_norm[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    // flush: word (idx + 1) * S' + r of a replica, r = hi * s + lo < S, is bin hi * (s * nbins) + idx * s + lo; the guard
    // rows and the pad word of each row (r = S, when S is even) are not bins
    {
      gptr<unsigned> gbins = to_global(d.bins);
      const bool more = si + 1u < seg_end;
      const unsigned nwords = nbk * oSp;
      const float invSp = 1.0f / (float)oSp;
      unsigned total = 0u;
#pragma unroll 4
      for (unsigned w = tid; w < nwords; w += nthreads) {
        const unsigned idx = fdiv(w, invSp), r = w - idx * oSp;
        unsigned n = 0u;
        for (unsigned rep = 0; rep < R; rep++) n += hist[rep * rstride + oSp + w];
        if (r < oS) {
          total += n;
          const unsigned h = fdiv(r, inv_in);
          if (n != 0u) {
            __hip_atomic_fetch_add(&gbins[h * s_out + idx * s_in + (r - h * s_in)], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kWave);
      if (lane == 0 && total != 0u) __hip_atomic_fetch_add(&s_norm[0], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __syncthreads();
      if (more) {
        for (unsigned b = 4u * tid; b < cstride; b += 4u * nthreads) {
          *reinterpret_cast<vuint4g*>(&hist[b]) = vuint4g{0u, 0u, 0u, 0u};
        }
      }
      if (tid == 0) {
        const unsigned n = s_norm[0];
        if (n != 0u) __hip_atomic_fetch_add(to_global(d.norm), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_norm[0] = 0u;
      }
    }
    __syncthreads();
    lds_clean = true;
  }
}

template <int NOBS, int NSLOT, bool LDS_HIST, typename PROG, int PREW>
__global__ __launch_bounds__(1024) void fill_kernel(const SxSignalDesc* __restrict__ descs,
                                                    const SxSegment* __restrict__ segs,
                                                    const unsigned* __restrict__ blk_off, unsigned hist_words,
                                                    unsigned dbg) {
  fill_body<NOBS, NSLOT, LDS_HIST, PROG, PREW>(descs, segs, blk_off, hist_words, dbg);
}

template <int NOBS, int NSLOT, typename PROG>
__global__ __launch_bounds__(1024) void fill_sparse_kernel(const SxSignalDesc* __restrict__ descs,
                                                           const SxSegment* __restrict__ segs,
                                                           const unsigned* __restrict__ blk_off, unsigned smax,
                                                           unsigned dbg) {
  fill_sparse_body<NOBS, NSLOT, PROG>(descs, segs, blk_off, smax, dbg);
}

#ifndef SXMC_ORDERED_BOUND
#define SXMC_ORDERED_BOUND 1024   // (measurement builds: make VARIANT=_b768 EXTRA=-DSXMC_ORDERED_BOUND=768)
#endif
template <int NOBS, int NSLOT, typename PROG, bool LDS_HIST = true>
__global__ __launch_bounds__(SXMC_ORDERED_BOUND) void fill_ordered_kernel(const SxSignalDesc* __restrict__ descs,
                                                            const SxSegment* __restrict__ segs,
                                                            const unsigned* __restrict__ blk_off, unsigned layout,
                                                            unsigned dbg) {
  SxChainDescs one;
  one.d[0] = one.d[1] = one.d[2] = one.d[3] = descs;
  fill_ordered_body<NOBS, NSLOT, PROG, 1, LDS_HIST>(one, segs, blk_off, layout, dbg);
}

// BOXED OR ORDERED.  The boxed form is fast while few granules' boxes straddle an edge -- config 3, one box: 59 us + 2.5 us
// per per cent of such granules, against 81 us for the ordered form whatever the parameters -- and that share grows with
// the resolution parameter (the box's image is |dx'/dx| dx + |dx'/dt| dt wide).  A plan with boxed tables therefore keeps
// the ordered plan beside it, and the HOST picks the form between flushes of a walk from the parameters it reads back
// (sxmc_group_adapt_fill_form).  Choosing on the device was built in three forms and measured (profiles/
// r05_boxed_dual_ab.log, r05_boxed_gated_pair.log): both bodies inlined in one kernel -- the ordered one spills, 95 us for
// 81; as functions of one kernel -- 116 and 87; as two launches of which one ends at once -- 85 and 70.
template <int NOBS, int NSLOT, typename PROG>
__global__ __launch_bounds__(SXMC_ORDERED_BOUND) void fill_boxed_kernel(const SxSignalDesc* __restrict__ descs,
                                                          const SxSegment* __restrict__ segs,
                                                          const unsigned* __restrict__ blk_off, unsigned layout,
                                                          unsigned dbg) {
  fill_boxed_body<NOBS, NSLOT, PROG>(descs, segs, blk_off, layout, dbg);
}

}  // namespace sxfill
