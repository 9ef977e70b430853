/*
 * sxmc_hip.h -- C ABI of libsxmc_hip.so: the MI355X (gfx950) implementation of sxmc's per-step
 * NLL evaluation (pdfz::EvalHist histogram fill with per-sample systematics + the nll_kernels
 * event log-sum, reduction and fused MCMC step).
 *
 * This is the drop-in boundary: plain pointers and sizes, `int` status returns, no exceptions,
 * no torch or C++ types.  Each entry point names the reference interface it replaces
 * (file:line relative to /root/reference).  The C++ mirror of the reference's own classes
 * (pdfz::Eval, pdfz::EvalHist, the device-mirror array and the kernel launch points mcmc.cpp
 * uses) lives in sxmc_amd/include/sxmc/ and is a header-only layer over this ABI.
 *
 * Unless stated otherwise pointers named `d_*` or documented "device" are HIP device pointers
 * (hipMalloc'd, or any pointer valid on the current device such as a torch tensor's data_ptr),
 * and everything else is host memory.  All functions return SXMC_OK (0) or an error code;
 * sxmc_last_error() returns the message of the last failure on the calling thread.
 */
#ifndef SXMC_HIP_H
#define SXMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SXMC_OK 0
#define SXMC_ERR_INVALID 1 /* argument validation failed: where the reference throws pdfz::Error */
#define SXMC_ERR_HIP 2     /* a HIP runtime call failed (reference: checkCuda) */
#define SXMC_ERR_STATE 3   /* call sequence error, e.g. eval before the buffers are bound */

#define SXMC_MAX_NFIELDS 10   /* pdfz.cpp:17 MAX_NFIELDS */
#define SXMC_MAX_SYST 16      /* systematics per evaluator (reference: unbounded array, pdfz.cpp:126-132) */
#define SXMC_MAX_SYST_PARS 8  /* polynomial coefficients per systematic */

/* pdfz.h:111-116 Systematic::Type */
#define SXMC_SYST_SHIFT 0
#define SXMC_SYST_SCALE 1
#define SXMC_SYST_RESOLUTION_SCALE 2
#define SXMC_SYST_CTSCALE 3

typedef struct sxmc_hist* sxmc_hist_t;   /* one pdfz::EvalHist (pdfz.h:402-574) */
typedef struct sxmc_group* sxmc_group_t; /* the set of evaluators MCMC steps together (mcmc.cpp:264-271) */
typedef void* sxmc_stream_t;             /* hipStream_t; NULL = the legacy default stream */
typedef void* sxmc_event_t;              /* hipEvent_t */

/* Counter-based generator state, one per parameter.  Replaces curandStateXORWOW
 * (nll_kernels.h:25-29 RNGState).  Philox4x32-10 keyed by `seed`, stream `subsequence`. */
typedef struct {
  uint64_t seed;
  uint64_t subsequence;
  uint64_t offset; /* draws consumed so far */
  uint64_t reserved;
} sxmc_rng_state;

/* ---------------------------------------------------------------- runtime / memory ---------- */
/* Replaces hemi's checkCuda error reporting. */
const char* sxmc_last_error(void);
const char* sxmc_version(void);

int sxmc_device_count(int* count);
int sxmc_set_device(int device);
int sxmc_get_device(int* device); /* the calling thread's current device */
/* name: at least 256 bytes. */
int sxmc_device_info(int device, char* name, int* compute_units, size_t* hbm_bytes,
                     int* lds_bytes_per_cu, int* clock_khz);
/* Host-side roctx ranges around the phases of a step (launch plan, fill, step end, SetEvalPoints, graph replay), for
 * rocprofv3 --marker-trace beside --kernel-trace.  Off by default; SXMC_ROCTX=1 in the environment turns them on too. */
int sxmc_set_tracing(int enable);
int sxmc_device_synchronize(void);
/* PCI bus id of a device ("0000:05:00.0"): tells two ranks on ONE card from two ranks on two cards of the same name. */
int sxmc_device_pci_bus_id(int device, char* out, size_t out_bytes);
/* Free and total device memory in bytes (hipMemGetInfo). */
int sxmc_mem_info(size_t* free_bytes, size_t* total_bytes);

/* Device side of hemi::Array<T>: allocation and the two copy directions
 * its accessors perform implicitly. */
int sxmc_malloc(void** d_ptr, size_t bytes);
int sxmc_free(void* d_ptr);
int sxmc_host_alloc(void** h_ptr, size_t bytes); /* pinned; hemi::Array(n, pinned=true) */
int sxmc_host_free(void* h_ptr);
int sxmc_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int sxmc_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int sxmc_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);
int sxmc_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, sxmc_stream_t s);
int sxmc_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, sxmc_stream_t s);
int sxmc_memset(void* d_dst, int value, size_t bytes);

/* cudaStreamCreate / cudaStreamSynchronize (pdfz.cpp:91, 493) */
int sxmc_stream_create(sxmc_stream_t* s);
/* A stream that does not synchronise with the legacy default stream: one per concurrent chain. */
int sxmc_stream_create_nonblocking(sxmc_stream_t* s);
int sxmc_stream_destroy(sxmc_stream_t s);
int sxmc_stream_synchronize(sxmc_stream_t s);
/* *done = 1 when everything queued on `s` has finished, 0 while work is pending (hipStreamQuery): lets a host thread
 * wait for a collective WITHOUT blocking inside the runtime, so that it can give up when a peer has failed. */
int sxmc_stream_query(sxmc_stream_t s, int* done);

/* HIP-graph capture of a launch sequence (the per-step sequence of mcmc.cpp:264-348 is the
 * same launches with the same arguments every step, so it can be recorded once and replayed).
 * Between begin and end, every sxmc_group_* evaluation and sxmc_launch_* call given `s` is recorded
 * instead of executed; `s` must be a created stream (the legacy default stream cannot be captured).
 * The group must already have evaluated once with its current configuration (descriptor uploads cannot
 * be recorded: SXMC_ERR_STATE otherwise), and a recorded graph is only valid until an evaluator of the
 * group changes (systematics, evaluation points, bindings, launch configuration).  Kernel arguments are
 * frozen at capture: device buffers are re-read at replay, scalar arguments are not.
 * Host threads: while one thread records, no other thread may allocate, free, copy through the legacy
 * default stream or synchronise the device (the ROCm runtime refuses those calls and fails the
 * recording); launching, and waiting on or copying through its own non-blocking stream, is fine.  A
 * program with one chain per host thread serialises set-up, recording and tear-down with a mutex
 * (sxmc::MCMC::exclusive, sxmc::ensemble_concurrent). */
typedef void* sxmc_graph_t;              /* hipGraphExec_t */
int sxmc_graph_begin_capture(sxmc_stream_t s);
int sxmc_graph_end_capture(sxmc_stream_t s, sxmc_graph_t* out);
/* Replays the recorded sequence `times` times on `s` (asynchronous). */
int sxmc_graph_launch(sxmc_graph_t graph, sxmc_stream_t s, int times);
int sxmc_graph_destroy(sxmc_graph_t graph);

int sxmc_event_create(sxmc_event_t* e);
int sxmc_event_destroy(sxmc_event_t e);
int sxmc_event_record(sxmc_event_t e, sxmc_stream_t s);
int sxmc_event_synchronize(sxmc_event_t e);
int sxmc_event_elapsed_ms(sxmc_event_t start, sxmc_event_t stop, float* ms);

/* ---------------------------------------------------------------- pdfz::EvalHist ------------ */
/* EvalHist::EvalHist + Eval::Eval (pdfz.cpp:57-97, 179-236).
 * samples: row-major [nsamples_floats / nfields][nfields] float32; host memory, or device
 * memory when samples_on_device != 0.  The evaluator keeps its own column-major copy
 * (the reference also copies, pdfz.cpp:197).  n_lower/n_upper/n_nbins are the lengths of the
 * three arrays so that the reference's size validation can be reproduced:
 * SXMC_ERR_INVALID for every condition under which the reference throws pdfz::Error
 * (pdfz.cpp:64-82, 189-195, 217-219) and additionally for upper[i] <= lower[i], nbins[i] < 0. */
int sxmc_hist_create(const float* samples, size_t nsamples_floats, int samples_on_device,
                     int nfields, int nobservables,
                     const double* lower, size_t n_lower,
                     const double* upper, size_t n_upper,
                     const int* nbins, size_t n_nbins,
                     unsigned dataset, sxmc_hist_t* out);
/* A second evaluator over the SAME sample table as `base` (shared, reference counted; nothing is
 * copied): own histogram, evaluation points, bindings and stream, systematics copied from `base`.
 * For several chains / fake experiments running concurrently on one GPU (BASELINE config 4: one
 * experiment per stream) without holding the MC tables more than once. */
int sxmc_hist_create_shared(sxmc_hist_t base, sxmc_hist_t* out);
/* EvalHist::~EvalHist (pdfz.cpp:239-242); also destroys the evaluator's stream, which the
 * reference leaks (pdfz.cpp:100-103). */
int sxmc_hist_destroy(sxmc_hist_t h);

/* Eval::AddSystematic (pdfz.cpp:126-174).  `pars` are indices into the parameter buffer
 * (npars of them, host memory, copied).  extra_field is used by RESOLUTION_SCALE only. */
int sxmc_hist_add_systematic(sxmc_hist_t h, int type, int obs, int extra_field,
                             int npars, const short* pars);

/* EvalHist::SetEvalPoints (pdfz.cpp:245-302).  points: host, rows of nobservables+1 floats
 * (last = dataset id).  SXMC_ERR_INVALID if npoints_floats % (nobservables+1) != 0.
 * The evaluator's previous evaluations must have finished (as for the reference: it replaces the arrays they
 * read).  A new data set that fits the buffers of the previous one -- every fake experiment after the first --
 * costs one host-to-device copy: no device-wide synchronisation and no allocation, so other chains running on
 * the same GPU are not stalled, and the groups the evaluator belongs to patch their descriptors instead of
 * re-planning their launches. */
int sxmc_hist_set_eval_points(sxmc_hist_t h, const float* points, size_t npoints_floats);

/* Eval::SetPDFValueBuffer / SetNormalizationBuffer / SetParameterBuffer (pdfz.cpp:106-124).
 * Device pointers, borrowed: they must outlive every evaluation that uses them. */
int sxmc_hist_set_pdf_value_buffer(sxmc_hist_t h, float* d_output, int offset, int stride);
int sxmc_hist_set_normalization_buffer(sxmc_hist_t h, unsigned* d_norm, int offset);
int sxmc_hist_set_parameter_buffer(sxmc_hist_t h, const double* d_params, int offset, int stride);

/* EvalHist::EvalAsync / EvalFinished (pdfz.cpp:441-495): zero, fill, (evaluate); eval_async returns before
 * completion, eval_finished when the evaluator's results are in its bound buffers.
 * TRANSPARENT BATCHING.  The reference's callers evaluate their S signals as "EvalAsync on all, then EvalFinished on
 * all" (mcmc.cpp:264-271, bench_sxmc.cpp:193-200).  Launched one by one that is S full-grid fills with S fixed
 * costs; so sxmc_hist_eval_async DEFERS: the evaluator joins the calling host thread's batch, and the batch is
 * launched as ONE group evaluation (what sxmc_group_eval_async does: one zero, ONE fill over all members' samples, one
 * lookup) -- at once when the last evaluator of a batch seen before has arrived (so the device works while the caller
 * goes on, as with the reference), otherwise at the first call that could tell the difference: sxmc_hist_eval_finished
 * of a member, any copy, launch, synchronisation or stream query through this ABI, a change of a member's bindings,
 * systematics or points, its destruction.  Results are identical (integer counters; the lookup is per evaluator).  An
 * unchanged mcmc.cpp gets the batched fill this way; the explicit group API below remains for callers that want
 * more (fused lookup + event sum, fused step end, HIP-graph replay).  A batch is per host thread: EvalFinished must
 * come from the thread that called EvalAsync (SXMC_ERR_STATE otherwise).  While a HIP graph is being recorded on the
 * calling thread, and after sxmc_set_deferred_eval(0) (or SXMC_DEFER_EVAL=0 in the environment), evaluations are
 * launched as asked: each at once, on its evaluator's own stream. */
int sxmc_hist_eval_async(sxmc_hist_t h, int do_eval_pdf);
int sxmc_hist_eval_finished(sxmc_hist_t h);
/* 0: sxmc_hist_eval_async launches at once on the evaluator's own stream (S separate launch sequences for S
 * evaluators: the reference's literal behaviour; measurement / tests).  Default 1.  Process-wide. */
int sxmc_set_deferred_eval(int enable);
/* LAZY EvalFinished (default 1; SXMC_LAZY_FINISH=0 in the environment changes the default).  A batch of two or more
 * deferred evaluations is launched on the legacy default stream -- where the reference's caller launches its NLL
 * kernels (mcmc.cpp:314-348) and with which every blocking stream orders.  What the caller does next through this ABI
 * on that stream or on a blocking stream (its kernels, blocking copies to the host, the next evaluation) is therefore
 * ordered after the batch ON THE DEVICE, and sxmc_hist_eval_finished of such a batch does not stop the host: the wait
 * is carried out by the first call that could tell the difference -- one that names a stream created non-blocking,
 * or that synchronises or queries a stream or the device.  The host then runs ahead of the device as a caller of the
 * group API does, instead of idling the device once per MCMC step.  NOT covered: device work the caller issues
 * outside this ABI, on a non-blocking stream of its own, straight after EvalFinished -- such a caller sets 0, and
 * EvalFinished blocks until the evaluation has finished, as in the reference. */
int sxmc_set_lazy_finish(int enable);
/* Process-wide counters of the batching above: group launches made for deferred evaluations, and the evaluations
 * (evaluator x EvalAsync) they carried -- 1 and S per MCMC step of an unchanged mcmc.cpp.  For tests and logs. */
int sxmc_deferred_eval_stats(unsigned long long* launches, unsigned long long* evaluations);

/* Introspection used by CreateHistogram / GetSamples / tests (pdfz.cpp:498-594, pdfz.h:542-556). */
int sxmc_hist_total_nbins(sxmc_hist_t h, int* total_nbins);
int sxmc_hist_bin_volume(sxmc_hist_t h, double* bin_volume);
int sxmc_hist_nsamples(sxmc_hist_t h, size_t* nsamples);
int sxmc_hist_npoints(sxmc_hist_t h, size_t* npoints);
int sxmc_hist_get_bins(sxmc_hist_t h, unsigned* h_bins, size_t n);       /* after eval_finished */
int sxmc_hist_get_read_bins(sxmc_hist_t h, int* h_read_bins, size_t n);
/* GetSamples: rows of nobservables+1 floats (observables, dataset id); n = nsamples*(nobs+1). */
int sxmc_hist_get_samples(sxmc_hist_t h, float* h_out, size_t n);
/* EvalHist::RandomSample's sampling step (pdfz.cpp:817-922; 1-3 observables, as there) on the device: nobserved
 * events drawn from the histogram of the evaluator's LAST evaluation (evaluate with do_eval_pdf = 0 first, what
 * CreateHistogram does) -- a bin with probability proportional to its content, a point uniform inside it (TH1::
 * GetRandom's rule) -- redrawn while outside [lowers, uppers] when those are given (pdfz.cpp:853-857).  The
 * histogram stays in HBM (the reference, and round 1 of this library, copied every signal's histogram to the
 * host per fake experiment); only the events come back: h_events receives nobserved rows of nobservables + 1
 * floats (last = the evaluator's dataset id), the layout sxmc_hist_set_eval_points takes.  Counter-based
 * generator (Philox4x32-10 keyed by `seed`): the same seed gives the same events.  The Poisson fluctuation of
 * the event count (pdfz.cpp:836-841) is the caller's, on the host: it decides array sizes. */
int sxmc_hist_random_sample(sxmc_hist_t h, size_t nobserved, unsigned long long seed, const float* lowers,
                            const float* uppers, float* h_events);
int sxmc_hist_get_stream(sxmc_hist_t h, sxmc_stream_t* s);

/* EvalHist's `optimize` constructor argument (pdfz.cpp:188; default 1).  The reference's evaluator runs its launch-shape
 * trials (Optimize, pdfz.cpp:622-814) inside its first EvalAsync once it has evaluation points (:441-448), never while
 * CreateHistogram evaluates (:503-504).  Here the trials are those of the BATCH the library forms behind the
 * per-evaluator calls (sxmc_group_optimize: a few timed fills pick lanes per CU, teams and codes on the box it runs on),
 * at the batch's first lookup evaluation, when EVERY member has optimize on; 0 pins the analytic launch shape. */
int sxmc_hist_set_optimize(sxmc_hist_t h, int enable);
/* EvalHist::Optimize (pdfz.cpp:622-628) called by hand: the trials run (again) at the next lookup evaluation of every
 * batch `h` is part of.  Nothing without evaluation points, as in the reference. */
int sxmc_hist_optimize(sxmc_hist_t h);
/* The launch plan of the batch the evaluator's evaluations go into (one line per fill launch, as
 * sxmc_group_launch_info) and a last line "tuned=<0|1> trial_launches=<n>".  Empty before the first evaluation. */
int sxmc_hist_launch_info(sxmc_hist_t h, char* out, size_t n);
/* Replaces EvalHist::Optimize/OptimizeBin/OptimizeEval (pdfz.cpp:622-814): the launch shape
 * is sized analytically from the device; 0 keeps the default.  threads: a multiple of 64 up to 1024. */
int sxmc_hist_set_launch_config(sxmc_hist_t h, int bin_threads, int bin_blocks_per_cu);

/* ---------------------------------------------------------------- evaluator group ----------- */
/* The "EvalAsync on all signals, then EvalFinished on all" of mcmc.cpp:264-271 and
 * bench_sxmc.cpp:193-200 as ONE batched launch sequence (zero, fill, evaluate) over all
 * members.  Members are borrowed; their bindings are re-read whenever they change. */
int sxmc_group_create(const sxmc_hist_t* members, int nmembers, sxmc_group_t* out);
int sxmc_group_destroy(sxmc_group_t g);
int sxmc_group_set_launch_config(sxmc_group_t g, int bin_threads, int bin_blocks_per_cu);
/* EvalHist::Optimize / OptimizeBin (pdfz.cpp:622-727) for the batched launch: times the fill with a few
 * lane counts per CU on stream `s` and keeps the fastest (the analytic default is within a few per cent;
 * which count wins differs from one box to the next).  Only acts when every member is a pure stream and no
 * launch configuration was set by hand; results never depend on the shape.  The members' histograms and
 * normalisation slots hold counts of the trial runs afterwards (the next evaluation zeroes them as usual).
 * Where the plan streams codes (sxmc_group_set_codes left at its default) the candidates run to 1024 lanes and the
 * codes are then timed against the float columns at the parameters bound now: the float stream is taken if it wins
 * by 3 % (whether codes pay is otherwise estimated from the binning when the table is laid out).
 * *chosen_threads (optional): the lane count kept, 0 when nothing was tried. */
int sxmc_group_optimize(sxmc_group_t g, sxmc_stream_t s, int* chosen_threads);
/* How the fill kernel's work is cut over workgroups: 0 = automatic, 1 = sliced (each workgroup one
 * contiguous slice of the concatenated members), 2 = interleaved (each member's workgroups stride
 * through it chunk by chunk, like a grid-stride copy). */
int sxmc_group_set_partition(sxmc_group_t g, int mode);
/* Interleaved partition of a BUCKETED table (rows sorted by bin): a member's workgroups split into `teams` teams, each
 * over a contiguous part of the member (so a workgroup sees a fraction of the bins and flushes as few).  0 = default
 * (one team); sxmc_group_optimize tries three on the box it runs on.  Results never depend on it. */
int sxmc_group_set_partition_teams(sxmc_group_t g, int teams);
/* Sparse counting (default on): a histogram too large for LDS (more than 40 832 bins) costs one scattered
 * HBM atomic per sample plus zeroing the whole array, yet an evaluation for lookup (do_eval_pdf != 0, the
 * fused evaluations, the MCMC step) reads it only at the data events' bins.  Such evaluations count into one
 * counter per distinct event bin instead; lut, normalisations and NLL are identical.  The dense histogram
 * is then NOT filled: sxmc_hist_get_bins returns SXMC_ERR_STATE until an evaluation with do_eval_pdf = 0
 * (what CreateHistogram does, pdfz.cpp:503-506) has run. */
int sxmc_group_set_sparse(sxmc_group_t g, int enable);
/* Pre-binning (default on): an observable that no systematic writes has the same bin index at every
 * evaluation, so for launches that run a static program the evaluators build, once, a 1- or 2-byte column
 * with the partial flat index of those observables and the fill streams it instead of their float
 * columns.  Results are identical; sxmc_group_algorithmic_bytes counts the bytes actually needed. */
int sxmc_group_set_prebinning(sxmc_group_t g, int enable);
/* Bucketing (default on; takes precedence over pre-binning where it applies): the same fact taken further.
 * Where the fill runs a static program and at least one observable is written by a systematic and at least one
 * is not, the evaluator keeps, once, a second copy of its table with the samples GROUPED by their bin indices
 * in the untouched observables (stable radix sort), holding only the columns that change from evaluation to
 * evaluation (the written observables + the truth fields referenced); samples outside the domain in an
 * untouched observable can never be counted and are left out.  The fill then solves the lower-dimensional
 * problem of the written observables and reads one bin offset per 256-sample granule for the rest: BASELINE
 * config 3 streams 12 bytes per sample instead of 16 (13 with pre-binning), config 5 12 instead of 24 (14).
 * Counters are integers, so visiting the samples in another order changes no count: histograms, norms, lookup
 * tables and NLL are bit-identical (the parity tests compare on/off and against the CPU restatement).  GetSamples keeps the
 * caller's row order (it reads the original table).  Skipped for a table whose granule padding would outweigh
 * the saving (few samples, many buckets).  sxmc_group_algorithmic_bytes counts the bytes actually needed. */
int sxmc_group_set_bucketing(sxmc_group_t g, int enable);
/* Ordering (default on; needs bucketing and a histogram that fits LDS): bucketing taken to ONE observable that is
 * written -- but only by one-coefficient shift / scale / cos-theta-scale systematics (pdfz.cpp:316-325) and read
 * by nothing else.  For the evaluation's parameters each of those is a monotone map of the sample's value (IEEE
 * addition of / multiplication by a constant rounds monotonically), and so is the binning after it
 * (pdfz.cpp:388-398).  The bucketed copy keeps the rows of every bucket sorted by the raw value; a 256-sample
 * granule whose first and last sample land in the same bin (or both below / both above the domain) therefore has
 * ALL its samples there.  The fill works that out per evaluation from two floats per granule, with the per-sample
 * arithmetic, and does not read the observable's column at all; only the granules that straddle a bin edge (at
 * most nbins + 1 per bucket) take the per-sample path over the column, granules outside the domain are skipped.
 * Any NaN (sample value or coefficient) sends a granule down the per-sample path.  BASELINE config 3 streams
 * 8 bytes per sample instead of 12; a 1-D histogram with one shift (bench_sxmc pdfz) streams 12 bytes per 256
 * samples.  Histograms, norms and NLL stay bit-identical (parity tests: on/off, the CPU restatement, samples
 * placed within ulps of the bin edges).  Of several eligible observables the one with the fewest bins is taken.
 * enable = 1 (default): where it pays -- the table must have at least twice as many granules as can straddle an edge
 * (buckets x (nbins + 1)); BASELINE config 5 with its 200 bins of r and 61 granules per bucket does not qualify and
 * keeps the unordered bucketed table.  enable = 2: wherever it applies (tests).  Histograms beyond LDS capacity: the
 * sparse counting over runs and the dense evaluation have ordered forms too (run-time compiled). */
int sxmc_group_set_ordering(sxmc_group_t g, int enable);
/* Codes (default on; needs an ordered table with its histogram in LDS, 2 to 4 streamed fields and only one-coefficient
 * systematics on them).  shift / scale / cos-theta scale / resolution scale with one coefficient are affine maps of
 * the sample's fields (pdfz.cpp:316-330), so for the evaluation's parameters the bin coordinate of a written
 * observable is ONE affine function of the fields, which the reference evaluates in double, operation by operation,
 * to within a few units in the last place of it.  The ordered copy therefore keeps each streamed field a second time
 * as a 16-bit code inside a window (value = base + (code + 1/2) step, off by at most step / 2: checked when the
 * table is laid out), two fields to a word, and the fill streams THOSE: it composes the program into
 * single-precision coefficients over the codes per evaluation, together with an error bound that covers the half
 * step, its own single-precision arithmetic and the reference's double roundings, and bins every sample whose bin
 * coordinate lies further than that bound from the nearest bin edge straight from its codes -- that bin IS the
 * reference's.  The few samples in 10^4 that lie closer go into a queue in LDS and are binned at the end of the
 * stream from their float values with the reference's arithmetic, as are rows outside the windows and the granules
 * that straddle an edge of the ordered observable.  Parameters that are not finite, or so large that the bound
 * reaches an eighth of a bin, switch the codes off for that evaluation (the float columns are streamed).  BASELINE
 * config 3 streams 4 bytes per sample instead of 8.  Histograms, norms and NLL stay bit-identical (parity tests:
 * on / off, the CPU restatement, samples within ulps of the transformed edges, rows outside the windows, queues
 * that overflow).  enable = 0: stream the float columns; 1: use the codes where they apply; -1: the library's
 * default (on unless the environment says SXMC_CODES=0). */
int sxmc_group_set_codes(sxmc_group_t g, int enable);
/* Boxes (default: where they pay; needs codes on, the histogram in LDS, exactly two written observables).  The ordered form
 * makes an observable written by ONE-field systematics a per-granule constant.  An observable that is resolution-scaled
 * (pdfz.cpp:326-329: x += p * (x - t), t a field nothing writes) depends on two fields and has no order -- but every
 * IEEE operation of its program is monotone in each operand, so the reference's own operations applied to the CORNERS
 * of a box [xmin, xmax] x [tmin, tmax] (the corner chosen by the coefficient's sign) bound the result of every row
 * inside the box from both sides, exactly: interval arithmetic whose endpoints round the way the values between them do.
 * The boxed copy sorts the rows of a bucket by (stratum of x - t, x), keeps the box of every 256-row granule (16 bytes), and
 * the fill (fill_boxed_kernel) works out per evaluation, one lane per granule, whether both ends of the box's image land
 * in one bin: then so does every row, and the observable costs nothing per sample.  Granules whose image straddles an
 * edge (a few per cent at BASELINE config 3, more for large resolution parameters) are binned from their float columns
 * with the reference's arithmetic.  The OTHER written observable is streamed as one 16-bit code per row (the codes above,
 * one field): BASELINE config 3 streams 2 bytes per sample instead of 4.  Any value or coefficient that is not finite
 * sends its granule (or the whole evaluation) to the float columns.  Histograms, norms and NLL stay bit-identical
 * (tests/test_gpu_boxed.py: against the ordered form, the float stream and the CPU restatement).  enable = -1 (default):
 * where it pays (at least four granules per bin of the boxed observable, stratum and bucket), TOGETHER WITH the ordered plan
 * (below); 1: wherever it applies, the boxed plan alone (tests); 0: never (the ordered / bucketed forms).  Lockstep sets
 * and the look-ahead pass use the ordered form. */
int sxmc_group_set_boxes(sxmc_group_t g, int enable);
/* With enable = -1 the boxed plan comes with an ORDERED TWIN.  How many boxes straddle an edge depends on the parameters of
 * the evaluation (the image of a box is |dx'/dx| dx + |dx'/dt| dt wide: it grows with the resolution parameter): at BASELINE
 * config 3 the boxed fill takes 65-67 us at a resolution parameter of 0, 76 at 0.05, 84 at 0.07, 131 at 0.2, the ordered
 * one 82-85 whatever the parameters (one box, profiles/r05_boxed_crossover.log).  Both plans stay resident -- the boxed tables with their partition
 * and launch shape, the ordered ones with theirs -- and every fill launches ONE of them: the ordered one until told
 * otherwise, so a caller that never asks runs the ordered form.
 *   sxmc_group_adapt_fill_form: waits for the group's stream, reads the parameters the evaluators are bound to back from
 *   the device, runs the reference's operations on the corners of a box of the tables' mean extents and takes the boxed
 *   form while its image is narrower than the limit (sxmc_group_set_box_limit, bins of the boxed observable; default
 *   0.12; back to boxed below 0.8 of it), the ordered form beyond.  *form: 1 boxed, 2 ordered, 0 the plan has one form
 *   only; *changed: it differs from the form of the launches so far -- recorded graphs of the group's steps keep
 *   replaying the OLD form (both stay valid) until they are recorded again.  The walks of this repository
 *   (sxmc::MCMC, sxmc_amd/mcmc.py) ask at set-up and after every flush of their jump buffer.  Not while a graph is being
 *   recorded.  Results do not depend on the form.
 *   sxmc_group_set_fill_form: 1 or 2 by hand (tests, measurements); sxmc_group_fill_form: the current one. */
int sxmc_group_set_box_limit(sxmc_group_t g, double bins);
int sxmc_group_adapt_fill_form(sxmc_group_t g, int* form, int* changed);
int sxmc_group_set_fill_form(sxmc_group_t g, int form);
int sxmc_group_fill_form(sxmc_group_t g, int* form);
/* What the codes of the group's current plan amount to: members whose fill streams codes, rows they hold, rows marked
 * "ask the exact columns" (outside a window) and rows marked "never counted" (not finite, or granule padding). */
int sxmc_group_codes_info(sxmc_group_t g, int* members, unsigned long long* rows, unsigned long long* exact_rows,
                          unsigned long long* never_rows);
/* The windows the codes of member `member` were cut from: field m of the streamed ones has code
 * floor((x - base[m]) / step[m]); *nfields = how many (0: the member's fill does not stream codes).  base, step: room for
 * SXMC_MAX_QSLOTS (4) doubles each.  For tests that place samples relative to the code cells. */
int sxmc_group_codes_windows(sxmc_group_t g, int member, int* nfields, double* base, double* step);
/* Cap on the per-workgroup queues of ambiguous rows of a fill over codes: 2^log2_entries entries, 9 .. 11; 0 (default):
 * as many as fit beside the histogram.  Smaller queues fill up and are emptied in the middle of the stream, and whole
 * granules are handed to the float columns; the RESULTS do not depend on it (tests/test_gpu_codes.py). */
int sxmc_group_set_codes_queue_log(sxmc_group_t g, int log2_entries);
/* Run-time kernels (default on).  The fill is fastest as straight-line code with the program of systematics
 * (apply_systematic, pdfz.cpp:306-331: which systematic writes which column, in which order, with how many
 * polynomial coefficients) fixed at compile time.  The library carries such kernels for a handful of programs;
 * for any other program of at most 8 systematics and 16 coefficients in all it compiles the same kernel template
 * with hiprtc when the group's launch plan is built (about half a second per program and process, cached) and
 * launches it like a built-in one -- bucketing and sparse counting included.  Off, or when hiprtc fails, such
 * programs run the kernel that decodes the program at run time (same results; sxmc_group_launch_info says why).
 * Results are bit-identical either way (same source, same flags: -ffp-contract=off). */
int sxmc_group_set_runtime_kernels(sxmc_group_t g, int enable);
/* One text line per fill launch of the group's current plan: members, shape, where the histogram lives, how the
 * program runs (builtin / runtime / decoded / generic), how the table is streamed (rows / prebinned / bucketed),
 * launch shape.  For logs and tests. */
int sxmc_group_launch_info(sxmc_group_t g, char* out, size_t n);
/* 1 (default): sxmc_group_eval_nll_async / sxmc_group_mcmc_step_async write the lookup table
 * (lut[j * E + i], the array eval_pdf produces at pdfz.cpp:411-436) as they consume it.  0: the table is
 * an intermediate nobody reads (the MCMC loop, mcmc.cpp:264-348), so it is not written, and the event
 * sum of nll_kernels.cpp:89-116 runs over the DISTINCT tuples of event bins, each log term weighted by
 * the number of events sharing the tuple: sum_i log(s_i) = sum_k n_k log(s_k).  Same value up to
 * rounding (the terms are added in another order); the work no longer grows with the number of
 * events.  sxmc_group_eval_async always writes the table. */
int sxmc_group_set_lut_output(sxmc_group_t g, int enable);
int sxmc_group_eval_async(sxmc_group_t g, int do_eval_pdf, sxmc_stream_t s);
/* As sxmc_group_eval_async(g, 1, s) followed by nll_event_chunks (nll_kernels.cpp:89-116) over
 * the members' lookup table, with the table lookup and the event sum fused in one kernel: the
 * lut is still written (it is the API contract, mcmc.cpp:232-236; unless sxmc_group_set_lut_output
 * switched that off) but not re-read.  Requires
 * all members to share the same number of evaluation points.  Writes one partial sum per
 * block: d_sums[0 .. *npartial_out).  d_sums must hold at least 1024 doubles. */
int sxmc_group_eval_nll_async(sxmc_group_t g, sxmc_stream_t s,
                              const double* d_pars, const double* d_nexpected,
                              const unsigned* d_n_mc, const short* d_source_id,
                              const unsigned* d_norms, double* d_sums, int* npartial_out);
/* EXPERIMENTAL (measured ~3% slower per step than sxmc_group_eval_nll_async + finish_nll_jump_pick_combo at
 * E = 1e5: every workgroup pays an agent-scope release and a ticket on one counter).
 * One whole MCMC step (mcmc.cpp:264-271 + 314-348) as three launches: zero, fill of all members, and
 * one kernel doing lookup + nll_event_chunks + finish_nll_jump_pick_combo (the workgroup that finishes
 * its event partial sum last also runs the step end).  The NLL is evaluated at d_v_proposed, which is
 * also the members' parameter buffer in an MCMC walk (mcmc.cpp:241).  Arguments as
 * finish_nll_jump_pick_combo (nll_kernels.h:190-207) without the partial-sum buffer. */
int sxmc_group_mcmc_step_async(sxmc_group_t g, sxmc_stream_t s, const double* d_means, const double* d_sigmas,
                               sxmc_rng_state* d_rng, double* d_nll_current, double* d_nll_proposed,
                               double* d_v_current, double* d_v_proposed, int* d_accepted, int* d_counter,
                               float* d_jump_buffer, int nparameters, size_t nsources,
                               const float* d_jump_width, const double* d_nexpected, const unsigned* d_n_mc,
                               const short* d_source_id, const unsigned* d_norms, int debug_mode);
/* finish_nll_jump_pick_combo (nll_kernels.cpp:230-271; arguments as sxmc_launch_finish_nll_jump_pick_combo,
 * one workgroup of 128) launched TOGETHER with the zeroing the group's next evaluation would start with:
 * that evaluation then skips its zero launch (3 launches per MCMC step instead of 4).  For a walk that does
 * not look at them between steps: the group's histograms and the members' normalisation slots are CLEARED
 * when this returns (sxmc_hist_get_bins gives SXMC_ERR_STATE until an evaluation with do_eval_pdf = 0).
 * Must follow an evaluation of the group on the same stream; d_norms is the array the members'
 * normalisations are bound into. */
int sxmc_group_finish_step_async(sxmc_group_t g, sxmc_stream_t s, size_t npartial_sums, const double* d_sums,
                                 const double* d_means, const double* d_sigmas, sxmc_rng_state* d_rng,
                                 double* d_nll_current, double* d_nll_proposed, double* d_v_current,
                                 double* d_v_proposed, int* d_accepted, int* d_counter, float* d_jump_buffer,
                                 int nparameters, size_t nsources, const float* d_jump_width,
                                 const double* d_nexpected, const unsigned* d_n_mc, const short* d_source_id,
                                 const unsigned* d_norms, int debug_mode);
/* One whole MCMC step (mcmc.cpp:264-271 + 314-348) for a walk that reads neither histograms, normalisations nor
 * the lookup table between steps: the fill of all members, then lookup + nll_event_chunks (nll_kernels.cpp:89-116)
 * + nll_event_reduce + nll_total + finish_nll_jump_pick_combo (:230-271) + the clearing the next evaluation
 * would start with (so that one skips its zero launch).  The step end runs as ONE workgroup in ONE launch
 * (2 launches per step) where it is small -- at most 256 row x member look-ups and 65 536 counters: BASELINE
 * config 1 -- and as two launches otherwise (lookup + event sum over many workgroups,
 * then step end beside the clearing: what sxmc_group_eval_nll_async + sxmc_group_finish_step_async do).  A row
 * costs one double division per member and a log, so beyond a few hundred of them one CU is slower than the
 * extra launch (measured: config 3, ~8000 x 12, 59 us against 15.6 us; bench_pdfz, 1000 x 1, 8 against 7;
 * config 1, 10 x 2, 7.0 against 7.6).
 * The NLL is evaluated at d_v_proposed, which is also the members' parameter buffer in a walk (mcmc.cpp:241).
 * Histograms and normalisations are CLEARED when this returns, as after sxmc_group_finish_step_async; the
 * lookup table is written only when sxmc_group_set_lut_output is on.  The one-workgroup form adds the log terms
 * in another order than the many-workgroup form (NLL equal to ~1e-15 relative). */
int sxmc_group_step_async(sxmc_group_t g, sxmc_stream_t s, const double* d_means, const double* d_sigmas,
                          sxmc_rng_state* d_rng, double* d_nll_current, double* d_nll_proposed,
                          double* d_v_current, double* d_v_proposed, int* d_accepted, int* d_counter,
                          float* d_jump_buffer, int nparameters, size_t nsources,
                          const float* d_jump_width, const double* d_nexpected, const unsigned* d_n_mc,
                          const short* d_source_id, const unsigned* d_norms, int debug_mode);
/* LOCKSTEP CHAINS.  Several chains over the SAME sample tables -- the fake experiments in flight on one GPU,
 * BASELINE config 4's per-GPU shape: evaluators made with sxmc_hist_create_shared, same systematics, each chain
 * with its own group, parameters, data and state -- evaluate different parameter vectors on identical samples.
 * Stepped one by one each evaluation streams the tables again while the vector units idle under the stream
 * (config 3: 175 us of stream against 122 us of arithmetic).  A multigroup steps 2-4 such chains TOGETHER: one
 * fill pass streams the tables once and bins every sample under each chain's parameters into that chain's LDS
 * histogram (bytes per evaluation divided by the number of chains; per chain exactly the operations of its own
 * fill, so its counts are bit-identical), then every chain's own step end runs as in sxmc_group_step_async.
 * Chains advanced this way walk exactly the chains they walk alone (tests).  The kernel is compiled through
 * hiprtc for the chains' program.  SXMC_ERR_STATE (with the reason) when the chains cannot share a pass --
 * different tables, systematics or launch configuration, histograms beyond LDS or not fitting LDS together, a
 * run-time decoded program: step them separately then.  args: one sxmc_step_args per chain, in group order. */
typedef struct {
  const double* d_means;
  const double* d_sigmas;
  sxmc_rng_state* d_rng;
  double* d_nll_current;
  double* d_nll_proposed;
  double* d_v_current;
  double* d_v_proposed;
  int* d_accepted;
  int* d_counter;
  float* d_jump_buffer;
  int nparameters;
  size_t nsources;
  const float* d_jump_width;
  const double* d_nexpected;
  const unsigned* d_n_mc;
  const short* d_source_id;
  const unsigned* d_norms;
  int debug_mode;
} sxmc_step_args; /* the arguments of finish_nll_jump_pick_combo (nll_kernels.h:190-207) */
typedef struct sxmc_multigroup* sxmc_multigroup_t;
/* groups: borrowed; they must outlive the multigroup (destroy it first). */
int sxmc_multigroup_create(const sxmc_group_t* groups, int ngroups, sxmc_multigroup_t* out);
int sxmc_multigroup_destroy(sxmc_multigroup_t mg);
int sxmc_multigroup_step_async(sxmc_multigroup_t mg, sxmc_stream_t s, const sxmc_step_args* args);
/* The chains' step ends (lookup + event sum; finish_nll_jump_pick_combo + clearing) share two launches for the whole
 * set -- chain = blockIdx.y, every workgroup doing what it does for a chain stepped alone, so the chains are the
 * same bit for bit -- instead of two launches per chain.  Default 1; 0: chain by chain (measurement / tests; in the
 * measurement build SXMC_JOINT_STEP_END=0 in the environment makes that the default of multigroups created afterwards,
 * for A/B runs of whole programs: profiles/r03_lockstep_step_ends.log). */
int sxmc_multigroup_set_joint_step_end(sxmc_multigroup_t mg, int enable);
/* LOOK-AHEAD WALK: one chain, two likelihood evaluations per pass over the tables.  A Metropolis step that rejects
 * (jump_decider, nll_kernels.cpp:56-86) leaves the chain where it was, and the next proposal -- current vector +
 * jump width x the next deviates (pick_new_vector, :30-53) -- is then known BEFORE the step is decided.  A
 * multigroup of exactly two groups over the same tables -- groups[0]'s evaluators bound to the chain's proposal
 * (args->d_v_proposed, args->d_norms), groups[1]'s to the look-ahead vector (d_v_lookahead, d_norms_lookahead) --
 * evaluates both in ONE fill pass (the lockstep kernel); the step end decides the step from the first and, if it
 * rejected, the FOLLOWING step from the second at once, then writes the next proposal and the next look-ahead
 * vector.  A pass advances the chain by 1 + P(reject) steps on average for about 1.2 x the time of a
 * single evaluation.  The chain is the sequential one bit for bit: every row of the jump buffer, the counters, the
 * generator states (pre-fetching: Brockwell, J. Comput. Graph. Stat. 15 (2006)).  d_cap (device, optional): the
 * value of *args->d_counter at which the walk stops -- a pass takes its second step only below it and does
 * nothing at or beyond it, so a caller that needs exactly n steps launches passes until the counter says n.
 * Needs the lookup table off (sxmc_group_set_lut_output(g, 0)), histograms in LDS and at most 256 parameters.
 * sxmc_lookahead_begin: the first look-ahead vector of a walk, from the state sxmc_launch_pick_new_vector left. */
int sxmc_multigroup_lookahead_step_async(sxmc_multigroup_t mg, sxmc_stream_t s, const sxmc_step_args* args,
                                         double* d_v_lookahead, const unsigned* d_norms_lookahead, const int* d_cap);
/* *ok = 1 when a walk over this group (evaluation points set, buffers bound) can be taken by the look-ahead pass and
 * stay the sequential chain bit for bit.  0: histograms beyond LDS, a materialised lookup table, members with
 * different points -- or a problem so small (at most 256 look-ups per step) that the sequential step ends in the
 * one-workgroup form, whose event sum is partitioned differently: walk sequentially then. */
int sxmc_group_lookahead_supported(sxmc_group_t g, int* ok);
int sxmc_lookahead_begin(sxmc_stream_t s, int nparameters, const sxmc_rng_state* d_rng, const float* d_jump_width,
                         const double* d_v_current, double* d_v_lookahead);
/* Kernels launched by the last sxmc_group_step_async (2 or 3, see there; +1 when the histograms had to be
 * zeroed first). */
int sxmc_group_last_step_launches(sxmc_group_t g, int* launches);
/* 0: sxmc_group_step_async always takes its three-launch route (measurement / tests).  Default 1. */
int sxmc_group_set_tail_kernel(sxmc_group_t g, int enable);
/* The step end of sxmc_group_step_async as ONE cooperative launch (default 1; SXMC_COOP_STEP_END=0 in the environment
 * changes the default): where the event sum is at most 128 workgroups of 128 rows (up to 16 384 rows: BASELINE
 * configs 2 and 3 with event classes), the look-ups + event sum (nll_event_chunks, nll_kernels.cpp:89-116), the step
 * end (finish_nll_jump_pick_combo, :230-271) and the clearing for the next evaluation run in one kernel: every
 * workgroup of the event sum hands its partial sum to a finisher workgroup through a slot of its own (one device-scope
 * store; no fences, no counters); the finisher -- which has meanwhile done everything of the step end that does not
 * need the sums -- polls the slots, finishes the step and empties them, which tells the other workgroups that every
 * look-up is done and the histograms may be cleared.  Same partial sums, same order: the chain is the one the
 * separate launches walk, bit for bit.  2 launches per step instead of 3: 14.2 us against 9.5 + 6.8 at BASELINE
 * config 3, 9.3 against 6.1 + 5.3 at config 2 (+10 % evaluations per second).  Every wait inside the kernel is bounded
 * (~0.3 s): a lane that gives up counts a timeout, which sxmc_group_step_end_timeouts reports (0 in any healthy run;
 * the results of a step that timed out are not valid). */
int sxmc_group_set_cooperative_step_end(sxmc_group_t g, int enable);
/* THE WHOLE STEP IN ONE LAUNCH (default 0: built, bit-identical and MEASURED SLOWER than the fill followed by the
 * cooperative step end -- 6 640-6 970 against 6 870-7 260 evaluations/s at BASELINE config 3, DESIGN.md section 4;
 * SXMC_FUSED_STEP=1 in the environment changes the default).  Where the step
 * end is cooperative (above), the plan is a single fill launch with a built-in kernel that has the form (the ordered
 * programs -- BASELINE config 3 --, the empty program over a pre-binned column -- config 2) and at most 256
 * parameters and signals, sxmc_group_step_async launches the fill's workgroups AND the step end's finisher and workers
 * as one grid: the roles start on CUs the fill's first workgroups have left, wait for the fill's workgroups to count
 * themselves done (each after its flush has landed), acquire, and go on as the cooperative step end.  Waits go from
 * later blocks to earlier ones (which the dispatcher has started before them) except the finisher's for its workers;
 * all are bounded and counted like the step end's.  Same arithmetic, same partial sums: the chain is the same bit for
 * bit.  1 launch per step.  A step whose fill is being timed (sxmc_group_profile) is launched unfused, so that the
 * fill's own duration stays measurable. */
int sxmc_group_set_fused_step(sxmc_group_t g, int enable);
/* s: the stream the count is read through (the chain's own; NULL = a blocking copy through the legacy stream, which
 * must not happen while another host thread records a graph). */
int sxmc_group_step_end_timeouts(sxmc_group_t g, sxmc_stream_t s, unsigned* timeouts);
/* Compiles (does not load or run) the fill kernel the library would specialise at run time for a program of
 * systematics -- see sxmc_group_set_runtime_kernels.  Needs no GPU: a build check, and the test hook of the
 * run-time compilation.  ops[i] = type | obs_slot << 4 | extra_slot << 8 | npars << 12 (npars 0 = one coefficient);
 * pre_width 0 or 3 (bucketed table); sparse_runs != 0: the kernel for histograms beyond LDS over a bucketed
 * table walked in runs.  *code_bytes (optional): size of the gfx950 code object. */
int sxmc_rtc_compile_check(int nobs, int nslot, int lds_hist, int pre_width, int sparse_runs, const unsigned* ops,
                           int nops, size_t* code_bytes);
/* The same for the lockstep-chains kernel (sxmc_multigroup_step_async), which exists only as a run-time kernel. */
int sxmc_rtc_compile_check_lockstep(int nobs, int nslot, int pre_width, int nchains, const unsigned* ops, int nops,
                                    size_t* code_bytes);
int sxmc_group_synchronize(sxmc_group_t g);
/* Live timing of the dominant kernel (the histogram fill) with HIP events on the stream it is
 * launched on.  enable!=0 starts recording (at most `capacity` launches are kept). */
int sxmc_group_profile(sxmc_group_t g, int enable, int capacity);
int sxmc_group_profile_read(sxmc_group_t g, double* fill_ms_total, int* nlaunches);
/* Algorithmic bytes of one group evaluation, SURVEY 8(d):
 * sum_j 4*N_j*U_j  (fill_read) ; sum_j 4*B_j*w (hist) ; 16*E*S (event) */
int sxmc_group_algorithmic_bytes(sxmc_group_t g, double* fill_read, double* hist, double* event);

/* ---------------------------------------------------------------- NLL / MCMC step kernels --- */
/* Launch points with the argument lists of nll_kernels.h:60-207, preceded by the
 * (grid, block, stream) triple the caller passes to HEMI_KERNEL_LAUNCH (mcmc.cpp:252, 314,
 * 326, 396, 404, 409).  All array arguments are device pointers. */

/* init_device_rngs (nll_kernels.cpp:18-27): state[i] = (seed, subsequence i, offset 0) */
int sxmc_launch_init_device_rngs(int grid, int block, sxmc_stream_t s, int nthreads,
                                 unsigned long long seed, sxmc_rng_state* d_state);
/* pick_new_vector (nll_kernels.cpp:191-197) */
int sxmc_launch_pick_new_vector(int grid, int block, sxmc_stream_t s, int nthreads,
                                sxmc_rng_state* d_rng, const float* d_jump_width,
                                const double* d_current_vector, double* d_proposed_vector);
/* jump_decider (nll_kernels.cpp:200-206) */
int sxmc_launch_jump_decider(int grid, int block, sxmc_stream_t s, sxmc_rng_state* d_rng,
                             double* d_nll_current, const double* d_nll_proposed,
                             double* d_v_current, const double* d_v_proposed,
                             unsigned nparameters, int* d_accepted, int* d_counter,
                             float* d_jump_buffer);
/* nll_event_chunks (nll_kernels.cpp:89-116): thread t of grid*block writes d_sums[t] */
int sxmc_launch_nll_event_chunks(int grid, int block, sxmc_stream_t s, const float* d_lut,
                                 const double* d_pars, size_t ne, size_t ns,
                                 const double* d_nexpected, const unsigned* d_n_mc,
                                 const short* d_source_id, const unsigned* d_norms,
                                 double* d_sums);
/* nll_event_reduce (nll_kernels.cpp:209-212); grid must be 1 */
int sxmc_launch_nll_event_reduce(int grid, int block, sxmc_stream_t s, size_t nthreads,
                                 const double* d_sums, double* d_total_sum);
/* nll_total (nll_kernels.cpp:215-227) */
int sxmc_launch_nll_total(int grid, int block, sxmc_stream_t s, size_t nparameters,
                          const double* d_pars, size_t nsignals, size_t nsources,
                          const double* d_means, const double* d_sigmas,
                          const double* d_events_total, const double* d_nexpected,
                          const unsigned* d_n_mc, const short* d_source_id,
                          const unsigned* d_norms, double* d_nll);
/* finish_nll_jump_pick_combo (nll_kernels.cpp:230-271); grid must be 1 */
int sxmc_launch_finish_nll_jump_pick_combo(int grid, int block, sxmc_stream_t s,
                                           size_t npartial_sums, const double* d_sums,
                                           size_t nsignals, size_t nsources,
                                           const double* d_means, const double* d_sigmas,
                                           sxmc_rng_state* d_rng, double* d_nll_current,
                                           double* d_nll_proposed, double* d_v_current,
                                           double* d_v_proposed, int* d_accepted,
                                           int* d_counter, float* d_jump_buffer,
                                           int nparameters, const float* d_jump_width,
                                           const double* d_nexpected, const unsigned* d_n_mc,
                                           const short* d_source_id, const unsigned* d_norms,
                                           int debug_mode);

/* ---------------------------------------------------------------- multi-GPU exchange (RCCL) ----- */
/* Fake experiments shard over the GPUs of a node one experiment per rank with NO data-path collective
 * (sxmc.cpp:59-145 is a loop of independent iterations; every rank holds a replica of the MC tables).  The only
 * exchange is one all-gather, at the end, of the per-experiment intervals (interval.h:22-27: point_estimate,
 * lower, upper, coverage per parameter) from which rank 0 takes the medians (sxmc.cpp:126-145, utils.h:76-90).
 * These entry points are that exchange on RCCL (xGMI between the GPUs of a node): plain float buffers. */
typedef struct sxmc_comm* sxmc_comm_t;
/* One communicator per device, all in THIS process (one host thread per GPU: sxmc::ensemble_multi_gpu).
 * out: ndevices handles, out[i] is rank i on devices[i]. */
int sxmc_comm_init_all(const int* devices, int ndevices, sxmc_comm_t* out);
/* One process per GPU: rank 0 makes the id (128 bytes), passes it to the others by any means (a file, MPI,
 * torch.distributed), and every rank joins with the current device. */
int sxmc_comm_unique_id(char* id, size_t id_bytes);
int sxmc_comm_init_rank(const char* id, size_t id_bytes, int nranks, int rank, sxmc_comm_t* out);
int sxmc_comm_rank(sxmc_comm_t c, int* rank, int* nranks);
/* Rank, rank count and device AS THE COMMUNICATOR REPORTS THEM (ncclCommUserRank / ncclCommCount / ncclCommCuDevice),
 * not what the caller passed in: what a bench line records to show how many ranks RCCL really joined.  Any of the
 * three pointers may be null. */
int sxmc_comm_query(sxmc_comm_t c, int* rank, int* nranks, int* device);
/* *failed = 1 when the communicator has recorded an asynchronous error (a peer died, a link failed). */
int sxmc_comm_async_error(sxmc_comm_t c, int* failed);
/* Fail-fast tear-down: ncclCommAbort -- ends a collective that is still waiting for a peer that will never arrive
 * (instead of sxmc_comm_destroy, which would wait for it) and frees the handle. */
int sxmc_comm_abort(sxmc_comm_t c);
/* d_recv[r * count .. (r + 1) * count) = rank r's d_send[0 .. count) on every rank; device buffers; asynchronous
 * on `s` of the communicator's device. */
int sxmc_comm_allgather_f32(sxmc_comm_t c, const float* d_send, float* d_recv, size_t count, sxmc_stream_t s);
int sxmc_comm_destroy(sxmc_comm_t c);
const char* sxmc_comm_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------
 * MEASUREMENT BUILD ONLY.  libsxmc_hip.so does not export what follows (tests/test_abi.py checks); it is compiled into
 * libsxmc_hip_measure.so (make -C sxmc_amd/csrc VARIANT=_measure EXTRA=-DSXMC_MEASURE=1), the library behind the
 * roofline decompositions in profiles/ and behind the tests that need a look inside (known answers of the generator
 * and of the rounded powers, the margin of the codes' error bound). */
#ifdef SXMC_MEASURE
/* Hooks of the histogram-fill kernels.  RESULTS ARE WRONG when mode != 0.
 * bit 0 = stream the columns, skip arithmetic and histogram; bit 1 = arithmetic and histogram run but every reload
 * hits one cached address; bit 2 = skip only the histogram update; bit 3 = stream the float columns regardless of
 * codes (fused step: the fill's part of the launch alone); bit 4 = drop what the queues of ambiguous rows hold;
 * bit 5 = no LDS additions; bits 8-15 = 1 + 64 s: the roundings' share of the codes' error bound scaled by s;
 * bits 16-23 = 1 + 64 t: the whole threshold scaled by t (fill_kernels.inc.h, "THE BOUND"). */
int sxmc_group_set_debug_mode(sxmc_group_t g, int mode);
/* d_out[k] = d_x[k]^i as the polynomial systematics form it (p = sum_i c_i * pow(x, i), pdfz.cpp:310-314): the power
 * rounded ONCE, like libm's pow for small integer exponents -- not the i - 1 roundings of repeated multiplication.
 * Exact for i <= 1, the plain product for i = 2. */
int sxmc_debug_pow_int(const double* d_x, int n, int i, double* d_out);
/* Raw Philox4x32-10 output of d_state[0], 4 words per draw; advances the state. */
int sxmc_debug_philox_dump(sxmc_rng_state* d_state, unsigned* d_out, int ndraws);
/* THE GATED STEP, an experiment ("one more look"): the group's next sxmc_group_step_async puts its
 * fill on `fill_stream` -- ordered after the previous fill, not after the previous step end -- as step `node` (0 .. 15)
 * of a recording; the fill does what does not depend on the proposal beside the step end and waits for it inside.
 * fill_stream null: back to one stream.  sxmc_measure_stream_fork: `to` waits for what `from` has queued (inside a
 * recording: brings `to` into it). */
int sxmc_measure_set_gated_step(sxmc_group_t g, sxmc_stream_t fill_stream, int node);
int sxmc_measure_stream_fork(sxmc_stream_t from, sxmc_stream_t to);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SXMC_HIP_H */
