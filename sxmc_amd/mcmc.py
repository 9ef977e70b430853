"""The caller of the hot path, in the shape of the reference's MCMC driver (src/mcmc.cpp).

ROOT-free: the chain goes to a numpy array instead of a TNtuple.  Buffer set-up follows
mcmc.cpp:159-242, the step sequence mcmc.cpp:261-348 and MCMC::nll mcmc.cpp:390-415.  Two step
forms are offered:
  * reference form   -- group.EvalAsync (zero, fill, lookup) ; nll_event_chunks ;
                        finish_nll_jump_pick_combo      (4 kernels, lut written and re-read)
  * fused form       -- group.EvalNllAsync (zero, fill, lookup+event sum) ;
                        finish_nll_jump_pick_combo      (4 kernels, lut written, not re-read)
  * step form        -- group.McmcStepAsync: zero, fill, lookup + event sum + step end (3 kernels)
  * drop-in form     -- fused="dropin": the reference's literal calls, mcmc.cpp:264-271 + 314-348 -- EvalAsync on
                        every evaluator, EvalFinished on every evaluator, nll_event_chunks,
                        finish_nll_jump_pick_combo -- with no group call at all; the library coalesces the S deferred
                        evaluations into one batched launch (sxmc_hist_eval_async).  The reference form's chain.
All produce the same numbers up to the order of the partial sums.
"""
import os

import numpy as np

from . import capi, nll, pdfz
from .capi import DeviceArray

NLL_BLOCKS, NLL_BLOCK_SIZE, REDUCE_THREADS = 64, 256, 128    # mcmc.cpp:37-45


def make_systematic(desc):
    t = desc["type"]
    if t == "shift":
        return pdfz.ShiftSystematic(desc["obs"], desc["pars"])
    if t == "scale":
        return pdfz.ScaleSystematic(desc["obs"], desc["pars"])
    if t == "ctscale":
        return pdfz.CosThetaScaleSystematic(desc["obs"], desc["pars"])
    if t == "resolution_scale":
        return pdfz.ResolutionScaleSystematic(desc["obs"], desc["true_obs"], desc["pars"])
    raise ValueError(t)


# Steps between the flushes at which a walk over a plan with two forms of the fill asks which one to take
# (sxmc_group_adapt_fill_form): config 3's chain moves its resolution parameter by ~0.05 in 5 000 steps.
ADAPT_INTERVAL = 1000


class MCMC:
    def __init__(self, workload, seed=1234, stream=None, fused=True, samples_on_device=None, share_with=None,
                 lut_output=True, consume=False):
        """share_with: another MCMC over the same workload -- this one's evaluators then share its sample
        tables (one copy in HBM) and only the per-chain state is new; give each such chain its own
        non-blocking `stream` to let their kernels overlap.
        lut_output=False: the fused step forms do not materialise the lookup table (self.lut then holds the
        values of setup() only) and sum over distinct event-bin tuples (sxmc_group_set_lut_output).
        consume=True (fused form): the step end also clears histograms and normalisations for the next step
        (3 launches per step; they cannot be read between steps: sxmc_group_finish_step_async)."""
        w = workload
        self.w = w
        self.stream = stream
        self.fused = fused
        if fused == "dropin" and stream is not None:
            # the evaluators launch on their own (blocking) streams, which order with the legacy default stream only --
            # where the reference launches its NLL kernels (HEMI_KERNEL_LAUNCH(..., 0, 0, ...), mcmc.cpp:314-348)
            raise ValueError("the drop-in form launches its NLL kernels on the legacy default stream, like the reference")
        self.consume = bool(consume) and fused is True
        self.tail = True        # consume: one call per step (sxmc_group_step_async) instead of EvalNllAsync + FinishStepAsync
        self.nsources, self.nsignals = w.nsources, w.nsignals
        self.nparameters = w.nparameters
        self.nnllthreads = NLL_BLOCKS * NLL_BLOCK_SIZE

        # evaluators (signal.cpp:112-133 build_pdfz + AddSystematic)
        self.pdfs = []
        for j, s in enumerate(w.signals):
            if share_with is not None:
                self.pdfs.append(pdfz.EvalHist.Shared(share_with.pdfs[j]))
                continue
            src = samples_on_device[j] if samples_on_device is not None else s.samples
            ev = pdfz.EvalHist(src, s.nfields, w.nobs, w.lower, w.upper, w.nbins, dataset=s.dataset)
            for d in w.systematics:
                ev.AddSystematic(make_systematic(d))
            self.pdfs.append(ev)
        self.group = nll.EvalGroup(self.pdfs)
        self.group.SetLutOutput(lut_output)

        # mcmc.cpp:53-98
        self.parameter_means = DeviceArray(w.parameter_means().astype(np.float64))
        self.parameter_sigma = DeviceArray(w.parameter_sigmas().astype(np.float64))
        self.nexpected = DeviceArray(np.array([s.nexpected for s in w.signals], dtype=np.float64))
        self.n_mc = DeviceArray(np.array([s.n_mc for s in w.signals], dtype=np.uint32))
        self.source_id = DeviceArray(np.array([s.source_id for s in w.signals], dtype=np.int16))
        self.rngs = nll.make_rngs(self.nparameters, seed, stream)

        # mcmc.cpp:159-198
        self.current_vector = DeviceArray(w.parameter_means().astype(np.float64))
        self.proposed_vector = DeviceArray(w.parameter_means().astype(np.float64))
        self.normalizations = DeviceArray.zeros(self.nsignals, np.uint32)
        self.event_partial_sums = DeviceArray.zeros(self.nnllthreads, np.float64)
        self.event_total_sum = DeviceArray.zeros(1, np.float64)
        self.jump_counter = DeviceArray.zeros(1, np.int32)
        self.accept_counter = DeviceArray.zeros(1, np.int32)
        self.current_nll = DeviceArray.zeros(1, np.float64)
        self.proposed_nll = DeviceArray.zeros(1, np.float64)
        self.jump_width = None
        self.jump_buffer = None
        self.lut = None
        self.nevents = 0
        self._graph, self._graph_steps = None, 0

    # mcmc.cpp:198-228 (keeps the reference's `i < nsignals` test at :217)
    def initial_jump_widths(self, fixed=None):
        w = self.w
        means = w.parameter_means().astype(np.float32)
        sigmas = w.parameter_sigmas().astype(np.float32)
        fixed = [False] * self.nparameters if fixed is None else fixed
        nfloat = sum(1 for f in fixed if not f)
        scale_factor = np.float32(2.4 * 2.4 / nfloat)
        out = np.zeros(self.nparameters, dtype=np.float32)
        for i in range(self.nparameters):
            if fixed[i]:
                out[i] = -1
                continue
            if sigmas[i] > 0:
                width = sigmas[i]
            elif i < self.nsignals:
                m = max(means[i], np.float32(10))
                width = np.sqrt(m) / m
            else:
                width = np.sqrt(max(means[i], np.float32(1)))
            out[i] = np.float32(0.1 * width * scale_factor)
        return out

    def setup(self, data=None, sync_interval=10000, jump_width=None):
        """mcmc.cpp:186, 230-256: bind buffers, first evaluation at the current vector, NLL of it,
        first proposal."""
        w = self.w
        data = w.events if data is None else data
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        self._graph = None                      # recorded steps hold the old evaluation-point tables
        self.nevents = data.size // (w.nobs + 1)
        self.sync_interval = sync_interval
        self._since_flush = 0                   # steps launched since the jump buffer was last read back
        self.jump_buffer = DeviceArray.zeros(sync_interval * (self.nparameters + 1), np.float32)
        jw = self.initial_jump_widths() if jump_width is None else np.asarray(jump_width, np.float32)
        self.jump_width = DeviceArray(jw)
        self.lut = DeviceArray.zeros(self.nevents * self.nsignals, np.float32)
        for i, p in enumerate(self.pdfs):
            p.SetEvalPoints(data)
            p.SetPDFValueBuffer(self.lut, i * self.nevents, 1)
            p.SetNormalizationBuffer(self.normalizations, i)
            p.SetParameterBuffer(self.current_vector, self.nsources)
        self.group.EvalAsync(True, self.stream)
        self.group.EvalFinished()
        for p in self.pdfs:
            p.SetParameterBuffer(self.proposed_vector, self.nsources)
        self.nll(self.current_vector, self.current_nll)
        nll.pick_new_vector(1, 64, self.stream, self.nparameters, self.rngs, self.jump_width,
                            self.current_vector, self.proposed_vector)
        self._adapt_pending = True               # (a plan with two forms of the fill: chosen at the first step)
        self._two_forms_cached = None

    def nll(self, v, out):
        """MCMC::nll (mcmc.cpp:390-415): three launches over an already evaluated lut."""
        nll.nll_event_chunks(NLL_BLOCKS, NLL_BLOCK_SIZE, self.stream, self.lut, v, self.nevents, self.nsignals,
                             self.nexpected, self.n_mc, self.source_id, self.normalizations,
                             self.event_partial_sums)
        nll.nll_event_reduce(1, REDUCE_THREADS, self.stream, self.nnllthreads, self.event_partial_sums,
                             self.event_total_sum)
        nll.nll_total(1, 1, self.stream, self.nparameters, v, self.nsignals, self.nsources,
                      self.parameter_means, self.parameter_sigma, self.event_total_sum, self.nexpected,
                      self.n_mc, self.source_id, self.normalizations, out)

    def step(self, debug_mode=False):
        """One pass of the hot path = one NLL evaluation at the proposed vector + the fused
        accept/reject/propose (mcmc.cpp:264-271, 314-348).  Asynchronous."""
        if getattr(self, "_adapt_pending", False) and not getattr(self, "_recording", False):
            # a plan with a boxed and an ordered form: the form of the first steps, from the first proposal
            # (sxmc_group_adapt_fill_form; flush() asks again)
            self._adapt_pending = False
            self.group.AdaptFillForm()
        self._launching(1)
        if self.fused == "step":
            self.group.McmcStepAsync(self.stream, self.parameter_means, self.parameter_sigma, self.rngs,
                                     self.current_nll, self.proposed_nll, self.current_vector,
                                     self.proposed_vector, self.accept_counter, self.jump_counter,
                                     self.jump_buffer, self.nparameters, self.nsources, self.jump_width,
                                     self.nexpected, self.n_mc, self.source_id, self.normalizations, debug_mode)
            return
        if self.consume and self.tail:
            self.group.StepAsync(self.stream, self.parameter_means, self.parameter_sigma, self.rngs,
                                 self.current_nll, self.proposed_nll, self.current_vector, self.proposed_vector,
                                 self.accept_counter, self.jump_counter, self.jump_buffer, self.nparameters,
                                 self.nsources, self.jump_width, self.nexpected, self.n_mc, self.source_id,
                                 self.normalizations, debug_mode)
            return
        if self.fused == "dropin":
            for p in self.pdfs:                  # mcmc.cpp:265-267
                p.EvalAsync()
            for p in self.pdfs:                  # mcmc.cpp:268-270
                p.EvalFinished()
            nll.nll_event_chunks(NLL_BLOCKS, NLL_BLOCK_SIZE, self.stream, self.lut, self.proposed_vector,
                                 self.nevents, self.nsignals, self.nexpected, self.n_mc, self.source_id,
                                 self.normalizations, self.event_partial_sums)
            npartial = self.nnllthreads
        elif self.fused:
            npartial = self.group.EvalNllAsync(self.stream, self.proposed_vector, self.nexpected, self.n_mc,
                                               self.source_id, self.normalizations, self.event_partial_sums)
        else:
            self.group.EvalAsync(True, self.stream)
            nll.nll_event_chunks(NLL_BLOCKS, NLL_BLOCK_SIZE, self.stream, self.lut, self.proposed_vector,
                                 self.nevents, self.nsignals, self.nexpected, self.n_mc, self.source_id,
                                 self.normalizations, self.event_partial_sums)
            npartial = self.nnllthreads
        if self.consume:
            self.group.FinishStepAsync(self.stream, npartial, self.event_partial_sums, self.parameter_means,
                                       self.parameter_sigma, self.rngs, self.current_nll, self.proposed_nll,
                                       self.current_vector, self.proposed_vector, self.accept_counter,
                                       self.jump_counter, self.jump_buffer, self.nparameters, self.nsources,
                                       self.jump_width, self.nexpected, self.n_mc, self.source_id,
                                       self.normalizations, debug_mode)
            return
        nll.finish_nll_jump_pick_combo(1, REDUCE_THREADS, self.stream, npartial, self.event_partial_sums,
                                       self.nsignals, self.nsources, self.parameter_means,
                                       self.parameter_sigma, self.rngs, self.current_nll, self.proposed_nll,
                                       self.current_vector, self.proposed_vector, self.accept_counter,
                                       self.jump_counter, self.jump_buffer, self.nparameters, self.jump_width,
                                       self.nexpected, self.n_mc, self.source_id, self.normalizations,
                                       debug_mode)

    def _launching(self, n):
        """Every step appends a row to the jump buffer (sync_interval rows, jump_decider nll_kernels.cpp:78-84, which
        does not check): refuse on the host what would run past its end."""
        if getattr(self, "_recording", False):
            return
        if self._since_flush + n > self.sync_interval:
            raise RuntimeError("%d step(s) after %d since the last flush do not fit the jump buffer of %d rows: "
                               "flush() first, or setup(sync_interval=...) for longer runs"
                               % (n, self._since_flush, self.sync_interval))
        self._since_flush += n

    def reseed(self, seed):
        """Fresh generator states (a new experiment on the same evaluators)."""
        self.rngs = nll.make_rngs(self.nparameters, seed, self.stream)

    def walk(self, data, nsteps, burnin_fraction, debug_mode=False, sync_interval=10000, graph_steps=0,
             lookahead=False):
        """MCMC::operator() (mcmc.cpp:143-387): start at the means, walk nsteps, re-tune the proposal
        widths from the chain's spread at burnin_steps and 2 * burnin_steps (dropping the steps so far
        unless debug_mode).  Returns (chain [nkept, P + 1] float32, accepted).
        graph_steps = K > 0 replays a HIP graph of K recorded steps wherever K steps fit between two
        points that need the host (re-tuning, jump-buffer flush); the chain is the same."""
        self.walk_begin(data, nsteps, burnin_fraction, debug_mode, sync_interval)
        self.lookahead_passes = 0
        # (a shape the look-ahead pass is not offered for -- see Group.LookaheadSupported -- walks sequentially)
        if lookahead and self.group.LookaheadSupported():
            # the look-ahead walk (LookaheadWalk below): two evaluations per pass over the tables, one or two steps
            # per pass; needs consume=True, lut_output=False and a created stream.  The same chain.
            la = LookaheadWalk(self)
            la.bind(data)
            i = 0
            try:
                for f in self.flush_schedule():
                    self._retune_if_due(i)
                    la.restart()              # (the look-ahead vector with the widths as they now are)
                    la.steps(f - i + 1, graph_passes=graph_steps, debug_mode=self._debug, count0=0)
                    self._flush_if_due(f)
                    i = f + 1
            finally:
                self.lookahead_passes = la.passes
                la.close()
            return self.walk_end()
        if graph_steps <= 0:
            for i in range(nsteps):
                self.walk_advance(i)
            return self.walk_end()
        i = 0
        for f in self.flush_schedule():       # steps i..f: host work only before step i and after step f
            self._retune_if_due(i)
            self.steps(f - i + 1, graph_steps, self._debug)
            self._flush_if_due(f)
            i = f + 1
        return self.walk_end()

    def flush_schedule(self):
        """Indices of the steps after which the jump buffer is read back (mcmc.cpp:351-377), ascending.  The
        re-tuning points (burnin_steps, 2 * burnin_steps) each directly follow one of them."""
        n, b = self._nsteps, self._burnin
        due = {k for k in range(0, n, self.sync_interval)} | {k for k in (n - 1, b - 1, 2 * b - 1) if 0 <= k < n}
        # a plan with two forms of the fill (sxmc_group_adapt_fill_form is asked at every flush): the choice is made from
        # the parameters at the flush, and a chain moves -- flushes every ADAPT_INTERVAL steps bound how stale it gets
        if self._two_forms():
            due |= {k for k in range(ADAPT_INTERVAL - 1, n, ADAPT_INTERVAL)}
        return sorted(due)

    def capture_steps(self, k, debug_mode=False):
        """Records k steps on this chain's stream as one HIP graph.  One step must have
        run since the last change to the evaluators; the graph is dropped by setup()."""
        if self.stream is None:
            raise ValueError("graph capture needs a created stream (MCMC(stream=capi.new_stream()))")
        self._recording = True                  # (recorded steps run, and are counted, when the graph is launched)
        # (measurement build + SXMC_GATED_STEP=1, an experiment: every step's fill on a second stream, beside the step
        # end of the step before, waiting inside for the proposal -- include/sxmc_hip.h, sxmc_measure_set_gated_step)
        gated = (os.environ.get("SXMC_GATED_STEP") == "1" and capi.is_measurement_build() and self.consume
                 and k <= 16)
        if gated and getattr(self, "_fill_stream", None) is None:
            self._fill_stream = capi.new_stream()
        try:
            with capi.Graph.capture(self.stream) as g:
                if gated:
                    capi.call("sxmc_measure_stream_fork", capi.ptr(self.stream), capi.ptr(self._fill_stream))
                for q in range(k):
                    if gated:
                        capi.call("sxmc_measure_set_gated_step", self.group._g, capi.ptr(self._fill_stream), q)
                    self.step(debug_mode)
                if gated:
                    capi.call("sxmc_measure_set_gated_step", self.group._g, None, 0)
        finally:
            self._recording = False
            if gated:
                capi.call("sxmc_measure_set_gated_step", self.group._g, None, 0)
        return g

    def steps(self, n, graph_steps=0, debug_mode=False):
        """n steps in a row: graph replays of graph_steps recorded steps, the remainder launched one by one."""
        if graph_steps > 0 and n >= graph_steps:
            if self._graph is None or self._graph_steps != graph_steps:
                self._graph, self._graph_steps = self.capture_steps(graph_steps, debug_mode), graph_steps
            self._launching((n // graph_steps) * graph_steps)
            self._graph.launch(n // graph_steps)
            n %= graph_steps
        for _ in range(n):
            self.step(debug_mode)

    # The same walk cut into per-step pieces, so that several chains can be advanced in turn and have
    # their kernels in flight together (one stream per chain).
    def walk_begin(self, data, nsteps, burnin_fraction, debug_mode=False, sync_interval=10000):
        w = self.w
        self._nsteps, self._debug = nsteps, debug_mode
        self._burnin = int(nsteps * burnin_fraction)
        self.current_vector.set(w.parameter_means().astype(np.float64))
        self.jump_counter.set(np.zeros(1, np.int32))
        self.accept_counter.set(np.zeros(1, np.int32))
        self.setup(data, sync_interval=sync_interval)
        # mcmc.cpp:199: nfloat = the parameters that are not fixed (a fixed one carries jump width <= 0)
        nfloat = max(1, int(np.count_nonzero(self.jump_width.get() > 0)))
        self._scale_factor = np.float32(2.4 * 2.4 / nfloat)
        self._rows, self._accepted = [np.zeros((0, self.nparameters + 1), np.float32)], 0

    def _retune_if_due(self, i):
        b = self._burnin
        if i == b or i == 2 * b:                                   # mcmc.cpp:274-311
            sofar = np.concatenate(self._rows, axis=0)
            jw = self.jump_width.get()
            for j in range(self.nparameters):
                if jw[j] <= 0:
                    continue
                sd = float(sofar[:, j].std()) if sofar.shape[0] > 1 else 0.0
                jw[j] = self._scale_factor * (sd if sd > 0 else jw[j])
            self.jump_width.set(jw)
            if not self._debug:
                self._rows = [np.zeros((0, self.nparameters + 1), np.float32)]

    def _flush_if_due(self, i):
        b = self._burnin
        if i % self.sync_interval == 0 or i == self._nsteps - 1 or i == b - 1 or i == 2 * b - 1 or \
                (i % ADAPT_INTERVAL == ADAPT_INTERVAL - 1 and self._two_forms()):
            r, nacc = self.flush(device_wide=False)                # mcmc.cpp:351-377
            self._rows.append(r)
            self._accepted += nacc

    def _two_forms(self):
        """Does the chain's plan hold a boxed and an ordered form of the fill (decided once per walk)?"""
        if getattr(self, "_two_forms_cached", None) is None:
            self._two_forms_cached = self.group.FillForm() != 0
        return self._two_forms_cached

    def walk_advance(self, i):
        self._retune_if_due(i)
        self.step(self._debug)
        self._flush_if_due(i)

    def walk_end(self):
        return np.concatenate(self._rows, axis=0), self._accepted

    def flush(self, device_wide=True):
        """mcmc.cpp:351-377: read back and reset the jump buffer.  Returns (rows, naccepted).
        device_wide=False waits for this chain's stream only (other chains keep running)."""
        if device_wide or self.stream is None:
            capi.synchronize()
        else:
            capi.call("sxmc_stream_synchronize", capi.ptr(self.stream))
        njumps = int(self.jump_counter.get()[0])
        nacc = int(self.accept_counter.get()[0])
        rows = self.jump_buffer.get()[: njumps * (self.nparameters + 1)].reshape(njumps, self.nparameters + 1)
        self.jump_counter.set(np.zeros(1, np.int32))
        self.accept_counter.set(np.zeros(1, np.int32))
        self._since_flush = 0
        if self.consume:
            # (the cooperative step end waits inside its kernel, with a bound; a wait that ran into it invalidates the run)
            timeouts = self.group.StepEndTimeouts(self.stream)
            if timeouts:
                raise RuntimeError("%d workgroup(s) of the cooperative step end gave up waiting: the chain is not valid"
                                   % timeouts)
        # a plan with a boxed and an ordered form: which one the steps up to the next flush take (include/sxmc_hip.h,
        # sxmc_group_adapt_fill_form); recorded steps replay the old form, so they are recorded again
        if self.group.AdaptFillForm()[1]:
            graph, self._graph = self._graph, None
            if graph is not None and hasattr(graph, "close"):
                graph.close()
        return rows.copy(), nacc

    def run(self, nsteps, debug_mode=False):
        """The step loop without burn-in re-tuning: returns the chain [nsteps, P+1] (float32)."""
        chunks, accepted = [], 0
        for i in range(nsteps):
            self.step(debug_mode)
            if (i + 1) % self.sync_interval == 0 or i == nsteps - 1:
                rows, nacc = self.flush()
                chunks.append(rows)
                accepted += nacc
        return np.concatenate(chunks, axis=0), accepted


class LookaheadWalk:
    """ONE chain stepped with two likelihood evaluations per pass over the tables (the look-ahead walk,
    sxmc_multigroup_lookahead_step_async): besides the step's proposal, the vector the NEXT step would propose if this
    one rejects -- known in advance, because a rejection leaves the chain where it was -- is evaluated in the same
    fill pass, and the step end decides one or two steps.  The chain is the sequential walk's, bit for bit.
    chain: an MCMC(lut_output=False, consume=True, stream=...) after setup(); a shadow set of evaluators over the
    same tables is created here."""

    def __init__(self, chain, threads=768, blocks_per_cu=1):
        assert chain.consume and chain.stream is not None, "the look-ahead walk needs consume=True and a created stream"
        self.chain = chain
        self.shadow = MCMC(chain.w, seed=1, fused=True, stream=chain.stream, share_with=chain, lut_output=False,
                           consume=True)
        # threads > 0: lanes per workgroup for BOTH sets of evaluators (their launch plans must agree; a pass of two
        # evaluations is bound by vector issue and needs more registers than 1 024 lanes leave each -- spills inside
        # the stream loop drain the loads in flight --: 768, the kernel being compiled for that bound).  0: leave both
        # as they are (defaults agree).
        self._threads, self._bpc = threads, blocks_per_cu
        self.cap = DeviceArray.zeros(1, np.int32)
        self._graph, self._graph_passes = None, 0
        self.mg = None
        self.passes = 0                       # passes launched (each: one fill of the tables, two evaluations)
        self.steps_seen, self.passes_seen = 0, 0

    def bind(self, data=None):
        """After chain.setup(): the shadow evaluators get the same data; (re)starts the look-ahead."""
        c, s = self.chain, self.shadow
        self.drop_graph()
        s.setup(data, sync_interval=8)
        for g in (c.group, s.group):
            if self._threads:
                g.SetLaunchConfig(self._threads, self._bpc)
        if self.mg is None:
            self.mg = nll.MultiGroup([c, s])
        self.restart()

    def restart(self):
        """The look-ahead vector for the chain's present state (after setup, a re-tuning of the jump widths, ...)."""
        c, s = self.chain, self.shadow
        capi.call("sxmc_lookahead_begin", capi.ptr(c.stream), c.nparameters, capi.ptr(c.rngs), capi.ptr(c.jump_width),
                  capi.ptr(c.current_vector), capi.ptr(s.proposed_vector))

    def one_pass(self, debug_mode=False):
        self.mg.LookaheadStepAsync(self.chain.stream, self.shadow.proposed_vector, self.shadow.normalizations, self.cap,
                                   debug_mode)
        self.passes += 1

    def drop_graph(self):
        if self._graph is not None:
            self._graph.close()
        self._graph = None

    def steps(self, n, graph_passes=0, debug_mode=False, count0=None):
        """Exactly n more steps of the chain: passes are launched in rounds of about what is still needed, the
        jump counter read back after each round (a pass advances the chain by one or two steps; passes launched
        beyond the stop do nothing).  count0: the jump counter now, if the caller knows it."""
        c = self.chain
        if count0 is None:
            capi.call("sxmc_stream_synchronize", capi.ptr(c.stream))
            count0 = int(c.jump_counter.get()[0])
        target = count0 + n
        if target > c.sync_interval:
            raise RuntimeError("%d more steps after %d in the jump buffer do not fit its %d rows: flush() first"
                               % (n, count0, c.sync_interval))
        c._since_flush = target
        self.cap.set(np.array([target], np.int32))
        done = count0
        while done < target:
            need = target - done
            # a pass advances the chain by 1 + P(reject) steps: aim a little short of what is needed (a pass beyond
            # the stop still streams the tables), finish with what the counter says is left
            rate = min(2.0, 1.03 * self.steps_seen / self.passes_seen) if self.passes_seen >= 16 else 1.75
            k = npass = max(1, int(need / rate))
            before, pdone = done, self.passes
            if graph_passes > 0 and k >= graph_passes:
                if self._graph is None or self._graph_passes != graph_passes:
                    self.drop_graph()
                    self.one_pass(debug_mode)                            # (plans in place before recording)
                    k -= 1
                    with capi.Graph.capture(c.stream) as g:
                        for _ in range(graph_passes):
                            self.one_pass(debug_mode)
                    self.passes -= graph_passes
                    self._graph, self._graph_passes = g, graph_passes
                reps = k // graph_passes
                if reps:
                    self._graph.launch(reps)
                    self.passes += reps * graph_passes
                    k -= reps * graph_passes
            for _ in range(k):
                self.one_pass(debug_mode)
            capi.call("sxmc_stream_synchronize", capi.ptr(c.stream))
            done = int(c.jump_counter.get()[0])
            self.steps_seen += done - before
            self.passes_seen += self.passes - pdone
        return done

    def close(self):
        self.drop_graph()
        if self.mg is not None:
            self.mg.close()
        for p in self.shadow.pdfs:
            p.close()
        self.shadow.group.close()


class LockstepChains:
    """2-4 chains over the same sample tables (MCMC(..., share_with=base), all on ONE stream) advanced together:
    every step is one fill pass over the tables for all of them (nll.MultiGroup / sxmc_multigroup_step_async)
    followed by each chain's own step end.  The chains walk exactly what they walk when stepped alone."""

    def __init__(self, chains):
        self.chains = list(chains)
        self.stream = self.chains[0].stream
        assert all(c.stream == self.stream for c in self.chains), "lockstep chains share one stream"
        assert all(c.consume for c in self.chains), "lockstep chains clear for the next step (consume=True)"
        self.mg = nll.MultiGroup(self.chains)
        self._graph, self._graph_steps = None, 0
        self._recording = False

    def step(self, debug_mode=False):
        if not self._recording:
            for c in self.chains:
                c._launching(1)      # (every chain's jump buffer must hold the run)
        self.mg.StepAsync(self.stream, debug_mode)

    def drop_graph(self):
        if self._graph is not None:
            self._graph.close()
        self._graph = None

    def steps(self, n, graph_steps=0, debug_mode=False):
        """n lockstep steps: graph replays of graph_steps recorded steps, the remainder launched one by one."""
        if graph_steps > 0 and n >= graph_steps:
            if self._graph is None or self._graph_steps != graph_steps:
                self.drop_graph()
                self._recording = True
                try:
                    with capi.Graph.capture(self.stream) as g:
                        for _ in range(graph_steps):
                            self.step(debug_mode)
                finally:
                    self._recording = False
                self._graph, self._graph_steps = g, graph_steps
            for c in self.chains:
                c._launching((n // graph_steps) * graph_steps)
            self._graph.launch(n // graph_steps)
            n %= graph_steps
        for _ in range(n):
            self.step(debug_mode)

    def close(self):
        self.drop_graph()
        self.mg.close()
