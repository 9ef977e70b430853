"""Python spelling of the reference's NLL / MCMC-step kernel launch points (src/nll_kernels.h)
and of the batched evaluator group, over the C ABI.

Each function takes the (grid, block, stream) triple mcmc.cpp passes to HEMI_KERNEL_LAUNCH
followed by the reference's argument list (nll_kernels.h:60-207).  Arrays are device buffers.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import ptr


def make_rngs(nparameters, seed, stream=None):
    """hemi::Array<RNGState>(nparameters) + init_device_rngs (mcmc.cpp:116-126)."""
    rng = capi.DeviceArray.empty(nparameters * 4, np.uint64)
    block = 128
    grid = nparameters // block + 1
    capi.call("sxmc_launch_init_device_rngs", grid, block, ptr(stream), nparameters, int(seed), ptr(rng))
    return rng


def pick_new_vector(grid, block, stream, nthreads, rng, jump_width, current_vector, proposed_vector):
    capi.call("sxmc_launch_pick_new_vector", grid, block, ptr(stream), int(nthreads), ptr(rng),
              ptr(jump_width), ptr(current_vector), ptr(proposed_vector))


def jump_decider(grid, block, stream, rng, nll_current, nll_proposed, v_current, v_proposed, nparameters,
                 accepted, counter, jump_buffer):
    capi.call("sxmc_launch_jump_decider", grid, block, ptr(stream), ptr(rng), ptr(nll_current),
              ptr(nll_proposed), ptr(v_current), ptr(v_proposed), int(nparameters), ptr(accepted),
              ptr(counter), ptr(jump_buffer))


def nll_event_chunks(grid, block, stream, lut, pars, ne, ns, nexpected, n_mc, source_id, norms, sums):
    capi.call("sxmc_launch_nll_event_chunks", grid, block, ptr(stream), ptr(lut), ptr(pars), int(ne), int(ns),
              ptr(nexpected), ptr(n_mc), ptr(source_id), ptr(norms), ptr(sums))


def nll_event_reduce(grid, block, stream, nthreads, sums, total_sum):
    capi.call("sxmc_launch_nll_event_reduce", grid, block, ptr(stream), int(nthreads), ptr(sums), ptr(total_sum))


def nll_total(grid, block, stream, nparameters, pars, nsignals, nsources, means, sigmas, events_total,
              nexpected, n_mc, source_id, norms, nll):
    capi.call("sxmc_launch_nll_total", grid, block, ptr(stream), int(nparameters), ptr(pars), int(nsignals),
              int(nsources), ptr(means), ptr(sigmas), ptr(events_total), ptr(nexpected), ptr(n_mc),
              ptr(source_id), ptr(norms), ptr(nll))


def finish_nll_jump_pick_combo(grid, block, stream, npartial_sums, sums, nsignals, nsources, means, sigmas,
                               rng, nll_current, nll_proposed, v_current, v_proposed, accepted, counter,
                               jump_buffer, nparameters, jump_width, nexpected, n_mc, source_id, norms,
                               debug_mode=False):
    capi.call("sxmc_launch_finish_nll_jump_pick_combo", grid, block, ptr(stream), int(npartial_sums),
              ptr(sums), int(nsignals), int(nsources), ptr(means), ptr(sigmas), ptr(rng), ptr(nll_current),
              ptr(nll_proposed), ptr(v_current), ptr(v_proposed), ptr(accepted), ptr(counter),
              ptr(jump_buffer), int(nparameters), ptr(jump_width), ptr(nexpected), ptr(n_mc), ptr(source_id),
              ptr(norms), int(bool(debug_mode)))


class EvalGroup:
    """All signals' evaluators stepped together: the batched form of the
    `EvalAsync on all, then EvalFinished on all` loop of mcmc.cpp:264-271."""

    def __init__(self, evaluators):
        self.evaluators = list(evaluators)
        arr = (C.c_void_p * max(1, len(self.evaluators)))(*[e.handle for e in self.evaluators])
        g = C.c_void_p(0)
        capi.call("sxmc_group_create", arr, len(self.evaluators), C.byref(g))
        self._g = g

    def SetLaunchConfig(self, bin_threads=0, bin_blocks_per_cu=0):
        capi.call("sxmc_group_set_launch_config", self._g, int(bin_threads), int(bin_blocks_per_cu))

    def Optimize(self, stream=None):
        """Times the fill with a few lane counts per CU and keeps the fastest (EvalHist::Optimize for the batched
        launch).  Returns the lane count kept (0: nothing to choose)."""
        n = C.c_int(0)
        capi.call("sxmc_group_optimize", self._g, ptr(stream), C.byref(n))
        return n.value

    def SetPartition(self, mode):
        """0 automatic, 1 sliced, 2 interleaved (see include/sxmc_hip.h)."""
        capi.call("sxmc_group_set_partition", self._g, int(mode))

    def SetPartitionTeams(self, teams):
        """Teams of workgroups per member over a bucketed table (0 = default, one; see include/sxmc_hip.h)."""
        capi.call("sxmc_group_set_partition_teams", self._g, int(teams))

    def SetSparse(self, enable):
        """Count only the event bins when a histogram beyond LDS capacity is evaluated for lookup (default on)."""
        capi.call("sxmc_group_set_sparse", self._g, int(bool(enable)))

    def SetPrebinning(self, enable):
        """Stream observables that no systematic writes as one pre-binned narrow column (default on)."""
        capi.call("sxmc_group_set_prebinning", self._g, int(bool(enable)))

    def SetBucketing(self, enable):
        """Stream a copy of the table grouped by the bins of the observables no systematic writes (default on)."""
        capi.call("sxmc_group_set_bucketing", self._g, int(bool(enable)))

    def SetOrdering(self, enable, force=False):
        """Keep each bucket's rows sorted by a monotonically written observable: its bin is then one constant per
        256-sample granule, worked out per evaluation from the granule's end values (default: on where the table
        has enough granules per bin edge to pay; force: wherever it applies)."""
        capi.call("sxmc_group_set_ordering", self._g, 2 if (enable and force) else int(bool(enable)))

    def SetBoxes(self, enable):
        """Group each bucket's rows into small boxes of a resolution-scaled observable and its truth field: the observable
        is then one constant per 256-sample granule wherever the box's image lands in one bin (None: where it pays, the
        default; True: wherever it applies; False: never).  See include/sxmc_hip.h."""
        capi.call("sxmc_group_set_boxes", self._g, -1 if enable is None else int(bool(enable)))

    def SetBoxLimit(self, bins):
        """AdaptFillForm takes the boxed form while the image of a mean box is narrower than this many bins."""
        capi.call("sxmc_group_set_box_limit", self._g, float(bins))

    def AdaptFillForm(self):
        """A boxed plan with an ordered twin: choose the form of the next fills from the parameters the evaluators read now
        (sxmc_group_adapt_fill_form).  Returns (form, changed): form 1 boxed, 2 ordered, 0 one form only."""
        f, ch = C.c_int(0), C.c_int(0)
        capi.call("sxmc_group_adapt_fill_form", self._g, C.byref(f), C.byref(ch))
        return f.value, bool(ch.value)

    def SetFillForm(self, form):
        capi.call("sxmc_group_set_fill_form", self._g, int(form))

    def FillForm(self):
        f = C.c_int(0)
        capi.call("sxmc_group_fill_form", self._g, C.byref(f))
        return f.value

    def SetCodes(self, enable):
        """Stream an ordered table's fields as 16-bit codes with an exact recheck of the samples near a bin edge
        (default on where it applies; None: the library's default).  See include/sxmc_hip.h."""
        capi.call("sxmc_group_set_codes", self._g, -1 if enable is None else int(bool(enable)))

    def CodesInfo(self):
        """(members streaming codes, rows, rows that always ask the exact columns, rows never counted)."""
        m = C.c_int(0)
        r, e, n = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        capi.call("sxmc_group_codes_info", self._g, C.byref(m), C.byref(r), C.byref(e), C.byref(n))
        return m.value, r.value, e.value, n.value

    def SetRuntimeKernels(self, enable):
        """Specialise the fill kernel through hiprtc for programs of systematics that are not built in (default on)."""
        capi.call("sxmc_group_set_runtime_kernels", self._g, int(bool(enable)))

    def LookaheadSupported(self):
        """Can a walk over this group be taken by the look-ahead pass and stay the sequential chain bit for bit?
        (sxmc_group_lookahead_supported: not for histograms beyond LDS, a materialised lookup table, or problems of at
        most 256 look-ups per step, whose sequential step ends in the one-workgroup form.)"""
        ok = C.c_int(0)
        capi.call("sxmc_group_lookahead_supported", self._g, C.byref(ok))
        return bool(ok.value)

    def LaunchInfo(self):
        buf = C.create_string_buffer(8192)
        capi.call("sxmc_group_launch_info", self._g, buf, len(buf))
        return buf.value.decode()

    def SetLutOutput(self, enable):
        """False: EvalNllAsync / McmcStepAsync do not write the lookup table and sum over the distinct tuples
        of event bins, weighted by multiplicity (see include/sxmc_hip.h).  Default True."""
        capi.call("sxmc_group_set_lut_output", self._g, int(bool(enable)))

    def SetDebugMode(self, mode):
        """The kernels' measurement hooks (RESULTS ARE WRONG when mode != 0): only the measurement build has them
        (SXMC_HIP_LIB=.../libsxmc_hip_measure.so; include/sxmc_hip.h, "MEASUREMENT BUILD ONLY")."""
        if not capi.is_measurement_build():
            raise capi.SxmcError(capi.ERR_STATE, "sxmc_group_set_debug_mode exists in libsxmc_hip_measure.so only: "
                                                 "run with SXMC_HIP_LIB pointing at it")
        capi.call("sxmc_group_set_debug_mode", self._g, int(mode))

    def SetCodesQueueLog(self, log2_entries):
        """Cap on the queues of ambiguous rows of a fill over codes (2^9 .. 2^11 entries, 0: what fits); results do not
        depend on it."""
        capi.call("sxmc_group_set_codes_queue_log", self._g, int(log2_entries))

    def CodesWindows(self, member):
        """(base[], step[]) of the windows member `member`'s codes were cut from; empty when its fill streams none."""
        n = C.c_int(0)
        base, step = (C.c_double * 4)(), (C.c_double * 4)()
        capi.call("sxmc_group_codes_windows", self._g, int(member), C.byref(n), base, step)
        return list(base[:n.value]), list(step[:n.value])

    def EvalAsync(self, do_eval_pdf=True, stream=None):
        capi.call("sxmc_group_eval_async", self._g, int(bool(do_eval_pdf)), ptr(stream))

    def EvalNllAsync(self, stream, pars, nexpected, n_mc, source_id, norms, sums):
        """Fill + lookup + nll_event_chunks fused; returns the number of partial sums written."""
        n = C.c_int(0)
        capi.call("sxmc_group_eval_nll_async", self._g, ptr(stream), ptr(pars), ptr(nexpected), ptr(n_mc),
                  ptr(source_id), ptr(norms), ptr(sums), C.byref(n))
        return n.value

    def McmcStepAsync(self, stream, means, sigmas, rng, nll_current, nll_proposed, v_current, v_proposed,
                      accepted, counter, jump_buffer, nparameters, nsources, jump_width, nexpected, n_mc,
                      source_id, norms, debug_mode=False):
        """One whole MCMC step in three launches (zero, fill, lookup + event sum + step end)."""
        capi.call("sxmc_group_mcmc_step_async", self._g, ptr(stream), ptr(means), ptr(sigmas), ptr(rng),
                  ptr(nll_current), ptr(nll_proposed), ptr(v_current), ptr(v_proposed), ptr(accepted),
                  ptr(counter), ptr(jump_buffer), int(nparameters), int(nsources), ptr(jump_width),
                  ptr(nexpected), ptr(n_mc), ptr(source_id), ptr(norms), int(bool(debug_mode)))

    def StepAsync(self, stream, means, sigmas, rng, nll_current, nll_proposed, v_current, v_proposed,
                  accepted, counter, jump_buffer, nparameters, nsources, jump_width, nexpected, n_mc,
                  source_id, norms, debug_mode=False):
        """One whole MCMC step in two launches: fill, then lookup + event sum + step end + clearing for the next
        step in one workgroup (sxmc_group_step_async; three launches where that is too much for one workgroup)."""
        capi.call("sxmc_group_step_async", self._g, ptr(stream), ptr(means), ptr(sigmas), ptr(rng),
                  ptr(nll_current), ptr(nll_proposed), ptr(v_current), ptr(v_proposed), ptr(accepted),
                  ptr(counter), ptr(jump_buffer), int(nparameters), int(nsources), ptr(jump_width),
                  ptr(nexpected), ptr(n_mc), ptr(source_id), ptr(norms), int(bool(debug_mode)))

    def LastStepLaunches(self):
        n = C.c_int(0)
        capi.call("sxmc_group_last_step_launches", self._g, C.byref(n))
        return n.value

    def SetTailKernel(self, enable):
        capi.call("sxmc_group_set_tail_kernel", self._g, int(bool(enable)))

    def SetCooperativeStepEnd(self, enable):
        """The step end of StepAsync as ONE launch whose workgroups wait for each other (step_end_kernel) where the
        event sum is small enough; off: look-ups + event sum, then step end + clearing (two launches)."""
        capi.call("sxmc_group_set_cooperative_step_end", self._g, int(bool(enable)))

    def SetFusedStep(self, enable):
        """The whole step as ONE launch (fill_step_kernel: the fill's workgroups + the step end's roles in one grid)
        where the plan allows it; off: the fill, then the step end."""
        capi.call("sxmc_group_set_fused_step", self._g, int(bool(enable)))

    def StepEndTimeouts(self, stream=None):
        """Workgroups of the cooperative step end that gave up waiting (0 in a healthy run), read through `stream`."""
        n = C.c_uint(0)
        capi.call("sxmc_group_step_end_timeouts", self._g, ptr(stream), C.byref(n))
        return n.value

    def FinishStepAsync(self, stream, npartial_sums, sums, means, sigmas, rng, nll_current, nll_proposed, v_current,
                        v_proposed, accepted, counter, jump_buffer, nparameters, nsources, jump_width, nexpected,
                        n_mc, source_id, norms, debug_mode=False):
        """finish_nll_jump_pick_combo launched together with the zeroing the next evaluation would start with
        (histograms and normalisations are cleared afterwards; see include/sxmc_hip.h)."""
        capi.call("sxmc_group_finish_step_async", self._g, ptr(stream), int(npartial_sums), ptr(sums), ptr(means),
                  ptr(sigmas), ptr(rng), ptr(nll_current), ptr(nll_proposed), ptr(v_current), ptr(v_proposed),
                  ptr(accepted), ptr(counter), ptr(jump_buffer), int(nparameters), int(nsources), ptr(jump_width),
                  ptr(nexpected), ptr(n_mc), ptr(source_id), ptr(norms), int(bool(debug_mode)))

    def EvalFinished(self):
        capi.call("sxmc_group_synchronize", self._g)

    def Profile(self, enable=True, capacity=4096):
        capi.call("sxmc_group_profile", self._g, int(bool(enable)), int(capacity))

    def ProfileRead(self):
        ms, n = C.c_double(0), C.c_int(0)
        capi.call("sxmc_group_profile_read", self._g, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def AlgorithmicBytes(self):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        capi.call("sxmc_group_algorithmic_bytes", self._g, C.byref(a), C.byref(b), C.byref(c))
        return dict(fill_read=a.value, hist=b.value, event=c.value)

    def close(self):
        if getattr(self, "_g", None):
            capi.load().sxmc_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGroup:
    """2-4 chains over the same sample tables stepped TOGETHER: one fill pass streams the tables once for all of
    them (sxmc_multigroup_*, include/sxmc_hip.h).  chains: objects with the MCMC attributes (sxmc_amd.mcmc.MCMC)."""

    def __init__(self, chains):
        self.chains = list(chains)
        arr = (C.c_void_p * len(self.chains))(*[c.group._g for c in self.chains])
        mg = C.c_void_p(0)
        capi.call("sxmc_multigroup_create", arr, len(self.chains), C.byref(mg))
        self._mg = mg
        self._args = (capi.StepArgs * len(self.chains))()

    def SetJointStepEnd(self, enable):
        """False: the chains' step ends are launched chain by chain instead of two launches for the set (the chains are
        the same either way; measurement / tests)."""
        capi.call("sxmc_multigroup_set_joint_step_end", self._mg, int(bool(enable)))

    def StepAsync(self, stream, debug_mode=False):
        for a, m in zip(self._args, self.chains):
            a.d_means, a.d_sigmas, a.d_rng = ptr(m.parameter_means).value, ptr(m.parameter_sigma).value, ptr(m.rngs).value
            a.d_nll_current, a.d_nll_proposed = ptr(m.current_nll).value, ptr(m.proposed_nll).value
            a.d_v_current, a.d_v_proposed = ptr(m.current_vector).value, ptr(m.proposed_vector).value
            a.d_accepted, a.d_counter = ptr(m.accept_counter).value, ptr(m.jump_counter).value
            a.d_jump_buffer, a.nparameters, a.nsources = ptr(m.jump_buffer).value, m.nparameters, m.nsources
            a.d_jump_width, a.d_nexpected, a.d_n_mc = ptr(m.jump_width).value, ptr(m.nexpected).value, ptr(m.n_mc).value
            a.d_source_id, a.d_norms = ptr(m.source_id).value, ptr(m.normalizations).value
            a.debug_mode = int(bool(debug_mode))
        capi.call("sxmc_multigroup_step_async", self._mg, ptr(stream), C.cast(self._args, C.c_void_p))

    def LookaheadStepAsync(self, stream, v_lookahead, norms_lookahead, cap=None, debug_mode=False):
        """One pass of the look-ahead walk (sxmc_multigroup_lookahead_step_async): chains[0] is THE chain, chains[1]'s
        evaluators are bound to v_lookahead / norms_lookahead.  cap: device int32, the jump counter at which to stop."""
        a, m = self._args[0], self.chains[0]
        a.d_means, a.d_sigmas, a.d_rng = ptr(m.parameter_means).value, ptr(m.parameter_sigma).value, ptr(m.rngs).value
        a.d_nll_current, a.d_nll_proposed = ptr(m.current_nll).value, ptr(m.proposed_nll).value
        a.d_v_current, a.d_v_proposed = ptr(m.current_vector).value, ptr(m.proposed_vector).value
        a.d_accepted, a.d_counter = ptr(m.accept_counter).value, ptr(m.jump_counter).value
        a.d_jump_buffer, a.nparameters, a.nsources = ptr(m.jump_buffer).value, m.nparameters, m.nsources
        a.d_jump_width, a.d_nexpected, a.d_n_mc = ptr(m.jump_width).value, ptr(m.nexpected).value, ptr(m.n_mc).value
        a.d_source_id, a.d_norms = ptr(m.source_id).value, ptr(m.normalizations).value
        a.debug_mode = int(bool(debug_mode))
        capi.call("sxmc_multigroup_lookahead_step_async", self._mg, ptr(stream), C.cast(self._args, C.c_void_p),
                  ptr(v_lookahead), ptr(norms_lookahead), ptr(cap) if cap is not None else None)

    def close(self):
        if getattr(self, "_mg", None):
            capi.load().sxmc_multigroup_destroy(self._mg)
            self._mg = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
