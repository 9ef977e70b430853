"""Python spelling of the reference's pdfz interface (src/pdfz.h) over the C ABI.

Same class names, method names, argument order and error behaviour as pdfz::Eval /
pdfz::EvalHist (pdfz.h:246-574): constructor validation raises `Error` where the reference
throws pdfz::Error; buffers are device arrays (capi.DeviceArray, or anything with data_ptr());
EvalAsync returns before completion and EvalFinished waits.  ROOT-returning methods
(CreateHistogram, RandomSample) are replaced by plain-array accessors (GetBins).
"""
import ctypes as C

import numpy as np

from . import capi


class Error(Exception):
    """pdfz::Error (pdfz.h:93-102)."""

    def __init__(self, msg):
        super().__init__(msg)
        self.msg = msg


class Systematic:
    SHIFT, SCALE, RESOLUTION_SCALE, CTSCALE = 0, 1, 2, 3   # pdfz.h:111-116

    def __init__(self, type_):
        self.type = type_


def _pars(pars):
    return [int(pars)] if np.isscalar(pars) else [int(p) for p in pars]


class ShiftSystematic(Systematic):
    """x' = x + p, p = sum p_i x^i (pdfz.h:145-157).  pars: parameter indices."""

    def __init__(self, obs, pars):
        super().__init__(Systematic.SHIFT)
        self.obs, self.pars = int(obs), _pars(pars)


class ScaleSystematic(Systematic):
    """x' = x (1 + p) (pdfz.h:168-180)."""

    def __init__(self, obs, pars):
        super().__init__(Systematic.SCALE)
        self.obs, self.pars = int(obs), _pars(pars)


class CosThetaScaleSystematic(Systematic):
    """x' = 1 + (x - 1)(1 + p) (pdfz.h:194-206)."""

    def __init__(self, obs, pars):
        super().__init__(Systematic.CTSCALE)
        self.obs, self.pars = int(obs), _pars(pars)


class ResolutionScaleSystematic(Systematic):
    """x' = x + p (x - x_true) (pdfz.h:218-233)."""

    def __init__(self, obs, true_obs, pars):
        super().__init__(Systematic.RESOLUTION_SCALE)
        self.obs, self.true_obs, self.pars = int(obs), int(true_obs), _pars(pars)


def _raise(rc):
    if rc == capi.ERR_INVALID:
        raise Error(capi.last_error())
    capi.check(rc)


class EvalHist:
    """pdfz::EvalHist (pdfz.h:402-574, pdfz.cpp:179-495)."""

    def __init__(self, samples, nfields, nobservables, lower, upper, nbins, dataset=0, optimize=True):
        lib = capi.load()
        self._h = None
        on_device = hasattr(samples, "data_ptr")
        if on_device:
            nfloats = int(samples.numel())
        else:
            samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
            nfloats = samples.size
        sp = capi.ptr(samples)
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        nbins = np.ascontiguousarray(nbins, dtype=np.int32)
        h = C.c_void_p(0)
        _raise(lib.sxmc_hist_create(sp, nfloats, int(on_device), int(nfields), int(nobservables),
                                    capi.ptr(lower), lower.size, capi.ptr(upper), upper.size,
                                    capi.ptr(nbins), nbins.size, int(dataset), C.byref(h)))
        self._h = h
        self.nfields, self.nobservables, self.dataset = int(nfields), int(nobservables), int(dataset)
        self._keep = {}

    @classmethod
    def Shared(cls, base):
        """A second evaluator over the SAME sample table as `base` (nothing copied; systematics copied):
        for concurrent chains / experiments on one GPU (sxmc_hist_create_shared)."""
        self = cls.__new__(cls)
        self._h = None
        h = C.c_void_p(0)
        _raise(capi.load().sxmc_hist_create_shared(base._h, C.byref(h)))
        self._h = h
        self.nfields, self.nobservables, self.dataset = base.nfields, base.nobservables, base.dataset
        self._keep = {"base": base}
        return self

    # -- Eval interface -------------------------------------------------------------------
    def SetEvalPoints(self, points):
        points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1)
        _raise(capi.load().sxmc_hist_set_eval_points(self._h, capi.ptr(points), points.size))

    def SetPDFValueBuffer(self, output, offset=0, stride=1):
        self._keep["pdf"] = output
        _raise(capi.load().sxmc_hist_set_pdf_value_buffer(self._h, capi.ptr(output), int(offset), int(stride)))

    def SetNormalizationBuffer(self, norm, offset=0):
        self._keep["norm"] = norm
        _raise(capi.load().sxmc_hist_set_normalization_buffer(self._h, capi.ptr(norm), int(offset)))

    def SetParameterBuffer(self, params, offset=0, stride=1):
        self._keep["params"] = params
        _raise(capi.load().sxmc_hist_set_parameter_buffer(self._h, capi.ptr(params), int(offset), int(stride)))

    def AddSystematic(self, syst):
        pars = np.asarray(syst.pars, dtype=np.int16)
        extra = getattr(syst, "true_obs", 0)
        _raise(capi.load().sxmc_hist_add_systematic(self._h, int(syst.type), int(syst.obs), int(extra),
                                                    pars.size, capi.ptr(pars)))

    def EvalAsync(self, do_eval_pdf=True):
        _raise(capi.load().sxmc_hist_eval_async(self._h, int(bool(do_eval_pdf))))

    def EvalFinished(self):
        _raise(capi.load().sxmc_hist_eval_finished(self._h))

    # -- replaces Optimize*: analytic launch sizing, optionally overridden -------------------
    def SetLaunchConfig(self, bin_threads=0, bin_blocks_per_cu=0):
        _raise(capi.load().sxmc_hist_set_launch_config(self._h, int(bin_threads), int(bin_blocks_per_cu)))

    def Optimize(self):
        pass

    # -- introspection ----------------------------------------------------------------------
    @property
    def total_nbins(self):
        v = C.c_int(0)
        _raise(capi.load().sxmc_hist_total_nbins(self._h, C.byref(v)))
        return v.value

    @property
    def bin_volume(self):
        v = C.c_double(0)
        _raise(capi.load().sxmc_hist_bin_volume(self._h, C.byref(v)))
        return v.value

    @property
    def nsamples(self):
        v = C.c_size_t(0)
        _raise(capi.load().sxmc_hist_nsamples(self._h, C.byref(v)))
        return v.value

    @property
    def npoints(self):
        v = C.c_size_t(0)
        _raise(capi.load().sxmc_hist_npoints(self._h, C.byref(v)))
        return v.value

    def GetBins(self):
        """Bin contents of the last evaluation (the array CreateHistogram reads, pdfz.cpp:511)."""
        out = np.empty(self.total_nbins, dtype=np.uint32)
        _raise(capi.load().sxmc_hist_get_bins(self._h, capi.ptr(out), out.size))
        return out

    def RandomSample(self, nobserved, seed, lowers=None, uppers=None):
        """EvalHist::RandomSample's sampling step on the device (pdfz.cpp:817-922): nobserved events drawn from
        the histogram of the last evaluation (EvalAsync(False) first), rows of nobservables + 1 floats."""
        out = np.empty((int(nobserved), self.nobservables + 1), dtype=np.float32)
        lo = None if lowers is None else np.ascontiguousarray(lowers, dtype=np.float32)
        hi = None if uppers is None else np.ascontiguousarray(uppers, dtype=np.float32)
        _raise(capi.load().sxmc_hist_random_sample(self._h, int(nobserved), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                   capi.ptr(lo), capi.ptr(hi), capi.ptr(out)))
        return out

    def GetReadBins(self):
        out = np.empty(self.npoints, dtype=np.int32)
        _raise(capi.load().sxmc_hist_get_read_bins(self._h, capi.ptr(out), out.size))
        return out

    def GetSamples(self):
        """pdfz.h:542-556: rows of nobservables + 1 floats (observables, dataset id)."""
        out = np.empty(self.nsamples * (self.nobservables + 1), dtype=np.float32)
        _raise(capi.load().sxmc_hist_get_samples(self._h, capi.ptr(out), out.size))
        return out

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            capi.load().sxmc_hist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
