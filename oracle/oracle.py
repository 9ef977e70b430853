"""ctypes loader for the CPU parity oracle (oracle/libsxmc_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else:
the product package (sxmc_amd/) must never import this module.  See sxmc_oracle.h for what the
oracle restates and its pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SXMC_ORACLE_LIB selects another build of the same oracle (the ASan + UBSan build: tests/test_plan_cpu.py)
_LIB_PATH = os.environ.get("SXMC_ORACLE_LIB") or os.path.join(_HERE, "libsxmc_oracle.so")

MAX_SYST_PARS = 8
SHIFT, SCALE, RESOLUTION_SCALE, CTSCALE = 0, 1, 2, 3   # pdfz.h:111-116
TYPE_BY_NAME = {"shift": SHIFT, "scale": SCALE, "resolution_scale": RESOLUTION_SCALE,
                "ctscale": CTSCALE}


class SystT(C.Structure):
    _fields_ = [("type", C.c_short), ("obs", C.c_short), ("extra_field", C.c_short),
                ("npars", C.c_short), ("pars", C.c_short * MAX_SYST_PARS)]


def build():
    """Compile the oracle with gcc (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsxmc_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_hist_geometry.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def make_systs(systs):
    """systs: list of dicts {type, obs, extra_field (or true_obs), pars:[...]} -> ctypes array."""
    arr = (SystT * max(1, len(systs)))()
    for i, s in enumerate(systs):
        t = s["type"]
        arr[i].type = TYPE_BY_NAME[t] if isinstance(t, str) else int(t)
        arr[i].obs = int(s["obs"])
        arr[i].extra_field = int(s.get("extra_field", s.get("true_obs", 0)))
        pars = list(s["pars"])
        assert len(pars) <= MAX_SYST_PARS
        arr[i].npars = len(pars)
        for k, p in enumerate(pars):
            arr[i].pars[k] = int(p)
    return arr


class HistGeometry:
    def __init__(self, lower, upper, nbins):
        self.lower = np.ascontiguousarray(lower, dtype=np.float64)
        self.upper = np.ascontiguousarray(upper, dtype=np.float64)
        self.nbins = np.ascontiguousarray(nbins, dtype=np.int32)
        self.nobs = len(self.nbins)
        self.bin_stride = np.zeros(self.nobs, dtype=np.int32)
        vol = C.c_double(0)
        self.total_nbins = lib().oracle_hist_geometry(
            self.nobs, _p(self.lower, C.c_double), _p(self.upper, C.c_double),
            _p(self.nbins, C.c_int), _p(self.bin_stride, C.c_int), C.byref(vol))
        self.bin_volume = vol.value


def set_eval_points(geom, points, dataset=0):
    points = np.ascontiguousarray(points, dtype=np.float32).reshape(-1)
    assert points.size % (geom.nobs + 1) == 0
    n = points.size // (geom.nobs + 1)
    rb = np.empty(n, dtype=np.int32)
    lib().oracle_set_eval_points(
        C.c_size_t(n), _p(points, C.c_float), geom.nobs, _p(geom.lower, C.c_double),
        _p(geom.upper, C.c_double), _p(geom.nbins, C.c_int), _p(geom.bin_stride, C.c_int),
        C.c_uint(dataset), _p(rb, C.c_int))
    return rb


def bin_samples(geom, samples, nfields, systs, params, param_stride=1, nthreads=1):
    """samples: row-major float32 [n, nfields]; params: float64 array already offset.
    Returns (bins uint32[B], norm int)."""
    samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
    assert samples.size % nfields == 0
    n = samples.size // nfields
    params = np.ascontiguousarray(params, dtype=np.float64)
    bins = np.zeros(geom.total_nbins, dtype=np.uint32)
    norm = C.c_uint(0)
    sarr = make_systs(systs)
    args = [C.c_size_t(n), _p(samples, C.c_float), geom.nobs, int(nfields),
            _p(geom.bin_stride, C.c_int), _p(geom.nbins, C.c_int),
            _p(geom.lower, C.c_double), _p(geom.upper, C.c_double),
            len(systs), sarr, _p(params, C.c_double), int(param_stride),
            geom.total_nbins, _p(bins, C.c_uint), C.byref(norm)]
    if nthreads > 1:
        lib().oracle_bin_samples_mt(int(nthreads), *args)
    else:
        lib().oracle_bin_samples(*args)
    return bins, norm.value


def eval_pdf(read_bins, bins, norm, bin_volume, out=None, offset=0, stride=1):
    read_bins = np.ascontiguousarray(read_bins, dtype=np.int32)
    n = read_bins.size
    if out is None:
        out = np.zeros(offset + n * stride, dtype=np.float32)
    nrm = C.c_uint(int(norm))
    sub = out[offset:]
    lib().oracle_eval_pdf(C.c_size_t(n), _p(read_bins, C.c_int),
                          _p(np.ascontiguousarray(bins, dtype=np.uint32), C.c_uint),
                          C.byref(nrm), C.c_double(bin_volume), _p(sub, C.c_float), int(stride))
    return out


def nll_event_chunks(lut, pars, ne, ns, nexpected, n_mc, source_id, norms):
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    nexpected = np.ascontiguousarray(nexpected, dtype=np.float64)
    n_mc = np.ascontiguousarray(n_mc, dtype=np.uint32)
    source_id = np.ascontiguousarray(source_id, dtype=np.int16)
    norms = np.ascontiguousarray(norms, dtype=np.uint32)
    sums = np.zeros(1, dtype=np.float64)
    lib().oracle_nll_event_chunks(_p(lut, C.c_float), _p(pars, C.c_double), C.c_size_t(ne),
                                  C.c_size_t(ns), _p(nexpected, C.c_double), _p(n_mc, C.c_uint),
                                  _p(source_id, C.c_short), _p(norms, C.c_uint),
                                  _p(sums, C.c_double))
    return sums


def nll_event_reduce(sums):
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    tot = np.zeros(1, dtype=np.float64)
    lib().oracle_nll_event_reduce(C.c_size_t(sums.size), _p(sums, C.c_double),
                                  _p(tot, C.c_double))
    return tot


def nll_total(pars, nsignals, nsources, means, sigmas, events_total, nexpected, n_mc,
              source_id, norms):
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    means = np.ascontiguousarray(means, dtype=np.float64)
    sigmas = np.ascontiguousarray(sigmas, dtype=np.float64)
    ev = np.ascontiguousarray(events_total, dtype=np.float64).reshape(1)
    nexpected = np.ascontiguousarray(nexpected, dtype=np.float64)
    n_mc = np.ascontiguousarray(n_mc, dtype=np.uint32)
    source_id = np.ascontiguousarray(source_id, dtype=np.int16)
    norms = np.ascontiguousarray(norms, dtype=np.uint32)
    out = np.zeros(1, dtype=np.float64)
    lib().oracle_nll_total(C.c_size_t(pars.size), _p(pars, C.c_double), C.c_size_t(nsignals),
                           C.c_size_t(nsources), _p(means, C.c_double), _p(sigmas, C.c_double),
                           _p(ev, C.c_double), _p(nexpected, C.c_double), _p(n_mc, C.c_uint),
                           _p(source_id, C.c_short), _p(norms, C.c_uint), _p(out, C.c_double))
    return out[0]


def full_nll(lut, pars, ne, ns, nsources, means, sigmas, nexpected, n_mc, source_id, norms):
    """MCMC::nll (mcmc.cpp:390-415) on the CPU path: chunks -> reduce -> total."""
    sums = nll_event_chunks(lut, pars, ne, ns, nexpected, n_mc, source_id, norms)
    tot = nll_event_reduce(sums)
    return nll_total(pars, ns, nsources, means, sigmas, tot, nexpected, n_mc, source_id, norms), tot[0]


def jump_decider(u, nll_current, nll_proposed, v_current, v_proposed, accepted, counter,
                 jump_buffer, debug_mode=False):
    """In-place on the numpy arrays passed (float64 / int32 / float32)."""
    n = v_current.size
    lib().oracle_jump_decider(C.c_double(u), _p(nll_current, C.c_double),
                              _p(nll_proposed, C.c_double), _p(v_current, C.c_double),
                              _p(v_proposed, C.c_double), C.c_uint(n), _p(accepted, C.c_int),
                              _p(counter, C.c_int), _p(jump_buffer, C.c_float),
                              int(bool(debug_mode)))


def pick_new_vector(z, jump_width, current):
    z = np.ascontiguousarray(z, dtype=np.float64)
    jump_width = np.ascontiguousarray(jump_width, dtype=np.float32)
    current = np.ascontiguousarray(current, dtype=np.float64)
    out = np.zeros_like(current)
    lib().oracle_pick_new_vector(int(current.size), _p(z, C.c_double), _p(jump_width, C.c_float),
                                 _p(current, C.c_double), _p(out, C.c_double))
    return out
