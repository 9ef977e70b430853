"""ctypes binding of libsxmc_hip.so (the C ABI in include/sxmc_hip.h).

Plumbing only: every evaluation goes through the hand-written gfx950 kernels in the shared
library.  There is no CPU or PyTorch fallback -- if the library is missing or a call fails this
module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SXMC_HIP_LIB selects another build of the same library (A/B experiments on kernel variants)
LIB_PATH = os.environ.get("SXMC_HIP_LIB") or os.path.join(_HERE, "csrc", "libsxmc_hip.so")

OK, ERR_INVALID, ERR_HIP, ERR_STATE = 0, 1, 2, 3
MAX_NFIELDS, MAX_SYST, MAX_SYST_PARS = 10, 16, 8
SYST_SHIFT, SYST_SCALE, SYST_RESOLUTION_SCALE, SYST_CTSCALE = 0, 1, 2, 3


class SxmcError(RuntimeError):
    """A libsxmc_hip call failed.  `code` is one of ERR_INVALID / ERR_HIP / ERR_STATE."""

    def __init__(self, code, msg):
        super().__init__("sxmc_hip error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class StepArgs(C.Structure):
    """sxmc_step_args (include/sxmc_hip.h): the arguments of finish_nll_jump_pick_combo for one chain."""
    _fields_ = [("d_means", C.c_void_p), ("d_sigmas", C.c_void_p), ("d_rng", C.c_void_p),
                ("d_nll_current", C.c_void_p), ("d_nll_proposed", C.c_void_p), ("d_v_current", C.c_void_p),
                ("d_v_proposed", C.c_void_p), ("d_accepted", C.c_void_p), ("d_counter", C.c_void_p),
                ("d_jump_buffer", C.c_void_p), ("nparameters", C.c_int), ("nsources", C.c_size_t),
                ("d_jump_width", C.c_void_p), ("d_nexpected", C.c_void_p), ("d_n_mc", C.c_void_p),
                ("d_source_id", C.c_void_p), ("d_norms", C.c_void_p), ("debug_mode", C.c_int)]


class RngState(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("subsequence", C.c_uint64), ("offset", C.c_uint64),
                ("reserved", C.c_uint64)]


_vp, _i, _u, _sz, _d, _ull = C.c_void_p, C.c_int, C.c_uint, C.c_size_t, C.c_double, C.c_ulonglong
_pi, _psz, _pd, _pvp = C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_double), C.POINTER(C.c_void_p)

# name -> argument types.  Every function returns int except the two string getters.
SIGNATURES = {
    "sxmc_device_count": [_pi],
    "sxmc_set_device": [_i],
    "sxmc_get_device": [C.POINTER(C.c_int)],
    "sxmc_device_info": [_i, C.c_char_p, _pi, _psz, _pi, _pi],
    "sxmc_set_tracing": [_i],
    "sxmc_device_synchronize": [],
    "sxmc_device_pci_bus_id": [_i, C.c_char_p, _sz],
    "sxmc_mem_info": [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
    "sxmc_malloc": [_pvp, _sz],
    "sxmc_free": [_vp],
    "sxmc_host_alloc": [_pvp, _sz],
    "sxmc_host_free": [_vp],
    "sxmc_memcpy_h2d": [_vp, _vp, _sz],
    "sxmc_memcpy_d2h": [_vp, _vp, _sz],
    "sxmc_memcpy_d2d": [_vp, _vp, _sz],
    "sxmc_memcpy_h2d_async": [_vp, _vp, _sz, _vp],
    "sxmc_memcpy_d2h_async": [_vp, _vp, _sz, _vp],
    "sxmc_memset": [_vp, _i, _sz],
    "sxmc_stream_create": [_pvp],
    "sxmc_stream_create_nonblocking": [_pvp],
    "sxmc_stream_destroy": [_vp],
    "sxmc_stream_synchronize": [_vp],
    "sxmc_stream_query": [_vp, _pi],
    "sxmc_graph_begin_capture": [_vp],
    "sxmc_graph_end_capture": [_vp, _pvp],
    "sxmc_graph_launch": [_vp, _vp, _i],
    "sxmc_graph_destroy": [_vp],
    "sxmc_event_create": [_pvp],
    "sxmc_event_destroy": [_vp],
    "sxmc_event_record": [_vp, _vp],
    "sxmc_event_synchronize": [_vp],
    "sxmc_event_elapsed_ms": [_vp, _vp, C.POINTER(C.c_float)],
    "sxmc_hist_create": [_vp, _sz, _i, _i, _i, _vp, _sz, _vp, _sz, _vp, _sz, _u, _pvp],
    "sxmc_hist_create_shared": [_vp, _pvp],
    "sxmc_hist_destroy": [_vp],
    "sxmc_hist_add_systematic": [_vp, _i, _i, _i, _i, _vp],
    "sxmc_hist_set_eval_points": [_vp, _vp, _sz],
    "sxmc_hist_set_pdf_value_buffer": [_vp, _vp, _i, _i],
    "sxmc_hist_set_normalization_buffer": [_vp, _vp, _i],
    "sxmc_hist_set_parameter_buffer": [_vp, _vp, _i, _i],
    "sxmc_hist_eval_async": [_vp, _i],
    "sxmc_hist_eval_finished": [_vp],
    "sxmc_set_deferred_eval": [_i],
    "sxmc_set_lazy_finish": [_i],
    "sxmc_deferred_eval_stats": [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)],
    "sxmc_hist_total_nbins": [_vp, _pi],
    "sxmc_hist_bin_volume": [_vp, _pd],
    "sxmc_hist_nsamples": [_vp, _psz],
    "sxmc_hist_npoints": [_vp, _psz],
    "sxmc_hist_get_bins": [_vp, _vp, _sz],
    "sxmc_hist_get_read_bins": [_vp, _vp, _sz],
    "sxmc_hist_get_samples": [_vp, _vp, _sz],
    "sxmc_hist_random_sample": [_vp, _sz, _ull, _vp, _vp, _vp],
    "sxmc_hist_get_stream": [_vp, _pvp],
    "sxmc_hist_set_launch_config": [_vp, _i, _i],
    "sxmc_hist_set_optimize": [_vp, _i],
    "sxmc_hist_optimize": [_vp],
    "sxmc_hist_launch_info": [_vp, C.c_char_p, _sz],
    "sxmc_group_create": [_vp, _i, _pvp],
    "sxmc_group_destroy": [_vp],
    "sxmc_group_set_launch_config": [_vp, _i, _i],
    "sxmc_group_optimize": [_vp, _vp, _pi],
    "sxmc_group_set_partition": [_vp, _i],
    "sxmc_group_set_partition_teams": [_vp, _i],
    "sxmc_group_set_sparse": [_vp, _i],
    "sxmc_group_set_prebinning": [_vp, _i],
    "sxmc_group_set_bucketing": [_vp, _i],
    "sxmc_group_set_ordering": [_vp, _i],
    "sxmc_group_set_codes": [_vp, _i],
    "sxmc_group_set_boxes": [_vp, _i],
    "sxmc_group_set_box_limit": [_vp, C.c_double],
    "sxmc_group_adapt_fill_form": [_vp, _pi, _pi],
    "sxmc_group_set_fill_form": [_vp, _i],
    "sxmc_group_fill_form": [_vp, _pi],
    "sxmc_group_codes_info": [_vp, _pi, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)],
    "sxmc_group_codes_windows": [_vp, _i, _pi, _pd, _pd],
    "sxmc_group_set_codes_queue_log": [_vp, _i],
    "sxmc_group_set_runtime_kernels": [_vp, _i],
    "sxmc_group_launch_info": [_vp, C.c_char_p, _sz],
    "sxmc_group_set_lut_output": [_vp, _i],
    "sxmc_group_eval_async": [_vp, _i, _vp],
    "sxmc_group_eval_nll_async": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pi],
    "sxmc_group_mcmc_step_async": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp,
                                   _vp, _vp, _vp, _vp, _i],
    "sxmc_group_step_async": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _sz, _vp,
                              _vp, _vp, _vp, _vp, _i],
    "sxmc_group_set_tail_kernel": [_vp, _i],
    "sxmc_group_set_cooperative_step_end": [_vp, _i],
    "sxmc_group_set_fused_step": [_vp, _i],
    "sxmc_group_step_end_timeouts": [_vp, _vp, C.POINTER(C.c_uint)],
    "sxmc_rtc_compile_check": [_i, _i, _i, _i, _i, _vp, _i, _psz],
    "sxmc_rtc_compile_check_lockstep": [_i, _i, _i, _i, _vp, _i, _psz],
    "sxmc_group_last_step_launches": [_vp, _pi],
    "sxmc_multigroup_create": [_vp, _i, _pvp],
    "sxmc_multigroup_destroy": [_vp],
    "sxmc_multigroup_step_async": [_vp, _vp, _vp],
    "sxmc_multigroup_set_joint_step_end": [_vp, C.c_int],
    "sxmc_multigroup_lookahead_step_async": [_vp, _vp, _vp, _vp, _vp, _vp],
    "sxmc_lookahead_begin": [_vp, _i, _vp, _vp, _vp, _vp],
    "sxmc_group_lookahead_supported": [_vp, _pi],
    "sxmc_group_finish_step_async": [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _sz,
                                     _vp, _vp, _vp, _vp, _vp, _i],
    "sxmc_group_synchronize": [_vp],
    "sxmc_group_profile": [_vp, _i, _i],
    "sxmc_group_profile_read": [_vp, _pd, _pi],
    "sxmc_group_algorithmic_bytes": [_vp, _pd, _pd, _pd],
    "sxmc_launch_init_device_rngs": [_i, _i, _vp, _i, _ull, _vp],
    "sxmc_launch_pick_new_vector": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "sxmc_launch_jump_decider": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp, _vp],
    "sxmc_launch_nll_event_chunks": [_i, _i, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp],
    "sxmc_launch_nll_event_reduce": [_i, _i, _vp, _sz, _vp, _vp],
    "sxmc_launch_nll_total": [_i, _i, _vp, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sxmc_launch_finish_nll_jump_pick_combo": [_i, _i, _vp, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i],
    "sxmc_comm_init_all": [_vp, _i, _pvp],
    "sxmc_comm_unique_id": [C.c_char_p, _sz],
    "sxmc_comm_init_rank": [C.c_char_p, _sz, _i, _i, _pvp],
    "sxmc_comm_rank": [_vp, _pi, _pi],
    "sxmc_comm_query": [_vp, _pi, _pi, _pi],
    "sxmc_comm_async_error": [_vp, _pi],
    "sxmc_comm_abort": [_vp],
    "sxmc_comm_allgather_f32": [_vp, _vp, _vp, _sz, _vp],
    "sxmc_comm_destroy": [_vp],
}
STRING_GETTERS = ("sxmc_last_error", "sxmc_version", "sxmc_comm_last_error")
# The measurement build (libsxmc_hip_measure.so: include/sxmc_hip.h, "MEASUREMENT BUILD ONLY") exports these besides;
# the product library must NOT (tests/test_abi.py).
MEASURE_SIGNATURES = {
    "sxmc_group_set_debug_mode": [_vp, _i],
    "sxmc_debug_pow_int": [_vp, _i, _i, _vp],
    "sxmc_debug_philox_dump": [_vp, _vp, _i],
    "sxmc_measure_set_gated_step": [_vp, _vp, _i],
    "sxmc_measure_stream_fork": [_vp, _vp],
}
MEASURE_LIB_PATH = os.path.join(_HERE, "csrc", "libsxmc_hip_measure.so")

_lib = None


def load():
    """Load libsxmc_hip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libsxmc_hip.so not found at %s: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (there is no fallback path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    for name, argtypes in MEASURE_SIGNATURES.items():   # (present when SXMC_HIP_LIB names the measurement build)
        if hasattr(lib, name):
            getattr(lib, name).argtypes = argtypes
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def is_measurement_build():
    """Does the loaded library carry the kernels' measurement hooks (SXMC_HIP_LIB = .../libsxmc_hip_measure.so)?"""
    return hasattr(load(), "sxmc_group_set_debug_mode")


_measure = None


def measure_lib():
    """The measurement build as a SECOND handle beside the product library, for the tests' look inside (known answers of
    the generator and of the rounded powers): its hooks take plain device pointers, whichever library allocated them."""
    global _measure
    if _measure is None:
        if not os.path.exists(MEASURE_LIB_PATH):
            raise ImportError("libsxmc_hip_measure.so not found at %s: __graft_entry__.build() builds it"
                              % MEASURE_LIB_PATH)
        lib = C.CDLL(MEASURE_LIB_PATH)
        for name, argtypes in MEASURE_SIGNATURES.items():
            getattr(lib, name).argtypes = argtypes
            getattr(lib, name).restype = C.c_int
        lib.sxmc_last_error.restype = C.c_char_p
        _measure = lib
    return _measure


def last_error():
    return load().sxmc_last_error().decode()


def check(rc):
    if rc != OK:
        raise SxmcError(rc, last_error())


def call(name, *args):
    check(getattr(load(), name)(*args))


def ptr(x):
    """Device/host address of x as c_void_p: DeviceArray, numpy array, int address, torch tensor or None."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, DeviceArray):
        return C.c_void_p(x.ptr)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, C.c_void_p):
        return x
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError("cannot take the address of %r" % type(x))


def device_count():
    n = C.c_int(0)
    rc = load().sxmc_device_count(C.byref(n))
    return n.value if rc == OK else 0


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus, lds, clk = C.c_int(0), C.c_int(0), C.c_int(0)
    hbm = C.c_size_t(0)
    call("sxmc_device_info", device, name, C.byref(cus), C.byref(hbm), C.byref(lds), C.byref(clk))
    bus = C.create_string_buffer(64)
    pci = bus.value.decode() if load().sxmc_device_pci_bus_id(device, bus, 64) == OK else None
    return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=hbm.value,
                lds_bytes_per_cu=lds.value, clock_khz=clk.value, pci_bus_id=pci)


def synchronize():
    call("sxmc_device_synchronize")


def new_stream(nonblocking=True):
    """A HIP stream handle (int address).  Non-blocking streams do not synchronise with the legacy
    default stream, so kernels of different chains can overlap."""
    s = C.c_void_p(0)
    call("sxmc_stream_create_nonblocking" if nonblocking else "sxmc_stream_create", C.byref(s))
    return s.value


class Graph:
    """A recorded launch sequence (HIP graph).  `with Graph.capture(stream) as g: ...launches on stream...`
    records instead of executing; g.launch(times) replays."""

    def __init__(self, stream):
        self.stream, self._g = stream, None

    @classmethod
    def capture(cls, stream):
        return cls(stream)

    def __enter__(self):
        call("sxmc_graph_begin_capture", ptr(self.stream))
        return self

    def __exit__(self, exc_type, exc, tb):
        g = C.c_void_p(0)
        rc = load().sxmc_graph_end_capture(ptr(self.stream), C.byref(g))
        if exc_type is None:
            check(rc)
            self._g = g
        return False

    def launch(self, times=1):
        call("sxmc_graph_launch", self._g, ptr(self.stream), int(times))

    def close(self):
        if self._g:
            load().sxmc_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray:
    """A typed device buffer (the device half of the reference's hemi::Array<T>).

    `DeviceArray(np_array)` uploads; `DeviceArray.empty(n, dtype)` allocates; `.get()` downloads.
    """

    def __init__(self, host=None, n=None, dtype=None):
        if host is not None:
            host = np.ascontiguousarray(host)
            n, dtype = host.size, host.dtype
        self.dtype = np.dtype(dtype)
        self.size = int(n)
        p = C.c_void_p(0)
        call("sxmc_malloc", C.byref(p), self.nbytes)
        self.ptr = p.value
        if host is not None and self.size:
            call("sxmc_memcpy_h2d", C.c_void_p(self.ptr), ptr(host), self.nbytes)

    @classmethod
    def empty(cls, n, dtype):
        return cls(n=n, dtype=dtype)

    @classmethod
    def zeros(cls, n, dtype):
        a = cls(n=n, dtype=dtype)
        if a.size:
            call("sxmc_memset", C.c_void_p(a.ptr), 0, a.nbytes)
        return a

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.size
        if self.size:
            call("sxmc_memcpy_h2d", C.c_void_p(self.ptr), ptr(host), self.nbytes)

    def get(self):
        out = np.empty(self.size, dtype=self.dtype)
        if self.size:
            call("sxmc_memcpy_d2h", ptr(out), C.c_void_p(self.ptr), self.nbytes)
        return out

    def offset_ptr(self, nelem):
        return self.ptr + int(nelem) * self.dtype.itemsize

    def free(self):
        if getattr(self, "ptr", None):
            load().sxmc_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
