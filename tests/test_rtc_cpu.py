"""No GPU needed: the fill-kernel template the library specialises at run time (hiprtc) compiles for gfx950 for
the shapes and programs the GPU tests use -- built-in programs, polynomial systematics, the empty program, bucketed
tables and the sparse counting over runs."""
import ctypes as C

import numpy as np
import pytest

from sxmc_amd import capi

SHIFT, SCALE, RES, CTSCALE = 0, 1, 2, 3


def op(type_, obs_slot, extra_slot=0, npars=0):
    return type_ | (obs_slot << 4) | (extra_slot << 8) | (npars << 12)


@pytest.mark.parametrize("nobs,nslot,lds,prew,runs,ops", [
    (3, 4, 1, 0, 0, [op(SHIFT, 1), op(SCALE, 0), op(RES, 0, 3), op(CTSCALE, 2)]),     # C3 + ctscale(c)
    (2, 3, 1, 3, 0, [op(SHIFT, 1), op(SCALE, 0), op(RES, 0, 2)]),                      # C3 bucketed
    (2, 3, 0, 3, 1, [op(SHIFT, 1), op(SCALE, 0), op(RES, 0, 2)]),                      # C5: sparse counting over runs
    (2, 2, 1, 0, 0, [op(SHIFT, 0, 0, 3), op(SCALE, 1, 0, 2)]),                         # polynomials
    (5, 7, 0, 0, 0, []),                                                               # no systematics, 5-D
    (1, 3, 1, 5, 0, [op(SHIFT, 2), op(SCALE, 0), op(RES, 0, 1)]),                      # C3 bucketed, r ordered
    (0, 1, 1, 5, 0, [op(SHIFT, 0)]),                                                   # bench_pdfz ordered: nothing streamed
    (2, 4, 1, 5, 0, [op(CTSCALE, 3), op(SCALE, 3), op(SCALE, 0), op(RES, 1, 2)]),      # two ops on the ordered observable
    (1, 3, 0, 5, 1, [op(SHIFT, 2), op(SCALE, 0), op(RES, 0, 1)]),                      # C5, r ordered: sparse counting over runs
    (1, 3, 0, 5, 0, [op(SHIFT, 2), op(SCALE, 0), op(RES, 0, 1)]),                      # ... and its dense evaluation
    (1, 3, 1, 6, 0, [op(SHIFT, 0), op(SCALE, 2), op(RES, 2, 1)]),                      # C3 bucketed, e BOXED, r streamed
    (1, 3, 1, 6, 0, [op(RES, 2, 1), op(CTSCALE, 0), op(SHIFT, 2), op(RES, 2, 1)]),     # two resolution scales on the boxed one
])
def test_runtime_specialisation_compiles_without_a_gpu(nobs, nslot, lds, prew, runs, ops):
    lib = capi.load()
    arr = np.asarray(ops, dtype=np.uint32)
    n = C.c_size_t(0)
    rc = lib.sxmc_rtc_compile_check(nobs, nslot, lds, prew, runs, capi.ptr(arr), len(ops), C.byref(n))
    assert rc == 0, capi.last_error()
    assert n.value > 1000            # a gfx950 code object came out


def test_ordered_kernel_refuses_a_systematic_that_is_not_monotone():
    lib = capi.load()
    arr = np.asarray([op(SHIFT, 1, 0, 2)], dtype=np.uint32)   # a 2-coefficient (polynomial) shift on the ordered slot
    rc = lib.sxmc_rtc_compile_check(1, 2, 1, 5, 0, capi.ptr(arr), 1, None)
    assert rc != 0 and "not a monotone systematic" in capi.last_error()


def test_boxed_kernel_refuses_programs_it_cannot_box():
    lib = capi.load()
    # a resolution scale on the STREAMED observable (its truth field would have to be streamed too)
    arr = np.asarray([op(RES, 0, 1), op(SCALE, 2)], dtype=np.uint32)
    rc = lib.sxmc_rtc_compile_check(1, 3, 1, 6, 0, capi.ptr(arr), 2, None)
    assert rc != 0 and "not a program the boxed form can run" in capi.last_error()
    # a polynomial on the boxed observable
    arr = np.asarray([op(SHIFT, 0), op(SCALE, 2, 0, 2), op(RES, 2, 1)], dtype=np.uint32)
    rc = lib.sxmc_rtc_compile_check(1, 3, 1, 6, 0, capi.ptr(arr), 3, None)
    assert rc != 0


def test_runtime_specialisation_reports_a_bad_shape():
    lib = capi.load()
    arr = np.asarray([op(RES, 0, 5)], dtype=np.uint32)        # truth field slot 5 of a 2-slot kernel
    rc = lib.sxmc_rtc_compile_check(1, 2, 1, 0, 0, capi.ptr(arr), 1, None)
    assert rc != 0 and "slot out of range" in capi.last_error()


@pytest.mark.parametrize("nobs,nslot,prew,nchains,ops", [
    (2, 3, 3, 4, [op(SHIFT, 1), op(SCALE, 0), op(RES, 0, 2)]),        # config 3, bucketed, four chains per pass
    (2, 3, 0, 2, [op(SHIFT, 1), op(RES, 0, 2)]),                      # rows, two chains
    (1, 1, 3, 3, [op(SHIFT, 0, 0, 3)]),                               # a polynomial, three chains
    (1, 3, 5, 4, [op(SHIFT, 2), op(SCALE, 0), op(RES, 0, 1)]),        # config 3 with r ordered, four chains
    (0, 1, 5, 2, [op(SCALE, 0)]),                                     # one ordered observable alone, two chains
])
def test_lockstep_kernel_compiles_without_a_gpu(nobs, nslot, prew, nchains, ops):
    lib = capi.load()
    arr = np.asarray(ops, dtype=np.uint32)
    n = C.c_size_t(0)
    rc = lib.sxmc_rtc_compile_check_lockstep(nobs, nslot, prew, nchains, capi.ptr(arr), len(ops), C.byref(n))
    assert rc == 0, capi.last_error()
    assert n.value > 1000
