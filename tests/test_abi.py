"""CPU: the C-ABI library loads and exports every symbol include/sxmc_hip.h declares, the ctypes
table covers them all, and argument validation (which needs no GPU) behaves like the reference's
constructor checks (test_pdfz.cpp:42-73, test_pdfz_2d.cpp:8-39)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sxmc_amd import capi, pdfz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sxmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sxmc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 55
    for n in names:
        assert getattr(lib, n) is not None, n


def test_ctypes_table_matches_header():
    assert set(declared_symbols()) == set(capi.SIGNATURES) | set(capi.STRING_GETTERS)


def test_no_oracle_in_product():
    # the product must never import, link or call the oracle (parity claims depend on it)
    for base, _, files in os.walk(os.path.join(ROOT, "sxmc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                assert "oracle" not in open(os.path.join(base, f), errors="ignore").read().lower(), f
    out = os.popen("ldd %s" % capi.LIB_PATH).read()
    assert "oracle" not in out and "libamdhip64" in out


def test_constructor_validation_matches_reference(golden):
    for c in golden["ctor_errors"]:
        with pytest.raises(pdfz.Error):
            pdfz.EvalHist(np.zeros(c["nsamples_floats"], np.float32), c["nfields"], c["nobs"], c["lower"],
                          c["upper"], c["nbins"])


def test_constructor_messages():
    def msg(**kw):
        a = dict(samples=np.zeros(7, np.float32), nfields=1, nobservables=1, lower=[0.0], upper=[1.0], nbins=[2])
        a.update(kw)
        with pytest.raises(pdfz.Error) as e:
            pdfz.EvalHist(**a)
        return e.value.msg
    assert msg(nfields=2) == "Length of samples array is not divisible by number of fields."   # pdfz.cpp:65
    assert msg(nobservables=0) == "Number of observables in PDF is zero."                       # pdfz.cpp:69
    assert msg(nbins=[0]) == "Cannot make histogram with zero bins."                            # pdfz.cpp:218
    assert "MAX_NFIELDS" in msg(samples=np.zeros(22, np.float32), nfields=11)                   # pdfz.cpp:194
    assert "too large" in msg(samples=np.zeros(8, np.float32), nfields=4, nobservables=4, lower=[0.0] * 4,
                              upper=[1.0] * 4, nbins=[1000] * 4)


def test_launch_argument_validation_needs_no_gpu():
    lib = capi.load()
    assert lib.sxmc_launch_nll_event_reduce(2, 128, None, 10, None, None) == capi.ERR_INVALID
    assert b"one workgroup" in lib.sxmc_last_error()
    assert lib.sxmc_launch_nll_total(1, 2048, None, 1, None, 1, 1, None, None, None, None, None, None, None,
                                     None) == capi.ERR_INVALID
    assert lib.sxmc_hist_set_launch_config(None, 256, 1) == capi.ERR_INVALID


def test_rng_state_layout():
    assert C.sizeof(capi.RngState) == 32
