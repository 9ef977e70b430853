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


def declared_symbols(measure_section=False):
    """Entry points include/sxmc_hip.h declares: the product's, or those inside its `#ifdef SXMC_MEASURE` section."""
    text = open(os.path.join(ROOT, "include", "sxmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = re.findall(r"#ifdef SXMC_MEASURE\n(.*?)#endif", text, flags=re.S)
    assert len(inside) == 1
    if not measure_section:
        text = text.replace(inside[0], "")
    return sorted(set(re.findall(r"\b(sxmc_[a-z0-9_]+)\s*\(", inside[0] if measure_section else text)))


def exported(path):
    out = os.popen("nm -D --defined-only %s" % path).read()
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln}


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 55
    for n in names:
        assert getattr(lib, n) is not None, n


def test_ctypes_table_matches_header():
    assert set(declared_symbols()) == set(capi.SIGNATURES) | set(capi.STRING_GETTERS)
    assert set(declared_symbols(measure_section=True)) == set(capi.MEASURE_SIGNATURES)


def test_product_library_has_no_measurement_hooks():
    """The entry points whose header comment says RESULTS ARE WRONG, the test hooks and anything else named *debug* live
    in libsxmc_hip_measure.so only; the product exports exactly what the header's product part declares."""
    product = exported(os.path.join(ROOT, "sxmc_amd", "csrc", "libsxmc_hip.so"))
    hooks = set(declared_symbols(measure_section=True))
    assert hooks == {"sxmc_group_set_debug_mode", "sxmc_debug_pow_int", "sxmc_debug_philox_dump",
                     "sxmc_measure_set_gated_step", "sxmc_measure_stream_fork"}
    assert not (hooks & product)
    assert not [n for n in product if ("debug" in n or "measure" in n) and n.startswith("sxmc_")]
    assert {n for n in product if n.startswith("sxmc_")} == set(declared_symbols())
    # no kernel of the product carries a hook either: sx_dbg() is the constant 0 there
    src = open(os.path.join(ROOT, "sxmc_amd", "csrc", "fill_kernels.inc.h")).read()
    assert "return SXMC_MEASURE ? dbg_arg : 0u;" in src and "#define SXMC_MEASURE 0" in src


def test_measurement_build_exports_the_hooks_and_the_product_abi():
    path = capi.MEASURE_LIB_PATH
    assert os.path.exists(path), "__graft_entry__.build() builds libsxmc_hip_measure.so"
    measure = exported(path)
    assert set(declared_symbols(measure_section=True)) <= measure
    assert set(declared_symbols()) <= measure


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
