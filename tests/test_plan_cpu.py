"""CPU, no device: the host side under sanitizers (SURVEY.md section 5 asks for them on the CPU build; GPU ASan is
not available on the pool).

* tests/cpp/test_plan.cpp -- the host planners of libsxmc_hip.so (sxmc_amd/csrc/sxmc_plan.h: work partitions, sparse
  tables, bucket layouts, per-bucket event tables, event classes, SetEvalPoints' loop) built from randomized shapes
  and walked the way the kernels index them; a plain build and one under AddressSanitizer + UndefinedBehaviorSanitizer.
* the CPU oracle's own known-answer tests re-run against oracle/libsxmc_oracle_asan.so (same sources, -fsanitize=
  address,undefined), so that the checker everything else is compared with is itself free of out-of-bounds reads and
  undefined arithmetic on the reference's fixtures.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _make(target, where=CPP):
    subprocess.check_call(["make", "-s", "-C", where, target])


@pytest.mark.parametrize("exe", ["test_plan", "test_plan_asan"])
def test_host_planners_device_free(exe):
    _make(exe)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(CPP, exe)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "9 tests, 0 failed" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_planner_header_needs_no_hip():
    """sxmc_plan.h is what the library's host side (sxmc_launch_plan.cpp, sxmc_evaluator.cpp) uploads from: it must stay free of HIP and of library state, or the
    device-free test above stops covering what runs in production."""
    text = open(os.path.join(ROOT, "sxmc_amd", "csrc", "sxmc_plan.h")).read()
    code = "\n".join(line.split("//")[0] for line in text.splitlines())      # (comments may name HIP)
    assert "hip" not in code.lower() and "#include <hip" not in text
    src = "".join(open(os.path.join(ROOT, "sxmc_amd", "csrc", f)).read()
                  for f in ("sxmc_launch_plan.cpp", "sxmc_evaluator.cpp", "sxmc_host.h"))
    for fn in ("build_partition", "interleaved_segments", "apportion_workgroups", "build_sparse_tables", "eval_point_bins",
               "bucket_granules", "bucket_key_offsets", "bucketed_layout", "bucket_tables", "event_classes"):
        assert "sxplan::" + fn in src, fn + " is not what the library calls"


def test_oracle_known_answers_under_asan_ubsan():
    _make("libsxmc_oracle_asan.so", os.path.join(ROOT, "oracle"))
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc's libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, SXMC_ORACLE_LIB=os.path.join(ROOT, "oracle", "libsxmc_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_oracle_nll.py")],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "failed" not in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
