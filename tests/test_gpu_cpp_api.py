"""Runs the C++ API test driver (tests/cpp/test_cpp_api.cpp): the reference's known-answer tests and
an MCMC walk written against the C++ mirror of the reference interface, in the reference's spelling
(hemi::Array, HEMI_KERNEL_LAUNCH, pdfz::EvalHist)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_cpp_api")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


def test_cpp_driver_builds_and_refuses_to_run_without_a_gpu():
    # not a GPU test: the header-only C++ layer compiles with plain g++ against the C ABI, and the
    # product has no CPU path -- without a device the driver says so and stops
    build()
    assert os.path.exists(BIN)
    from sxmc_amd import capi
    if capi.device_count() == 0:
        r = subprocess.run([BIN], capture_output=True, text=True)
        assert r.returncode == 2 and "no GPU" in r.stdout


@pytest.mark.gpu
def test_cpp_api_on_gpu():
    build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(r.stdout[-4000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_bench_driver_on_gpu():
    """tests/cpp/bench_cpp.cpp: BASELINE config 3's shape built and walked entirely through the C++ host layer
    (build_pdfz, sxmc::MCMC on its own stream, graph-replayed steps), here at 1 % of the samples."""
    import json
    build()
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "bench_cpp"), "--scale", "0.01", "--steps", "300",
                        "--graph-steps", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["steps"] == 300 and out["nsignals"] == 12 and out["steps_per_sec"] > 100
    assert 0 < out["accepted"] < 300


@pytest.mark.gpu
def test_cpp_bench_driver_multi_rank_rehearsal_on_gpu():
    """bench_cpp --device-list 0,0 --host-staging: sxmc::ensemble_multi_gpu with two device threads on this box's one
    card (everything but the RCCL call: RCCL refuses two ranks on one card), and the same with one rank through RCCL;
    the set-up lock's waiting and holding times are in the record."""
    import json
    build()
    exe = os.path.join(ROOT, "tests", "cpp", "bench_cpp")
    common = ["--scale", "0.01", "--no-walk", "--graph-steps", "8", "--experiments", "4", "--exp-steps", "300",
              "--chains", "2", "--sets", "1"]
    r = subprocess.run([exe] + common + ["--device-list", "0,0", "--host-staging"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ranks"] == 2 and out["experiments"] == 4 and out["rccl_nranks"] == 0
    assert out["exchange"].startswith("host staging") and out["gathered_floats"] == 4 * 15 * 4
    assert len(out["setup_locks"]) == 1 and out["setup_locks"][0]["acquisitions"] > 0
    r = subprocess.run([exe] + common + ["--devices", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    one = json.loads(r.stdout.strip().splitlines()[-1])
    assert one["rccl_nranks"] == 1 and one["rccl_devices"] == [0] and one["exchange"].startswith("ncclAllGather")
    # same experiments, same seeds: the medians do not depend on how the experiments were sharded
    assert one["median_upper_limit_source0"] == out["median_upper_limit_source0"]
    # a rank count RCCL cannot form on this box fails loudly instead of hanging
    r = subprocess.run([exe] + common + ["--device-list", "0,0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "RCCL communicators" in r.stderr
