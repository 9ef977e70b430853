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


FIT = """
{
  // a fit in the reference's schema (config.cpp:19-297): three signals, two observables + a cut, three systematics
  "fit": {"nexperiments": 3, "nsteps": 400, "burnin_fraction": 0.2, "seed": 42, "confidence": 0.9,
          "signals": ["bkg_b", "sig", "bkg_a"], "observables": ["energy", "radius"], "cuts": ["valid"],
          "signal_name": "sig"},
  "pdfs": {
    "observables": {
      "energy": {"title": "E", "field": "e", "bins": 16, "min": 0.0, "max": 10.0},
      "radius": {"title": "R", "field": "r", "bins": 12, "min": 0.0, "max": 6.0},
      "valid": {"title": "fit valid", "field": "valid", "bins": 1, "min": 0.5, "max": 1.5}
    },
    "systematics": {
      "r_shift": {"title": "s", "type": "shift", "observable_field": "r", "mean": [0.0], "sigma": [0.05]},
      "e_scale": {"title": "s", "type": "scale", "observable_field": "e", "mean": [0.0], "sigma": [0.01]},
      "e_res": {"title": "s", "type": "resolution_scale", "observable_field": "e", "truth_field": "e_true",
                "mean": [0.0], "sigma": [0.05]}
    }
  },
  "signals": {
    "sig": {"title": "S", "filename": "sig.npz", "dataset": 0, "rate": 400.0, "systematics": ["e_scale", "r_shift", "e_res"]},
    "bkg_a": {"title": "A", "filename": "a.npz", "dataset": 0, "scale": 40.0, "systematics": ["e_scale", "r_shift", "e_res"]},
    "bkg_b": {"title": "B", "filename": "b.npz", "dataset": 0, "rate": 900.0, "source": "bkg",
              "systematics": ["e_scale", "r_shift", "e_res"]}
  },
  "sources": {"bkg": {"mean": 1.0, "sigma": 0.2}}
}
"""


@pytest.mark.gpu
def test_cpp_bench_driver_from_a_fit_configuration(tmp_path):
    """bench_cpp --config fit.json: sxmc::load_config (the reference's JSON schema, tables from .npz files, cuts) ->
    sxmc::ensemble_multi_gpu, end to end in C++, through RCCL with the ranks this box has and as
    a two-rank rehearsal; the same experiments give the same medians either way."""
    import json

    import numpy as np
    build()
    rng = np.random.default_rng(8)
    for name, n, mu in (("sig.npz", 60000, 5.0), ("a.npz", 40000, 2.0), ("b.npz", 80000, 7.5)):
        e_true = rng.normal(mu, 1.5, n)
        np.savez(tmp_path / name, e_true=e_true, e=(e_true + rng.normal(0, 0.3, n)).astype(np.float32),
                 r=6.0 * rng.uniform(0, 1, n) ** (1 / 3), valid=rng.integers(0, 4, n) > 0, junk=np.arange(n))
    (tmp_path / "fit.json").write_text(FIT)
    exe = os.path.join(ROOT, "tests", "cpp", "bench_cpp")
    outs = []
    (tmp_path / "out").mkdir()
    for extra in (["--devices", "1", "--output-dir", str(tmp_path / "out")], ["--device-list", "0,0", "--host-staging"]):
        r = subprocess.run([exe, "--config", str(tmp_path / "fit.json"), "--chains", "2", "--sets", "1", "--graph-steps", "8"]
                           + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
        assert lines[0]["driver"].startswith("sxmc::load_config") and lines[0]["signals"] == 3
        assert lines[0]["nfields"] == 4 and lines[0]["experiments"] == 3 and lines[0]["steps"] == 400
        assert 0 < lines[0]["rows_total"] < 180000            # the cut on `valid` removed about a quarter
        outs.append(lines[-1])
        assert lines[-1]["experiments"] == 3 and lines[-1]["steps_each"] == 400
        assert lines[-1]["gathered_floats"] == 3 * (3 + 3) * 4   # 3 sources (bkg, bkg_a, sig) + 3 systematic parameters
    # --output-dir: every experiment's chain as <prefix>_<k>.npz, columns = parameter names + likelihood (sxmc.cpp:130-141)
    for k in range(3):
        with np.load(tmp_path / "out" / ("lspace_%d.npz" % k)) as z:
            assert list(z.files) == ["bkg_a", "bkg", "sig", "e_scale_0", "r_shift_0", "e_res_0", "likelihood"]
            assert z["likelihood"].shape == z["sig"].shape and 100 < z["sig"].shape[0] <= 400
            assert np.all(np.isfinite(z["likelihood"]))
            chain = np.stack([z[f] for f in z.files], axis=1).astype(np.float32)
        # ... and beside it what sxmc.cpp:100-101 prints per experiment (LikelihoodSpace::print_best_fit +
        # print_correlations): the C++ text equals the Python forms' on the saved chain, character for character
        from sxmc_amd import ensemble
        names = ["bkg_a", "bkg", "sig", "e_scale_0", "r_shift_0", "e_res_0"]
        cl = float(np.float32(0.9))
        want = ensemble.format_best_fit(names, ensemble.contour_intervals(chain, cl), chain[:, -1].min(), cl) + \
            ensemble.format_correlations(names, ensemble.correlation_matrix(chain))
        assert (tmp_path / "out" / ("lspace_%d.txt" % k)).read_text() == want
    assert outs[0]["rccl_nranks"] == 1 and outs[1]["ranks"] == 2
    assert outs[0]["median_upper_limit_source0"] == outs[1]["median_upper_limit_source0"]
    # configured data sets (sxmc.cpp:71-80): experiment i fits file i of every data set, clipped to the PDF boundaries
    # and the cuts, instead of a fake data set; a data set with too few files is refused before anything runs
    from sxmc_amd import io
    kept = []
    for i in range(3):
        n = 700 + 150 * i
        e = rng.normal(5.0, 2.5, n).astype(np.float32)
        np.savez(tmp_path / ("data%d.npz" % i), e=e, e_true=e, r=(6.0 * rng.uniform(0, 1, n) ** (1 / 3)).astype(np.float32),
                 valid=rng.integers(0, 5, n) > 0)
    cfg = json.loads("\n".join(line.split("//")[0] for line in FIT.splitlines()))
    cfg["data"] = {"0": [{"title": "d%d" % i, "filename": "data%d.npz" % i} for i in range(3)]}
    (tmp_path / "withdata.json").write_text(json.dumps(cfg))
    fc = io.load_config(str(tmp_path / "withdata.json"))
    w = io.build_workload(fc)
    kept = [io.load_data(fc, w, i).shape[0] for i in range(3)]
    assert all(0 < k < 700 + 150 * i for i, k in enumerate(kept))
    r = subprocess.run([exe, "--config", str(tmp_path / "withdata.json"), "--chains", "2", "--sets", "1", "--devices", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")][-1]
    assert last["data"] == "configured data sets" and last["nevents"] == kept
    cfg["data"]["0"].pop()
    (tmp_path / "short.json").write_text(json.dumps(cfg))
    r = subprocess.run([exe, "--config", str(tmp_path / "short.json"), "--devices", "1"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 1 and "data set 0 lists 2 file(s): experiment 2 has none" in r.stderr
    # a configuration the batched drivers cannot take is refused with the reason, not walked wrongly
    bad = json.loads("\n".join(line.split("//")[0] for line in FIT.splitlines()))
    bad["signals"]["sig"]["systematics"] = ["e_scale"]
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    r = subprocess.run([exe, "--config", str(tmp_path / "bad.json"), "--devices", "1"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 1 and "every signal to list every systematic" in r.stderr
