"""CPU: the ROOT-free input/output layer (sxmc_amd/io.py): the reference's JSON schema
(config/example.json, with its defects repaired: missing comma at line 42, `true_field` ->
`truth_field`, `output_file` -> `output_prefix`), cut / column-packing rules, tables and chains."""
import json

import numpy as np

from sxmc_amd import io

EXAMPLE = """
{
  // fit parameters (config/example.json:2-14)
  "fit": {
    "nexperiments": 1, "nsteps": 1000, "burnin_fraction": 0.1, "output_prefix": "fit_test",
    "signal_name": "sig_a", "confidence": 0.9,
    "signals": ["sig_a", "sig_b"],
    "observables": ["energy"],
    "cuts": ["radius"]
  },
  "pdfs": {
    "observables": {
      "energy": {"title": "Energy (MeV)", "units": "MeV", "field": "energy", "bins": 10, "min": 5.0, "max": 15.0},
      "radius": {"title": "Radius", "field": "radius", "bins": 1, "min": 0.0, "max": 10.0} /* a cut */
    },
    "systematics": {
      "energy_scale": {"title": "E scale", "type": "scale", "observable_field": "energy", "mean": [0.0], "sigma": [0.01]},
      "energy_resolution": {"title": "E res", "type": "resolution_scale", "observable_field": "energy",
                            "truth_field": "mc_energy", "mean": [0.0, 0.0], "sigma": [0.001, 0.0]}
    }
  },
  "signals": {
    "sig_a": {"title": "A", "filename": "a.npz", "dataset": 0, "rate": 50.0,
              "systematics": ["energy_scale", "energy_resolution"]},
    "sig_b": {"title": "B", "filename": "b.npz", "dataset": 0, "scale": 20.0, "source": "shared",
              "systematics": ["energy_scale", "energy_resolution"]}
  },
  "sources": {"shared": {"mean": 1.0, "sigma": 0.25, "fixed": false}}
}
"""


def test_comments_are_stripped_outside_strings():
    assert json.loads(io.strip_comments('{"a": "x//y", /* c */ "b": 1 // t\n}')) == {"a": "x//y", "b": 1}


def test_config_schema_field_order_and_parameter_indices():
    fc = io.load_config(EXAMPLE)
    # (confidence, burnin_fraction, rates and scales are floats in the reference: asFloat())
    assert fc.nsteps == 1000 and fc.confidence == float(np.float32(0.9)) and fc.error_type == "contour"
    # sample fields: observables, extra truth fields, DATASET (config.cpp:153-194)
    assert fc.sample_fields == ["energy", "mc_energy", "DATASET"]
    es, er = fc.systematics
    assert es["pidx"] == [0] and er["pidx"] == [1, 2] and er["npars"] == 2        # config.cpp:119
    assert es["observable_field_index"] == 0 and er["truth_field_index"] == 1
    assert [s["name"] for s in fc.sources] == ["sig_a", "shared"]                 # a signal is its own source
    assert fc.sources[1]["sigma"] == np.float32(0.25)
    assert fc.signals[1]["source"]["index"] == 1 and fc.signals[1]["scale"] == 20.0
    assert fc.cuts[0]["field"] == "radius"


def test_cuts_are_inclusive_and_columns_are_packed():
    data = np.array([[6.0, 9.0, 5.5], [7.0, 10.0, 6.5], [8.0, 10.5, 7.5], [9.0, -0.1, 8.5], [1.0, 0.0, 1.5]],
                    np.float32)                                                   # energy, radius, mc_energy
    out = io.read_dataset_to_samples(data, ["energy", "radius", "mc_energy"], 3,
                                     ["energy", "mc_energy", "DATASET"], [("radius", 0.0, 10.0)])
    # radius 10.0 passes (data > upper rejects), 10.5 and -0.1 fail; energy is not a cut here
    assert out.tolist() == [[6.0, 5.5, 3.0], [7.0, 6.5, 3.0], [1.0, 1.5, 3.0]]


def test_last_cut_on_a_field_wins_and_data_sets_need_no_truth_field():
    data = np.array([[6.0, 9.0, 5.5], [7.0, 10.0, 6.5], [8.0, 10.5, 7.5], [9.0, -0.1, 8.5], [1.0, 0.0, 1.5]], np.float32)
    fields = ["energy", "radius", "mc_energy"]
    # two cuts on one field: signal.cpp:57-69 overwrites the field's bounds cut by cut -- only the LAST one is applied
    # (the first alone would keep nothing)
    out = io.read_dataset_to_samples(data, fields, 3, ["energy", "mc_energy", "DATASET"],
                                     [("radius", 100.0, 200.0), ("radius", 0.0, 10.0)])
    assert out.tolist() == [[6.0, 5.5, 3.0], [7.0, 6.5, 3.0], [1.0, 1.5, 3.0]]
    # real data carries no Monte Carlo truth branch: as a data set (required = the observables) the field is zeros ...
    real = data[:, :2]
    out = io.read_dataset_to_samples(real, fields[:2], 0, ["energy", "mc_energy", "DATASET"], [], required=1)
    assert out[:, 0].tolist() == [6.0, 7.0, 8.0, 9.0, 1.0] and not out[:, 1].any() and not out[:, 2].any()
    # ... but an observable must be there, and an MC table must carry every field
    with pytest.raises(KeyError):
        io.read_dataset_to_samples(real[:, 1:], ["radius"], 0, ["energy", "mc_energy", "DATASET"], [], required=1)
    with pytest.raises(KeyError):
        io.read_dataset_to_samples(real, fields[:2], 0, ["energy", "mc_energy", "DATASET"], [])


def test_tables_and_chains_roundtrip(tmp_path):
    m = np.arange(12, dtype=np.float32).reshape(4, 3)
    io.write_table(tmp_path / "t.npz", m, ["a", "b", "c"])
    got, fields = io.read_table(tmp_path / "t.npz")
    assert fields == ["a", "b", "c"] and np.array_equal(got, m)
    np.savez(tmp_path / "mixed.npz", i=np.arange(3), b=np.array([True, False, True]), d=np.ones(3))
    got, fields = io.read_table(tmp_path / "mixed.npz")                           # int / bool / double -> float32
    assert got.dtype == np.float32 and got[:, 1].tolist() == [1.0, 0.0, 1.0]
    io.write_chain(tmp_path / "c.npz", ["x", "likelihood"], np.array([[1.0, 2.0], [3.0, 4.0]], np.float32))
    with np.load(tmp_path / "c.npz") as z:
        assert z["likelihood"].tolist() == [2.0, 4.0]


def test_workload_from_config(tmp_path):
    rng = np.random.default_rng(0)
    for name, n in (("a.npz", 1000), ("b.npz", 2000)):
        mc = rng.uniform(4, 16, n).astype(np.float32)
        io.write_table(tmp_path / name, np.stack([mc + rng.normal(0, 0.5, n).astype(np.float32),
                                                  rng.uniform(0, 12, n).astype(np.float32), mc], axis=1),
                       ["energy", "radius", "mc_energy"])
    (tmp_path / "fit.json").write_text(EXAMPLE)
    fc = io.load_config(str(tmp_path / "fit.json"))
    w = io.build_workload(fc)
    assert w.nobs == 1 and w.nbins == [10] and w.nsources == 2 and w.nparameters == 5
    a, b = w.signals
    assert a.nfields == 3 and a.samples.shape[0] < 1000 and a.n_mc == 1000        # n_mc counts events before cuts
    # scale -> nexpected: -1 / scale kept in a FLOAT (config.cpp:221), times -n_mc in double (signal.cpp:31-35)
    assert b.nexpected == float(np.float32(-1.0) / np.float32(20.0)) * -2000.0 and abs(b.nexpected - 100.0) < 1e-5
    assert np.all(a.samples[:, 2] == 0) and a.samples[:, 0].max() > 15            # observables are not cuts for MC
    assert w.systematics[1] == dict(type="resolution_scale", obs=0, true_obs=1, pars=[1, 2])
    assert list(w.parameter_sigmas()) == [0.0, 0.25, 0.01, 0.001, 0.0]
    assert w.parameter_names == ["sig_a", "shared", "energy_scale_0", "energy_resolution_0", "energy_resolution_1",
                                 "likelihood"]


# ---- the C++ input layer (sxmc_amd/include/sxmc/config.h) against this one, on the same files --------------------
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP = os.path.join(ROOT, "tests", "cpp", "config_dump")
DUMP_ASAN = os.path.join(ROOT, "tests", "cpp", "config_dump_asan")

# signals named so that KEY order (what jsoncpp 0.6 iterates in: config.cpp:97) differs from file order and from
# fit.signals order; different systematics lists; a source shared by two signals; a cut; two data sets; mixed dtypes
KEYED = """
{
  "fit": {"nexperiments": 3, "nsteps": 500, "seed": 1234567890123, "confidence": 0.9, "error_type": "projection",
          "signals": ["zeta", "alpha", "Mid"], "observables": ["energy", "radius"], "cuts": ["fitvalid"],
          "signal_name": "alpha", "debug_mode": true},
  "pdfs": {
    "observables": {
      "energy": {"title": "E", "field": "e", "bins": 12, "min": 0.1, "max": 10.3},
      "radius": {"title": "R", "field": "r", "bins": 7, "min": 0.0, "max": 6.0},
      "fitvalid": {"title": "ok", "field": "valid", "bins": 1, "min": 0.5, "max": 1.5}
    },
    "systematics": {
      "r_shift": {"title": "s", "type": "shift", "observable_field": "r", "mean": [0.0], "sigma": [0.05]},
      "e_scale": {"title": "s", "type": "scale", "observable_field": "e", "mean": [0.0, 0.1], "sigma": [0.01, 0.0], "fixed": true},
      "e_res": {"title": "s", "type": "resolution_scale", "observable_field": "e", "truth_field": "e_true", "mean": [0.0]},
      "c_ct": {"title": "s", "type": "ctscale", "observable_field": "r", "mean": [0.0], "sigma": [0.3]}
    }
  },
  "signals": {
    "zeta": {"title": "Z", "filename": "z.npz", "dataset": 1, "rate": 12.3, "systematics": ["r_shift", "e_res"]},
    "alpha": {"title": "A", "filename": "a.npz", "dataset": 0, "scale": 3.0, "source": "common",
              "systematics": ["e_scale", "r_shift"]},
    "Mid": {"title": "M", "filename": "m.npz", "dataset": 0, "rate": 7, "source": "common", "mean": 2.0,
            "systematics": ["c_ct"]}
  },
  "sources": {"common": {"mean": 1.5, "sigma": 0.1}},
  "data": {"1": [{"title": "d1", "filename": "d1.npz"}], "0": [{"title": "d0", "filename": "d0.npz"}, {"title": "d0b", "filename": "z.npz"}]}
}
"""


def _write_keyed_files(tmp_path):
    rng = np.random.default_rng(5)
    for name, n in (("z.npz", 700), ("a.npz", 1300), ("m.npz", 10), ("d1.npz", 90), ("d0.npz", 60)):
        e_true = rng.uniform(0, 11, n)
        # the archive's field order differs from the sample-field order; dtypes: double, float, int, bool
        np.savez(tmp_path / name, valid=rng.integers(0, 3, n), r=rng.uniform(-0.5, 6.5, n).astype(np.float32),
                 e_true=e_true, junk=rng.integers(0, 2, n).astype(bool), e=(e_true + rng.normal(0, 0.4, n)))
    # d1.npz again as REAL data would come: no Monte Carlo truth branch (e_true), which a resolution_scale systematic
    # of the fit names -- a data set needs the observables only (signal.cpp:72-77 + GetSamples)
    with np.load(tmp_path / "d1.npz") as f:
        kept = {k: f[k] for k in f.files if k != "e_true"}
    np.savez(tmp_path / "d1.npz", **kept)
    (tmp_path / "fit.json").write_text(KEYED)
    return str(tmp_path / "fit.json")


def _python_summary(path):
    fc = io.load_config(path)
    out = {"sample_fields": fc.sample_fields, "nexperiments": fc.nexperiments, "nsteps": fc.nsteps, "seed": fc.seed,
           "error_type": fc.error_type, "confidence": fc.confidence, "burnin_fraction": fc.burnin_fraction,
           "debug_mode": fc.debug_mode, "signal_name": fc.signal_name,
           "sources": [(s["name"], s["index"], float(s["mean"]), float(s["sigma"]), s["fixed"]) for s in fc.sources],
           # (type as pdfz::Systematic::Type numbers it, pdfz.h:109-116: SHIFT, SCALE, RESOLUTION_SCALE, CTSCALE)
           "systematics": [(s["name"], {"shift": 0, "scale": 1, "resolution_scale": 2, "ctscale": 3}[s["type"]],
                            s["observable_field_index"],
                            s["truth_field_index"], s["npars"], s["fixed"], s["pidx"], s["means"], s["sigmas"])
                           for s in fc.systematics],
           "observables": [(o["name"], o["field_index"], o["bins"], float(o["lower"]), float(o["upper"]))
                           for o in fc.observables]}
    cuts = [(c["field"], c["lower"], c["upper"]) for c in fc.cuts]
    sigs = []
    for s in fc.signals:
        table, fields = io.read_table(os.path.join(fc.base_dir, s["filename"]))
        n_mc = table.shape[0]
        samples = io.read_dataset_to_samples(table, fields, s["dataset"], fc.sample_fields, cuts)
        nexp = s["rate"] if s["rate"] is not None else float(np.float32(-1.0) / np.float32(s["scale"])) * (-1.0 * n_mc)
        sigs.append((s["name"], s["dataset"], s["source"]["index"], nexp, n_mc, _fingerprint(samples)))
    out["signals"] = sigs
    return out, fc


def _fingerprint(a):
    flat = np.ascontiguousarray(a, np.float32).ravel()
    total = 0.0
    for x in flat.astype(np.float64):          # (the C++ side adds in index order, in double)
        total += x
    return (int(a.shape[0]), total, int(np.bitwise_xor.reduce(flat.view(np.uint32))) if flat.size else 0)


@pytest.mark.parametrize("exe", [DUMP, DUMP_ASAN])
def test_cpp_load_config_matches_python_on_the_same_files(tmp_path, exe):
    if not os.path.exists(exe):
        pytest.skip("tests/cpp is not built")
    path = _write_keyed_files(tmp_path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, path], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cpp = json.loads(r.stdout)
    py, fc = _python_summary(path)
    # key order: Mid < alpha < zeta (strcmp), so the union numbers c_ct first; sources: common (Mid), then zeta
    assert [s["name"] for s in cpp["systematics"]] == ["c_ct", "e_scale", "r_shift", "e_res"]
    assert [s["pidx"] for s in cpp["systematics"]] == [[0], [1, 2], [3], [4]]
    assert [(s["name"], s["index"]) for s in cpp["sources"]] == [("common", 0), ("zeta", 1)]
    assert cpp["sample_fields"] == py["sample_fields"] == ["e", "r", "e_true", "DATASET"]
    for k in ("nexperiments", "nsteps", "seed", "error_type", "debug_mode", "signal_name"):
        assert cpp[k] == py[k], k
    assert cpp["confidence"] == pytest.approx(py["confidence"], rel=1e-8)
    assert [(s["name"], s["index"], s["mean"], s["sigma"], s["fixed"]) for s in cpp["sources"]] == \
        [(n, i, pytest.approx(m, rel=1e-8), pytest.approx(sg, rel=1e-8), f) for n, i, m, sg, f in py["sources"]]
    assert [(s["name"], s["type"], s["observable_field_index"], s["truth_field_index"], s["npars"], s["fixed"],
             s["pidx"], s["means"], s["sigmas"]) for s in cpp["systematics"]] == [tuple(x) for x in py["systematics"]]
    assert [(o["name"], o["field_index"], o["bins"]) for o in cpp["observables"]] == [x[:3] for x in py["observables"]]
    for got, want in zip(cpp["signals"], py["signals"]):
        name, dataset, src, nexp, n_mc, (rows, total, xor) = want
        assert (got["name"], got["dataset"], got["source_index"], got["n_mc"]) == (name, dataset, src, n_mc)
        assert got["nexpected"] == nexp                       # bit for bit, incl. the float in -1 / scale
        assert (got["table"]["rows"], got["table"]["sum"], got["table"]["xor"]) == (rows, total, xor)
        assert rows < n_mc                                    # the cut removed something
    # data sets: clipped to the PDF boundaries, observables + dataset id
    w = io.build_workload(fc)
    # (experiment i fits file i of every data set, sxmc.cpp:71-80: data set 0 lists two files, data set 1 one)
    data = io.load_data(fc, w, 0)
    assert [len(cpp["data"][k]) for k in ("0", "1")] == [2, 1]
    for ds in (0, 1):
        rows, total, xor = _fingerprint(data[data[:, -1] == ds])
        t = cpp["data"][str(ds)][0]
        assert (t["rows"], t["sum"], t["xor"]) == (rows, total, xor)
    assert np.all(np.diff(data[:, -1]) >= 0)                   # data set 0's rows, then data set 1's
    with pytest.raises(ValueError, match="data set 1 lists 1 file"):
        io.load_data(fc, w, 1)
    assert cpp["same_systematics_everywhere"] is False


def test_cpp_json_reader_accepts_what_the_reference_accepts(tmp_path):
    if not os.path.exists(DUMP_ASAN):
        pytest.skip("tests/cpp is not built")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    doc = '{ // c\n "a": [1, -2.5e3, true, null, "x\\"y // not a comment", {"k": {}}], /* c */ "b": {"z": 1, "A": 2}\n}'
    (tmp_path / "d.json").write_text(doc)
    r = subprocess.run([DUMP_ASAN, "--json", str(tmp_path / "d.json")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == json.loads(io.strip_comments(doc))
    assert list(json.loads(r.stdout)["b"]) == ["A", "z"]       # members in key order, like jsoncpp 0.6
    # the defect of the reference's own config/example.json (a missing comma) is reported the
    # way jsoncpp reports it, not swallowed
    (tmp_path / "bad.json").write_text('{"a": {"x": 1 "y": 2}}')
    r = subprocess.run([DUMP_ASAN, "--json", str(tmp_path / "bad.json")], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "missing ',' or '}' in object declaration" in r.stderr and "line 1" in r.stderr
    for bad in ('{"a": [1, 2}', '{"a": "unterminated}', '{"a": 1} trailing', '/* open', '', "[" * 100000, '{"a":' * 5000):
        (tmp_path / "bad.json").write_text(bad)
        r = subprocess.run([DUMP_ASAN, "--json", str(tmp_path / "bad.json")], capture_output=True, text=True, env=env)
        assert r.returncode == 1 and "JSON parse error" in r.stderr, bad


def test_cpp_table_reader_rejects_what_it_cannot_read(tmp_path):
    if not os.path.exists(DUMP_ASAN):
        pytest.skip("tests/cpp is not built")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    cfg = json.loads(io.strip_comments(EXAMPLE))

    def run():
        (tmp_path / "fit.json").write_text(json.dumps(cfg))
        return subprocess.run([DUMP_ASAN, str(tmp_path / "fit.json")], capture_output=True, text=True, env=env)
    good = dict(energy=np.ones(4, np.float32), radius=np.ones(4, np.float32), mc_energy=np.ones(4, np.float32))
    np.savez(tmp_path / "b.npz", **good)
    np.savez_compressed(tmp_path / "a.npz", **good)
    r = run()
    assert r.returncode == 1 and "compressed" in r.stderr
    np.savez(tmp_path / "a.npz", energy=np.ones(4), radius=np.ones(5), mc_energy=np.ones(4))
    r = run()
    assert r.returncode == 1 and "differ in length" in r.stderr
    np.savez(tmp_path / "a.npz", energy=np.ones((2, 2)), radius=np.ones(4), mc_energy=np.ones(4))
    r = run()
    assert r.returncode == 1 and "1-D" in r.stderr
    np.savez(tmp_path / "a.npz", energy=np.ones(4), radius=np.ones(4))            # a sample field is missing
    r = run()
    assert r.returncode == 1 and "mc_energy" in r.stderr
    (tmp_path / "a.npz").write_bytes(b"PK\x03\x04 truncated")
    r = run()
    assert r.returncode == 1
    np.savez(tmp_path / "a.npz", **good)                                          # and the good file loads
    r = run()
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["signals"][0]["n_mc"] == 4
    # a 2-D float32 .npy with the field names in the configuration
    np.save(tmp_path / "a.npy", np.array([[0, 1, 2], [3, 4, 5], [6, 10.5, 8], [9, 10, 11]], np.float32))
    cfg["signals"]["sig_a"]["filename"] = "a.npy"
    cfg["signals"]["sig_a"]["fields"] = ["energy", "radius", "mc_energy"]
    r = run()
    assert r.returncode == 0, r.stderr
    t = json.loads(r.stdout)["signals"][0]["table"]
    assert t["rows"] == 3                      # radius 10 passes (bounds inclusive), 10.5 does not


def test_cpp_input_layer_survives_damaged_files_under_asan(tmp_path):
    """Byte-level damage to a valid configuration and to valid .npz / .npy tables (truncation, flipped bytes, overwritten
    length fields): sxmc::load_config must either load or refuse with a message -- under AddressSanitizer +
    UndefinedBehaviorSanitizer never read out of bounds, overflow or crash."""
    if not os.path.exists(DUMP_ASAN):
        pytest.skip("tests/cpp is not built")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    rng = np.random.default_rng(99)
    cfg = json.loads(io.strip_comments(EXAMPLE))
    good = dict(energy=np.linspace(5, 15, 64).astype(np.float32), radius=np.linspace(0, 12, 64), mc_energy=np.arange(64))
    np.savez(tmp_path / "b.npz", **good)
    np.savez(tmp_path / "a.npz", **good)
    (tmp_path / "fit.json").write_text(json.dumps(cfg))
    r = subprocess.run([DUMP_ASAN, str(tmp_path / "fit.json")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    pristine = {"fit.json": (tmp_path / "fit.json").read_bytes(), "a.npz": (tmp_path / "a.npz").read_bytes()}
    np.save(tmp_path / "a.npy", np.arange(30, dtype=np.float32).reshape(10, 3))
    pristine["a.npy"] = (tmp_path / "a.npy").read_bytes()
    outcomes = {0: 0, 1: 0}
    for trial in range(240):
        name = ("fit.json", "a.npz", "a.npy")[trial % 3]
        data = bytearray(pristine[name])
        kind = rng.integers(0, 4)
        if kind == 0:
            data = data[: rng.integers(0, len(data))]                       # truncated
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))):
                data[rng.integers(0, len(data))] = rng.integers(0, 256)     # flipped bytes
        elif kind == 2:
            at = rng.integers(0, max(1, len(data) - 8))
            data[at:at + 4] = (0xFFFFFFFF if rng.integers(0, 2) else int(rng.integers(0, 1 << 31))).to_bytes(4, "little")
        else:
            at = rng.integers(0, len(data))
            data[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))   # inserted bytes
        for k, v in pristine.items():
            (tmp_path / k).write_bytes(bytes(data) if k == name else v)
        if name == "a.npy":          # (the .npy is only read when the configuration points at it)
            c2 = json.loads(pristine["fit.json"])
            c2["signals"]["sig_a"].update(filename="a.npy", fields=["energy", "radius", "mc_energy"])
            (tmp_path / "fit.json").write_text(json.dumps(c2))
        r = subprocess.run([DUMP_ASAN, str(tmp_path / "fit.json")], capture_output=True, text=True, env=env, timeout=60,
                           errors="replace")      # (a damaged file's bytes may come back in the message)
        assert r.returncode in (0, 1), (trial, name, kind, r.returncode, r.stderr[-1500:])
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (trial, name, kind, r.stderr[-1500:])
        if r.returncode == 1:
            assert r.stderr.startswith("config_dump: ")                      # refused with a message
        outcomes[r.returncode] += 1
    assert outcomes[1] > 60 and outcomes[0] > 5, outcomes       # most damage is noticed; harmless damage still loads


def test_cpp_chain_files_are_read_by_numpy_and_by_the_python_layer(tmp_path):
    """sxmc::write_chain_npz (the "ls" ntuple of sxmc.cpp:130-141 without ROOT): numpy.load verifies every member's
    CRC-32, so a clean load of identical columns says the archive is a well-formed .npz; and the C++ reader reads back
    what the C++ writer wrote."""
    if not os.path.exists(DUMP_ASAN):
        pytest.skip("tests/cpp is not built")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = np.random.default_rng(4)
    names = ["source_a", "bkg", "e_scale_0", "likelihood"]
    chain = rng.normal(size=(1237, 4)).astype(np.float32)
    io.write_chain(tmp_path / "py.npz", names, chain)
    r = subprocess.run([DUMP_ASAN, "--rewrite", str(tmp_path / "py.npz"), str(tmp_path / "cpp.npz")], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    with np.load(tmp_path / "cpp.npz") as z:
        assert list(z.files) == names
        for i, n in enumerate(names):
            assert z[n].dtype == np.float32 and np.array_equal(z[n], chain[:, i])
    got, fields = io.read_table(tmp_path / "cpp.npz")
    assert fields == names and np.array_equal(got, chain)
    r = subprocess.run([DUMP_ASAN, "--rewrite", str(tmp_path / "cpp.npz"), str(tmp_path / "cpp2.npz")], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "cpp2.npz").read_bytes() == (tmp_path / "cpp.npz").read_bytes()
    # an empty chain (no kept rows) is still a valid file
    io.write_chain(tmp_path / "e.npz", names, np.zeros((0, 4), np.float32))
    r = subprocess.run([DUMP_ASAN, "--rewrite", str(tmp_path / "e.npz"), str(tmp_path / "e2.npz")], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stderr
    with np.load(tmp_path / "e2.npz") as z:
        assert z["likelihood"].shape == (0,)


def test_fit_samples_reloads_a_saved_chain_instead_of_walking(tmp_path):
    """fit.samples (config.cpp:51, sxmc.cpp:84-94): the intervals come from a saved likelihood space, no walk, no
    device -- in Python (io.run_config) and in C++ (bench_cpp --config), same numbers."""
    exe = os.path.join(ROOT, "tests", "cpp", "bench_cpp")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp is not built")
    from tests.test_intervals import synthetic_chain
    chain = synthetic_chain(3, 3000, -348086.3)
    names = ["sig_a", "shared", "energy_scale_0", "energy_resolution_0", "likelihood"]
    io.write_chain(tmp_path / "saved.npz", names, chain)
    for error_type in ("contour", "projection"):
        cfg = json.loads(io.strip_comments(EXAMPLE))
        cfg["fit"].update(samples="saved.npz", signal_name="shared", error_type=error_type)
        (tmp_path / "fit.json").write_text(json.dumps(cfg))
        import io as pyio
        said = pyio.StringIO()
        iv, limits, nm = io.run_config(str(tmp_path / "fit.json"), report=said)
        assert nm == names and iv.shape == (1, 4, 4) and limits == [float(iv[0, 1, 2])]
        r = subprocess.run([exe, "--config", str(tmp_path / "fit.json")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        # best fit + correlation matrix as sxmc.cpp:100-101 prints them: the same text from both (the fitted mean of a
        # projection interval agrees to 1e-6 only, so that block is compared for the contour estimator)
        assert "-- Best fit --" in r.stderr and "-- Correlation matrix --" in r.stderr
        if error_type == "contour":
            assert r.stderr == said.getvalue(), "\n" + r.stderr + "---\n" + said.getvalue()
        else:
            assert r.stderr.split("-- Correlation matrix --")[1] == said.getvalue().split("-- Correlation matrix --")[1]
        rec = json.loads(r.stdout)
        assert rec["rows"] == 3000 and rec["error_type"] == error_type
        for p, n in enumerate(names[:-1]):
            got = rec["intervals"][n]
            assert np.float32(got[1]) == iv[0, p, 1] and np.float32(got[2]) == iv[0, p, 2]       # limits: bit for bit
            assert abs(got[0] - iv[0, p, 0]) <= 2e-6 * max(abs(iv[0, p, 0]), float(np.ptp(chain[:, p])))
