"""Input / output formats either side of the hot path, ROOT-free.

  read_table              src/io/ttree_io.cpp:21-159   first TTree of a ROOT file -> row-major float matrix
                                                       + field names; here: an .npz with one 1-D array per
                                                       field (int / float / double / bool -> float32)
  read_dataset_to_samples src/signal.cpp:50-109        cuts + column packing [fields..., DATASET]
  load_config             src/config.cpp:19-297, observable.cpp, systematic.cpp, source.cpp
                                                       the reference's JSON schema (C-style comments allowed,
                                                       README.md:64-65)
  write_chain             src/sxmc.cpp:130-141, mcmc.cpp:100-114  the "ls" ntuple: one column per parameter
                                                       + "likelihood"; here an .npz with the same columns

Config parsing and file formats are not compute; they exist so that a fit described in the reference's
schema can be run end to end (run_config; `fit.samples` re-loads a saved chain instead of walking, as sxmc.cpp:84-94
does).  Plots and ROOT files are not supported.
"""
import json
import os
import re

import numpy as np

from . import workloads

TYPE_NAMES = ("shift", "scale", "ctscale", "resolution_scale")     # systematic.cpp:21-36


def strip_comments(text):
    """Remove // and /* */ comments outside strings."""
    out, i, n, in_str = [], 0, len(text), False
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            out.append(c)
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            i = text.index("*/", i) + 2
            continue
        else:
            out.append(c)
        i += 1
    return "".join(out)


def read_table(path):
    """-> (float32 matrix [n, nfields] row-major, field names).  One 1-D array per field in an .npz."""
    with np.load(path) as z:
        fields = list(z.files)
        cols = []
        for f in fields:
            a = np.asarray(z[f])
            if a.ndim != 1 or a.dtype.kind not in "iufb":
                raise ValueError("field %r of %s: only 1-D int/float/double/bool branches are supported" % (f, path))
            cols.append(a.astype(np.float32))
    n = cols[0].size if cols else 0
    if any(c.size != n for c in cols):
        raise ValueError("fields of %s differ in length" % path)
    return (np.stack(cols, axis=1) if cols else np.zeros((0, 0), np.float32)), fields


def write_table(path, matrix, fields):
    np.savez(path, **{f: np.asarray(matrix)[:, i] for i, f in enumerate(fields)})


def read_dataset_to_samples(dataset, dataset_fields, dataset_id, sample_fields, cuts, required=None):
    """signal.cpp:50-109.  dataset: [n, len(dataset_fields)]; sample_fields ends with "DATASET";
    cuts: list of (field, lower, upper) -- an event is dropped when a cut field is < lower or > upper
    (bounds inclusive); when several cuts name one field the LAST one counts, as in the reference's lookup table
    (signal.cpp:57-69 overwrites the field's bounds cut by cut).
    required: how many of the leading sample fields must exist in the data set (default: all of them, the MC tables).
    A data set of real events carries no Monte Carlo truth branch: the reference maps such a field past the row
    (signal.cpp:72-77) and never looks at it again -- GetSamples keeps the observables only -- so for data sets
    (required = number of observables) a missing non-observable field is filled with 0 instead of being refused.
    Returns float32 [nkept, len(sample_fields)]."""
    dataset = np.asarray(dataset, np.float32)
    keep = np.ones(dataset.shape[0], bool)
    last = {}
    for field, lower, upper in cuts:
        last[field] = (lower, upper)
    for j, name in enumerate(dataset_fields):
        if name in last:
            lower, upper = last[name]
            col = dataset[:, j]
            keep &= ~((col < np.float64(lower)) | (col > np.float64(upper)))
    required = len(sample_fields) - 1 if required is None else required
    out = np.zeros((int(keep.sum()), len(sample_fields)), np.float32)
    kept = dataset[keep]
    for k, f in enumerate(sample_fields[:-1]):
        if f in dataset_fields:
            out[:, k] = kept[:, dataset_fields.index(f)]
        elif k < required:
            raise KeyError("sample field %r not in data set fields %r" % (f, dataset_fields))
    out[:, -1] = dataset_id
    return out


class FitConfig:
    """What FitConfig::FitConfig (config.cpp:19-297) extracts, as plain Python data."""


def load_config(path_or_text, base_dir=None):
    if os.path.exists(path_or_text):
        base_dir = base_dir or os.path.dirname(os.path.abspath(path_or_text))
        text = open(path_or_text).read()
    else:
        text = path_or_text
    root = json.loads(strip_comments(text))
    fit, pdfs = root["fit"], root["pdfs"]
    obs_params, sys_params = pdfs["observables"], pdfs.get("systematics", {})
    sig_params, src_params = root["signals"], root.get("sources", {})

    fc = FitConfig()
    fc.nexperiments = int(fit["nexperiments"])                    # config.cpp:43-49
    fc.nsteps = int(fit["nsteps"])
    assert fc.nexperiments > 0 and fc.nsteps > 0
    fc.error_type = fit.get("error_type", "contour")
    assert fc.error_type in ("contour", "projection")
    fc.burnin_fraction = float(np.float32(fit.get("burnin_fraction", 0.1)))     # (floats in the reference: asFloat())
    fc.debug_mode = bool(fit.get("debug_mode", False))
    fc.output_prefix = fit.get("output_prefix", "lspace")
    fc.seed = int(fit.get("seed", 0))
    fc.confidence = float(np.float32(fit.get("confidence", 0.683)))
    fc.signal_name = fit.get("signal_name", "")
    fc.samples = fit.get("samples", "")      # config.cpp:51: a saved chain to use INSTEAD of walking (sxmc.cpp:84-94)

    def observable(name):
        c = obs_params[name]
        return dict(name=name, field=c["field"], bins=int(c["bins"]), lower=np.float32(c["min"]),
                    upper=np.float32(c["max"]))

    fc.observables = [observable(n) for n in fit["observables"]]
    fc.cuts = [observable(n) for n in fit.get("cuts", [])]
    assert not {o["name"] for o in fc.observables} & {c["name"] for c in fc.cuts}

    # systematics and sources: union over the signals (config.cpp:97-151).  The reference walks the JSON object
    # `signals` with jsoncpp 0.6's iterator, i.e. in KEY order (a std::map compared with strcmp), not in file order:
    # the numbering of the sources and of the systematic parameters -- the layout of the parameter vector -- follows it
    fc.systematics, fc.sources = [], []
    pidx = 0
    for sname in sorted(sig_params, key=lambda k: k.encode()):
        sconf = sig_params[sname]
        for sys_name in sconf.get("systematics", []):
            if any(s["name"] == sys_name for s in fc.systematics):
                continue
            c = sys_params[sys_name]
            if c["type"] not in TYPE_NAMES:
                raise ValueError("Unknown systematic type %s" % c["type"])
            means = [float(x) for x in c["mean"]]
            sigmas = [float(x) for x in c["sigma"]] if "sigma" in c else [0.0] * len(means)
            assert len(sigmas) == len(means)
            s = dict(name=sys_name, type=c["type"], observable_field=c["observable_field"],
                     truth_field=c.get("truth_field") if c["type"] == "resolution_scale" else None,
                     means=means, sigmas=sigmas, npars=len(means), fixed=bool(c.get("fixed", False)),
                     pidx=list(range(pidx, pidx + len(means))))
            if c["type"] == "resolution_scale":
                assert s["truth_field"] is not None
            pidx += len(means)
            fc.systematics.append(s)
        if "source" in sconf:
            src_name = sconf["source"]
            if not any(s["name"] == src_name for s in fc.sources):
                p = src_params[src_name]
                fc.sources.append(dict(name=src_name, index=len(fc.sources), mean=np.float32(p.get("mean", 1.0)),
                                       sigma=np.float32(p.get("sigma", 0.0)), fixed=bool(p.get("fixed", False))))
        else:                                                   # the signal is a source for itself
            fc.sources.append(dict(name=sname, index=len(fc.sources), mean=np.float32(sconf.get("mean", 1.0)),
                                   sigma=np.float32(sconf.get("sigma", 0.0)), fixed=bool(sconf.get("fixed", False))))

    # order of the sampled fields: observables, then extra truth fields, then DATASET (config.cpp:153-194)
    fc.sample_fields = []

    def index_with_append(name):
        if name not in fc.sample_fields:
            fc.sample_fields.append(name)
        return fc.sample_fields.index(name)

    for o in fc.observables:
        o["field_index"] = index_with_append(o["field"])
    for s in fc.systematics:
        assert s["observable_field"] in fc.sample_fields, "systematic observable must be an observable"
        s["observable_field_index"] = fc.sample_fields.index(s["observable_field"])
        s["truth_field_index"] = index_with_append(s["truth_field"]) if s["type"] == "resolution_scale" else 0
    fc.sample_fields.append("DATASET")

    fc.signals = []
    for name in fit["signals"]:                                   # config.cpp:197-258
        c = sig_params[name]
        assert ("rate" in c) != ("scale" in c)
        src_name = c.get("source", name)
        fc.signals.append(dict(
            name=name, dataset=int(c["dataset"]), filename=c["filename"],
            # config.cpp:216-222: both go through a float
            rate=float(np.float32(c["rate"])) if "rate" in c else None,
            scale=float(np.float32(c["scale"])) if "scale" in c else None,
            systematics=[s for s in c.get("systematics", [])],
            source=next(s for s in fc.sources if s["name"] == src_name)))
    fc.data = {int(k): [dict(filename=row["filename"], title=row.get("title", "")) for row in rows]
               for k, rows in root.get("data", {}).items()}
    fc.base_dir = base_dir or "."
    return fc


def build_workload(fc):
    """Load every signal's table (Signal::Signal, signal.cpp:11-47) and assemble the Workload the MCMC
    driver takes.  All signals must carry the same systematics list (one batched launch)."""
    nobs = len(fc.observables)
    nfields = len(fc.sample_fields)
    cuts = [(c["field"], c["lower"], c["upper"]) for c in fc.cuts]
    signals = []
    for s in fc.signals:
        table, fields = read_table(os.path.join(fc.base_dir, s["filename"]))
        n_mc = table.shape[0]                                    # before cuts (signal.cpp:28)
        # config.cpp:221 keeps -1 / scale in a float, signal.cpp:31-35 multiplies it by -n_mc in double
        nexpected = s["rate"] if s["rate"] is not None else \
            float(np.float32(-1.0) / np.float32(s["scale"])) * (-1.0 * n_mc)
        samples = read_dataset_to_samples(table, fields, s["dataset"], fc.sample_fields, cuts)
        sig = workloads.Signal(samples, nfields, nexpected, s["source"]["index"], dataset=s["dataset"])
        sig.n_mc_total = n_mc
        sig.name = s["name"]
        signals.append(sig)
    systs = [dict(type=s["type"], obs=s["observable_field_index"], true_obs=s["truth_field_index"],
                  pars=s["pidx"]) for s in fc.systematics]
    order = sorted(range(nobs), key=lambda i: fc.observables[i]["field_index"])
    lower = [float(fc.observables[i]["lower"]) for i in order]
    upper = [float(fc.observables[i]["upper"]) for i in order]
    nbins = [fc.observables[i]["bins"] for i in order]
    sigmas = [x for s in fc.systematics for x in s["sigmas"]]
    w = workloads.Workload(fc.output_prefix, nobs, lower, upper, nbins, signals, systs, sigmas,
                           np.zeros((0, nobs + 1), np.float32), "from config")
    w.source_means = [float(s["mean"]) for s in fc.sources]
    w.source_sigmas = [float(s["sigma"]) for s in fc.sources]
    w.syst_means = [x for s in fc.systematics for x in s["means"]]
    w.parameter_names = [s["name"] for s in fc.sources] + \
        ["%s_%d" % (s["name"], j) for s in fc.systematics for j in range(s["npars"])] + ["likelihood"]
    nsrc = len(fc.sources)
    w.parameter_means = lambda: np.array(w.source_means + w.syst_means, np.float64)
    w.parameter_sigmas = lambda: np.array(w.source_sigmas + sigmas, np.float64)
    assert w.nsources == nsrc
    return w


def load_data(fc, w, experiment=0):
    """The data experiment i fits when the configuration lists data sets (config.cpp:261-296, sxmc.cpp:71-80): file i
    of EVERY data set, in the order of the data set ids, clipped to the PDF boundaries (observables act as cuts).
    None: no data sets configured (the caller samples a fake one)."""
    if not fc.data:
        return None
    cuts = [(o["field"], o["lower"], o["upper"]) for o in fc.observables] + \
           [(c["field"], c["lower"], c["upper"]) for c in fc.cuts]
    rows = []
    for dataset, files in sorted(fc.data.items()):
        if experiment >= len(files):        # (the reference indexes past the end of its list here)
            raise ValueError("data set %d lists %d file(s): experiment %d has none to fit (one file per experiment)"
                             % (dataset, len(files), experiment))
        f = files[experiment]
        table, fields = read_table(os.path.join(fc.base_dir, f["filename"]))
        s = read_dataset_to_samples(table, fields, dataset, fc.sample_fields, cuts, required=len(fc.observables))
        rows.append(np.concatenate([s[:, :w.nobs], s[:, -1:]], axis=1))      # GetSamples: observables + dataset
    return np.concatenate(rows, axis=0)


def write_chain(path, names, chain):
    """The "ls" ntuple of one experiment: columns = parameter names + "likelihood"."""
    np.savez(path, **{n: np.asarray(chain)[:, i] for i, n in enumerate(names)})


def run_config(path, out_dir=None, nexperiments=None, nsteps=None, report=None):
    """ensemble() of sxmc.cpp:44-145 for a config in the reference's schema: per experiment fake data
    (or the configured data sets), MCMC, contour intervals; chains written as <prefix>_<i>.npz.
    report: a text stream that receives, per experiment, what sxmc.cpp:100-101 prints (best fit + correlation matrix).
    Returns (intervals [nexp, P, 4], limits of fit.signal_name, parameter names)."""
    from . import ensemble

    def say(names, chain, iv, one_sided=None):
        if report is not None:
            report.write(ensemble.format_best_fit(names, iv, np.asarray(chain, np.float32)[:, -1].min(), fc.confidence, one_sided))
            report.write(ensemble.format_correlations(names, ensemble.correlation_matrix(chain)))
    fc = load_config(path)
    if fc.samples:                                               # sxmc.cpp:84-94: no walk, the saved likelihood space
        chain, names = read_table(os.path.join(fc.base_dir, fc.samples))
        assert names and names[-1] == "likelihood" and chain.shape[0] > 0
        one_sided = None
        if fc.error_type == "projection":
            cols = [ensemble.projection_interval(chain[:, p], fc.confidence) for p in range(len(names) - 1)]
            iv = np.array([c[:4] for c in cols], np.float32)
            one_sided = [bool(c[4]) for c in cols]
        else:
            iv = ensemble.contour_intervals(chain, fc.confidence)
        say(names[:-1], chain, iv, one_sided)
        limits = [float(iv[names.index(fc.signal_name), 2])] if fc.signal_name in names[:-1] else []
        return iv[None], limits, names
    from .mcmc import MCMC
    w = build_workload(fc)
    m = MCMC(w, seed=fc.seed & 0xFFFFFFFF, fused=True, lut_output=False, consume=True)
    nexp = nexperiments or fc.nexperiments
    nsteps = nsteps or fc.nsteps
    allint, limits = [], []
    for i in range(nexp):
        rng = np.random.default_rng((fc.seed << 20) + i)
        events = load_data(fc, w, i)                             # sxmc.cpp:71-80: file i of every configured data set
        if events is None:
            events, _ = ensemble.make_fake_dataset(rng, w, m.pdfs, poisson=True)
        m.reseed(((fc.seed << 20) + i) & 0xFFFFFFFF)
        chain, _ = m.walk(events, nsteps, fc.burnin_fraction, debug_mode=fc.debug_mode)
        if fc.error_type == "projection":                       # likelihood.cpp:104-137: the chosen estimator
            cols = [ensemble.projection_interval(chain[:, p], fc.confidence) for p in range(chain.shape[1] - 1)]
            iv = np.array([c[:4] for c in cols], np.float32)
        else:
            cols = None
            iv = ensemble.contour_intervals(chain, fc.confidence)
        say(w.parameter_names, chain, iv, [bool(c[4]) for c in cols] if cols else None)
        allint.append(iv)
        if fc.signal_name in w.parameter_names:
            limits.append(float(iv[w.parameter_names.index(fc.signal_name), 2]))
        if out_dir:
            write_chain(os.path.join(out_dir, "%s_%d.npz" % (fc.output_prefix, i)), w.parameter_names, chain)
    return np.array(allint), limits, w.parameter_names
