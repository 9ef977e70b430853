"""CPU model of the CODES argument (fill_ordered_body, sxmc_amd/csrc/fill_kernels.inc.h; layout_kernels.hip).

The fill over codes bins a sample from 16-bit codes of its fields whenever its bin coordinate lies further from a
bin edge than a bound `eps`; this test restates, in numpy, (a) the table's coding (windows, codes, the check of the
half step), (b) the kernel's composition of the program into coefficients over the codes and its bound -- the same
formulas, operation by operation --, (c) the single-precision evaluation, and holds them against the reference's
per-sample arithmetic in double (pdfz.cpp:306-331, 388-398: restated here with numpy's IEEE operations and tied to the
oracle's histogram): EVERY sample the model calls unambiguous must be in the bin, or outside the domain, exactly as
the reference has it, for random programs, parameters from tiny to large, and samples placed within ulps of the
transformed bin edges.  No GPU: the argument itself is what is tested."""
import numpy as np
import pytest

from oracle import oracle

QMAX = 65533


def reference_bins(tab, systs, params, lo, hi, nb):
    """Per sample: the index of every observable as bin_samples computes it, and whether the sample is in the domain."""
    f = [tab[:, k].astype(np.float64) for k in range(tab.shape[1])]
    with np.errstate(all="ignore"):
        for s in systs:
            p = 0.0 + params[s["pars"][0]] * 1.0
            k = s["obs"]
            if s["type"] == "shift":
                f[k] = f[k] + p
            elif s["type"] == "scale":
                f[k] = f[k] * (1 + p)
            elif s["type"] == "ctscale":
                f[k] = 1 + (f[k] - 1) * (1 + p)
            else:
                f[k] = f[k] + (p * (f[k] - f[s["true_obs"]]))
        idx, ind, ind_k = [], np.ones(tab.shape[0], bool), []
        for k in range(len(nb)):
            scale = nb[k] / (hi[k] - lo[k])
            ind_k.append((f[k] >= lo[k]) & (f[k] < hi[k]))
            ind &= ind_k[k]
            idx.append(((f[k] - lo[k]) * scale).astype(np.int64, casting="unsafe"))
    return idx, ind, ind_k


def windows(tab, fields, nobs, lo, hi):
    """get_bucket_codes (sxmc_launch_plan.cpp): a window per streamed field."""
    base, step = [], []
    ulo, uhi = 0.0, -1.0
    for m, fld in enumerate(fields):
        col = tab[:, fld]
        fin = col[np.isfinite(col)]
        none = fin.size == 0
        wlo, whi = (1.0, -1.0) if none else (float(fin.min()), float(fin.max()))
        if fld < nobs:
            w = hi[fld] - lo[fld]
            wlo = lo[fld] - w if none else max(wlo, lo[fld] - w)
            whi = hi[fld] + w if none else min(whi, hi[fld] + w)
            if not wlo < whi:
                wlo, whi = lo[fld] - w, hi[fld] + w
            ulo, uhi = (wlo, whi) if uhi < ulo else (min(ulo, wlo), max(uhi, whi))
        elif none:
            wlo, whi = 0.0, 1.0
        elif ulo <= uhi:
            w = uhi - ulo
            clo, chi = max(wlo, ulo - 3 * w), min(whi, uhi + 3 * w)
            if clo < chi:
                wlo, whi = clo, chi
        st = (whi - wlo) / 65532.0
        if not (st > 0 and np.isfinite(st)):
            st = max(abs(wlo), 1.0) * 2.0 ** -20
        base.append(wlo)
        step.append(st)
    return np.array(base), np.array(step)


def encode(tab, fields, base, step):
    """column_codes_kernel: codes, and the rows marked "ask the exact columns" (1) / "never counted" (2)."""
    n = tab.shape[0]
    codes = np.zeros((n, len(fields)), np.int64)
    mark = np.zeros(n, np.int8)
    for m, fld in enumerate(fields):
        x = tab[:, fld].astype(np.float64)
        fin = np.isfinite(x)
        with np.errstate(all="ignore"):
            t = (x - base[m]) / step[m]
            inside = fin & (t >= 0.0) & (t < QMAX + 1.0)
            q = np.where(inside, t, 0.0).astype(np.int64)
            centre = base[m] + (q + 0.5) * step[m]
            ok = inside & (np.abs(x - centre) <= 0.5 * step[m] * (1.0 + 2.0 ** -20))
        codes[:, m] = np.where(ok, q, 0)
        mark = np.where(~fin, 2, np.where(~ok & (mark < 2), np.maximum(mark, 1), mark)).astype(np.int8)
    return codes, mark


EPS_FACTOR = [1.0]      # (the negative control shrinks the bound through this)


def compose(systs, params, fields, base, step, lo, hi, nb, binned, details=None):
    """The kernel's AffineForm, coefficient by coefficient; returns per binned observable (alpha32[], gamma32, eps32)
    or None when the bound rules the codes out.  details: a list that receives, per binned observable, the terms of
    the bound (Q, S, D budgets), the double-precision constant g and the composed row of A."""
    nq = len(fields)
    slot = {fld: m for m, fld in enumerate(fields)}
    a = np.eye(nq)
    c = np.zeros(nq)
    mag = np.array([max(abs(base[m]), abs(base[m] + 65534.0 * step[m])) for m in range(nq)])
    if not np.all(np.isfinite(params)):
        return None
    for s in systs:
        if s["obs"] not in slot:
            continue                                           # (the ordered observable's own systematics)
        k = slot[s["obs"]]
        p = 0.0 + params[s["pars"][0]] * 1.0
        ap = abs(p)
        if s["type"] == "shift":
            c[k] = c[k] + p
            mag[k] = mag[k] + ap
        elif s["type"] == "scale":
            a[k] = a[k] * (1 + p)
            c[k] = c[k] * (1 + p)
            mag[k] = mag[k] * (1 + ap)
        elif s["type"] == "ctscale":
            a[k] = a[k] * (1 + p)
            c[k] = 1 + (c[k] - 1) * (1 + p)
            mag[k] = 1 + (mag[k] + 1) * (1 + ap)
        else:
            e = slot[s["true_obs"]]
            a[k] = a[k] + p * (a[k] - a[e])
            c[k] = c[k] + p * (c[k] - c[e])
            mag[k] = mag[k] + ap * (mag[k] + mag[e])
    out = []
    for obs in binned:
        k = slot[obs]
        sc = nb[obs] / (hi[obs] - lo[obs])
        alpha = a[k] * step * sc
        sum_abs = float(np.sum(np.abs(alpha)))
        g = (c[k] - lo[obs] + float(np.sum(a[k] * (base + 0.5 * step)))) * sc
        mu = sum_abs * 65536.0 + abs(g) + nb[obs] + 0.25
        eps = 0.5 * sum_abs * (1.0 + 2.0 ** -19) + mu * 2.0 ** -21 + (mag[k] + abs(lo[obs])) * sc * 2.0 ** -44
        if not eps < 0.125:
            return None
        # the kernel evaluates u' = u + e (e in the constant term) and asks fract(u') >= 2e + 2^-23
        e = eps * 1.01 * EPS_FACTOR[0]
        out.append((alpha.astype(np.float32), np.float32(g + e), np.float32(2.0 * e + 2.0 ** -23)))
        if details is not None:
            details.append(dict(Q=0.5 * sum_abs * (1.0 + 2.0 ** -19), S=mu * 2.0 ** -21,
                                D=(mag[k] + abs(lo[obs])) * sc * 2.0 ** -44, g=g, e=e, a=a[k].copy(), c=c[k], sc=sc, mu=mu))
    return out


def fma32(a, b, c):
    """A single-precision fused multiply-add: the product of two floats is exact in double."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def classify(codes, coef):
    """The kernel's per-sample test: (index per binned observable, unambiguous, in every domain)."""
    n = codes.shape[0]
    idx, clear, inside = [], np.ones(n, bool), np.ones(n, bool)
    for (alpha, g, thr32), nbk in coef:
        u = np.full(n, g, np.float32)
        for m in reversed(range(codes.shape[1])):            # (the order does not matter to the bound)
            u = fma32(np.full(n, alpha[m], np.float32), codes[:, m].astype(np.float32), u)
        fl = np.floor(u)
        with np.errstate(all="ignore"):
            fr = np.minimum((u - fl).astype(np.float32), np.float32(1.0 - 2.0 ** -24))   # (v_fract_f32's clamp)
        clear &= fr >= thr32
        i = fl.astype(np.int64)
        inside &= (i >= 0) & (i < nbk)
        idx.append(i)
    return idx, clear, inside


def make_case(seed):
    rng = np.random.default_rng(seed)
    nobs = int(rng.integers(2, 4))
    nextra = int(rng.integers(1, 3))
    nb = [int(rng.choice([3, 7, 20, 50, 200])) for _ in range(nobs)]
    lo = [float(rng.choice([0.0, -1.0, 5.0])) for _ in range(nobs)]
    hi = [lo[k] + float(rng.choice([1.0, 2.0, 10.0])) for k in range(nobs)]
    o = int(rng.integers(0, nobs))                             # the ordered observable: only shifted
    systs, npar = [dict(type="shift", obs=o, pars=[0])], 1
    binned = [k for k in range(nobs) if k != o][:int(rng.integers(1, 3))]
    fields = set(binned)
    for j, k in enumerate(binned):
        for i in range(int(rng.integers(1, 4))):
            kind = ["shift", "scale", "ctscale", "resolution_scale"][int(rng.integers(0, 4))]
            if j == 0 and i == 0:
                kind = "resolution_scale"
            d = dict(type=kind, obs=k, pars=[npar])
            npar += 1
            if kind == "resolution_scale":
                d["true_obs"] = int(rng.choice([f for f in list(range(nobs, nobs + nextra)) + binned if f != k]))
                fields.add(d["true_obs"])
            systs.append(d)
    fields = sorted(fields, key=lambda f: (f not in binned, f))
    n = 60000
    tab = np.empty((n, nobs + nextra), np.float32)
    for k in range(nobs + nextra):
        kk = k if k < nobs else binned[0]
        w = hi[kk] - lo[kk]
        tab[:, k] = rng.uniform(lo[kk] - 0.3 * w, hi[kk] + 0.3 * w, size=n)
    for f in fields:                                           # outliers, values that are not finite
        far = rng.uniform(size=n) < 0.003
        tab[far, f] = rng.uniform(-400, 400, size=int(far.sum())).astype(np.float32)
        bad = rng.uniform(size=n) < 0.002
        tab[bad, f] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), size=int(bad.sum()))
    return rng, nobs, nb, lo, hi, systs, npar, binned, fields, tab


@pytest.mark.parametrize("seed", range(40))
def test_unambiguous_samples_land_where_the_reference_puts_them(seed):
    rng, nobs, nb, lo, hi, systs, npar, binned, fields, tab = make_case(7000 + seed)
    base_params = rng.normal(0, 0.05, npar)
    # samples on the transformed edges of the first binned observable (float64 composition, then ulps around)
    k0 = binned[0]
    A, C = np.eye(tab.shape[1]), np.zeros(tab.shape[1])
    for s in systs:
        p, k = base_params[s["pars"][0]], s["obs"]
        if s["type"] == "shift":
            C[k] += p
        elif s["type"] == "scale":
            A[k] *= 1 + p
            C[k] *= 1 + p
        elif s["type"] == "ctscale":
            A[k] *= 1 + p
            C[k] = 1 + (C[k] - 1) * (1 + p)
        else:
            e = s["true_obs"]
            A[k] = A[k] + p * (A[k] - A[e])
            C[k] = C[k] + p * (C[k] - C[e])
    if abs(A[k0][k0]) > 1e-3:
        n = tab.shape[0]
        edges = lo[k0] + rng.integers(0, nb[k0] + 1, size=n) / nb[k0] * (hi[k0] - lo[k0])
        with np.errstate(all="ignore"):
            rest = C[k0] + sum(A[k0][f] * tab[:, f].astype(np.float64) for f in range(tab.shape[1]) if f != k0)
            x = ((edges - rest) / A[k0][k0]).astype(np.float32)
        for _ in range(2):
            up = rng.uniform(size=n) < 0.5
            mv = rng.uniform(size=n) < 0.5
            x = np.where(mv, np.nextafter(x, np.where(up, np.float32(1e9), np.float32(-1e9))), x).astype(np.float32)
        take = (rng.uniform(size=n) < 0.5) & np.isfinite(x)
        tab[take, k0] = x[take]
    base, step = windows(tab, fields, nobs, lo, hi)
    codes, mark = encode(tab, fields, base, step)
    assert (mark == 1).mean() < 0.02
    used = 0
    for trial in range(6):
        params = base_params.copy()
        if trial == 1:
            params = rng.normal(0, 0.05, npar)
        if trial == 2:
            params = rng.normal(0, 0.7, npar)
        if trial == 3:
            params = rng.normal(0, 8.0, npar)
        if trial == 4:
            params[int(rng.integers(0, npar))] = -1.0                  # a scale that flattens an observable
        if trial == 5:
            params[int(rng.integers(0, npar))] = np.nextafter(base_params[0], 1.0)
        coef = compose(systs, params, fields, base, step, lo, hi, nb, binned)
        if coef is None:
            continue                                                   # (the kernel streams the float columns)
        used += 1
        idx_ref, ind_ref, ind_k = reference_bins(tab, systs, params, lo, hi, nb)
        idx, clear, inside = classify(codes, [(c, nb[obs]) for c, obs in zip(coef, binned)])
        decided = clear & (mark == 0)
        # in the domain of every BINNED observable, as the reference has it (the ordered one is the granule's business)
        ind_binned = np.ones(tab.shape[0], bool)
        for obs in binned:
            ind_binned &= ind_k[obs]
        assert np.array_equal(inside[decided], ind_binned[decided]), (seed, trial)
        both = decided & inside
        for j, obs in enumerate(binned):
            assert np.array_equal(idx[j][both], idx_ref[obs][both]), (seed, trial, obs)
        # rows that are never counted really are outside every domain
        assert not ind_ref[mark == 2].any()
        # the bound is not vacuous: most samples are decided from their codes
        if trial < 2:
            assert decided.mean() > 0.4
    assert used >= 2


def reference_product(tab, systs, params, lo, nb, hi, obs):
    """(x - lo) * scale of observable `obs` as bin_samples forms it in double, before the truncation (pdfz.cpp:388-398)."""
    f = [tab[:, k].astype(np.float64) for k in range(tab.shape[1])]
    with np.errstate(all="ignore"):
        for s in systs:
            p = 0.0 + params[s["pars"][0]] * 1.0
            k = s["obs"]
            if s["type"] == "shift":
                f[k] = f[k] + p
            elif s["type"] == "scale":
                f[k] = f[k] * (1 + p)
            elif s["type"] == "ctscale":
                f[k] = 1 + (f[k] - 1) * (1 + p)
            else:
                f[k] = f[k] + (p * (f[k] - f[s["true_obs"]]))
        return (f[obs] - lo[obs]) * (nb[obs] / (hi[obs] - lo[obs]))


@pytest.mark.parametrize("seed", range(12))
def test_the_terms_of_the_bound_hold_one_by_one(seed):
    """THE BOUND, term by term (fill_kernels.inc.h): u_codes - u_ref = (what the codes leave unknown) + (roundings).
    The first part, evaluated in extended precision from the sample's position in its code cells, must stay within Q;
    what is left -- the single-precision evaluation over the codes and the double roundings of the reference's own
    arithmetic -- within S + D.  Measured on the way: how much of S + D random samples use (a few per cent -- what the
    GPU measurement of tests/test_gpu_codes_margin.py finds with samples built on the threshold)."""
    rng, nobs, nb, lo, hi, systs, npar, binned, fields, tab = make_case(8000 + seed)
    base, step = windows(tab, fields, nobs, lo, hi)
    codes, mark = encode(tab, fields, base, step)
    worst_q = worst_r = 0.0
    used = 0
    for scale_of_params in (0.05, 0.4):
        params = rng.normal(0, scale_of_params, npar)
        det = []
        coef = compose(systs, params, fields, base, step, lo, hi, nb, binned, details=det)
        if coef is None:
            continue
        used += 1
        ok = mark == 0
        for (alpha32, g32, thr32), d, obs in zip(coef, det, binned):
            # u_codes as the lanes form it, WITHOUT e: single-precision FMAs from the rounded constant term
            u = np.full(tab.shape[0], np.float32(d["g"]), np.float32)
            for m in range(codes.shape[1]):
                u = fma32(np.full(tab.shape[0], alpha32[m], np.float32), codes[:, m].astype(np.float32), u)
            with np.errstate(all="ignore"):
                u_ref = reference_product(tab, systs, params, lo, nb, hi, obs)
                # what the codes leave unknown, in extended precision: sum_m alpha_m ((c_m + 1/2) - (x_m - base_m) / step_m)
                q = np.zeros(tab.shape[0], np.longdouble)
                for m, fld in enumerate(fields):
                    x = tab[:, fld].astype(np.longdouble)
                    pos = (x - np.longdouble(base[m])) / np.longdouble(step[m])
                    q += np.longdouble(d["a"][m]) * np.longdouble(step[m]) * np.longdouble(d["sc"]) * (codes[:, m] + np.longdouble(0.5) - pos)
                r = u.astype(np.longdouble) - u_ref.astype(np.longdouble) - q
            fin = ok & np.isfinite(u_ref) & (np.abs(u_ref) < 1e6)
            assert fin.sum() > 0.5 * tab.shape[0]
            assert np.all(np.abs(q[fin]) <= d["Q"]), (seed, obs, float(np.max(np.abs(q[fin]))), d["Q"])
            assert np.all(np.abs(r[fin]) <= d["S"] + d["D"]), (seed, obs, float(np.max(np.abs(r[fin]))), d["S"] + d["D"])
            worst_q = max(worst_q, float(np.max(np.abs(q[fin])) / d["Q"]))
            worst_r = max(worst_r, float(np.max(np.abs(r[fin])) / (d["S"] + d["D"])))
    assert used >= 1
    assert worst_q > 0.9            # the half code step is reached: Q is no margin
    assert worst_r < 0.5            # the roundings use a small part of what the bound provides for


def test_reference_restatement_agrees_with_the_oracle():
    """The numpy restatement of the per-sample arithmetic used above gives the oracle's histogram."""
    rng, nobs, nb, lo, hi, systs, npar, binned, fields, tab = make_case(7100)
    params = rng.normal(0, 0.05, npar)
    tabo = np.concatenate([tab, np.zeros((tab.shape[0], 1), np.float32)], axis=1)
    geom = oracle.HistGeometry(lo, hi, nb)
    bins, norm = oracle.bin_samples(geom, tabo, tabo.shape[1], systs, params)
    idx, ind, _ = reference_bins(tab, systs, params, lo, hi, nb)
    flat = np.zeros(tab.shape[0], np.int64)
    for k in range(nobs):
        flat += idx[k] * int(geom.bin_stride[k])
    ok = ind & (flat >= 0) & (flat < geom.total_nbins)
    mine = np.bincount(flat[ok], minlength=geom.total_nbins).astype(np.uint32)
    assert int(ind.sum()) == norm and np.array_equal(mine, bins)


def test_the_bound_is_needed():
    """Negative control: with the bound cut to a hundredth, samples on the transformed edges ARE misplaced -- the test
    above would see a bound that is too small."""
    wrong = 0
    for seed in range(12):
        rng, nobs, nb, lo, hi, systs, npar, binned, fields, tab = make_case(7000 + seed)
        params = rng.normal(0, 0.05, npar)
        base, step = windows(tab, fields, nobs, lo, hi)
        codes, mark = encode(tab, fields, base, step)
        EPS_FACTOR[0] = 0.01
        try:
            small = compose(systs, params, fields, base, step, lo, hi, nb, binned)
        finally:
            EPS_FACTOR[0] = 1.0
        if small is None:
            continue
        idx_ref, ind_ref, ind_k = reference_bins(tab, systs, params, lo, hi, nb)
        idx, clear, inside = classify(codes, [(c, nb[obs]) for c, obs in zip(small, binned)])
        decided = clear & (mark == 0) & inside
        for j, obs in enumerate(binned):
            wrong += int(np.sum(idx[j][decided] != idx_ref[obs][decided]))
    assert wrong > 0
