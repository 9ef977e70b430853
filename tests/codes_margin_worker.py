"""Worker of tests/test_gpu_codes_margin.py: measures, on the GPU, how much room the codes' error bound has.

Run as a child process with SXMC_HIP_LIB = sxmc_amd/csrc/libsxmc_hip_measure.so (the measurement build: only it can
scale the bound -- sxmc_group_set_debug_mode bits 8-23, fill_kernels.inc.h "THE BOUND").  Prints one JSON object.

The fill over codes (fill_ordered_body) trusts a sample's codes when fract(u') >= 2e, u' = u_codes + e, and hands every
other sample to the reference's arithmetic.  e = Q + R: Q = half a code step per field in bins (an identity, checked per
row when the table is built: the worst case IS reached), R = everything that bounds roundings (single precision:
mu 2^-21, double: 2^-44, the slack 1.01 and 2^-23).  A wrong bin needs a sample that
  (A) has u_ref just BELOW an integer k while its fields sit at the edges of their code cells that push u_codes UP by Q
      (then u' - k = r - (k - u_ref) + R: trusted, with the floor one too high, iff the rounding error r of the single-
      precision evaluation exceeds what R left for it), or
  (B) has u_ref just ABOVE k with the fields at the opposite edges and r negative.
Random samples almost never do (position in both cells, distance to the edge and rounding error all extreme at once), so
the samples are BUILT: for every bin edge k and every code cell of the binned observable near it, the truth field's
value that puts u_ref a hair from k is solved for, and the pairs whose truth value falls at the right edge of its own
cell are kept -- a few thousand per parameter set, each with its own rounding error r.  The histogram is then filled
with R scaled by s = 1, 3/4, ... 0 and compared with the oracle's: s_min = the smallest s at which every sample is still
binned like the oracle.  Scaling the WHOLE threshold (Q too) by t < 1 must misplace samples: the negative control.
Reference arithmetic: /root/reference/src/pdfz.cpp:306-331, 388-398 (restated in oracle/sxmc_oracle.c)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle                      # noqa: E402  (the checker)
from sxmc_amd import capi, nll, pdfz           # noqa: E402
from sxmc_amd.capi import DeviceArray          # noqa: E402
from sxmc_amd.mcmc import make_systematic      # noqa: E402

LO, HI, NB = [0.3, 0.0, -1.0], [9.1, 6.0, 1.0], [20, 20, 20]       # e, r, c
SYSTS = [dict(type="shift", obs=1, pars=[0]), dict(type="scale", obs=0, pars=[1]),
         dict(type="resolution_scale", obs=0, true_obs=3, pars=[2])]
NFIELDS = 5
R_VALUES = [1.05, 2.55, 3.45, 4.65]                                 # middles of bins of r
T_RANGE = (-3.0, 15.0)                                              # the truth field's values (pins its window)
PARAM_SETS = [[0.02, 0.01, 0.05], [-0.03, -0.02, -0.07], [0.0, 0.1, 0.3], [0.01, -0.05, -0.4], [0.0, 0.0, 0.1137],
              [0.04, 0.3, 0.02]]
ROUNDING_SCALES = [1.0, 0.75, 0.5, 0.375, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.0]
TOTAL_SCALES = [0.85, 0.75, 0.5]


def background(rng, n):
    """Ordinary rows + the rows that pin the windows of the codes (observable: domain widened by its width on either
    side; truth field: its finite range)."""
    w = HI[0] - LO[0]
    tab = np.zeros((n, NFIELDS), np.float32)
    t = rng.uniform(T_RANGE[0] + 1.0, T_RANGE[1] - 1.0, n)
    tab[:, 3] = t
    tab[:, 0] = t + rng.normal(0, 0.4, n)
    # r (the ordered observable) takes four values only: inside a bucket the rows are sorted by r, and a 256-row granule
    # whose rows straddle an r-bin edge takes the float path whatever the codes say -- with runs of equal r nearly every
    # granule is decided from its codes (the built rows below join these runs)
    tab[:, 1] = np.array(R_VALUES, np.float32)[rng.integers(0, 4, n)]
    tab[:, 2] = rng.uniform(-1.0, 1.0, n)
    tab[0, 0], tab[1, 0] = LO[0] - 2 * w, HI[0] + 2 * w            # beyond the window: marked "ask the exact columns"
    tab[2, 3], tab[3, 3] = T_RANGE
    return tab


def reference_u(e, t, p1, p2, sc):
    """The reference's arithmetic on observable 0, operation by operation in IEEE double (numpy), up to the product
    (x - lo) * scale whose truncation is the bin index (pdfz.cpp:316-330, 388-398)."""
    e = e.astype(np.float64)
    t = t.astype(np.float64)
    pc1, pc2 = 0.0 + p1 * 1.0, 0.0 + p2 * 1.0
    x = e * (1 + pc1)
    x = x + (pc2 * (x - t))
    return (x - LO[0]) * sc, x


def adversarial(rng, base, step, params, nmax=6000, tol_t=0.01):
    """Rows built to sit on the threshold of the codes' test, both kinds (see the module docstring): columns e, truth
    value, kind (0: A, 1: B), the bin edge k they sit on."""
    p0, p1, p2 = params
    sc = NB[0] / (HI[0] - LO[0])
    a_e, a_t = (1 + p1) * (1 + p2), -p2
    alpha_e, alpha_t = a_e * step[0] * sc, a_t * step[1] * sc
    rprime = 2.0 ** -21 * (abs(alpha_e) + abs(alpha_t)) * 65536.0      # (order of the roundings' share, in bins)
    rows = []
    for side in (+1, -1):          # +1: kind A (u_ref below k, cells pushed up); -1: kind B
        for k in range(0, NB[0] + 1):
            # every code cell of e whose solved truth value can lie in the truth window
            x_edge = LO[0] + k / sc                                     # the transformed value at the edge
            e_lo = (x_edge + p2 * (T_RANGE[0] + 0.5 if p2 > 0 else T_RANGE[1] - 0.5)) / a_e
            e_hi = (x_edge + p2 * (T_RANGE[1] - 0.5 if p2 > 0 else T_RANGE[0] + 0.5)) / a_e
            q_lo, q_hi = sorted((int((e_lo - base[0]) / step[0]), int((e_hi - base[0]) / step[0])))
            q = np.arange(max(q_lo, 1), min(q_hi, 65000))
            if q.size == 0:
                continue
            # e at the edge of its cell that moves the cell's centre in the direction `side` (in u)
            at_low = (alpha_e > 0) == (side > 0)
            e_real = base[0] + (q + (0.0 if at_low else 1.0)) * step[0]
            # the FIRST float32 inside the cell from that edge (a cell holds ~1000 of them): checked against the table's
            # own coding, code = floor((x - base) / step) in double
            e32 = e_real.astype(np.float32)
            if at_low:
                e32 = np.where(e32.astype(np.float64) < e_real, np.nextafter(e32, np.float32(1e9)), e32).astype(np.float32)
            else:
                e32 = np.where(e32.astype(np.float64) >= e_real, np.nextafter(e32, np.float32(-1e9)), e32).astype(np.float32)
            coded = np.floor((e32.astype(np.float64) - base[0]) / step[0]) == q
            delta = rng.uniform(0.0, 0.002, q.size) * rprime
            target = k - side * delta                                   # u_true
            # a_e e - p2 t = lo + target / sc
            t_real = (a_e * e32.astype(np.float64) - LO[0] - target / sc) / p2
            ft = (t_real - base[1]) / step[1]
            frac = ft - np.floor(ft)
            t_low = (alpha_t > 0) == (side > 0)
            near = ((frac < tol_t) if t_low else (frac > 1.0 - tol_t)) & coded
            inside = (t_real > T_RANGE[0] + 0.01) & (t_real < T_RANGE[1] - 0.01)
            sel = near & inside
            if not sel.any():
                continue
            rows.append(np.stack([e32[sel], t_real[sel].astype(np.float32),
                                  np.full(int(sel.sum()), 0.0 if side > 0 else 1.0, np.float32),
                                  np.full(int(sel.sum()), float(k), np.float32)], axis=1))
    if not rows:
        return np.zeros((0, 4), np.float32)
    rows = np.concatenate(rows)
    if rows.shape[0] > nmax:
        rows = rows[rng.choice(rows.shape[0], nmax, replace=False)]
    return rows


def one_parameter_set(rng, params, nbackground, scales=True):
    """scales False: only the unscaled evaluation (what the product library can do: no hook is touched)."""
    tab0 = background(rng, nbackground)
    geom = oracle.HistGeometry(LO, HI, NB)

    def make(tab):
        ev = pdfz.EvalHist(tab, NFIELDS, 3, LO, HI, NB)
        for s in SYSTS:
            ev.AddSystematic(make_systematic(s))
        norm, pbuf = DeviceArray.zeros(1, np.uint32), DeviceArray(np.asarray(params, np.float64))
        ev.SetNormalizationBuffer(norm)
        ev.SetParameterBuffer(pbuf)
        group = nll.EvalGroup([ev])
        group.SetOrdering(True, force=True)
        group.SetBoxes(False)     # (the two-field codes of the ordered form: where the bound's S term is largest)
        group.SetCodes(True)
        assert "ordered+codes" in group.LaunchInfo(), group.LaunchInfo()
        return ev, group, norm, pbuf

    ev, group, norm, pbuf = make(tab0)
    base, step = group.CodesWindows(0)
    group.close()
    ev.close()
    assert len(base) == 2
    adv = adversarial(rng, base, step, params)
    # the built rows: e, truth; r one of the four values of the runs, c in the middle of a bin.  Every (edge, kind) gets
    # (r, c) bins OF ITS OWN: the comparison is between histograms, and there misplaced rows cancel -- at one edge a
    # kind A row moves a count from bin k - 1 to k and a kind B row one from k to k - 1; at neighbouring edges one row's
    # gain is the next one's loss.  Edge k -> c bin k (edge 20 shares c bin 0 under another r), kind -> r.
    rows = np.zeros((adv.shape[0], NFIELDS), np.float32)
    rows[:, 0], rows[:, 3] = adv[:, 0], adv[:, 1]
    kind, edge = adv[:, 2].astype(np.int64), adv[:, 3].astype(np.int64)
    rows[:, 1] = np.array(R_VALUES, np.float32)[2 * kind + (edge == NB[0])]
    rows[:, 2] = (LO[2] + ((edge % NB[2]) + 0.5) * (HI[2] - LO[2]) / NB[2]).astype(np.float32)
    # (shuffled: rows of equal r keep their table order inside a bucket, and rows appended at the end of the table would
    # all sit at the END of their run -- in the one granule that straddles the next r value and takes the float path)
    tab = np.concatenate([tab0, rows])
    order = rng.permutation(tab.shape[0])
    tab, rows = tab[order], None
    ev, group, norm, pbuf = make(tab)
    base2, step2 = group.CodesWindows(0)
    assert base2 == base and step2 == step, "the built rows moved the windows"
    want_bins, want_norm = oracle.bin_samples(geom, tab, NFIELDS, SYSTS, np.asarray(params, np.float64))
    # how close to the edges the built rows are, by the reference's own arithmetic
    sc = NB[0] / (HI[0] - LO[0])
    built = np.flatnonzero(order >= tab0.shape[0])
    u, _ = reference_u(tab[built, 0], tab[built, 3], params[1], params[2], sc)
    dist = np.abs(u - np.rint(u))

    def misplaced(mode):
        if scales:
            group.SetDebugMode(mode)
        group.EvalAsync(False)
        group.EvalFinished()
        got = ev.GetBins()
        return int(np.abs(got.astype(np.int64) - want_bins.astype(np.int64)).sum() // 2 +
                   abs(int(norm.get()[0]) - int(want_norm)))

    out = {"params": list(params), "rows_built": int(built.size), "rows": int(tab.shape[0]),
           "median_distance_to_edge_bins": float(np.median(dist)) if dist.size else None,
           "windows": {"base": base, "step": step},
           "unscaled": misplaced(0)}
    if scales:
        out["rounding_scale"] = {str(s): misplaced((1 + int(round(64 * s))) << 8) for s in ROUNDING_SCALES}
        out["total_scale"] = {str(t): misplaced((1 + int(round(64 * t))) << 16) for t in TOTAL_SCALES}
        # how many samples the exact path decides (hook 16 drops what the queues hold: they go missing), as shipped and
        # with the threshold halved -- the built rows leave the queues when the threshold no longer covers them
        out["decided_by_exact_path"] = {"unscaled": misplaced(16), "half_threshold": misplaced(16 | ((1 + 32) << 16))}
        group.SetDebugMode(0)
    group.close()
    ev.close()
    return out


def main():
    """argv: [background rows] [number of RANDOM parameter sets instead of the fixed six] [seed]"""
    if not capi.is_measurement_build():
        raise SystemExit("codes_margin_worker.py needs SXMC_HIP_LIB = .../libsxmc_hip_measure.so")
    if capi.device_count() < 1:
        raise SystemExit("codes_margin_worker.py needs a GPU")
    nbackground = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    nrandom = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 20251)
    param_sets = PARAM_SETS
    if nrandom > 0:
        # shift of r small (its granules must stay in their bins), scale of e ordinary to large, resolution scale of
        # either sign from small to large (never near 0: the truth value is solved for through it)
        param_sets = [[float(rng.normal(0, 0.02)), float(rng.normal(0, 0.1)),
                       float(rng.choice([-1.0, 1.0]) * rng.uniform(0.03, 0.5))] for _ in range(nrandom)]
    sets = [one_parameter_set(rng, p, nbackground) for p in param_sets]
    ok = [s for s in ROUNDING_SCALES if all(r["rounding_scale"][str(s)] == 0 for r in sets)]
    # s_min: the smallest scale from which upwards nothing is misplaced
    s_min = None
    for s in ROUNDING_SCALES:
        if s in ok:
            s_min = s
        else:
            break
    print(json.dumps({"sets": sets, "s_min": s_min, "rounding_scales": ROUNDING_SCALES,
                      "misplaced_at_zero": sum(r["rounding_scale"]["0.0"] for r in sets),
                      "misplaced_at_half_threshold": sum(r["total_scale"]["0.5"] for r in sets),
                      "misplaced_at_85_percent_threshold": sum(r["total_scale"]["0.85"] for r in sets),
                      "misplaced_unscaled": sum(r["unscaled"] for r in sets),
                      "rows_built": sum(r["rows_built"] for r in sets)}))


if __name__ == "__main__":
    main()
