#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  summarize_rocprof.py stats <dir> <out.csv>     kernel_stats.csv -> per-kernel calls / avg / share
  summarize_rocprof.py pmc <out.csv> <dir>...    counter_collection.csv (one dir per --pmc pass)
                                                 -> per-kernel mean counter values per launch
  summarize_rocprof.py timeline <dir> <out.csv> [skip_fraction]
                                                 kernel_trace.csv -> per kernel: calls, mean duration, mean IDLE GAP
                                                 on the device before it starts (start minus the latest end of any
                                                 earlier kernel, clamped at 0), and the mean period of the repeating
                                                 sequence; the first skip_fraction (default 0.3) of the dispatches is
                                                 left out (set-up, warm-up)
Torch's data-generation kernels are dropped; kernel names are shortened to the function name.
HBM bytes per launch of the fill kernel follow MI355X_MICROARCH.md (HBM section):
  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced
  streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for the atomics here.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    if "at::native" in name or "rocclr" in name or "Cijk" in name:
        return None
    m = re.search(r"(\w+_kernel\w*)(<[^(]*)?", name)
    if not m:
        return name[:60]
    base = m.group(1)
    if base == "fill_kernel":
        t = re.search(r"fill_kernel<(\d+), (\d+), (true|false), .*?(StaticProg|DynamicProg)", name)
        if t:
            return "fill_kernel<nobs=%s,nslot=%s,lds=%s,%s>" % (t.group(1), t.group(2), t.group(3), t.group(4))
    return base


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pattern, d))
    return hits[0]


def stats(d, out):
    rows = []
    for r in csv.DictReader(open(find(d, "*kernel_stats.csv"))):
        s = short(r["Name"])
        if s:
            rows.append((s, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                         float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
    tot = sum(r[5] for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_us,share_of_listed\n")
        for r in sorted(rows, key=lambda r: -r[5]):
            f.write("%s,%d,%.2f,%.2f,%.2f,%.1f,%.4f\n" % (r + (r[5] / tot,)))
    print(open(out).read())


def pmc(out, dirs):
    acc = collections.defaultdict(list)
    for d in dirs:
        for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
            s = short(r["Kernel_Name"])
            if s:
                acc[(s, r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), v in sorted(acc.items()):
            f.write("%s,%s,%d,%.6g\n" % (k, c, len(v), sum(v) / len(v)))
        for (k, c), v in sorted(acc.items()):
            if c == "FETCH_SIZE" and k.startswith("fill_kernel"):
                fetch = sum(v) / len(v)
                w = acc.get((k, "WRITE_SIZE"))
                write = sum(w) / len(w) if w else 0.0
                f.write("%s,HBM_READ_BYTES_corrected(2*FETCH_SIZE*1024),%d,%.6g\n" % (k, len(v), 2 * fetch * 1024))
                f.write("%s,HBM_WRITE_BYTES(WRITE_SIZE*1024),%d,%.6g\n" % (k, len(w or []), write * 1024))
                f.write("%s,HBM_TRAFFIC_BYTES,%d,%.6g\n" % (k, len(v), 2 * fetch * 1024 + write * 1024))
    print(open(out).read())


def timeline(d, out, skip=0.3):
    rows = []
    for r in csv.DictReader(open(find(d, "*kernel_trace.csv"))):
        s = short(r["Kernel_Name"])
        if s:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), s))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    acc = collections.OrderedDict()
    latest_end = None
    for start, end, name in rows:
        gap = max(0, start - latest_end) if latest_end is not None else 0
        a = acc.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (end - start) / 1e3
        a[2] += gap / 1e3
        latest_end = end if latest_end is None else max(latest_end, end)
    span = (rows[-1][1] - rows[0][0]) / 1e3 if rows else 0.0
    busy = sum(a[1] for a in acc.values())
    with open(out, "w") as f:
        f.write("kernel,calls,avg_us,avg_idle_gap_before_us\n")
        for k, a in acc.items():
            f.write("%s,%d,%.2f,%.2f\n" % (k, a[0], a[1] / a[0], a[2] / a[0]))
        f.write("# window %.1f us, kernels busy %.1f us (%.1f %%), idle %.1f us\n"
                % (span, busy, 100.0 * busy / max(span, 1e-9), span - busy))
        most = max((a[0] for a in acc.values()), default=0)
        if most:
            f.write("# period of the most frequent kernel: %.2f us\n" % (span / most))
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 0.3)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
