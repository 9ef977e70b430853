#!/usr/bin/env python3
"""Write one workload's entry of profiles/traffic.json from a PMC summary (tools/summarize_rocprof.py pmc ...):
HBM bytes per launch of the fill kernel = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (separate --pmc passes, the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md), together with the fingerprints of the kernel and planner sources the
profiled library was built from -- bench.py compares them with the sources it runs and says so when they differ.

  update_traffic.py <key> <pmc_summary.csv> <kernel name prefix> <committed copy under profiles/> [note]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    key, summary, prefix, committed = sys.argv[1:5]
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    from bench import source_fingerprints
    fetch = write = None
    for line in open(summary).read().splitlines()[1:]:
        kernel, counter, _, mean = line.rsplit(",", 3)      # (kernel names contain commas)
        if not kernel.startswith(prefix):
            continue
        if counter == "FETCH_SIZE":
            fetch = float(mean)
        if counter == "WRITE_SIZE":
            write = float(mean)
    if fetch is None or write is None:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for %r in %s" % (prefix, summary))
    path = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(path))
    t[key] = {"bytes_per_launch": 2 * fetch * 1024 + write * 1024, "read_bytes": 2 * fetch * 1024,
              "write_bytes": write * 1024,
              "source": "%s: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (separate --pmc passes, gfx950 FETCH_SIZE x2 "
                        "correction)%s" % (committed, "; " + note if note else ""),
              "profiled_sources": source_fingerprints()}
    json.dump(t, open(path, "w"), indent=1)
    print(key, json.dumps(t[key], indent=1))


if __name__ == "__main__":
    main()
