set -o pipefail
mkdir -p gpurun_out/r3i
for wl in c2 c3 c5; do
  extra=""; [ $wl = c5 ] && extra="--steps 40"
  timeout -k 10 300 python3 bench.py --workload $wl --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline $extra > gpurun_out/r3i/$wl.json 2> gpurun_out/r3i/$wl.err || { echo $wl failed; tail -5 gpurun_out/r3i/$wl.err; exit 1; }
  python3 - gpurun_out/r3i/$wl.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); rf = r["roofline"]
print("%s: %.0f evals/s step %.1f us fill %.2f us (%d) in-region %.2f us frac %.3f empty bracket %.2f us" % (r["config"]["workload"][:2], r["value"], 1e3*r["ms_per_step"], 1e3*rf["avg_launch_ms"], rf["launches_timed"], 1e3*rf["in_timed_region"]["avg_launch_ms"], rf["frac"], 1e3*rf["empty_event_bracket_ms"]))
PY
done
timeout -k 10 300 python3 bench.py --lookahead --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline > gpurun_out/r3i/la.json 2> gpurun_out/r3i/la.err && python3 - <<'PY'
import json
r = json.loads(open("gpurun_out/r3i/la.json").read().strip().splitlines()[-1]); rf = r["roofline"]
print("lookahead: %.0f steps/s pass %.2f us (%d)" % (r["value"], 1e3*rf["avg_launch_ms"], rf["launches_timed"]))
PY
