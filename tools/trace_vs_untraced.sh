export TMPDIR=/tmp
B="python3 bench.py --steps 150 --warmup 10 --no-cpu-baseline --experiments 0 --also none"
val() { python3 -c 'import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.1f evals/s fill %.2f us (%d launches)" % (r["value"], 1e3*r["roofline"]["avg_launch_ms"], r["roofline"]["launches_timed"]))'; }
echo "U1 untraced B-args: $($B 2>/dev/null | val)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr1 -- $B > /tmp/tr1.log 2>&1; echo "T1 traced: $(grep -h '^{"metric"' /tmp/tr1.log | val)"; python3 tools/summarize_rocprof.py stats /tmp/tr1 /tmp/tr1.csv | head -3
echo "U2 untraced B-args: $($B 2>/dev/null | val)"
echo "U3 untraced 1000 steps + cpu baseline: $(python3 bench.py --experiments 0 --also none 2>/dev/null | val)"
echo "U4 untraced B-args: $($B 2>/dev/null | val)"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr2 -- $B > /tmp/tr2.log 2>&1; echo "T2 traced: $(grep -h '^{"metric"' /tmp/tr2.log | val)"; python3 tools/summarize_rocprof.py stats /tmp/tr2 /tmp/tr2.csv | head -3
