export TMPDIR=/tmp
for form in "SXMC_FUSED_STEP=1" "SXMC_FUSED_STEP=0"; do for dm in 0 8; do
  if [ "$form" = "SXMC_FUSED_STEP=0" ] && [ $dm = 8 ]; then continue; fi
  env $form rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 bench.py --steps 400 --warmup 50 --also none --experiments 0 --no-cpu-baseline --debug-mode $dm > /tmp/tp.log 2>&1
  echo "== $form debug $dm"; python3 tools/summarize_rocprof.py timeline /tmp/tp /tmp/tp.csv 0.5 | head -6; rm -rf /tmp/tp
done; done
