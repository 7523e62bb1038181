#!/bin/bash
# A/B of one environment switch on one GPU box: scripts/ab_env.sh VAR=VALUE [bench args...]
# Runs bench.py without and with the variable, twice each, and prints fp32 / bf16 ms per step.
kv=$1; shift
run() { timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16']['ms_per_step'] if d.get('bf16') else '')"; }
for i in 1 2; do
  echo "default   $(run "$@")"
  echo "$kv $(export $kv; run "$@")"
done
