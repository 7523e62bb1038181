#!/bin/bash
# A/B of several environment settings on one GPU box: scripts/ab_envs.sh "A=1 B=2" "C=3" ... [-- bench args]
# Runs bench.py with no setting and with each quoted group of VAR=VALUE pairs in turn, two rounds, and prints
# fp32 / bf16 ms per step.
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" == "--" ] && shift
run() { timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>>gpurun_out/ab_envs.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16']['ms_per_step'] if d.get('bf16') else '')"; }
for i in 1 2; do
  echo "default: $(run "$@")"
  for kv in "${sets[@]}"; do
    echo "$kv: $(export $kv; run "$@")"
  done
done
