#!/bin/bash
# A/B of several environment settings on one GPU box: scripts/ab_envs.sh "A=1 B=2" "C=3" ... [-- bench args]
# Runs bench.py with no setting and with each quoted group of VAR=VALUE pairs in turn, two rounds, and prints
# fp32 / bf16 ms per step, or the exit code when a run left no result line (139 = the process crashed).
sw=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sw+=("$1"); shift; done; [ "$1" == "--" ] && shift
run() { out=$(timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null); rc=$?
  if [ -z "$out" ]; then echo "no output, exit code $rc"; else echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16']['ms_per_step'] if d.get('bf16') else '')"; fi; }
for i in 1 2; do
  echo "default   $(run "$@")"
  for kv in "${sw[@]}"; do echo "$kv $(export $kv; run "$@")"; done
done
