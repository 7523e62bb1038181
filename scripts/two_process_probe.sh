#!/bin/bash
# Two INDEPENDENT single-GPU benches (no process group, no collective) sharing one GPU, against the same bench alone:
# separates "two processes time-slice one GPU" from anything the gradient all-reduce does.
#   scripts/two_process_probe.sh            (on the GPU box; writes gpurun_out/two_proc_*.json)
set -u
cd "$GRAFT_REPO_ROOT"
A="--size 128 --batch 2 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --dtype f32"
timeout -k 10 120 python bench.py $A > gpurun_out/two_proc_solo.json 2> gpurun_out/two_proc_solo.err || exit 1
timeout -k 10 120 python bench.py $A --eager > gpurun_out/two_proc_solo_eager.json 2>> gpurun_out/two_proc_solo.err || exit 1
for mode in graph eager; do
  extra=""; [ $mode = eager ] && extra="--eager"
  timeout -k 10 200 python bench.py $A $extra > gpurun_out/two_proc_${mode}_a.json 2> gpurun_out/two_proc_${mode}_a.err &
  pa=$!
  timeout -k 10 200 python bench.py $A $extra > gpurun_out/two_proc_${mode}_b.json 2> gpurun_out/two_proc_${mode}_b.err &
  pb=$!
  wait $pa; ra=$?
  wait $pb; rb=$?
  echo "$mode: exit codes $ra $rb"
  [ $ra -eq 0 ] && [ $rb -eq 0 ] || exit 1
done
python - <<'PY'
import json
for n in ("solo", "solo_eager", "graph_a", "graph_b", "eager_a", "eager_b"):
    d = json.loads(open("gpurun_out/two_proc_%s.json" % n).read().strip().splitlines()[-1])
    print("%-11s ms_per_step %8.3f" % (n, d["ms_per_step"]))
PY
