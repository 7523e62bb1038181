#!/bin/bash
# usage: scripts/eval_trace.sh TAG [f32|bf16]: kernel trace of the 1024x1024 batch-10 evaluation step, top kernels per step
tag=$1; dt=${2:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/et_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --mode eval --size 1024 --batch 10 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $out/trace.log 2>&1 || exit 1
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 scripts/prof_summary.py $out/kernel_stats.csv 12 30
