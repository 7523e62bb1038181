#!/bin/bash
# usage: scripts/quick_trace.sh TAG MODEL BATCH DTYPE  - kernel trace of 12 steps (weight gradients on the main stream), top kernels printed
set -u
tag=$1; model=$2; batch=$3; dt=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/qt_$tag
mkdir -p $out
export SSMI_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/step_only.py $model $dt $batch 12 > $out/trace.log 2>&1 || exit 1
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 scripts/prof_summary.py $out/kernel_stats.csv 12 2>/dev/null | head -45 || head -30 $out/kernel_stats.csv
