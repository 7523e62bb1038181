#!/bin/bash
# Per-kernel evidence with ONE kernel on the chip at a time (weight gradients kept on the main stream):
#   scripts/clean_evidence.sh <tag> <model> <batch> [f32|bf16]
# Every pass is its own process running scripts/step_only.py in ONE activation dtype (so a table never mixes the float32 and
# bfloat16 kernels of one run), with mangled kernel names (-M; scripts/kernel_names.py demangles them, bfloat16 included).
# 1 kernel trace (durations), 2 MFMA busy counters, 3+4 HBM fetch / write counters, 5 algorithmic flops / bytes per
# convolution variant.  scripts/evidence_table.py joins them.  Every profiler pass has its own limit.
set -u
tag=$1; model=$2; batch=$3; dt=${4:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ev_$tag
mkdir -p $out
export SSMI_NO_OVERLAP=1
timeout -k 10 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/step_only.py $model $dt $batch 12 > $out/trace.log 2>&1 || exit 1
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
export SSMI_EAGER=1
timeout -s KILL 300 rocprofv3 -M --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pass -- python3 scripts/step_only.py $model $dt $batch 3 > $out/pass.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 -M --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 scripts/step_only.py $model $dt $batch 3 > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 -M --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 scripts/step_only.py $model $dt $batch 3 > $out/write.log 2>&1 || exit 1
unset SSMI_EAGER
python3 scripts/pmc_traffic.py $out > $out/traffic_summary.txt || exit 1
timeout -k 10 200 python3 scripts/conv_records.py $model $batch $out/conv_records.json $dt > $out/conv_records.txt 2>&1 || exit 1
python3 scripts/evidence_table.py $out 12 > $out/evidence.md && head -60 $out/evidence.md
