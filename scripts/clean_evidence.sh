#!/bin/bash
# Per-kernel evidence with ONE kernel on the chip at a time (weight gradients kept on the main stream):
#   scripts/clean_evidence.sh <tag> <model> <batch> [f32|bf16]      (bf16: the float32 step runs first in the same process)
# 1 kernel trace (durations), 2 MFMA busy counters, 3+4 HBM fetch / write counters, 5 algorithmic flops / bytes per
# convolution variant.  scripts/evidence_table.py joins them.  Every profiler pass is its own process with its own limit.
set -u
tag=$1; model=$2; batch=$3; dt=${4:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ev_$tag
mkdir -p $out
B="--model $model --batch $batch --dtype $dt --no-overlap-wgrad --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 10 --warmup 2 $B > $out/trace.log 2>&1 || exit 1
cp $(ls $out/trace/*/*_kernel_stats.csv | head -1) $out/kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pass -- python3 bench.py --steps 2 --warmup 1 --eager $B > $out/pass.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --eager $B > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --eager $B > $out/write.log 2>&1 || exit 1
python3 scripts/pmc_traffic.py $out > $out/traffic_summary.txt || exit 1
timeout -k 10 200 python3 scripts/conv_records.py $model $batch $out/conv_records.json > $out/conv_records.txt 2>&1 || exit 1
python3 scripts/evidence_table.py $out 12 > $out/evidence.md && head -50 $out/evidence.md
