#!/bin/bash
# Kernel trace of scripts/step_only.py on the GPU box: scripts/profile_step.sh <tag> [model] [dtype] [batch]
# Writes gpurun_out/prof_<tag>/ and prints the per-kernel table (12 steps).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 scripts/step_only.py "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*/*_kernel_stats.csv | head -1)
cp "$f" gpurun_out/prof_$tag/kernel_stats.csv
python3 scripts/prof_summary.py gpurun_out/prof_$tag/kernel_stats.csv 12 40
