#!/bin/bash
# Kernel-trace profile of the headline bench on the GPU box:  scripts/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ (rocprofv3 csv) ; summarise with scripts/prof_summary.py.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 -M --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*/*_kernel_stats.csv | head -1)
cp "$f" gpurun_out/prof_$tag/kernel_stats.csv
python3 scripts/prof_summary.py gpurun_out/prof_$tag/kernel_stats.csv 12 30
