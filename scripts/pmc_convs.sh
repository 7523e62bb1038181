#!/bin/bash
# SQ counters for selected layer shapes of scripts/bench_convs.py:  scripts/pmc_convs.sh <tag> "<shape indices>" "<counters>"
set -e
tag=$1; export SSMI_SHAPES=$2; counters=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 -M --pmc $counters --output-format csv -d $out/pass -- python3 scripts/bench_convs.py 16 > $out/pass.log 2>&1
python3 scripts/pmc_table.py $out
