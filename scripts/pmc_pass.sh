#!/bin/bash
# One rocprofv3 -M --pmc pass over an eager bench run, averaged per kernel:  scripts/pmc_pass.sh <tag> "<counters>" [bench args]
set -e
tag=$1; counters=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 240 rocprofv3 -M --pmc $counters --output-format csv -d $out/pass -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline "$@" > $out/pass.log 2>&1
python3 scripts/pmc_table.py $out
