#!/bin/bash
# HBM traffic per kernel from the memory-side L2 counters, one rocprofv3 -M --pmc pass per counter (FETCH_SIZE takes 3 of the
# 4 TCC slots, WRITE_SIZE 2, so they cannot share a pass).  Run on the GPU box:
#     scripts/pmc_traffic.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/{fetch,write}/... csv and gpurun_out/pmc_<tag>/traffic.json (scripts/pmc_traffic.py).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 240 rocprofv3 -M --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline "$@" > $out/fetch.log 2>&1
timeout -s KILL 240 rocprofv3 -M --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline "$@" > $out/write.log 2>&1
python3 scripts/pmc_traffic.py $out
