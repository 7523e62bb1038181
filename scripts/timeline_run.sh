#!/bin/bash
# usage: scripts/timeline_run.sh TAG [VAR=VALUE ...]: kernel trace of the overlapped float32 ENet step, timeline of the last step
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl_$tag
mkdir -p $out
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 scripts/step_only.py enet-naive-upsampling f32 16 8 > $out/trace.log 2>&1 || exit 1
python3 scripts/step_timeline.py $(ls $out/trace/*/*_kernel_trace.csv | head -1) 30
