#!/bin/bash
# usage: scripts/eval_ms.sh MODEL BATCH SIZE DTYPE [VAR=VALUE ...]  - ms per evaluation step of one bench run
model=$1; batch=$2; size=$3; dt=$4; shift 4
env "$@" timeout -k 10 300 python bench.py --mode eval --model "$model" --batch "$batch" --size "$size" --dtype "$dt" --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('eval $model b$batch @$size $dt $*', d['ms_per_step'], 'ms', d['config']['miou'])"
