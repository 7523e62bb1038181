#!/bin/bash
# usage: scripts/bf16_ms.sh MODEL BATCH [VAR=VALUE ...]  - prints the f32 and bf16 ms/step of one bench run
model=$1; batch=$2; shift 2
env "$@" timeout -k 10 300 python bench.py --model "$model" --batch "$batch" --dtype both --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d.get('bf16') or {}
print('$model b$batch $*', 'f32', d['ms_per_step'], 'bf16', b.get('ms_per_step'))"
