#!/bin/bash
# A/B of two builds of the library on one GPU box: scripts/ab_lib.sh <old .so> [bench args...]
# Runs bench.py with the in-tree library, with <old .so> swapped in, and with the in-tree library again.
set -e
old=$1; shift
lib=semantic-segmentation_amd/libssmi.so
cp $lib /tmp/libssmi_new.so
run() { timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16']['ms_per_step'] if d.get('bf16') else '')"; }
echo "new $(run "$@")"
cp $old $lib
echo "old $(run "$@")"
cp /tmp/libssmi_new.so $lib
echo "new $(run "$@")"
cp $old $lib
echo "old $(run "$@")"
cp /tmp/libssmi_new.so $lib
