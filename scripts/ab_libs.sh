#!/bin/bash
# usage: scripts/ab_libs.sh "SHAPE,INDICES" VARIANT...  - per-layer timings (scripts/bench_convs.py, SSMI_BENCH_DTYPE honoured) and the ENet
# step with ab_libs/libssmi_VARIANT.so swapped in for the in-tree library, one after the other; the in-tree file is put back at the end
lib=semantic-segmentation_amd/libssmi.so
export SSMI_SHAPES=$1; shift
cp $lib /tmp/libssmi_keep.so
for v in "$@"; do
  cp ab_libs/libssmi_$v.so $lib; echo "== $v"
  timeout -k 10 120 python scripts/bench_convs.py 16 2>/dev/null | cut -c1-62 || exit 1
  scripts/bf16_ms.sh enet-naive-upsampling 16 || exit 1
done
cp /tmp/libssmi_keep.so $lib
