#!/bin/bash
# Per-layer convolution timings (scripts/bench_convs.py, batch 16) with the in-tree library and with another build swapped in:
#   scripts/ab_convs.sh <other .so> [shape indices]
other=$1; export SSMI_SHAPES=${2:-2,3,4,6,8,10,11}
lib=semantic-segmentation_amd/libssmi.so
cp $lib /tmp/libssmi_keep.so
echo "in-tree"; python scripts/bench_convs.py 16 2>/dev/null | cut -c1-62
cp $other $lib
echo "$other"; python scripts/bench_convs.py 16 2>/dev/null | cut -c1-62
cp /tmp/libssmi_keep.so $lib
