#!/bin/bash
# Step time (fp32, bf16) for combinations of weight-gradient streams and strips on one box: scripts/wgrad_combo.sh
run() { timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --dtype both 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16']['ms_per_step'])"; }
echo "default        $(run)"
for combo in "4 512" "2 512" "2 256" "8 256" "4 384" "4 192"; do
  set -- $combo
  echo "s$1 x $2       $(SSMI_WGRAD_STREAMS=$1 SSMI_WG_STRIPS=$2 run)"
done
echo "default        $(run)"
