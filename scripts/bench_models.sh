#!/bin/bash
# Step time of every model at 512x512 on one GPU (no CPU baseline):  scripts/bench_models.sh
cd "$GRAFT_REPO_ROOT"
for spec in "enet-naive-upsampling 16" "enet-max-unpooling 16" "enet-naive-upsampling-enhanced 16" "segnet 4" "unet 4"; do
  set -- $spec
  timeout -k 10 280 python bench.py --model $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-34s batch %2d  %8.2f ms/step  %8.1f img/s  loss %.4f' % ('$1', $2, d['ms_per_step'], d['value'], d['config']['final_loss']))" || exit 1
done
