#!/bin/bash
# Data-parallel step on the one-GPU box: (1) one-rank RCCL group, ENet and SegNet (multi-bucket), final loss against the
# plain single-process run; (2) two gloo ranks sharing the GPU (host-staged buckets), once.
set -u
cd "$GRAFT_REPO_ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29551
C="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --dtype f32"
for m in "enet-naive-upsampling --batch 16" "segnet --batch 4" "unet --batch 4"; do
  tag=$(echo $m | cut -d' ' -f1)
  timeout -k 10 200 python bench.py --model $m $C > gpurun_out/dpc_${tag}_plain.json 2> gpurun_out/dpc_${tag}_plain.err || { echo "plain $tag failed"; tail -5 gpurun_out/dpc_${tag}_plain.err; exit 1; }
  SSMI_FORCE_DP=1 timeout -k 10 200 python bench.py --model $m $C > gpurun_out/dpc_${tag}_dp1.json 2> gpurun_out/dpc_${tag}_dp1.err || { echo "dp1 $tag failed"; tail -5 gpurun_out/dpc_${tag}_dp1.err; exit 1; }
done
python - <<'PY'
import json
for m in ("enet-naive-upsampling", "segnet", "unet"):
    a = json.loads(open("gpurun_out/dpc_%s_plain.json" % m).read().strip().splitlines()[-1])
    b = json.loads(open("gpurun_out/dpc_%s_dp1.json" % m).read().strip().splitlines()[-1])
    print(m, "plain ms", a["ms_per_step"], "loss", a["config"]["final_loss"], "| one-rank RCCL ms", b["ms_per_step"], "loss", b["config"]["final_loss"])
    print("   ", json.dumps(b["data_parallel"]))
PY
bash scripts/dp_rehearsal.sh r03 100
