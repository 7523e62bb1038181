cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 -M --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_final.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_final/*/*_kernel_stats.csv | head -1) gpurun_out/prof_final/kernel_stats.csv
for spec in "enet-max-unpooling 16 maxunpool" "enet-naive-upsampling-enhanced 16 enhanced" "segnet 4 segnet" "unet 4 unet"; do
  set -- $spec
  timeout -k 10 280 python bench.py --model $1 --batch $2 --no-cpu-baseline > gpurun_out/$3_bench.json 2>/dev/null || exit 1
done
timeout -k 10 280 python bench.py --mode eval --size 1024 --batch 10 --no-cpu-baseline > gpurun_out/eval1024_bench.json 2>/dev/null || exit 1
timeout -k 10 280 python bench.py --mode eval --size 1024 --batch 10 --dtype bf16 --no-cpu-baseline > gpurun_out/eval1024_bf16_bench.json 2>/dev/null || exit 1
cut -c1-400 gpurun_out/*_bench.json
