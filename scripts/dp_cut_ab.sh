cd "$GRAFT_REPO_ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29551
for dt in f32 bf16; do
for e in PLAIN=1 SSMI_FORCE_DP=1 "SSMI_FORCE_DP=1 SSMI_DP_CUTS=0"; do
  env $e timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --dtype $dt 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); b=d.get('bf16') or d
dp=(b.get('data_parallel') or d.get('data_parallel') or {})
print('$dt $e', b['ms_per_step'], dp.get('buckets_per_step'), dp.get('segments_ms'), dp.get('bucket_wait_ms'))" || exit 1
done; done
