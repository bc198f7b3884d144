cd /root/repo
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 30 --warmup 5"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), d.get("exposed_exchange_ms"))'
$B 2>/dev/null | tail -1 | python -c "$P" plain
PGGAN_FORCE_DP=1 $B 2>/dev/null | tail -1 | python -c "$P" dp1
PGGAN_FORCE_DP=1 PGGAN_DP_BUCKETS=0 $B 2>/dev/null | tail -1 | python -c "$P" dp1-nobuckets
PGGAN_FORCE_DP=1 GPU_MAX_HW_QUEUES=4 $B 2>/dev/null | tail -1 | python -c "$P" dp1-4queues
GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | tail -1 | python -c "$P" plain-8queues
PGGAN_FORCE_DP=1 PGGAN_DP_TORCH_ALLREDUCE=1 $B 2>/dev/null | tail -1 | python -c "$P" dp1-torch-allreduce
