set -x
mkdir -p gpurun_out/r6_dp0
B="python bench.py --depth 0 --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 200 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["d_step_gp"]["ms"], d.get("step_issue"), d.get("allreduce_ms"), d.get("exposed_exchange_ms"))'
run() { echo -n "[$1] "; env $1 timeout 300 $B 2>gpurun_out/r6_dp0/err.txt | python -c "$P"; }
{
run A=1; run PGGAN_GRAPH_STAGE0=0; run PGGAN_FORCE_DP=1; run "PGGAN_FORCE_DP=1 PGGAN_GRAPH_STAGE0=0"; run "PGGAN_FORCE_DP=1 GPU_MAX_HW_QUEUES=4"; run "PGGAN_FORCE_DP=1 PGGAN_DP_BUCKETS=0 PGGAN_GRAPH_STAGE0=0"
} 2>&1 | tee gpurun_out/r6_dp0/depth0.txt
