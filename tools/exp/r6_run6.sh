set -x
mkdir -p gpurun_out/r6_sweep
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 5"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["d_step_gp"]["ms"])'
run() { echo -n "[$1] "; env $1 timeout 300 $B 2>/dev/null | python -c "$P"; }
{
for i in 1 2; do run PGGAN_MAIN_PRIORITY=0; run PGGAN_MAIN_PRIORITY=1; done
for v in "PG_WINO_KS_TARGET=864" "PG_WINO_KS_TARGET=1024" "PG_WINO_KS_TARGET=768" "PG_WINO_KS_TARGET=640" "PG_WINO_KS_MAX=16" "PG_WINO_KS_MINCH=2" "PG_WINO_KS_MINCH=8" "PG_WINO_KS_PAIRS=600" "PG_WINO_KS_PAIRS=300" "PG_WINO_KS_TARGET=864"; do run $v; done
} 2>&1 | tee gpurun_out/r6_sweep/sweep1.txt
