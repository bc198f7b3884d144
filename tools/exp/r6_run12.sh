set -x
mkdir -p gpurun_out/r6_first
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 40 --warmup 5"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["d_step_gp"]["ms"])'
run() { echo -n "[$1] "; env $1 timeout 300 $B 2>/dev/null | python -c "$P"; }
{
for i in 1 2 3; do run A=0; run PGGAN_EARLY_REAL=1; run "PGGAN_EARLY_REAL=1 PGGAN_EARLY_G_FIRST=1"; done
} 2>&1 | grep -v "^+" | tee gpurun_out/r6_first/ab.txt
PGGAN_EARLY_REAL=1 PGGAN_EARLY_G_FIRST=1 python tools/phase_timeline.py > gpurun_out/r6_first/phase_timeline.txt 2>&1; tail -8 gpurun_out/r6_first/phase_timeline.txt
