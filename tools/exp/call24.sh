cd /root/repo
B="python bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --steps 20 --warmup 5"
$B 2>/dev/null | tail -1 > gpurun_out/r2_b_prio0.log
PGGAN_MAIN_PRIORITY=-1 $B 2>/dev/null | tail -1 > gpurun_out/r2_b_prio1.log
python tools/sweep_wino.py thin > gpurun_out/r2_sweep_wino_thin.txt 2>&1
$B 2>/dev/null | tail -1 > gpurun_out/r2_b_prio0b.log
PGGAN_MAIN_PRIORITY=-1 $B 2>/dev/null | tail -1 > gpurun_out/r2_b_prio1b.log
for f in prio0 prio1 prio0b prio1b; do python -c "import json,sys; d=json.loads(open('gpurun_out/r2_b_$f.log').read()); print('$f', d['value'], d['ms_per_step'])"; done
cat gpurun_out/r2_sweep_wino_thin.txt
